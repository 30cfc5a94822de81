// fuzz_io.cpp — mutation fuzz drivers for the host-side decoders on the from-file path (§8f-2 / §8f-3), built with
// -fsanitize=address,undefined by `make -C contextsv_amd/csrc asan` (CPU only: the GPU pool takes no sanitizer runs).
//
//   fuzz_io inflate <iterations> <seed>   raw DEFLATE streams (zlib-made, then corrupted / truncated / extended) through fastz::inflate:
//                                          intact streams must decode to the input; anything else must be refused or decoded without
//                                          touching memory outside the two buffers (the sanitizer is the judge; the CRC behind it in
//                                          bgzf::inflate_block is what catches wrong outputs of accepted streams)
//   fuzz_io bam <iterations> <seed>       BAM files whose RECORDS are corrupted while the BGZF framing and CRCs stay valid (bad block
//                                          sizes, counts, name / CIGAR / tag lengths), and BAI files with flipped bytes: BamReader must
//                                          fail with an error or succeed — never read outside its buffers
//   fuzz_io vcf <iterations> <seed>       SNP VCFs with mutated bytes, dropped columns, huge numbers: SNPFile::load must not crash
//
// Exit code 0 = no crash and no wrong result on intact inputs. Reference seams: htslib's bgzf / sam / vcf readers as used at
// src/sv_caller.cpp:48-181 and src/cnv_caller.cpp:558-809 (the reference delegates all of it to htslib).
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../contextsv_amd/csrc/host/bam_io.h"
#include "../../contextsv_amd/csrc/host/bgzf.h"
#include "../../contextsv_amd/csrc/host/fast_inflate.h"
#include "../../contextsv_amd/csrc/host/snp_io.h"

namespace {

std::string tmp_path(const char *stem)
{
    const char *d = getenv("TMPDIR");
    return std::string(d && *d ? d : "/tmp") + "/csv_fuzz_" + std::to_string((long)getpid()) + "_" + stem;
}

std::vector<uint8_t> raw_deflate(const std::vector<uint8_t> &src, int level)
{
    z_stream z;
    memset(&z, 0, sizeof z);
    if (deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) abort();
    std::vector<uint8_t> out(deflateBound(&z, (uLong)src.size()) + 16);
    z.next_in = (Bytef *)src.data(); z.avail_in = (uInt)src.size();
    z.next_out = out.data(); z.avail_out = (uInt)out.size();
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) abort();
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}

int fuzz_inflate(long iters, uint64_t seed)
{
    std::mt19937_64 g(seed);
    long accepted = 0, intact_ok = 0;
    for (long it = 0; it < iters; it++) {
        const size_t n = 1 + g() % 20000;
        std::vector<uint8_t> src(n);
        const int kind = (int)(g() % 4);
        for (size_t i = 0; i < n; i++)
            src[i] = kind == 0 ? (uint8_t)g() : kind == 1 ? (uint8_t)("ACGT"[g() & 3]) : kind == 2 ? (uint8_t)(i % 7 ? 'A' : (uint8_t)g()) : (uint8_t)((i * 31) >> 3);
        std::vector<uint8_t> z = raw_deflate(src, (int)(g() % 10));
        {   // intact: exact sizes -> the input back (or a refusal, which the caller answers with zlib)
            std::vector<uint8_t> in(z), out(n);
            if (fastz::inflate(in.data(), in.size(), out.data(), out.size())) {
                if (memcmp(out.data(), src.data(), n) != 0) { fprintf(stderr, "inflate: wrong output on an intact stream (iteration %ld)\n", it); return 1; }
                intact_ok++;
            }
        }
        for (int m = 0; m < 6; m++) {
            std::vector<uint8_t> in(z);
            size_t out_len = n;
            switch (g() % 5) {
            case 0: for (int k = 0; k < 1 + (int)(g() % 4); k++) in[g() % in.size()] ^= (uint8_t)(1u << (g() % 8)); break;
            case 1: in.resize(g() % (in.size() + 1)); break;
            case 2: for (int k = 0; k < 1 + (int)(g() % 64); k++) in.push_back((uint8_t)g()); break;
            case 3: out_len = g() % (2 * n + 2); break;
            default: for (size_t k = g() % in.size(); k < in.size(); k++) in[k] = (uint8_t)g(); break;
            }
            // exact-size heap buffers: one byte beyond either end is an AddressSanitizer report
            std::vector<uint8_t> a(in.begin(), in.end());
            uint8_t *out = out_len ? (uint8_t *)malloc(out_len) : nullptr;
            accepted += fastz::inflate(a.data(), a.size(), out, out_len) ? 1 : 0;
            free(out);
        }
    }
    printf("inflate: %ld iterations, %ld intact streams decoded by the fast path, %ld mutated streams accepted (left to the CRC)\n", iters, intact_ok, accepted);
    return 0;
}

// a small valid BAM (three contigs, every op, tags in front of nothing in particular) through the library's own writer
bool make_bam(const std::string &path, std::mt19937_64 &g, std::string *err)
{
    BamHeader h;
    h.text = "@HD\tVN:1.6\tSO:coordinate\n";
    for (int c = 0; c < 3; c++) { h.names.push_back("chr" + std::to_string(c + 1)); h.lens.push_back(100000 + 1000 * c); h.text += "@SQ\tSN:" + h.names.back() + "\tLN:" + std::to_string(h.lens.back()) + "\n"; }
    BamWriter w;
    if (!w.open(path, h, 1, 2)) { *err = w.error(); return false; }
    for (int c = 0; c < 3; c++) {
        int32_t pos = 10;
        for (int r = 0; r < 40; r++) {
            std::vector<uint32_t> cig;
            const int nops = 1 + (int)(g() % 12);
            for (int k = 0; k < nops; k++) cig.push_back((uint32_t)((1 + g() % 300) << 4) | (uint32_t)(g() % 9));
            const int lseq = (int)(g() % 40);
            std::vector<uint8_t> seq((size_t)(lseq + 1) / 2), qual((size_t)lseq, 30);
            for (auto &b : seq) b = (uint8_t)g();
            w.add(c, pos, (uint8_t)(g() % 61), (uint16_t)(g() % 4096), "read" + std::to_string(c) + "_" + std::to_string(r), cig.data(), (uint32_t)cig.size(), lseq ? seq.data() : nullptr, lseq,
                  lseq ? qual.data() : nullptr);
            pos += (int32_t)(g() % 500);
        }
    }
    if (!w.close()) { *err = w.error(); return false; }
    return true;
}

std::vector<uint8_t> slurp(const std::string &p)
{
    std::vector<uint8_t> v;
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return v;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
    fclose(f);
    return v;
}
bool spit(const std::string &p, const std::vector<uint8_t> &v)
{
    FILE *f = fopen(p.c_str(), "wb");
    if (!f) return false;
    const bool ok = v.empty() || fwrite(v.data(), 1, v.size(), f) == v.size();
    fclose(f);
    return ok;
}

int fuzz_bam(long iters, uint64_t seed)
{
    std::mt19937_64 g(seed);
    const std::string good = tmp_path("good.bam"), bad = tmp_path("bad.bam");
    std::string err;
    if (!make_bam(good, g, &err)) { fprintf(stderr, "bam: cannot write the seed file: %s\n", err.c_str()); return 1; }
    {   // the seed file reads back
        BamReader r;
        if (!r.open(good) || !r.loadIndex()) { fprintf(stderr, "bam: the seed file does not open: %s\n", r.error().c_str()); return 1; }
        BamReadOptions o; o.want_qnames = true; o.want_seq = true; o.threads = 2;
        BamShard s;
        if (!r.readContig("chr2", o, s) || s.n_reads() != 40) { fprintf(stderr, "bam: the seed file reads %lu records of chr2: %s\n", (unsigned long)s.n_reads(), r.error().c_str()); return 1; }
    }
    // the uncompressed payload of the seed file
    const std::vector<uint8_t> file = slurp(good), bai = slurp(good + ".bai");
    std::vector<bgzf::Block> blocks;
    if (!bgzf::scan_blocks(file.data(), file.size(), 0, blocks, &err)) { fprintf(stderr, "bam: block scan failed: %s\n", err.c_str()); return 1; }
    std::vector<uint8_t> payload;
    for (const bgzf::Block &b : blocks) {
        const size_t at = payload.size();
        payload.resize(at + b.isize);
        if (b.isize && !bgzf::inflate_block(file.data(), b, payload.data() + at, &err)) { fprintf(stderr, "bam: inflate failed: %s\n", err.c_str()); return 1; }
    }
    long opened = 0, read_ok = 0;
    for (long it = 0; it < iters; it++) {
        std::vector<uint8_t> p(payload);
        std::vector<uint8_t> idx(bai);
        const int what = (int)(g() % 4);
        if (what < 3) {
            for (int k = 0; k < 1 + (int)(g() % 6); k++) {
                const size_t at = g() % p.size();
                switch (g() % 3) {
                case 0: p[at] = (uint8_t)g(); break;
                case 1: p[at] = 0xff; if (at + 1 < p.size()) p[at + 1] = 0xff; break;                 // large counts / lengths
                default: if (at + 4 <= p.size()) { const uint32_t v = (uint32_t)g() % 70000; memcpy(&p[at], &v, 4); } break;
                }
            }
            if (g() % 8 == 0) p.resize(g() % p.size());                                                  // truncated records
        } else if (!idx.empty()) {
            for (int k = 0; k < 1 + (int)(g() % 8); k++) idx[g() % idx.size()] ^= (uint8_t)(1u << (g() % 8));
        }
        // valid BGZF framing and CRCs around whatever the records now say
        std::vector<uint8_t> out;
        const size_t blk = 700 + g() % 60000;
        for (size_t at = 0; at < p.size(); at += blk) bgzf::deflate_block(p.data() + at, (uint32_t)std::min(blk, p.size() - at), 1, out);
        bgzf::deflate_block(nullptr, 0, 1, out);                                                         // EOF marker
        if (!spit(bad, out) || !spit(bad + ".bai", idx)) { fprintf(stderr, "bam: cannot write %s\n", bad.c_str()); return 1; }
        BamReader r;
        if (!r.open(bad)) continue;
        opened++;
        BamReadOptions o; o.want_qnames = (g() & 1) != 0; o.want_seq = (g() & 1) != 0; o.threads = 1 + (int)(g() % 3);
        if (r.loadIndex()) { BamShard s; if (r.readContig("chr" + std::to_string(1 + g() % 3), o, s)) read_ok++; }
        uint64_t unplaced = 0;
        r.readAll(o, [](BamShard &&) {}, &unplaced);
    }
    remove(good.c_str()); remove((good + ".bai").c_str()); remove(bad.c_str()); remove((bad + ".bai").c_str());
    printf("bam: %ld mutated files, %ld opened, %ld contig reads succeeded\n", iters, opened, read_ok);
    return 0;
}

int fuzz_vcf(long iters, uint64_t seed)
{
    std::mt19937_64 g(seed);
    const std::string path = tmp_path("snps.vcf");
    std::string base = "##fileformat=VCFv4.2\n##contig=<ID=chr1,length=100000>\n##FORMAT=<ID=GT,Number=1,Type=String,Description=\"g\">\n"
                       "##FORMAT=<ID=DP,Number=1,Type=Integer,Description=\"d\">\n##FORMAT=<ID=AD,Number=R,Type=Integer,Description=\"a\">\n"
                       "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS1\n";
    for (int i = 0; i < 60; i++)
        base += "chr1\t" + std::to_string(100 + 37 * i) + "\t.\t" + "ACGT"[i & 3] + "\t" + "CGTA"[i & 3] + "\t" + std::to_string(20 + i) + "\t" + (i % 5 ? "PASS" : ".") +
                "\t.\tGT:DP:AD\t0/1:" + std::to_string(8 + i) + ":" + std::to_string(4 + i / 2) + "," + std::to_string(4 + i / 3) + "\n";
    {
        std::vector<uint8_t> v(base.begin(), base.end());
        if (!spit(path, v)) return 1;
        SNPFile f; std::string err;
        if (!f.load(path, 2, &err) || f.records_kept() == 0) { fprintf(stderr, "vcf: the seed file does not load: %s\n", err.c_str()); return 1; }
    }
    long loaded = 0;
    for (long it = 0; it < iters; it++) {
        std::string s = base;
        for (int k = 0; k < 1 + (int)(g() % 8); k++) {
            const size_t at = g() % s.size();
            switch (g() % 5) {
            case 0: s[at] = (char)g(); break;
            case 1: s[at] = '\t'; break;
            case 2: s[at] = '\n'; break;
            case 3: s.insert(at, "99999999999999999999"); break;
            default: s.erase(at, g() % 40); break;
            }
            if (s.empty()) s = "#";
        }
        std::vector<uint8_t> v(s.begin(), s.end());
        if (!spit(path, v)) return 1;
        SNPFile f; std::string err;
        if (f.load(path, 1 + (int)(g() % 3), &err)) {
            loaded++;
            const SNPFileTable &t = f.table("chr1", "", "", 1);
            std::vector<uint32_t> pos; std::unordered_map<uint32_t, double> baf, pfb;
            t.query(1, 100000, pos, baf, pfb);
        }
    }
    remove(path.c_str());
    printf("vcf: %ld mutated files, %ld loaded\n", iters, loaded);
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: fuzz_io inflate|bam|vcf [iterations] [seed]\n"); return 2; }
    const long iters = argc > 2 ? atol(argv[2]) : 200;
    const uint64_t seed = argc > 3 ? strtoull(argv[3], nullptr, 10) : 1;
    const std::string mode = argv[1];
    if (mode == "inflate") return fuzz_inflate(iters, seed);
    if (mode == "bam") return fuzz_bam(iters, seed);
    if (mode == "vcf") return fuzz_vcf(iters, seed);
    fprintf(stderr, "unknown mode %s\n", argv[1]);
    return 2;
}
