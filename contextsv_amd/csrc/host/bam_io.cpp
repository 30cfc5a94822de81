#include "bam_io.h"

#include <algorithm>
#include <cstring>
#include <future>
#include <map>
#include <thread>

namespace {
inline uint16_t le16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t le64(const uint8_t *p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }
inline void put16(std::vector<uint8_t> &v, uint16_t x) { v.push_back((uint8_t)x); v.push_back((uint8_t)(x >> 8)); }
inline void put32(std::vector<uint8_t> &v, uint32_t x) { for (int i = 0; i < 4; i++) v.push_back((uint8_t)(x >> (8 * i))); }
inline void put64(std::vector<uint8_t> &v, uint64_t x) { for (int i = 0; i < 8; i++) v.push_back((uint8_t)(x >> (8 * i))); }

constexpr uint32_t OP_S = 4, OP_N = 3;
constexpr int kPseudoBin = 37450;
constexpr int kLinearShift = 14;

// Locate the CG:B,I array among the auxiliary fields (bam_aux_get + the checks of htslib's bam_tag2cigar).
// Returns the element pointer and count, or nullptr.
const uint8_t *find_cg(const uint8_t *aux, const uint8_t *end, uint32_t *count)
{
    const uint8_t *p = aux;
    while (p + 3 <= end) {
        const bool is_cg = p[0] == 'C' && p[1] == 'G';
        const uint8_t type = p[2];
        p += 3;
        size_t len;
        switch (type) {
            case 'A': case 'c': case 'C': len = 1; break;
            case 's': case 'S': len = 2; break;
            case 'i': case 'I': case 'f': len = 4; break;
            case 'd': len = 8; break;
            case 'Z': case 'H': {
                const uint8_t *z = (const uint8_t *)memchr(p, 0, (size_t)(end - p));
                if (!z) return nullptr;
                len = (size_t)(z - p) + 1;
                break;
            }
            case 'B': {
                if (p + 5 > end) return nullptr;
                const uint8_t sub = p[0];
                const uint32_t n = le32(p + 1);
                size_t es;
                switch (sub) { case 'c': case 'C': es = 1; break; case 's': case 'S': es = 2; break; case 'i': case 'I': case 'f': es = 4; break; default: return nullptr; }
                if ((size_t)(end - p - 5) < (size_t)n * es) return nullptr;
                if (is_cg) {
                    if (sub != 'I' && sub != 'i') return nullptr;
                    *count = n;
                    return p + 5;
                }
                len = 5 + (size_t)n * es;
                break;
            }
            default: return nullptr;
        }
        if (is_cg) return nullptr;                     // a CG tag of another type is not a CIGAR
        if ((size_t)(end - p) < len) return nullptr;
        p += len;
    }
    return nullptr;
}
}  // namespace

int32_t bam_ref_len(const uint32_t *cigar, uint32_t n_cigar)
{
    // M, D, N, =, X consume the reference (bam_cigar_type bit 2: 0x3C1A7 >> (op << 1) & 2)
    int32_t l = 0;
    for (uint32_t i = 0; i < n_cigar; i++)
        if ((0x3C1A7u >> ((cigar[i] & 15) << 1)) & 2) l += (int32_t)(cigar[i] >> 4);
    return l;
}

int bam_reg2bin(int64_t beg, int64_t end)
{
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

int BamHeader::tid(const std::string &name) const
{
    for (size_t i = 0; i < names.size(); i++) if (names[i] == name) return (int)i;
    return -1;
}

csv_reads BamShard::view() const
{
    csv_reads r{};
    r.n_reads = pos.size();
    r.n_cigar = cigar.size();
    r.pos = pos.data(); r.flag = flag.data(); r.mapq = mapq.data(); r.tid = nullptr;
    r.cigar_off = cigar_off.data(); r.cigar = cigar.data();
    return r;
}

void BamShard::clear()
{
    pos.clear(); flag.clear(); mapq.clear(); cigar.clear(); seq.clear(); qnames.clear();
    cigar_off.assign(1, 0);
    seq_off.assign(1, 0);
}

// ---- BAI -----------------------------------------------------------------------------------------------
struct BamReader::Index {
    struct Ref { bool any = false; uint64_t min_beg = ~0ull, max_end = 0; };
    std::vector<Ref> refs;
};

BamReader::BamReader() = default;
BamReader::~BamReader() = default;

bool BamReader::open(const std::string &p)
{
    path = p;
    if (!file.open(p, &err)) return false;
    if (file.size() == 0) { err = "BAM: empty file " + p; return false; }
    // The header is read block by block (it is small); remember where the first record starts.
    std::vector<uint8_t> text;
    uint64_t coff = 0;
    std::vector<uint64_t> block_start;         // uncompressed offset -> block bookkeeping for the first-record voffset
    std::vector<uint64_t> block_coff;
    auto need = [&](size_t n) -> bool {
        while (text.size() < n) {
            if (coff >= file.size()) { err = "BAM: truncated header in " + p; return false; }
            bgzf::Block b;
            if (!bgzf::parse_block(file.data() + coff, file.size() - coff, coff, b, &err)) return false;
            block_start.push_back(text.size());
            block_coff.push_back(coff);
            const size_t at = text.size();
            text.resize(at + b.isize);
            if (!bgzf::inflate_block(file.data(), b, text.data() + at, &err)) return false;
            coff += b.csize;
        }
        return true;
    };
    if (!need(12)) return false;
    if (memcmp(text.data(), "BAM\1", 4) != 0) { err = "BAM: bad magic in " + p; return false; }
    const uint32_t l_text = le32(text.data() + 4);
    if (!need(12 + (size_t)l_text)) return false;
    hdr.text.assign((const char *)text.data() + 8, l_text);
    while (!hdr.text.empty() && hdr.text.back() == '\0') hdr.text.pop_back();
    size_t o = 8 + (size_t)l_text;
    const uint32_t n_ref = le32(text.data() + o);
    o += 4;
    for (uint32_t i = 0; i < n_ref; i++) {
        if (!need(o + 4)) return false;
        const uint32_t l_name = le32(text.data() + o);
        if (!need(o + 4 + (size_t)l_name + 4)) return false;
        std::string name((const char *)text.data() + o + 4, l_name ? l_name - 1 : 0);
        hdr.names.push_back(name);
        hdr.lens.push_back(le32(text.data() + o + 4 + l_name));
        o += 8 + (size_t)l_name;
    }
    // o = uncompressed offset of the first record
    size_t bi = block_start.size() - 1;
    while (bi > 0 && block_start[bi] > o) bi--;
    if (o == text.size()) first_record_voffset = bgzf::voffset(coff, 0);       // starts with the next block
    else first_record_voffset = bgzf::voffset(block_coff[bi], (uint32_t)(o - block_start[bi]));
    return true;
}

bool BamReader::loadIndex(const std::string &index_path)
{
    std::vector<std::string> candidates;
    if (!index_path.empty()) candidates.push_back(index_path);
    else {
        candidates.push_back(path + ".bai");
        if (path.size() > 4 && path.compare(path.size() - 4, 4, ".bam") == 0) candidates.push_back(path.substr(0, path.size() - 4) + ".bai");
    }
    bgzf::MappedFile f;
    bool ok = false;
    std::string e;
    for (const std::string &c : candidates) if (f.open(c, &e)) { ok = true; break; }
    if (!ok) { err = "BAM: could not load index for " + path; return false; }
    const uint8_t *p = f.data();
    const size_t n = f.size();
    if (n < 8 || memcmp(p, "BAI\1", 4) != 0) { err = "BAI: bad magic"; return false; }
    auto idx = std::make_unique<Index>();
    const uint32_t n_ref = le32(p + 4);
    size_t o = 8;
    if ((uint64_t)n_ref * 8 > n - 8) { err = "BAI: truncated"; return false; }      // (every reference has at least its two counts: found by tools/fuzz — a flipped count asked for 2^32 entries)
    idx->refs.resize(n_ref);
    for (uint32_t r = 0; r < n_ref; r++) {
        if (o + 4 > n) { err = "BAI: truncated"; return false; }
        const uint32_t n_bin = le32(p + o);
        o += 4;
        for (uint32_t b = 0; b < n_bin; b++) {
            if (o + 8 > n) { err = "BAI: truncated"; return false; }
            const uint32_t bin = le32(p + o), n_chunk = le32(p + o + 4);
            o += 8;
            if (o + 16ull * n_chunk > n) { err = "BAI: truncated"; return false; }
            if (bin != (uint32_t)kPseudoBin) {
                for (uint32_t c = 0; c < n_chunk; c++) {
                    const uint64_t beg = le64(p + o + 16ull * c), end = le64(p + o + 16ull * c + 8);
                    idx->refs[r].any = true;
                    idx->refs[r].min_beg = std::min(idx->refs[r].min_beg, beg);
                    idx->refs[r].max_end = std::max(idx->refs[r].max_end, end);
                }
            }
            o += 16ull * n_chunk;
        }
        if (o + 4 > n) { err = "BAI: truncated"; return false; }
        const uint32_t n_intv = le32(p + o);
        o += 4 + 8ull * n_intv;
        if (o > n) { err = "BAI: truncated"; return false; }
    }
    index = std::move(idx);
    return true;
}

namespace {
// run fn(t) for t in [0, nt) on nt threads (the caller's included)
template <class F>
void parallel_chunks(int nt, F fn)
{
    if (nt <= 1) { fn(0); return; }
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; t++) pool.emplace_back(fn, t);
    fn(0);
    for (auto &t : pool) t.join();
}
}  // namespace

// Decompressed record stream from a virtual offset: batches of blocks are inflated by the pool while the previous batch is parsed.
bool BamReader::stream(uint64_t start_voffset, const BamReadOptions &opt, const std::function<size_t(const RecRef *, size_t)> &on_batch)
{
    struct Batch { std::vector<bgzf::Block> blocks; HugeVec<uint8_t> data; size_t head = 0; bool ok = true; std::string err; uint64_t next_coff = 0; };
    const uint32_t W = std::max<uint32_t>(opt.window_blocks, 1);
    const size_t kHead = 1 << 20;                        // room in front of each batch for the carried partial record (grown when needed)
    auto load = [&](uint64_t coff, Batch &b, size_t head) {
        b.blocks.clear();
        b.ok = true;
        uint64_t total = 0;
        while (coff < file.size() && b.blocks.size() < W) {
            bgzf::Block blk;
            if (!bgzf::parse_block(file.data() + coff, file.size() - coff, coff, blk, &b.err)) { b.ok = false; return; }
            b.blocks.push_back(blk);
            total += blk.isize;
            coff += blk.csize;
        }
        b.next_coff = coff;
        b.head = head;
        b.data.resize_uninit(head + total);
        if (!bgzf::inflate_range(file.data(), b.blocks, 0, b.blocks.size(), b.data.data() + head, opt.threads, &b.err)) b.ok = false;
    };
    Batch batch[2];
    uint64_t coff = start_voffset >> 16;
    size_t skip = (size_t)(start_voffset & 0xffff);
    if (coff >= file.size()) return true;
    load(coff, batch[0], kHead);
    int cur = 0;
    std::vector<uint8_t> carry;
    HugeVec<uint8_t> joined;                             // only when a carried record is larger than the head room
    std::vector<RecRef> recs;
    for (;;) {
        Batch &b = batch[cur];
        if (!b.ok) { err = b.err; return false; }
        if (b.blocks.empty()) break;
        std::future<void> ahead;
        const bool more = b.next_coff < file.size();
        if (more) ahead = std::async(std::launch::async, load, b.next_coff, std::ref(batch[cur ^ 1]), kHead);
        auto fail = [&](const std::string &m) { err = m; if (more) ahead.get(); return false; };
        // records of this batch = carried bytes + inflated bytes
        uint8_t *p = b.data.data() + b.head;
        size_t n = b.data.size() - b.head;
        if (skip) { if (skip > n) return fail("BAM: virtual offset beyond its block"); p += skip; n -= skip; skip = 0; }
        if (!carry.empty()) {
            if (carry.size() <= (size_t)(p - b.data.data())) { p -= carry.size(); memcpy(p, carry.data(), carry.size()); n += carry.size(); }
            else {
                joined.clear();
                joined.append(carry.data(), carry.size());
                joined.append(p, n);
                p = joined.data(); n = joined.size();
            }
            carry.clear();
        }
        recs.clear();
        size_t o = 0;
        while (o + 4 <= n) {
            const uint32_t bs = le32(p + o);
            if (bs < 32) return fail("BAM: record shorter than its fixed fields");
            if (bs > (1u << 30)) return fail("BAM: implausible record length");
            if (o + 4 + (size_t)bs > n) break;
            recs.push_back(RecRef{p + o + 4, bs});
            o += 4 + (size_t)bs;
        }
        const size_t took = on_batch(recs.data(), recs.size());
        if (took < recs.size() || !err.empty()) { if (more) ahead.get(); return err.empty(); }
        carry.assign(p + o, p + n);
        if (!more) break;
        ahead.get();
        cur ^= 1;
    }
    if (!carry.empty()) { err = "BAM: truncated record at end of file"; return false; }
    return true;
}

// Records -> arrays: a serial pass sizes every record (and resolves CG-tag CIGARs), then the copies run on the pool.
bool BamReader::append(const RecRef *recs, size_t n, const BamReadOptions &opt, BamShard &out)
{
    if (n == 0) return true;
    struct Src { const uint8_t *cig; uint32_t n_cig; };
    std::vector<Src> src(n);
    const size_t r0 = out.pos.size();
    out.cigar_off.resize(r0 + 1 + n);
    if (opt.want_seq) out.seq_off.resize(r0 + 1 + n);
    uint64_t cig_at = out.cigar.size(), seq_at = out.seq.size();
    for (size_t i = 0; i < n; i++) {
        const uint8_t *rec = recs[i].p;
        const uint32_t len = recs[i].len;
        const uint32_t l_name = rec[8], n_cigar = le16(rec + 12);
        const int32_t l_seq = (int32_t)le32(rec + 16);
        if (l_seq < 0) { err = "BAM: negative sequence length"; return false; }
        const size_t fixed = 32 + (size_t)l_name + 4ull * n_cigar + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
        if (fixed > len) { err = "BAM: record fields exceed the record length"; return false; }
        const uint8_t *cig = rec + 32 + l_name;
        src[i] = Src{cig, n_cigar};
        // long CIGAR in the CG tag behind a <l_seq>S<n>N placeholder (bam_tag2cigar)
        if (n_cigar > 0 && (int32_t)le32(rec) >= 0 && (int32_t)le32(rec + 4) >= 0) {
            const uint32_t c0 = le32(cig);
            if ((c0 & 15) == OP_S && (int32_t)(c0 >> 4) == l_seq) {
                uint32_t cnt = 0;
                const uint8_t *cg = find_cg(rec + fixed, rec + len, &cnt);
                if (cg && cnt >= n_cigar && cnt < (1u << 29)) src[i] = Src{cg, cnt};
            }
        }
        cig_at += src[i].n_cig;
        out.cigar_off[r0 + 1 + i] = cig_at;
        if (opt.want_seq) { seq_at += ((size_t)l_seq + 1) / 2; out.seq_off[r0 + 1 + i] = seq_at; }
    }
    out.pos.resize(r0 + n);
    out.flag.resize(r0 + n);
    out.mapq.resize(r0 + n);
    out.cigar.resize_uninit(cig_at);
    if (opt.want_seq) out.seq.resize_uninit(seq_at);
    if (opt.want_qnames) out.qnames.resize(r0 + n);
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(opt.threads, 1), n / 256 + 1));
    parallel_chunks(nt, [&](int t) {
        const size_t a = n * (size_t)t / (size_t)nt, b = n * ((size_t)t + 1) / (size_t)nt;
        for (size_t i = a; i < b; i++) {
            const uint8_t *rec = recs[i].p;
            const size_t r = r0 + i;
            out.pos[r] = (int32_t)le32(rec + 4);
            out.mapq[r] = rec[9];
            out.flag[r] = le16(rec + 14);
            if (src[i].n_cig) memcpy(out.cigar.data() + out.cigar_off[r], src[i].cig, 4ull * src[i].n_cig);   // BAM is little-endian, as is the target
            const uint32_t l_name = rec[8];
            if (opt.want_seq) {
                const uint8_t *seq = rec + 32 + l_name + 4ull * le16(rec + 12);
                const size_t nb = (size_t)(out.seq_off[r + 1] - out.seq_off[r]);
                if (nb) memcpy(out.seq.data() + out.seq_off[r], seq, nb);
            }
            if (opt.want_qnames) out.qnames[r].assign((const char *)rec + 32, l_name ? strnlen((const char *)rec + 32, l_name) : 0);
        }
    });
    return true;
}

bool BamReader::readContig(const std::string &chr, const BamReadOptions &opt, BamShard &out)
{
    err.clear();
    const int tid = hdr.tid(chr);
    if (tid < 0) { err = "BAM: unknown contig " + chr; return false; }
    if (!index) { err = "BAM: no index loaded for " + path; return false; }
    out.clear();
    out.tid = tid; out.name = chr; out.target_len = hdr.lens[tid];
    if ((size_t)tid >= index->refs.size() || !index->refs[tid].any) return true;
    const Index::Ref &ref = index->refs[tid];
    // size the big arrays once: the contig's compressed span bounds its records (BGZF rarely expands; CIGAR words dominate)
    // (offsets come from the index file: a corrupt one must not size an allocation — found by tools/fuzz)
    if ((ref.max_end >> 16) < (ref.min_beg >> 16) || (ref.min_beg >> 16) > file.size()) { err = "BAI: chunk offsets outside " + path; return false; }
    const uint64_t span = std::min<uint64_t>((ref.max_end >> 16) - (ref.min_beg >> 16), file.size()) + bgzf::kMaxBlock;
    out.cigar.reserve((size_t)(span * 4 / 4));            // ~4x compression of CIGAR words is typical; growth is by mremap anyway
    const int64_t end = hdr.lens[tid];
    return stream(ref.min_beg, opt, [&](const RecRef *recs, size_t n) -> size_t {
        // the contig's records are one run in a sorted file: skip an earlier contig's tail, stop at a later one / unplaced / past the end
        size_t a = 0;
        while (a < n && (int32_t)le32(recs[a].p) >= 0 && (int32_t)le32(recs[a].p) < tid) a++;
        size_t b = a;
        while (b < n && (int32_t)le32(recs[b].p) == tid && (int32_t)le32(recs[b].p + 4) < end) b++;
        if (!append(recs + a, b - a, opt, out)) return 0;
        return b == n ? n : b;                            // anything left over ends the iteration
    }) && err.empty();
}

bool BamReader::readAll(const BamReadOptions &opt, const std::function<void(BamShard &&)> &sink, uint64_t *n_unplaced)
{
    err.clear();
    BamShard cur;
    cur.clear();
    uint64_t unplaced = 0;
    auto flush = [&] {
        if (cur.tid >= 0 && cur.n_reads()) sink(std::move(cur));
        cur = BamShard();
        cur.clear();
    };
    const bool ok = stream(first_record_voffset, opt, [&](const RecRef *recs, size_t n) -> size_t {
        size_t i = 0;
        while (i < n) {
            const int32_t rtid = (int32_t)le32(recs[i].p);
            size_t j = i + 1;
            while (j < n && (int32_t)le32(recs[j].p) == rtid) j++;
            if (rtid < 0 || (size_t)rtid >= hdr.names.size()) unplaced += j - i;
            else {
                if (rtid != cur.tid) {
                    flush();
                    cur.tid = rtid; cur.name = hdr.names[rtid]; cur.target_len = hdr.lens[rtid];
                }
                if (!append(recs + i, j - i, opt, cur)) return 0;
            }
            i = j;
        }
        return n;
    }) && err.empty();
    if (ok) flush();
    if (n_unplaced) *n_unplaced = unplaced;
    return ok;
}

// ---- writer ------------------------------------------------------------------------------------------------
BamWriter::BamWriter() = default;
BamWriter::~BamWriter() = default;

bool BamWriter::open(const std::string &p, const BamHeader &header, int level, int threads)
{
    path = p;
    hdr = header;
    if (!out.open(p, level, threads, &err)) return false;
    std::vector<uint8_t> h;
    h.insert(h.end(), {'B', 'A', 'M', 1});
    put32(h, (uint32_t)hdr.text.size());
    h.insert(h.end(), hdr.text.begin(), hdr.text.end());
    put32(h, (uint32_t)hdr.names.size());
    for (size_t i = 0; i < hdr.names.size(); i++) {
        put32(h, (uint32_t)hdr.names[i].size() + 1);
        h.insert(h.end(), hdr.names[i].begin(), hdr.names[i].end());
        h.push_back(0);
        put32(h, hdr.lens[i]);
    }
    out.append(h.data(), h.size());
    opened = true;
    return true;
}

void BamWriter::add(int32_t tid, int32_t pos, uint8_t mapq, uint16_t flag, const std::string &qname, const uint32_t *cigar, uint32_t n_cigar,
                    const uint8_t *seq4, int32_t l_seq, const uint8_t *qual)
{
    const int32_t end = bam_end_pos(pos, flag, cigar, n_cigar);
    const bool long_cigar = n_cigar > 65535;
    const uint32_t l_name = (uint32_t)qname.size() + 1;
    rec.clear();
    put32(rec, 0);                                   // block_size, patched below
    put32(rec, (uint32_t)tid);
    put32(rec, (uint32_t)pos);
    rec.push_back((uint8_t)l_name);
    rec.push_back(mapq);
    put16(rec, (uint16_t)bam_reg2bin(pos, end));
    put16(rec, (uint16_t)(long_cigar ? 2 : n_cigar));
    put16(rec, flag);
    put32(rec, (uint32_t)l_seq);
    put32(rec, (uint32_t)-1);                        // next_refID
    put32(rec, (uint32_t)-1);                        // next_pos
    put32(rec, 0);                                   // tlen
    rec.insert(rec.end(), qname.begin(), qname.end());
    rec.push_back(0);
    if (long_cigar) {
        put32(rec, ((uint32_t)l_seq << 4) | OP_S);
        put32(rec, ((uint32_t)bam_ref_len(cigar, n_cigar) << 4) | OP_N);
    } else {
        const size_t at = rec.size();
        rec.resize(at + 4ull * n_cigar);
        if (n_cigar) memcpy(rec.data() + at, cigar, 4ull * n_cigar);
    }
    if (l_seq > 0) {
        rec.insert(rec.end(), seq4, seq4 + ((size_t)l_seq + 1) / 2);
        if (qual) rec.insert(rec.end(), qual, qual + l_seq);
        else rec.insert(rec.end(), (size_t)l_seq, (uint8_t)0xff);
    }
    if (long_cigar) {
        rec.insert(rec.end(), {'C', 'G', 'B', 'I'});
        put32(rec, n_cigar);
        const size_t at = rec.size();
        rec.resize(at + 4ull * n_cigar);
        memcpy(rec.data() + at, cigar, 4ull * n_cigar);
    }
    const uint32_t bs = (uint32_t)rec.size() - 4;
    rec[0] = (uint8_t)bs; rec[1] = (uint8_t)(bs >> 8); rec[2] = (uint8_t)(bs >> 16); rec[3] = (uint8_t)(bs >> 24);
    IndexEntry e;
    e.tid = tid; e.beg = pos; e.end = end; e.flag = flag;
    out.where(e.blk0, e.uo0);
    out.append(rec.data(), rec.size());
    out.where(e.blk1, e.uo1);
    entries.push_back(e);
    n_records++;
}

bool BamWriter::close()
{
    if (!opened) return true;
    opened = false;
    if (!out.close(&err)) return false;
    return writeIndex();
}

bool BamWriter::writeIndex()
{
    struct Chunk { uint64_t beg, end; };
    struct Ref {
        std::map<uint32_t, std::vector<Chunk>> bins;
        std::vector<uint64_t> linear;
        uint64_t off_beg = ~0ull, off_end = 0, n_mapped = 0, n_unmapped = 0;
        uint32_t last_bin = ~0u;
    };
    std::vector<Ref> refs(hdr.names.size());
    uint64_t n_no_coor = 0;
    for (const IndexEntry &e : entries) {
        if (e.tid < 0 || (size_t)e.tid >= refs.size()) { n_no_coor++; continue; }
        Ref &r = refs[e.tid];
        const uint64_t v0 = bgzf::voffset(out.block_coffset(e.blk0), e.uo0), v1 = bgzf::voffset(out.block_coffset(e.blk1), e.uo1);
        const uint32_t bin = (uint32_t)bam_reg2bin(e.beg, e.end);
        std::vector<Chunk> &ch = r.bins[bin];
        if (r.last_bin == bin && !ch.empty()) ch.back().end = v1;      // a run of records in one bin is one chunk
        else ch.push_back(Chunk{v0, v1});
        r.last_bin = bin;
        const size_t w0 = (size_t)(std::max(e.beg, 0) >> kLinearShift), w1 = (size_t)(std::max(e.end - 1, 0) >> kLinearShift);
        if (r.linear.size() <= w1) r.linear.resize(w1 + 1, ~0ull);
        for (size_t w = w0; w <= w1; w++) if (r.linear[w] == ~0ull) r.linear[w] = v0;
        r.off_beg = std::min(r.off_beg, v0);
        r.off_end = std::max(r.off_end, v1);
        if (e.flag & 4) r.n_unmapped++; else r.n_mapped++;
    }
    std::vector<uint8_t> b;
    b.insert(b.end(), {'B', 'A', 'I', 1});
    put32(b, (uint32_t)refs.size());
    for (Ref &r : refs) {
        const bool any = !r.bins.empty();
        put32(b, (uint32_t)r.bins.size() + (any ? 1 : 0));
        for (const auto &kv : r.bins) {
            put32(b, kv.first);
            put32(b, (uint32_t)kv.second.size());
            for (const Chunk &c : kv.second) { put64(b, c.beg); put64(b, c.end); }
        }
        if (any) {                                       // metadata pseudo-bin (SAMv1 §5.2)
            put32(b, (uint32_t)kPseudoBin);
            put32(b, 2);
            put64(b, r.off_beg); put64(b, r.off_end);
            put64(b, r.n_mapped); put64(b, r.n_unmapped);
        }
        for (size_t w = r.linear.size(); w-- > 0;)       // windows nothing starts in take the next window's offset
            if (r.linear[w] == ~0ull) r.linear[w] = w + 1 < r.linear.size() ? r.linear[w + 1] : 0;
        put32(b, (uint32_t)r.linear.size());
        for (uint64_t v : r.linear) put64(b, v);
    }
    put64(b, n_no_coor);
    FILE *f = fopen((path + ".bai").c_str(), "wb");
    if (!f) { err = "cannot create " + path + ".bai"; return false; }
    const bool ok = fwrite(b.data(), 1, b.size(), f) == b.size();
    if (fclose(f) != 0 || !ok) { err = "short write on " + path + ".bai"; return false; }
    return true;
}
