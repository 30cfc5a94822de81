#include "snp_io.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <future>
#include <limits>
#include <sstream>

#include "log.h"

namespace {
constexpr double kMinPfb = 0.01, kMaxPfb = 0.99;             // cnv_caller.cpp:33-34
constexpr int32_t kIntMissing = std::numeric_limits<int32_t>::min();        // bcf_int32_missing
constexpr int32_t kIntVectorEnd = std::numeric_limits<int32_t>::min() + 1;  // bcf_int32_vector_end

using sv = std::string_view;

// tab-separated field k of a line, or false
struct Fields {
    sv f[10];
    sv samples;          // everything after FORMAT
    int n = 0;
};
bool split_fields(sv line, Fields &out)
{
    out.n = 0;
    size_t a = 0;
    while (out.n < 9) {
        const size_t t = line.find('\t', a);
        if (t == sv::npos) { out.f[out.n++] = line.substr(a); out.samples = sv(); return out.n >= 8; }
        out.f[out.n++] = line.substr(a, t - a);
        a = t + 1;
    }
    out.samples = line.substr(a);
    return true;
}

// bcf_is_snp: every allele (REF and ALTs) is one base other than '*', or the symbolic <X> / <*>
bool is_snp(sv ref, sv alt)
{
    auto one = [](sv a) { return (a.size() == 1 && a[0] != '*') || a == "<X>" || a == "<*>"; };
    if (!one(ref)) return false;
    if (alt == ".") return true;                          // no ALT allele: only REF is tested
    size_t a = 0;
    for (;;) {
        const size_t c = alt.find(',', a);
        if (!one(alt.substr(a, c == sv::npos ? sv::npos : c - a))) return false;
        if (c == sv::npos) return true;
        a = c + 1;
    }
}

// index of `key` among the ':'-separated FORMAT keys, or -1
int format_index(sv format, sv key)
{
    int i = 0;
    size_t a = 0;
    for (;;) {
        const size_t c = format.find(':', a);
        if (format.substr(a, c == sv::npos ? sv::npos : c - a) == key) return i;
        if (c == sv::npos) return -1;
        a = c + 1;
        i++;
    }
}

// field `idx` of one sample column ("" when the sample has fewer fields: trailing fields may be dropped)
sv sample_field(sv sample, int idx)
{
    size_t a = 0;
    for (int i = 0; i < idx; i++) {
        const size_t c = sample.find(':', a);
        if (c == sv::npos) return sv();
        a = c + 1;
    }
    const size_t c = sample.find(':', a);
    return sample.substr(a, c == sv::npos ? sv::npos : c - a);
}

// integer vector of one FORMAT field value: "." / "" -> one missing value
void parse_ints(sv v, std::vector<int32_t> &out)
{
    out.clear();
    if (v.empty() || v == ".") { out.push_back(kIntMissing); return; }
    size_t a = 0;
    for (;;) {
        const size_t c = v.find(',', a);
        sv t = v.substr(a, c == sv::npos ? sv::npos : c - a);
        if (t.empty() || t == ".") out.push_back(kIntMissing);
        else out.push_back((int32_t)strtol(std::string(t).c_str(), nullptr, 10));
        if (c == sv::npos) return;
        a = c + 1;
    }
}

// what bcf_get_format_int32 hands back for `key`: sample-major, every sample padded to the longest with vector-end marks
bool format_ints(sv format, sv samples, sv key, std::vector<int32_t> &flat)
{
    const int idx = format_index(format, key);
    if (idx < 0) return false;
    std::vector<std::vector<int32_t>> per;
    size_t a = 0;
    for (;;) {
        const size_t t = samples.find('\t', a);
        per.emplace_back();
        parse_ints(sample_field(samples.substr(a, t == sv::npos ? sv::npos : t - a), idx), per.back());
        if (t == sv::npos) break;
        a = t + 1;
    }
    size_t width = 0;
    for (const auto &p : per) width = std::max(width, p.size());
    flat.clear();
    for (const auto &p : per) {
        flat.insert(flat.end(), p.begin(), p.end());
        flat.insert(flat.end(), width - p.size(), kIntVectorEnd);
    }
    return true;
}

// "##FORMAT=<ID=DP,Number=1,Type=Integer,...>" -> (kind, id, type)
bool header_decl(sv line, sv kind, std::string &id, std::string &type)
{
    const std::string prefix = "##" + std::string(kind) + "=<";
    if (line.substr(0, prefix.size()) != prefix) return false;
    auto grab = [&](sv key) -> std::string {
        const std::string k = std::string(key) + "=";
        size_t p = line.find(k, prefix.size() - 1);
        while (p != sv::npos && line[p - 1] != '<' && line[p - 1] != ',') p = line.find(k, p + 1);
        if (p == sv::npos) return "";
        const size_t e = line.find_first_of(",>", p + k.size());
        return std::string(line.substr(p + k.size(), e == sv::npos ? sv::npos : e - p - k.size()));
    };
    id = grab("ID");
    type = grab("Type");
    return !id.empty();
}

bool file_exists(const std::string &p) { std::ifstream f(p); return (bool)f; }

// iterate the lines of a chunk
template <class F>
bool for_lines(sv chunk, F fn)
{
    size_t a = 0;
    while (a < chunk.size()) {
        size_t e = chunk.find('\n', a);
        if (e == sv::npos) e = chunk.size();
        sv line = chunk.substr(a, e - a);
        if (!line.empty() && line.back() == '\r') line.remove_suffix(1);
        if (!line.empty() && !fn(line)) return false;
        a = e + 1;
    }
    return true;
}
}  // namespace

// ---- TextFile -----------------------------------------------------------------------------------------------
bool TextFile::open(const std::string &path, std::string *err)
{
    if (!file.open(path, err)) return false;
    const uint8_t *p = file.data();
    is_bgzf = false;
    if (file.size() >= 2 && p[0] == 0x1f && p[1] == 0x8b) {
        bgzf::Block b;
        if (!bgzf::parse_block(p, file.size(), 0, b, err)) { if (err) *err = path + " is gzip but not BGZF (use bgzip): " + *err; return false; }
        is_bgzf = true;
    }
    return true;
}

bool TextFile::forEachChunk(int threads, const std::function<bool(std::string_view)> &on_lines, std::string *err)
{
    if (!is_bgzf) {                                       // plain text: one chunk
        if (file.size()) on_lines(sv((const char *)file.data(), file.size()));
        return true;
    }
    struct Batch { std::vector<bgzf::Block> blocks; HugeVec<uint8_t> data; bool ok = true; std::string err; uint64_t next = 0; };
    const size_t kHead = 1 << 20, W = 1024;
    auto load = [&](uint64_t coff, Batch &b) {
        b.blocks.clear(); b.ok = true;
        uint64_t total = 0;
        while (coff < file.size() && b.blocks.size() < W) {
            bgzf::Block blk;
            if (!bgzf::parse_block(file.data() + coff, file.size() - coff, coff, blk, &b.err)) { b.ok = false; return; }
            b.blocks.push_back(blk);
            total += blk.isize;
            coff += blk.csize;
        }
        b.next = coff;
        b.data.resize_uninit(kHead + total);
        if (!bgzf::inflate_range(file.data(), b.blocks, 0, b.blocks.size(), b.data.data() + kHead, threads, &b.err)) b.ok = false;
    };
    Batch batch[2];
    load(0, batch[0]);
    int cur = 0;
    std::string carry;                                    // the unterminated tail of the previous batch
    HugeVec<uint8_t> joined;
    for (;;) {
        Batch &b = batch[cur];
        if (!b.ok) { if (err) *err = b.err; return false; }
        if (b.blocks.empty()) break;
        std::future<void> ahead;
        const bool more = b.next < file.size();
        if (more) ahead = std::async(std::launch::async, load, b.next, std::ref(batch[cur ^ 1]));
        char *p = (char *)b.data.data() + kHead;
        size_t n = b.data.size() - kHead;
        if (!carry.empty()) {
            if (carry.size() <= kHead) { p -= carry.size(); memcpy(p, carry.data(), carry.size()); n += carry.size(); }
            else { joined.clear(); joined.append((const uint8_t *)carry.data(), carry.size()); joined.append((const uint8_t *)p, n); p = (char *)joined.data(); n = joined.size(); }
            carry.clear();
        }
        size_t whole = n;
        if (more) {                                       // keep the last partial line for the next batch
            while (whole > 0 && p[whole - 1] != '\n') whole--;
            carry.assign(p + whole, n - whole);
        }
        const bool go = whole == 0 || on_lines(sv(p, whole));
        if (more) ahead.get();
        if (!go || !more) break;
        cur ^= 1;
    }
    return true;
}

// ---- --pfb table ----------------------------------------------------------------------------------------------
bool AlleleFreqFiles::load(const std::string &table_path, std::string *err)
{
    if (table_path.empty()) return true;
    std::ifstream in(table_path);
    if (!in.is_open()) { if (err) *err = "Population allele frequency file does not exist: " + table_path; return false; }
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line[0] == '#') continue;
        line = line.substr(0, 255);                       // the reference reads through a 256-byte fgets buffer (:248-250)
        std::vector<std::string> parts;
        std::istringstream ss(line);
        std::string tok;
        while (std::getline(ss, tok, '=')) parts.push_back(tok);
        if (parts.size() != 2) continue;
        const std::string vcf = parts[1].substr(0, parts[1].find_first_of("\r\n"));
        if (!file_exists(vcf)) { if (err) *err = "Error: Allele frequency file does not exist: " + vcf; return false; }
        paths[parts[0]] = vcf;
    }
    return true;
}

std::string AlleleFreqFiles::get(std::string chr) const
{
    if (chr.find("chr") != std::string::npos) chr = chr.substr(3, chr.size() - 3);     // wherever "chr" was found (:297-300)
    auto it = paths.find(chr);
    return it == paths.end() ? "" : it->second;
}

std::string gnomadContigName(const std::string &chr, const std::string &pfb_filepath)
{
    std::string g = chr;
    if (pfb_filepath.find("chr") == std::string::npos) {
        if (g.find("chr") != std::string::npos) g = g.substr(3);
    } else if (g.find("chr") == std::string::npos) {
        g = "chr" + chr;
    }
    return g;
}

// ---- tables -----------------------------------------------------------------------------------------------------
void SNPFileTable::query(uint32_t start_pos, uint32_t end_pos, std::vector<uint32_t> &snp_pos, std::unordered_map<uint32_t, double> &snp_baf,
                         std::unordered_map<uint32_t, double> &snp_pfb) const
{
    // pos is ascending for an indexed file; tolerate local disorder by scanning from the first position >= start
    const size_t a = std::lower_bound(pos.begin(), pos.end(), start_pos) - pos.begin();
    uint32_t lo = UINT32_MAX, hi = 0;
    bool any = false;
    for (size_t i = a; i < pos.size() && pos[i] <= end_pos; i++) {
        snp_pos.push_back(pos[i]);
        snp_baf[pos[i]] = baf[i];
        lo = std::min(lo, pos[i]); hi = std::max(hi, pos[i]);
        any = true;
    }
    if (!any) return;                                     // :734-740
    const size_t g = std::lower_bound(af_pos.begin(), af_pos.end(), lo) - af_pos.begin();
    if (g < af_pos.size() && af_pos[g] <= hi) snp_pfb[af_pos[g]] = af[g];     // first acceptable record, then the reference breaks (:800-801)
}

bool SNPFile::load(const std::string &snp_vcf, int threads, std::string *err)
{
    if (snp_vcf.empty()) { if (err) *err = "ERROR: SNP file path is empty."; return false; }
    TextFile tf;
    if (!tf.open(snp_vcf, err)) return false;
    if (tf.compressed() && !file_exists(snp_vcf + ".tbi") && !file_exists(snp_vcf + ".csi")) {
        if (err) *err = "ERROR: Could not add SNP file to reader: " + snp_vcf + " (no .tbi / .csi index)";
        return false;
    }
    bool dp_int = false, ad_int = false;
    std::vector<int32_t> dp, ad;
    std::string last_chr;
    SNPFileTable *cur = nullptr;
    const bool ok = tf.forEachChunk(threads, [&](sv chunk) {
        return for_lines(chunk, [&](sv line) {
            if (line[0] == '#') {
                std::string id, type;
                if (header_decl(line, "FORMAT", id, type)) {
                    if (id == "DP") dp_int = type == "Integer";
                    if (id == "AD") ad_int = type == "Integer";
                }
                return true;
            }
            Fields f;
            if (!split_fields(line, f) || f.n < 9) return true;                 // no FORMAT / sample columns: DP lookup fails (:696-701)
            if (!is_snp(f.f[3], f.f[4])) return true;                            // :679-683
            if (f.f[5] == ".") return true;                                      // QUAL missing (:686)
            const float qual = (float)strtod(std::string(f.f[5]).c_str(), nullptr);
            if (qual <= 30) return true;
            if (!dp_int || !format_ints(f.f[8], f.samples, "DP", dp) || dp[0] <= 10) return true;    // :693-701 (missing = INT32_MIN)
            if (f.f[6] != "." && f.f[6] != "PASS") {                             // bcf_has_filter(.., "PASS") (:704-707)
                bool pass = false;
                size_t a = 0;
                for (;;) {
                    const size_t c = f.f[6].find(';', a);
                    if (f.f[6].substr(a, c == sv::npos ? sv::npos : c - a) == "PASS") pass = true;
                    if (c == sv::npos) break;
                    a = c + 1;
                }
                if (!pass) return true;
            }
            if (!ad_int || !format_ints(f.f[8], f.samples, "AD", ad) || ad.size() < 2) return true;  // :710-717
            const double b = (double)ad[1] / (double)(int32_t)((uint32_t)ad[0] + (uint32_t)ad[1]);   // :720
            if (cur == nullptr || f.f[0] != last_chr) {
                last_chr.assign(f.f[0]);
                cur = &tables[last_chr];
            }
            cur->pos.push_back((uint32_t)strtol(std::string(f.f[1]).c_str(), nullptr, 10));
            cur->baf.push_back(b);
            n_kept++;
            return true;
        });
    }, err);
    return ok;
}

const SNPFileTable &SNPFile::table(const std::string &chr, const std::string &pfb_vcf, const std::string &ethnicity, int threads)
{
    auto it = tables.find(chr);
    if (it == tables.end()) return none;
    SNPFileTable &t = it->second;
    if (af_done[chr]) return t;
    af_done[chr] = true;
    if (pfb_vcf.empty() || !file_exists(pfb_vcf) || t.pos.empty()) return t;        // use_pfb = false (:596-611)
    TextFile tf;
    std::string err;
    if (!tf.open(pfb_vcf, &err) || (tf.compressed() && !file_exists(pfb_vcf + ".tbi") && !file_exists(pfb_vcf + ".csi"))) {
        printError("ERROR: Could not add population allele frequency file to reader: " + pfb_vcf);
        return t;
    }
    const std::string key = ethnicity.empty() ? std::string("AF") : "AF_" + ethnicity;       // :616-623
    const std::string contig = gnomadContigName(chr, pfb_vcf);
    std::vector<uint32_t> sorted = t.pos;
    std::sort(sorted.begin(), sorted.end());
    const uint32_t last_pos = sorted.back();
    bool key_is_float = false;
    const std::string needle = key + "=";
    tf.forEachChunk(threads, [&](sv chunk) {
        return for_lines(chunk, [&](sv line) {
            if (line[0] == '#') {
                std::string id, type;
                if (header_decl(line, "INFO", id, type) && id == key) key_is_float = type == "Float";
                return true;
            }
            if (!key_is_float) return false;                                     // bcf_get_info_float fails on every record (:789-793)
            Fields f;
            if (!split_fields(line, f)) return true;
            if (f.f[0] != contig) return true;
            const uint32_t p = (uint32_t)strtol(std::string(f.f[1]).c_str(), nullptr, 10);
            if (p > last_pos) return false;                                      // sorted file: nothing further can match
            if (!std::binary_search(sorted.begin(), sorted.end(), p)) return true;   // :782-786
            if (!is_snp(f.f[3], f.f[4])) return true;                            // :775-779
            // INFO key at the start of the field or behind ';'
            sv info = f.f[7];
            size_t k = info.find(needle);
            while (k != sv::npos && k != 0 && info[k - 1] != ';') k = info.find(needle, k + 1);
            if (k == sv::npos) return true;                                      // status < 0
            const size_t e = info.find_first_of(";,", k + needle.size());
            const sv val = info.substr(k + needle.size(), e == sv::npos ? sv::npos : e - k - needle.size());
            if (val.empty()) return true;                                        // count == 0
            const double v = val == "." ? std::nan("") : (double)(float)strtod(std::string(val).c_str(), nullptr);
            if (v <= kMinPfb || v >= kMaxPfb) return true;                       // :795-799 (NaN passes, as there)
            t.af_pos.push_back(p);
            t.af.push_back(v);
            return true;
        });
    }, &err);
    return t;
}
