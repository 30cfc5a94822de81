// snp_oracle.cpp — TEST INFRASTRUCTURE ONLY. NOT PART OF THE PRODUCT PATH.
//
// Restatement of CNVCaller::readSNPAlleleFrequencies (src/cnv_caller.cpp:558-809) for ONE region, reading plain-text VCFs
// from top to bottom the way the reference's synced reader walks a region: per call, no tables, no caching. The htslib
// accessors it relies on are restated from their documented behaviour (bcf_is_snp, bcf_has_filter(PASS),
// bcf_get_format_int32's padded sample-major layout and missing / vector-end marks, bcf_get_info_float, QUAL stored as float).
//
// PARITY UNPINNED by a reference build (htslib absent) and by reference fixtures (its SNP test files are not in the repository).
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <fstream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

namespace {

std::vector<std::string> split(const std::string &s, char d)
{
    std::vector<std::string> out;
    std::string cur;
    for (char c : s) { if (c == d) { out.push_back(cur); cur.clear(); } else cur += c; }
    out.push_back(cur);
    return out;
}

bool allele_is_base(const std::string &a) { return (a.size() == 1 && a[0] != '*') || a == "<X>" || a == "<*>"; }

bool record_is_snp(const std::string &ref, const std::string &alt)
{
    std::vector<std::string> alleles{ref};
    if (alt != ".") for (const auto &a : split(alt, ',')) alleles.push_back(a);
    for (const auto &a : alleles) if (!allele_is_base(a)) return false;
    return true;
}

// header: is `id` declared under `kind` with type `type`?
bool declared(const std::vector<std::string> &header, const std::string &kind, const std::string &id, const std::string &type)
{
    bool ok = false;
    for (const auto &h : header) {
        if (h.rfind("##" + kind + "=<", 0) != 0) continue;
        const std::string body = h.substr(h.find('<') + 1, h.rfind('>') - h.find('<') - 1);
        std::string hid, htype;
        for (const auto &kv : split(body, ',')) {
            if (kv.rfind("ID=", 0) == 0) hid = kv.substr(3);
            if (kv.rfind("Type=", 0) == 0) htype = kv.substr(5);
        }
        if (hid == id) ok = htype == type;
    }
    return ok;
}

// the array bcf_get_format_int32 fills for `key`, or false when the record has no such FORMAT field
bool format_array(const std::vector<std::string> &cols, const std::string &key, std::vector<int> &out)
{
    if (cols.size() < 9) return false;
    const std::vector<std::string> keys = split(cols[8], ':');
    int idx = -1;
    for (size_t i = 0; i < keys.size(); i++) if (keys[i] == key) { idx = (int)i; break; }
    if (idx < 0) return false;
    std::vector<std::vector<int>> per;
    size_t width = 0;
    const size_t n_samples = cols.size() > 9 ? cols.size() - 9 : 1;
    for (size_t s = 0; s < n_samples; s++) {
        const std::string col = cols.size() > 9 ? cols[9 + s] : "";
        const std::vector<std::string> vals = split(col, ':');
        std::vector<int> v;
        const std::string f = (size_t)idx < vals.size() ? vals[(size_t)idx] : "";
        if (f.empty() || f == ".") v.push_back(INT_MIN);
        else for (const auto &t : split(f, ',')) v.push_back((t.empty() || t == ".") ? INT_MIN : (int)strtol(t.c_str(), nullptr, 10));
        width = std::max(width, v.size());
        per.push_back(v);
    }
    out.clear();
    for (auto &v : per) { while (v.size() < width) v.push_back(INT_MIN + 1); out.insert(out.end(), v.begin(), v.end()); }
    return true;
}

}  // namespace

extern "C" {

// Returns the number of positions (file order, duplicates kept) or a negative code. baf_out[i] is the map value at pos_out[i]
// after the whole region was read (a later duplicate overwrites). At most one (pfb_pos, pfb_val): *has_pfb.
int64_t orc_read_snp_af(const char *snp_txt, const char *pfb_txt, const char *chr, const char *chr_gnomad, uint32_t start_pos, uint32_t end_pos,
                        const char *af_key, uint32_t *pos_out, double *baf_out, uint64_t cap, int *has_pfb, uint32_t *pfb_pos, double *pfb_val)
{
    *has_pfb = 0;
    std::ifstream in(snp_txt);
    if (!in.is_open()) return -1;
    std::vector<std::string> header;
    std::vector<uint32_t> snp_pos;
    std::map<uint32_t, double> snp_baf;
    std::string line;
    bool dp_ok = false, ad_ok = false, decided = false;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '#') { header.push_back(line); continue; }
        if (!decided) { dp_ok = declared(header, "FORMAT", "DP", "Integer"); ad_ok = declared(header, "FORMAT", "AD", "Integer"); decided = true; }
        const std::vector<std::string> c = split(line, '\t');
        if (c.size() < 8 || c[0] != chr) continue;
        const uint32_t pos = (uint32_t)strtol(c[1].c_str(), nullptr, 10);
        if (pos < start_pos || pos > end_pos) {                       // region chr:start-end; only single-base records can pass is_snp anyway
            continue;
        }
        if (!record_is_snp(c[3], c[4])) continue;                     // :679-683
        if (c[5] == ".") continue;                                    // :686-689
        const float qual = (float)strtod(c[5].c_str(), nullptr);
        if (qual <= 30) continue;
        std::vector<int> dp, ad;
        if (!dp_ok || !format_array(c, "DP", dp) || dp[0] <= 10) continue;          // :693-701
        bool pass = c[6] == "." || c[6] == "PASS";                    // :704-707
        if (!pass) for (const auto &f : split(c[6], ';')) if (f == "PASS") pass = true;
        if (!pass) continue;
        if (!ad_ok || !format_array(c, "AD", ad) || ad.size() < 2) continue;        // :710-717
        const double baf = (double)ad[1] / (double)(int)((unsigned)ad[0] + (unsigned)ad[1]);   // :720
        snp_pos.push_back(pos);
        snp_baf[pos] = baf;
    }
    if (snp_pos.empty()) return 0;                                     // :734-740
    if (pfb_txt && *pfb_txt) {
        std::ifstream pf(pfb_txt);
        if (pf.is_open()) {
            const uint32_t lo = *std::min_element(snp_pos.begin(), snp_pos.end()), hi = *std::max_element(snp_pos.begin(), snp_pos.end());
            const std::set<uint32_t> pos_set(snp_pos.begin(), snp_pos.end());
            std::vector<std::string> ph;
            bool key_ok = false, key_decided = false;
            while (std::getline(pf, line)) {
                if (!line.empty() && line.back() == '\r') line.pop_back();
                if (line.empty()) continue;
                if (line[0] == '#') { ph.push_back(line); continue; }
                if (!key_decided) { key_ok = declared(ph, "INFO", af_key, "Float"); key_decided = true; }
                const std::vector<std::string> c = split(line, '\t');
                if (c.size() < 8 || c[0] != chr_gnomad) continue;
                const uint32_t p = (uint32_t)strtol(c[1].c_str(), nullptr, 10);
                if (p < lo || p > hi) continue;                        // region chr_gnomad:min-max (:749-754)
                if (!record_is_snp(c[3], c[4])) continue;              // :775-779
                if (!pos_set.count(p)) continue;                       // :782-786
                if (!key_ok) continue;                                 // bcf_get_info_float < 0 (:789-793)
                std::string val;
                bool found = false;
                for (const auto &kv : split(c[7], ';')) {
                    const size_t eq = kv.find('=');
                    if (eq != std::string::npos && kv.substr(0, eq) == af_key) { val = kv.substr(eq + 1); found = true; break; }
                }
                if (!found || val.empty()) continue;
                const std::string first = split(val, ',')[0];
                const double pfb = first == "." ? std::nan("") : (double)(float)strtod(first.c_str(), nullptr);
                if (pfb <= 0.01 || pfb >= 0.99) continue;              // :795-799
                *has_pfb = 1; *pfb_pos = p; *pfb_val = pfb;            // :800-801
                break;
            }
        }
    }
    if (snp_pos.size() > cap) return -2;
    for (size_t i = 0; i < snp_pos.size(); i++) { pos_out[i] = snp_pos[i]; baf_out[i] = snp_baf[snp_pos[i]]; }
    return (int64_t)snp_pos.size();
}

}  // extern "C"
