// cnv_oracle.cpp — TEST INFRASTRUCTURE ONLY. NOT PART OF THE PRODUCT PATH.
//
// Restatement of the copy-number pass around the HMM: CNVCaller::querySNPRegion (cnv_caller.cpp:53-164),
// runCIGARCopyNumberPrediction (:290-387), runCopyNumberPrediction (:166-287, without the --save-cnv JSON side output) and
// SVCaller::runSplitReadCopyNumberPredictions (sv_caller.cpp:983-1064), on a HOST depth array, with the window sums and the
// Viterbi of csv_oracle.c. C++ because the observation order is the iteration order of a libstdc++
// std::unordered_map<std::string,double> (cnv_caller.cpp:77,124).
// PARITY UNPINNED by reference fixtures (cnv_caller.cpp / sv_caller.cpp need htslib): line-by-line restatement only.
#include <algorithm>
#include <cstdint>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

extern "C" {
struct orc_hmm { double A[36], pi[6], B1_mean[6], B1_sd[6], B1_uf, B2_mean[5], B2_sd[5], B2_uf; };
void orc_window_log2(const uint32_t *depth, uint32_t depth_len, uint32_t start_pos, uint32_t end_pos, int32_t sample_size, double mean_chr_cov,
                     double *log2_out, uint32_t *win_start, uint32_t *win_end);
void orc_viterbi(const orc_hmm *hmm, int32_t T, const double *O1, const double *O2, const double *pfb, int32_t *states, double *loglik);
}

namespace {

struct Obs { std::vector<uint32_t> pos; std::vector<double> baf, pfb, log2; std::vector<uint8_t> is_snp; };
struct Snps { const uint32_t *pos; const double *baf, *pfb; const uint8_t *has; uint64_t n; };
struct Call { uint32_t start, end; int32_t sv_type, cluster_size; double hmm_likelihood; int64_t id; uint32_t aln_flags; int32_t genotype, cn_state, aln_offset; };

enum { UNKNOWN = -1, DEL = 0, DUP = 1, INV = 2, INS = 3, BND = 4, NEUTRAL = 5, LOH = 6 };
int type_from_state(int s) { static const int m[7] = {UNKNOWN, DEL, DEL, NEUTRAL, LOH, DUP, DUP}; return m[s]; }          // sv_types.h:96-104
int genotype_from_state(int s) { static const int m[7] = {3, 2, 1, 0, 2, 1, 2}; return m[s]; }                              // cnv_caller.h:76-84
bool valid_update(int t, int u) { if (u == UNKNOWN) return false; if (t == DEL && u != DEL) return false; if (t == INS && u != DUP) return false; return true; }

void query_snp_region(const uint32_t *depth, uint32_t depth_len, uint32_t start_pos, uint32_t end_pos, double mean_cov, int sample_size,
                      const Snps &s, Obs &o)
{
    std::vector<uint32_t> snp_pos;
    std::unordered_map<uint32_t, double> snp_baf_map, snp_pfb_map;
    for (uint64_t i = 0; i < s.n; i++)                       // the region's SNPs in file order (readSNPAlleleFrequencies' result)
        if (s.pos[i] >= start_pos && s.pos[i] <= end_pos) { snp_pos.push_back(s.pos[i]); snp_baf_map[s.pos[i]] = s.baf[i]; if (s.has[i]) snp_pfb_map[s.pos[i]] = s.pfb[i]; }
    sample_size = std::max((int)snp_pos.size(), sample_size);                               // :65
    if (start_pos > end_pos) return;                                                        // :69-73
    std::vector<double> l2(sample_size); std::vector<uint32_t> ws(sample_size), we(sample_size);
    orc_window_log2(depth, depth_len, start_pos, end_pos, sample_size, mean_cov, l2.data(), ws.data(), we.data());   // :76-108
    std::unordered_map<std::string, double> window_log2_map;
    for (int i = 0; i < sample_size; i++) window_log2_map[std::to_string(ws[i]) + "-" + std::to_string(we[i])] = l2[i];   // :111-112
    for (const auto &window : window_log2_map) {                                            // :124
        uint32_t window_start = std::stoi(window.first.substr(0, window.first.find('-')));
        uint32_t window_end = std::stoi(window.first.substr(window.first.find('-') + 1));
        double log2_cov = window.second;
        bool snp_found = false;
        for (uint32_t pos : snp_pos) {
            if (pos >= window_start && pos <= window_end) {
                o.pos.push_back(pos); o.baf.push_back(snp_baf_map[pos]); o.pfb.push_back(snp_pfb_map[pos]); o.log2.push_back(log2_cov); o.is_snp.push_back(1);
                snp_found = true;
            }
        }
        if (!snp_found) {
            uint32_t window_center = (window_start + window_end) / 2;
            o.pos.push_back(window_center); o.baf.push_back(-1.0); o.pfb.push_back(0.5); o.log2.push_back(log2_cov); o.is_snp.push_back(0);
        }
    }
}

std::tuple<double, int, int, int> run_cn_prediction(const uint32_t *depth, uint32_t depth_len, const orc_hmm *hmm, uint32_t start_pos, uint32_t end_pos,
                                                    double mean_cov, int sample_size, const Snps &s)
{
    if (start_pos > end_pos) return std::make_tuple(0.0, (int)UNKNOWN, 3, 0);
    Obs o;
    query_snp_region(depth, depth_len, start_pos, end_pos, mean_cov, sample_size, s, o);
    const int T = (int)o.pos.size();
    std::vector<int32_t> seq(T > 0 ? T : 1); double lh = 0.0;
    orc_viterbi(hmm, T, o.log2.data(), o.baf.data(), o.pfb.data(), seq.data(), &lh);
    if (T == 0) return std::make_tuple(0.0, (int)UNKNOWN, 3, 0);
    double pct[7] = {0}; double largest = 0.0; int non_neutral = 0;
    for (int i = 0; i < 6; i++) {
        pct[i + 1] = (double)std::count(seq.begin(), seq.begin() + T, i + 1) / (double)T;
        if (i + 1 != 3 && pct[i + 1] > largest) { largest = pct[i + 1]; non_neutral = i + 1; }
    }
    int max_state = 0;
    if (largest > 0.3) max_state = non_neutral; else if (pct[3] > 0.3) max_state = 3;
    return std::make_tuple(lh, type_from_state(max_state), genotype_from_state(max_state), max_state);
}

}  // namespace

extern "C" {

int64_t orc_query_snp_region(const uint32_t *depth, uint32_t depth_len, uint32_t start, uint32_t end, double mean_cov, int sample_size,
                             const uint32_t *snp_pos, const double *snp_baf, const double *snp_pfb, const uint8_t *snp_has, uint64_t n_snp,
                             uint32_t *pos_out, double *baf_out, double *pfb_out, double *log2_out, uint8_t *is_snp_out, uint64_t cap)
{
    Obs o; Snps s{snp_pos, snp_baf, snp_pfb, snp_has, n_snp};
    query_snp_region(depth, depth_len, start, end, mean_cov, sample_size, s, o);
    for (size_t i = 0; i < o.pos.size() && i < cap; i++) { pos_out[i] = o.pos[i]; baf_out[i] = o.baf[i]; pfb_out[i] = o.pfb[i]; log2_out[i] = o.log2[i]; is_snp_out[i] = o.is_snp[i]; }
    return (int64_t)o.pos.size();
}

// runCIGARCopyNumberPrediction (cnv_caller.cpp:290-387), in place
void orc_cigar_cn_prediction(const uint32_t *depth, uint32_t depth_len, Call *calls, uint64_t n, const orc_hmm *hmm, double mean_cov, int sample_size,
                             uint32_t min_cnv, const uint32_t *snp_pos, const double *snp_baf, const double *snp_pfb, const uint8_t *snp_has, uint64_t n_snp)
{
    Snps s{snp_pos, snp_baf, snp_pfb, snp_has, n_snp};
    for (uint64_t k = 0; k < n; k++) {
        Call &c = calls[k];
        if (c.start > c.end) continue;
        if ((c.end - c.start) < min_cnv) continue;
        Obs o;
        query_snp_region(depth, depth_len, c.start, c.end, mean_cov, sample_size, s, o);
        if (o.pos.empty()) continue;
        const int T = (int)o.pos.size();
        std::vector<int32_t> seq(T); double lh = 0.0;
        orc_viterbi(hmm, T, o.log2.data(), o.baf.data(), o.pfb.data(), seq.data(), &lh);
        std::vector<int> sv_states;
        for (int i = 0; i < T; i++) if (o.pos[i] >= c.start && o.pos[i] <= c.end) sv_states.push_back(seq[i]);
        int max_state = 0, max_count = 0;
        for (int i = 0; i < 6; i++) { int cnt = (int)std::count(sv_states.begin(), sv_states.end(), i + 1); if (cnt > max_count) { max_state = i + 1; max_count = cnt; } }
        if ((double)max_count / (double)(int)sv_states.size() < 0.50) max_state = 0;
        int updated = type_from_state(max_state);
        updated = (updated == LOH) ? c.sv_type : updated;
        if (valid_update(c.sv_type, updated)) { c.sv_type = updated; c.aln_flags |= 1u << 8; c.hmm_likelihood = lh; c.genotype = genotype_from_state(max_state); c.cn_state = max_state; }
    }
}

// runSplitReadCopyNumberPredictions (sv_caller.cpp:983-1064); calls has capacity cap; returns the new count
int64_t orc_split_cn_prediction(const uint32_t *depth, uint32_t depth_len, Call *calls, uint64_t n, uint64_t cap, const orc_hmm *hmm, double mean_cov,
                                int sample_size, const uint32_t *snp_pos, const double *snp_baf, const double *snp_pfb, const uint8_t *snp_has, uint64_t n_snp)
{
    Snps s{snp_pos, snp_baf, snp_pfb, snp_has, n_snp};
    std::vector<Call> v(calls, calls + n), additional;
    auto take = [](Call &c, double lh, int g, int cn) { c.aln_flags |= 1u << 8; c.hmm_likelihood = lh; c.genotype = g; c.cn_state = cn; };
    for (Call &c : v) {
        double lh; int t, g, cn;
        std::tie(lh, t, g, cn) = run_cn_prediction(depth, depth_len, hmm, c.start, c.end, mean_cov, sample_size, s);
        if (t == UNKNOWN) continue;
        if (c.sv_type == UNKNOWN && (t == DEL || t == DUP)) { c.sv_type = t; c.id = -1; take(c, lh, g, cn); }
        else if (c.sv_type != UNKNOWN && (t == c.sv_type || t == LOH || t == NEUTRAL)) take(c, lh, g, cn);
        else if (c.sv_type != UNKNOWN && (t != c.sv_type && (t == DEL || t == DUP))) {
            if (c.sv_type == INV) take(c, lh, g, cn);
            else if (c.sv_type == INS && t == DUP) { c.sv_type = t; c.id = -1; take(c, lh, g, cn); }
            else { Call x = c; x.sv_type = t; x.id = -1; take(x, lh, g, cn); additional.push_back(x); }
        }
    }
    for (Call &x : additional) {
        bool found = false;
        for (Call &e : v) if (e.start == x.start && e.end == x.end && e.sv_type == x.sv_type) { e = x; found = true; break; }
        if (!found) {   // addSVCall: lower_bound on (start,end), reject start > end
            if (x.start > x.end) continue;
            auto it = std::lower_bound(v.begin(), v.end(), x, [](const Call &a, const Call &b) { return a.start < b.start || (a.start == b.start && a.end < b.end); });
            v.insert(it, x);
        }
    }
    for (size_t i = 0; i < v.size() && i < cap; i++) calls[i] = v[i];
    return (int64_t)v.size();
}

}  // extern "C"
