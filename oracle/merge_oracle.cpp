// merge_oracle.cpp — TEST INFRASTRUCTURE ONLY. NOT PART OF THE PRODUCT PATH.
//
// Restatement of mergeSVs / mergeDuplicateSVs (sv_object.cpp:45-269, 324-350). C++ rather than C
// because the observable result depends on libstdc++'s (unstable) std::sort and on std::map's
// iteration order, which must be the same library calls on the same input order (SURVEY §7 hard
// part 2). Cluster labels come from the caller (oracle DBSCAN, reference DBSCAN or the GPU), so
// this file restates only the bucketing + representative choice.
//
// PARITY UNPINNED by reference fixtures: sv_object.cpp includes utils.h -> <htslib/sam.h>, absent
// from this image, so it cannot be built here. Pinned by the known answer recorded in SURVEY.md
// §8a row a5 (tests/test_merge.py::test_survey_known_answer) and by line-by-line restatement.
#include <algorithm>
#include <cstdint>
#include <map>
#include <tuple>
#include <vector>

extern "C" {

// POD mirror of the SVCall fields that mergeSVs reads or forwards (sv_object.h:16-35).
// `id` = index of the call in the caller's input array, so any other field (alt allele,
// evidence flags, genotype ...) can be recovered by the caller.
struct orc_call {
    uint32_t start, end;
    int32_t  sv_type;        // sv_types.h:16-25: UNKNOWN -1, DEL 0, DUP 1, INV 2, INS 3, BND 4, NEUTRAL 5, LOH 6
    int32_t  cluster_size;
    double   hmm_likelihood;
    int64_t  id;
};

typedef void (*orc_label_fn)(const uint32_t *start, const uint32_t *end, uint64_t n, double eps,
                             int32_t min_pts, int32_t *labels);

// mergeSVs (sv_object.cpp:45-269). `dbscan` supplies DBSCAN::fit labels. Output written to `out`
// (capacity >= n); returns the merged count.
int64_t orc_merge_svs(const orc_call *calls, uint64_t n, double epsilon, int32_t min_pts, int keep_noise,
                      orc_label_fn dbscan, orc_call *out)
{
    if (n < 2) {                                         // :49-51 (vector left untouched)
        for (uint64_t i = 0; i < n; i++) out[i] = calls[i];
        return (int64_t)n;
    }
    std::vector<orc_call> merged;
    const int types[5] = {0, 1, 2, 3, 4};                // :62-68 DEL, DUP, INV, INS, BND
    for (int sv_type : types) {
        std::vector<orc_call> merged_type;
        std::vector<orc_call> type_calls;
        for (uint64_t i = 0; i < n; i++)                 // :80-82 std::copy_if
            if (calls[i].sv_type == sv_type) type_calls.push_back(calls[i]);
        if (type_calls.size() < 2) {                     // :85-92
            for (const auto &c : type_calls) merged.push_back(c);
            continue;
        }
        std::vector<uint32_t> s(type_calls.size()), e(type_calls.size());
        std::vector<int32_t> labels(type_calls.size());
        for (size_t i = 0; i < type_calls.size(); i++) { s[i] = type_calls[i].start; e[i] = type_calls[i].end; }
        dbscan(s.data(), e.data(), type_calls.size(), epsilon, min_pts, labels.data());   // :94

        std::map<int, std::vector<orc_call>> cluster_map;        // :98-101
        for (size_t i = 0; i < labels.size(); i++) cluster_map[labels[i]].push_back(type_calls[i]);

        for (auto &cluster : cluster_map) {                      // :121
            int cluster_id = cluster.first;
            std::vector<orc_call> &cc = cluster.second;
            if (cc.size() < 2) continue;                         // :126-128
            if (cluster_id < 0 && keep_noise) {                  // :131-146
                for (const auto &c : cc) merged_type.push_back(c);
            } else {
                bool has_nonzero = false;                        // :155-166
                for (const auto &c : cc) if (c.hmm_likelihood != 0.0) { has_nonzero = true; break; }
                orc_call m = cc[0];
                if (has_nonzero) {
                    std::sort(cc.begin(), cc.end(), [](const orc_call &a, const orc_call &b) {   // :172-174
                        return a.cluster_size > b.cluster_size ||
                               (a.cluster_size == b.cluster_size && a.end - a.start > b.end - b.start);
                    });
                    auto it = std::find_if(cc.begin(), cc.end(), [](const orc_call &c) { return c.hmm_likelihood != 0.0; });
                    m = *it;                                     // :180 (cluster_size kept)
                    merged_type.push_back(m);
                } else {
                    std::sort(cc.begin(), cc.end(), [](const orc_call &a, const orc_call &b) {   // :190-192
                        return (a.end - a.start) > (b.end - b.start);
                    });
                    double top_pct = 0.2;                        // :208-211
                    size_t top_pct_size = std::max(1, (int)(cc.size() * top_pct));
                    size_t median_index = top_pct_size / 2;      // :229
                    m = cc[median_index];
                    m.cluster_size = (int)cc.size();             // :242
                    merged_type.push_back(m);
                }
            }
        }
        merged.insert(merged.end(), merged_type.begin(), merged_type.end());   // :263-264
    }
    for (size_t i = 0; i < merged.size(); i++) out[i] = merged[i];
    return (int64_t)merged.size();
}

// mergeDuplicateSVs (sv_object.cpp:324-350); in/out in place, returns the new count.
int64_t orc_merge_duplicates(orc_call *calls, uint64_t n)
{
    std::vector<orc_call> v(calls, calls + n), combined;
    std::sort(v.begin(), v.end(), [](const orc_call &a, const orc_call &b) {     // :330-332
        return std::tie(a.start, a.sv_type) < std::tie(b.start, b.sv_type);
    });
    for (size_t i = 0; i < v.size(); i++) {
        orc_call &c = v[i];
        if (i > 0 && c.start == v[i - 1].start && c.end == v[i - 1].end) {       // :337
            c.cluster_size += v[i - 1].cluster_size;
            combined.back() = c;
        } else {
            combined.push_back(c);
        }
    }
    for (size_t i = 0; i < combined.size(); i++) calls[i] = combined[i];
    return (int64_t)combined.size();
}

}  // extern "C"
