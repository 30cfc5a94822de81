// split_oracle.cpp — TEST INFRASTRUCTURE ONLY. NOT PART OF THE PRODUCT PATH.
//
// Literal restatement of SVCaller::findSplitSVSignatures (sv_caller.cpp:68-504) from the point where the BAM records have
// been read: same containers (their iteration order is observable), the recursive interval tree (:948-980), six
// sequential DBSCAN1D fits per group with the oracle's own DBSCAN1D (csv_oracle.c), the early `continue`s, the medians.
// Alignment intervals (getAlignmentReadPositions / bam_endpos) are inputs here; they have their own oracle (orc_aln_intervals).
// PARITY UNPINNED by reference fixtures (sv_caller.cpp needs htslib): line-by-line restatement only.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <vector>

extern "C" {
void orc_dbscan_1d(const int32_t *pts, uint64_t n, double eps, int32_t min_pts, int32_t *cl);
int64_t orc_largest_cluster(const int32_t *pts, const int32_t *cl, uint64_t n, int32_t *out);
}

namespace {

struct PrimaryAlignment { int start, end, query_start, query_end; bool strand; int cluster_size; };
struct SuppAlignment { int tid, start, end, query_start, query_end; bool strand; };
struct IntervalNode {
    PrimaryAlignment region; std::string qname; int max_end;
    std::unique_ptr<IntervalNode> left, right;
    IntervalNode(PrimaryAlignment r, std::string name) : region(r), qname(name), max_end(r.end), left(nullptr), right(nullptr) {}
};
void findOverlaps(const std::unique_ptr<IntervalNode> &root, const PrimaryAlignment &query, std::vector<std::string> &result)
{
    if (!root) return;
    if (query.start <= root->region.end && query.end >= root->region.start) result.push_back(root->qname);
    if (root->left && root->left->max_end >= query.start) findOverlaps(root->left, query, result);
    findOverlaps(root->right, query, result);
}
void insert(std::unique_ptr<IntervalNode> &root, const PrimaryAlignment &region, std::string qname)
{
    if (!root) { root = std::make_unique<IntervalNode>(region, qname); return; }
    if (region.start < root->region.start) insert(root->left, region, qname); else insert(root->right, region, qname);
    root->max_end = std::max(root->max_end, region.end);
}

struct Call {   // fields of the SVCall this function produces
    uint32_t start, end; int32_t sv_type, cluster_size; int32_t aln_offset; uint32_t aln_flags; int32_t tid;
};

std::vector<int> fit_largest(const std::vector<int> &v, double eps, int min_pts)
{
    std::vector<int32_t> lab(v.size() ? v.size() : 1), out(v.size() ? v.size() : 1);
    orc_dbscan_1d(v.data(), v.size(), eps, min_pts, lab.data());
    int64_t m = orc_largest_cluster(v.data(), lab.data(), v.size(), out.data());
    return std::vector<int>(out.begin(), out.begin() + m);
}

void add_call(std::vector<Call> &v, const Call &c)   // addSVCall (sv_object.cpp:22-33)
{
    if (c.start > c.end) return;
    auto it = std::lower_bound(v.begin(), v.end(), c, [](const Call &a, const Call &b) { return a.start < b.start || (a.start == b.start && a.end < b.end); });
    v.insert(it, c);
}

}  // namespace

extern "C" int64_t orc_split_signatures(uint64_t n, const int32_t *tid, const int32_t *pos, const uint16_t *flag, const uint8_t *mapq,
                                        const int32_t *ref_end, const int32_t *q_start, const int32_t *q_end, const uint32_t *qname_id,
                                        int min_mapq, Call *out, uint64_t cap)
{
    std::unordered_map<int, std::unordered_map<std::string, PrimaryAlignment>> primary_map;
    std::unordered_map<std::string, std::vector<SuppAlignment>> supp_map;
    std::unordered_set<std::string> supp_qnames;
    for (uint64_t i = 0; i < n; i++) {                                                       // :137-172
        if (flag[i] & 0x100 || flag[i] & 0x4 || flag[i] & 0x400 || flag[i] & 0x200 || mapq[i] < min_mapq) continue;
        const std::string qname = "r" + std::to_string(qname_id[i]);
        if (!(flag[i] & 0x800)) {
            primary_map[tid[i]][qname] = PrimaryAlignment{pos[i] + 1, ref_end[i], q_start[i], q_end[i], !(flag[i] & 0x10), 0};
        } else {
            supp_map[qname].push_back(SuppAlignment{tid[i], pos[i] + 1, ref_end[i], q_start[i], q_end[i], !(flag[i] & 0x10)});
            supp_qnames.insert(qname);
        }
    }
    std::unordered_map<int, std::unordered_set<std::string>> to_remove;                      // :183-202
    for (auto &chr_primary : primary_map)
        for (const auto &entry : chr_primary.second)
            if (supp_qnames.find(entry.first) == supp_qnames.end()) to_remove[chr_primary.first].insert(entry.first);
    for (auto &chr_primary : primary_map)
        for (const auto &qname : to_remove[chr_primary.first]) chr_primary.second.erase(qname);

    std::vector<Call> all;
    for (const auto &chr_primary : primary_map) {                                            // :205
        int primary_tid = chr_primary.first;
        std::vector<Call> chr_sv_calls;
        const std::unordered_map<std::string, PrimaryAlignment> &chr_primary_map = chr_primary.second;
        std::unique_ptr<IntervalNode> root = nullptr;
        for (const auto &entry : chr_primary_map) insert(root, entry.second, entry.first);
        std::vector<std::vector<std::string>> primary_clusters;
        std::set<std::string> processed;
        for (const auto &entry : chr_primary_map) {
            const std::string &qname = entry.first;
            if (processed.find(qname) != processed.end()) continue;
            std::vector<std::string> overlap_group;
            findOverlaps(root, entry.second, overlap_group);
            for (const std::string &q : overlap_group) processed.insert(q);
            if (overlap_group.size() > 1) primary_clusters.push_back(overlap_group);
        }
        int min_length = 2000, max_length = 1000000;
        for (const auto &primary_cluster : primary_clusters) {
            bool inversion = false;
            int num_primary = (int)primary_cluster.size(), num_supp_opposite_strand = 0;
            for (const std::string &qname : primary_cluster) {
                const std::vector<SuppAlignment> &supp_alns = supp_map[qname];
                bool primary_strand = chr_primary_map.at(qname).strand, has_opposite_strand = false;
                for (const SuppAlignment &s : supp_alns) if (s.tid == primary_tid && s.strand != primary_strand) has_opposite_strand = true;
                if (has_opposite_strand) num_supp_opposite_strand++;
            }
            if ((double)num_supp_opposite_strand / (double)num_primary > 0.5) inversion = true;
            std::vector<int> starts, ends;
            for (const std::string &qname : primary_cluster) { const PrimaryAlignment &p = chr_primary_map.at(qname); starts.push_back(p.start); ends.push_back(p.end); }
            std::vector<int> primary_start_cluster = fit_largest(starts, 100, 5);
            std::vector<int> primary_end_cluster = fit_largest(ends, 100, 5);
            if (primary_start_cluster.empty() && primary_end_cluster.empty()) continue;
            std::vector<int> supp_starts, supp_ends, read_distances, ref_distances;
            for (const std::string &qname : primary_cluster) {
                const PrimaryAlignment &primary_aln = chr_primary_map.at(qname);
                const std::vector<SuppAlignment> &supp_alns = supp_map.at(qname);
                for (const SuppAlignment &supp_aln : supp_alns) {
                    if (supp_aln.tid == primary_tid) {
                        int read_distance = 0, ref_distance = 0;
                        supp_starts.push_back(supp_aln.start); supp_ends.push_back(supp_aln.end);
                        if (supp_aln.strand == primary_aln.strand) {
                            bool primary_5p = false;
                            if (primary_aln.start < supp_aln.start) primary_5p = true;
                            read_distance = std::max(0, std::max(supp_aln.query_start, primary_aln.query_start) - std::min(supp_aln.query_end, primary_aln.query_end));
                            ref_distance = std::max(0, std::max(supp_aln.start, primary_aln.start) - std::min(supp_aln.end, primary_aln.end));
                            if (!primary_5p) read_distance = -read_distance;
                            read_distances.push_back(read_distance); ref_distances.push_back(ref_distance);
                        }
                    }
                }
            }
            std::vector<int> supp_start_cluster = fit_largest(supp_starts, 100, 5), supp_end_cluster = fit_largest(supp_ends, 100, 5);
            std::vector<int> read_distance_cluster = fit_largest(read_distances, 100, 5), ref_distance_cluster = fit_largest(ref_distances, 100, 5);
            if (supp_start_cluster.empty() && supp_end_cluster.empty() && read_distance_cluster.empty() && ref_distance_cluster.empty()) continue;
            std::vector<int> primary_positions; int primary_cluster_size = 0; bool primary_end = false;
            if (!primary_start_cluster.empty()) { std::sort(primary_start_cluster.begin(), primary_start_cluster.end()); primary_positions.push_back(primary_start_cluster[primary_start_cluster.size() / 2]); primary_cluster_size = primary_start_cluster.size(); }
            if (!primary_end_cluster.empty()) { std::sort(primary_end_cluster.begin(), primary_end_cluster.end()); primary_positions.push_back(primary_end_cluster[primary_end_cluster.size() / 2]); primary_cluster_size = std::max(primary_cluster_size, (int)primary_end_cluster.size()); primary_end = true; }
            std::vector<int> supp_positions; bool supp_end = false; int supp_cluster_size = 0;
            if (!supp_start_cluster.empty()) { std::sort(supp_start_cluster.begin(), supp_start_cluster.end()); supp_positions.push_back(supp_start_cluster[supp_start_cluster.size() / 2]); supp_cluster_size = supp_start_cluster.size(); }
            if (!supp_end_cluster.empty()) { std::sort(supp_end_cluster.begin(), supp_end_cluster.end()); supp_positions.push_back(supp_end_cluster[supp_end_cluster.size() / 2]); supp_cluster_size = std::max(supp_cluster_size, (int)supp_end_cluster.size()); supp_end = true; }
            int read_distance = 0, ref_distance = 0;
            if (!read_distance_cluster.empty() && !ref_distance_cluster.empty()) {
                std::sort(read_distance_cluster.begin(), read_distance_cluster.end());
                read_distance = read_distance_cluster[read_distance_cluster.size() / 2];
                bool primary_5p_most = read_distance > 0;
                read_distance = std::abs(read_distance);
                std::sort(ref_distance_cluster.begin(), ref_distance_cluster.end());
                ref_distance = ref_distance_cluster[ref_distance_cluster.size() / 2];
                int sv_start = 0; bool split_candidate_sv = false;
                if (primary_5p_most && primary_end) { std::sort(primary_positions.begin(), primary_positions.end()); sv_start = primary_positions.back(); split_candidate_sv = true; }
                else if (!primary_5p_most && supp_end) { std::sort(supp_positions.begin(), supp_positions.end()); sv_start = supp_positions.back(); split_candidate_sv = true; }
                if (split_candidate_sv) {
                    int aln_offset = ref_distance - read_distance;
                    if (read_distance > ref_distance && read_distance >= min_length && read_distance <= max_length)
                        add_call(chr_sv_calls, Call{(uint32_t)sv_start, (uint32_t)(sv_start + (read_distance - 1)), 3, primary_cluster_size, aln_offset, 1u << 4, primary_tid});
                    else if (ref_distance > read_distance && ref_distance >= min_length && ref_distance <= max_length)
                        add_call(chr_sv_calls, Call{(uint32_t)sv_start, (uint32_t)(sv_start + (ref_distance - 1)), -1, primary_cluster_size, aln_offset, 1u << 4, primary_tid});
                }
            }
            int cluster_size = std::max(primary_cluster_size, supp_cluster_size);
            int sv_type = inversion ? 2 : -1;
            for (int primary_pos : primary_positions)
                for (int supp_pos : supp_positions) {
                    int sv_start = std::min(primary_pos, supp_pos), sv_end = std::max(primary_pos, supp_pos) - 1, sv_length = sv_end - sv_start + 1;
                    if (sv_length >= min_length && sv_length <= max_length)
                        add_call(chr_sv_calls, Call{(uint32_t)sv_start, (uint32_t)sv_end, sv_type, cluster_size, 0, 1u << 3, primary_tid});
                }
        }
        std::sort(chr_sv_calls.begin(), chr_sv_calls.end(), [](const Call &a, const Call &b) { return a.start < b.start || (a.start == b.start && a.end < b.end); });
        // mergeDuplicateSVs (sv_object.cpp:324-350)
        std::sort(chr_sv_calls.begin(), chr_sv_calls.end(), [](const Call &a, const Call &b) { return std::tie(a.start, a.sv_type) < std::tie(b.start, b.sv_type); });
        std::vector<Call> combined;
        for (size_t i = 0; i < chr_sv_calls.size(); i++) {
            Call &c = chr_sv_calls[i];
            if (i > 0 && c.start == chr_sv_calls[i - 1].start && c.end == chr_sv_calls[i - 1].end) { c.cluster_size += chr_sv_calls[i - 1].cluster_size; combined.back() = c; }
            else combined.push_back(c);
        }
        all.insert(all.end(), combined.begin(), combined.end());
    }
    // deterministic presentation: by contig id (the reference stores per-contig vectors in a hash map)
    std::stable_sort(all.begin(), all.end(), [](const Call &a, const Call &b) { return a.tid < b.tid; });
    for (size_t i = 0; i < all.size() && i < cap; i++) out[i] = all[i];
    return (int64_t)all.size();
}
