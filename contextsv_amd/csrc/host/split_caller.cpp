#include "split_caller.h"

#include <algorithm>
#include <cstdlib>
#include <memory>
#include <set>
#include <unordered_set>

#include "dbscan.h"
#include "log.h"

namespace {

enum : uint16_t { FLAG_UNMAP = 0x4, FLAG_REVERSE = 0x10, FLAG_SECONDARY = 0x100, FLAG_QCFAIL = 0x200, FLAG_DUP = 0x400, FLAG_SUPP = 0x800 };

struct PrimaryAlignment { int start, end, query_start, query_end; bool strand; int cluster_size; };   // sv_caller.h:33-40
struct SuppAlignment { int tid, start, end, query_start, query_end; bool strand; };                  // sv_caller.h:42-49

// The reference's unbalanced BST keyed by start with a max_end annotation (sv_caller.cpp:948-980): nodes are inserted in the
// iteration order of the qname hash map, and the pre-order walk of findOverlaps is a group's member order, so the SHAPE of the tree
// is part of the result. libstdc++ iterates a freshly filled map roughly in reverse insertion order, i.e. by descending start for
// a coordinate-sorted BAM: the tree degenerates into a spine, insertion is quadratic (0.9 s for 2e5 primaries) and the reference's
// recursive insert / findOverlaps recurse once per node. The same shape is built here in O(n log n): the tree a sequence of
// BST insertions produces is the Cartesian tree of the keys (start, insertion index) with the insertion index as heap priority.
struct IntervalTree {
    struct Node { PrimaryAlignment region; const std::string *qname; int max_end; int32_t left, right; };
    std::vector<Node> nodes;                     // in insertion order
    int32_t root = -1;

    void add(const PrimaryAlignment &r, const std::string *q) { nodes.push_back(Node{r, q, r.end, -1, -1}); }

    void build()
    {
        const int32_t n = (int32_t)nodes.size();
        std::vector<int32_t> order((size_t)n);
        for (int32_t i = 0; i < n; i++) order[(size_t)i] = i;
        // equal starts go to the right of the earlier node (`region.start < root->region.start` else right, :969-975)
        std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            return nodes[(size_t)a].region.start != nodes[(size_t)b].region.start ? nodes[(size_t)a].region.start < nodes[(size_t)b].region.start : a < b;
        });
        std::vector<int32_t> spine;              // right spine of the tree built so far, insertion indices increasing
        for (int32_t i : order) {
            int32_t last = -1;
            while (!spine.empty() && spine.back() > i) { last = spine.back(); spine.pop_back(); }
            nodes[(size_t)i].left = last;
            if (!spine.empty()) nodes[(size_t)spine.back()].right = i;
            spine.push_back(i);
        }
        root = spine.empty() ? -1 : spine.front();
        for (int32_t i = n - 1; i >= 0; i--) {   // children were inserted later than their parent: their maxima are final
            Node &x = nodes[(size_t)i];
            if (x.left >= 0) x.max_end = std::max(x.max_end, nodes[(size_t)x.left].max_end);
            if (x.right >= 0) x.max_end = std::max(x.max_end, nodes[(size_t)x.right].max_end);
        }
    }

    // findOverlaps: node, then left (if it can overlap), then right, on an explicit stack
    void overlaps(const PrimaryAlignment &q, std::vector<const std::string *> &out, std::vector<int32_t> &stack) const
    {
        stack.clear();
        if (root >= 0) stack.push_back(root);
        while (!stack.empty()) {
            const Node &n = nodes[(size_t)stack.back()];
            stack.pop_back();
            if (q.start <= n.region.end && q.end >= n.region.start) out.push_back(n.qname);
            // The reference always descends to the right (:961). Subtrees that cannot hold an overlap contribute nothing, so skipping
            // them keeps the result and its order: right descendants all start at or after this node, and max_end bounds every end.
            if (n.right >= 0 && n.region.start <= q.end && nodes[(size_t)n.right].max_end >= q.start) stack.push_back(n.right);
            if (n.left >= 0 && nodes[(size_t)n.left].max_end >= q.start) stack.push_back(n.left);
        }
    }
};

// DBSCAN1D::getLargestCluster on precomputed labels (dbscan1d.cpp:72-90)
std::vector<int> largest_cluster(const std::vector<int> &points, const std::vector<int> &labels)
{
    int max_id = -1;
    for (int c : labels) max_id = std::max(max_id, c);
    std::vector<size_t> sizes((size_t)(max_id + 1), 0);
    for (int c : labels) if (c >= 0) sizes[(size_t)c]++;
    int best = -1; size_t best_n = 0;
    for (int c = 0; c <= max_id; c++) if (sizes[(size_t)c] > best_n) { best_n = sizes[(size_t)c]; best = c; }
    std::vector<int> out;
    if (best < 0) return out;
    for (size_t i = 0; i < labels.size(); i++) if (labels[i] == best) out.push_back(points[i]);
    return out;
}

int sorted_median(std::vector<int> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

struct Group {
    bool inversion = false;
    std::vector<int> sets[6];   // primary starts, primary ends, supp starts, supp ends, read distances, ref distances
};

}  // namespace

void findSplitSVSignatures(const std::vector<SplitRecord> &records, const std::vector<std::string> &qnames,
                           const std::vector<std::string> &target_names, const SplitParams &params,
                           std::unordered_map<std::string, std::vector<SVCall>> &sv_calls)
{
    // ---- collect primaries and supplementaries (sv_caller.cpp:137-172) ------------------------------------------------
    std::unordered_map<int, std::unordered_map<std::string, PrimaryAlignment>> primary_map;
    std::unordered_map<std::string, std::vector<SuppAlignment>> supp_map;
    std::unordered_set<std::string> supp_qnames;
    for (size_t i = 0; i < records.size(); i++) {
        const SplitRecord &r = records[i];
        if ((r.flag & (FLAG_SECONDARY | FLAG_UNMAP | FLAG_DUP | FLAG_QCFAIL)) || r.mapq < params.min_mapq) continue;
        const bool strand = !(r.flag & FLAG_REVERSE);
        if (!(r.flag & FLAG_SUPP)) {
            primary_map[r.tid][qnames[i]] = PrimaryAlignment{r.pos + 1, r.ref_end, r.q_start, r.q_end, strand, 0};   // later record wins
        } else {
            supp_map[qnames[i]].push_back(SuppAlignment{r.tid, r.pos + 1, r.ref_end, r.q_start, r.q_end, strand});
            supp_qnames.insert(qnames[i]);
        }
    }
    // ---- drop primaries that have no supplementary record (:183-202) ---------------------------------------------------
    {
        std::unordered_map<int, std::unordered_set<std::string>> to_remove;
        for (auto &chr_primary : primary_map)
            for (const auto &entry : chr_primary.second)
                if (supp_qnames.find(entry.first) == supp_qnames.end()) to_remove[chr_primary.first].insert(entry.first);
        int total_removed = 0;
        for (auto &chr_primary : primary_map) {
            total_removed += (int)to_remove[chr_primary.first].size();
            for (const auto &q : to_remove[chr_primary.first]) chr_primary.second.erase(q);
        }
        printMessage("Removed " + std::to_string(total_removed) + " primary alignments without supplementary alignments");
    }

    for (const auto &chr_primary : primary_map) {
        const int primary_tid = chr_primary.first;
        const std::string chr_name = target_names.at((size_t)primary_tid);
        const std::unordered_map<std::string, PrimaryAlignment> &chr_primary_map = chr_primary.second;
        printMessage("Processing chromosome " + chr_name + " with " + std::to_string(chr_primary_map.size()) + " primary alignments");

        // ---- overlap groups (:215-238): direct overlaps of the first unprocessed read in hash order, not transitive ----
        IntervalTree tree;
        tree.nodes.reserve(chr_primary_map.size());
        for (const auto &entry : chr_primary_map) tree.add(entry.second, &entry.first);
        tree.build();
        std::vector<std::vector<const std::string *>> primary_clusters;
        {
            std::set<std::string> processed;
            std::vector<int32_t> walk;
            for (const auto &entry : chr_primary_map) {
                if (processed.find(entry.first) != processed.end()) continue;
                std::vector<const std::string *> group;
                tree.overlaps(entry.second, group, walk);
                for (const std::string *q : group) processed.insert(*q);
                if (group.size() > 1) primary_clusters.push_back(std::move(group));
            }
        }

        // ---- the six point sets of every group (:248-347), then ONE batched DBSCAN1D launch -----------------------------
        std::vector<Group> groups(primary_clusters.size());
        for (size_t g = 0; g < primary_clusters.size(); g++) {
            Group &G = groups[g];
            const auto &members = primary_clusters[g];
            int n_opposite = 0;
            for (const std::string *q : members) {
                const PrimaryAlignment &p = chr_primary_map.at(*q);
                const std::vector<SuppAlignment> &supps = supp_map[*q];
                bool opposite = false;
                for (const SuppAlignment &s : supps) if (s.tid == primary_tid && s.strand != p.strand) opposite = true;
                n_opposite += opposite;
                G.sets[0].push_back(p.start);
                G.sets[1].push_back(p.end);
            }
            G.inversion = (double)n_opposite / (double)(int)members.size() > 0.5;                   // :265
            for (const std::string *q : members) {
                const PrimaryAlignment &p = chr_primary_map.at(*q);
                for (const SuppAlignment &s : supp_map.at(*q)) {
                    if (s.tid != primary_tid) continue;                                                // translocations: ignored (:352-354)
                    G.sets[2].push_back(s.start);
                    G.sets[3].push_back(s.end);
                    if (s.strand != p.strand) continue;
                    const bool primary_5p = p.start < s.start;                                         // :322-325
                    int read_distance = std::max(0, std::max(s.query_start, p.query_start) - std::min(s.query_end, p.query_end));
                    const int ref_distance = std::max(0, std::max(s.start, p.start) - std::min(s.end, p.end));
                    if (!primary_5p) read_distance = -read_distance;                                   // :343-345
                    G.sets[4].push_back(read_distance);
                    G.sets[5].push_back(ref_distance);
                }
            }
        }
        std::vector<std::vector<int>> flat_sets, flat_labels;
        flat_sets.reserve(groups.size() * 6);
        for (Group &G : groups) for (int k = 0; k < 6; k++) flat_sets.push_back(G.sets[k]);
        if (!flat_sets.empty()) DBSCAN1D::fitBatch(flat_sets, params.eps, params.min_pts, flat_labels);

        // ---- medians, SPLITDIST1 candidates, SPLIT dummies (:283-486) ----------------------------------------------------
        std::vector<SVCall> chr_sv_calls;
        chr_sv_calls.reserve(1000);
        for (size_t g = 0; g < groups.size(); g++) {
            Group &G = groups[g];
            std::vector<int> cl[6];
            for (int k = 0; k < 6; k++) cl[k] = largest_cluster(G.sets[k], flat_labels[g * 6 + k]);
            std::vector<int> &p_start = cl[0], &p_end = cl[1], &s_start = cl[2], &s_end = cl[3], &read_d = cl[4], &ref_d = cl[5];
            if (p_start.empty() && p_end.empty()) continue;                                          // :291-293
            if (s_start.empty() && s_end.empty() && read_d.empty() && ref_d.empty()) continue;        // :375-377

            std::vector<int> primary_positions, supp_positions;
            int primary_cluster_size = 0, supp_cluster_size = 0;
            bool primary_end = false, supp_end = false;
            if (!p_start.empty()) { primary_positions.push_back(sorted_median(p_start)); primary_cluster_size = (int)p_start.size(); }
            if (!p_end.empty()) { primary_positions.push_back(sorted_median(p_end)); primary_cluster_size = std::max(primary_cluster_size, (int)p_end.size()); primary_end = true; }
            if (!s_start.empty()) { supp_positions.push_back(sorted_median(s_start)); supp_cluster_size = (int)s_start.size(); }
            if (!s_end.empty()) { supp_positions.push_back(sorted_median(s_end)); supp_cluster_size = std::max(supp_cluster_size, (int)s_end.size()); supp_end = true; }

            if (!read_d.empty() && !ref_d.empty()) {                                                  // :422-468
                int read_distance = sorted_median(read_d);
                const bool primary_5p_most = read_distance > 0;
                read_distance = std::abs(read_distance);
                const int ref_distance = sorted_median(ref_d);
                int sv_start = 0;
                bool candidate = false;
                if (primary_5p_most && primary_end) {
                    std::sort(primary_positions.begin(), primary_positions.end());
                    sv_start = primary_positions.back(); candidate = true;
                } else if (!primary_5p_most && supp_end) {
                    std::sort(supp_positions.begin(), supp_positions.end());
                    sv_start = supp_positions.back(); candidate = true;
                }
                if (candidate) {
                    SVEvidenceFlags aln_type;
                    aln_type.set((size_t)SVDataType::SPLITDIST1);
                    const int aln_offset = ref_distance - read_distance;
                    if (read_distance > ref_distance && read_distance >= params.min_length && read_distance <= params.max_length) {
                        SVCall c((uint32_t)sv_start, (uint32_t)(sv_start + (read_distance - 1)), SVType::INS, getSVTypeSymbol(SVType::INS), aln_type,
                                 Genotype::UNKNOWN, 0.0, 0, aln_offset, primary_cluster_size);
                        addSVCall(chr_sv_calls, c);
                    } else if (ref_distance > read_distance && ref_distance >= params.min_length && ref_distance <= params.max_length) {
                        SVCall c((uint32_t)sv_start, (uint32_t)(sv_start + (ref_distance - 1)), SVType::UNKNOWN, getSVTypeSymbol(SVType::UNKNOWN), aln_type,
                                 Genotype::UNKNOWN, 0.0, 0, aln_offset, primary_cluster_size);
                        addSVCall(chr_sv_calls, c);
                    }
                }
            }
            // dummy call per (primary median, supplementary median) pair for the copy-number pass (:470-486)
            const int cluster_size = std::max(primary_cluster_size, supp_cluster_size);
            const SVType sv_type = G.inversion ? SVType::INV : SVType::UNKNOWN;
            const std::string alt = (sv_type == SVType::INV) ? "<INV>" : ".";
            for (int primary_pos : primary_positions)
                for (int supp_pos : supp_positions) {
                    const int sv_start = std::min(primary_pos, supp_pos), sv_end = std::max(primary_pos, supp_pos) - 1;
                    const int sv_length = sv_end - sv_start + 1;
                    if (sv_length >= params.min_length && sv_length <= params.max_length) {
                        SVEvidenceFlags aln_type;
                        aln_type.set((size_t)SVDataType::SPLIT);
                        SVCall c((uint32_t)sv_start, (uint32_t)sv_end, sv_type, alt, aln_type, Genotype::UNKNOWN, 0.0, 0, 0, cluster_size);
                        addSVCall(chr_sv_calls, c);
                    }
                }
        }
        std::sort(chr_sv_calls.begin(), chr_sv_calls.end(), [](const SVCall &a, const SVCall &b) {   // :491-493
            return a.start < b.start || (a.start == b.start && a.end < b.end);
        });
        mergeDuplicateSVs(chr_sv_calls);
        sv_calls[chr_name] = std::move(chr_sv_calls);
        printMessage(chr_name + ": Found " + std::to_string(sv_calls[chr_name].size()) + " SV candidates");
    }
}
