#include "split_caller.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <memory>
#include <stdexcept>
#include <thread>

#include "dbscan.h"
#include "log.h"
#include "par.h"
#include "umap_order.h"

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
    struct Node { PrimaryAlignment region; uint32_t member; int max_end; int32_t left, right; };
    std::vector<Node> nodes;                     // in insertion order
    int32_t root = -1;

    void add(const PrimaryAlignment &r, uint32_t member) { nodes.push_back(Node{r, member, r.end, -1, -1}); }

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
    void overlaps(const PrimaryAlignment &q, std::vector<uint32_t> &out, std::vector<int32_t> &stack) const
    {
        stack.clear();
        if (root >= 0) stack.push_back(root);
        while (!stack.empty()) {
            const Node &n = nodes[(size_t)stack.back()];
            stack.pop_back();
            if (q.start <= n.region.end && q.end >= n.region.start) out.push_back(n.member);
            // The reference always descends to the right (:961). Subtrees that cannot hold an overlap contribute nothing, so skipping
            // them keeps the result and its order: right descendants all start at or after this node, and max_end bounds every end.
            if (n.right >= 0 && n.region.start <= q.end && nodes[(size_t)n.right].max_end >= q.start) stack.push_back(n.right);
            if (n.left >= 0 && nodes[(size_t)n.left].max_end >= q.start) stack.push_back(n.left);
        }
    }
};

// DBSCAN1D::getLargestCluster on precomputed labels (dbscan1d.cpp:72-90); labels[i] belongs to points[i]
std::vector<int> largest_cluster(const std::vector<int> &points, const int *labels)
{
    const size_t n = points.size();
    int max_id = -1;
    for (size_t i = 0; i < n; i++) max_id = std::max(max_id, labels[i]);
    std::vector<size_t> sizes((size_t)(max_id + 1), 0);
    for (size_t i = 0; i < n; i++) if (labels[i] >= 0) sizes[(size_t)labels[i]]++;
    int best = -1; size_t best_n = 0;
    for (int c = 0; c <= max_id; c++) if (sizes[(size_t)c] > best_n) { best_n = sizes[(size_t)c]; best = c; }
    std::vector<int> out;
    if (best < 0) return out;
    for (size_t i = 0; i < n; i++) if (labels[i] == best) out.push_back(points[i]);
    return out;
}

int sorted_median(std::vector<int> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

struct Group {
    bool inversion = false;
    std::vector<int> sets[6];   // primary starts, primary ends, supp starts, supp ends, read distances, ref distances
};


// ---- query-name equality across contigs (names themselves, or run-wide dictionary ids) ---------------------------------
bool same_name(const SplitContig &a, uint64_t i, const SplitContig &b, uint64_t j)
{
    if (a.name_bytes && b.name_bytes) {
        const uint64_t la = a.name_off[i + 1] - a.name_off[i], lb = b.name_off[j + 1] - b.name_off[j];
        return la == lb && memcmp(a.name_bytes + a.name_off[i], b.name_bytes + b.name_off[j], (size_t)la) == 0;
    }
    return a.name_id[i] == b.name_id[j];
}

// a supplementary record in file order, found again by its name's hash (supp_map, sv_caller.cpp:162-165)
struct SuppRef { uint64_t hash; uint64_t ord; uint32_t contig; uint32_t rec; };   // ord: position in the file

// What one contig's thread keeps between the phases.
struct ContigWork {
    const SplitContig *in = nullptr;
    csvhost::UMapOrder order;                      // chr_primary_map's node list, all primaries
    std::vector<uint32_t> first_rec, last_rec;     // by node: the record that created the key / the last record of that name (its values win)
    std::vector<SuppRef> supps;                    // this contig's supplementary records, file order
    // survivors (primaries with a supplementary record) in the map's iteration order
    const std::vector<uint32_t> *dev_order = nullptr;                  // survivors' records in iteration order, when they came from params.device_order
    std::vector<uint32_t> member_rec;                                  // the primary record
    std::vector<std::pair<uint32_t, uint32_t>> member_supp_ref;       // (contig, record) of the supplementary records, all members back to back
    std::vector<size_t> member_supp_off;                               // end of member m's supplementary records in member_supp_ref
    std::vector<uint32_t> need;                                        // records whose intervals are gathered (sorted)
    std::vector<int32_t> got[3];
    std::vector<PrimaryAlignment> member;
    std::vector<uint32_t> member_slot, supp_slot;      // where member m's / supplementary reference q's record sits in its contig's `need` list (prepare())
    std::vector<SuppAlignment> member_supps;           // member m's: [member_supp_off[m - 1], member_supp_off[m]) (one array: 1e4 one-element vectors cost more than the pass's arithmetic)
    std::vector<Group> groups;
    size_t set_base = 0;                           // first of this contig's point sets in the genome-wide batch
    size_t n_primary = 0;
    std::vector<SVCall> calls;
};

template <class F>
void parallel_over(size_t n, int threads, F f) { csvhost::parallel_for(n, threads, f); }

}  // namespace

// findSplitSVSignatures in two halves, so that a run can do the first — everything that needs no alignment intervals: the primaries and
// supplementaries, the qname map's iteration order, the survivors (sv_caller.cpp:137-202) — while the device is still busy with the
// CIGAR pass, and the second — intervals, tree, groups, fits, calls (:205-498) — once the scan's per-read outputs exist.
struct SplitPass::Impl {
    const std::vector<SplitContig> &contigs;
    const std::vector<std::string> &target_names;
    SplitParams params;
    std::vector<size_t> by_size;
    std::vector<ContigWork> work;
    std::vector<SuppRef> supp_index;
    std::vector<std::vector<uint32_t>> dev_recs;
    std::vector<char> grouped, called;                  // per contig: intervals gathered and overlap groups built; calls made and handed out
    void gatherFor(const std::vector<size_t> &ids);
    void groupsOf(size_t c, bool trace);
    void finishEarly(const std::vector<size_t> &ids);
    void finishFor(const std::vector<size_t> &ids, std::unordered_map<std::string, std::vector<SVCall>> &sv_calls);
    Impl(const std::vector<SplitContig> &c, const std::vector<std::string> &t, const SplitParams &p) : contigs(c), target_names(t), params(p) {}
    void prepare();
    void finish(std::unordered_map<std::string, std::vector<SVCall>> &sv_calls);
};

SplitPass::SplitPass(const std::vector<SplitContig> &contigs, const std::vector<std::string> &target_names, const SplitParams &params)
    : p(new Impl(contigs, target_names, params)) {}
SplitPass::~SplitPass() = default;
void SplitPass::prepare() { p->prepare(); prepared = true; }
void SplitPass::finish(std::unordered_map<std::string, std::vector<SVCall>> &sv_calls) { if (!prepared) prepare(); p->finish(sv_calls); }
void SplitPass::finishEarly(const std::vector<size_t> &contig_ids) { if (prepared) p->finishEarly(contig_ids); }
void SplitPass::finishFor(const std::vector<size_t> &contig_ids, std::unordered_map<std::string, std::vector<SVCall>> &sv_calls) { if (!prepared) prepare(); p->finishFor(contig_ids, sv_calls); }

void findSplitSVSignatures(const std::vector<SplitContig> &contigs, const std::vector<std::string> &target_names, const SplitParams &params,
                           std::unordered_map<std::string, std::vector<SVCall>> &sv_calls)
{
    SplitPass pass(contigs, target_names, params);
    pass.prepare();
    pass.finish(sv_calls);
}

void SplitPass::Impl::prepare()
{
    for (const SplitContig &c : contigs)
        if (c.n && (!c.qhash || (!c.name_id && !(c.name_bytes && c.name_off)) || (!params.intervals && !(c.ref_end && c.q_start && c.q_end))))
            throw std::runtime_error("findSplitSVSignatures: a contig without query-name hashes / identities or without alignment intervals");
    // larger contigs first: the wall time of a parallel phase is the largest contig's
    by_size.assign(contigs.size(), 0);
    for (size_t i = 0; i < by_size.size(); i++) by_size[i] = i;
    std::sort(by_size.begin(), by_size.end(), [&](size_t a, size_t b) { return contigs[a].n != contigs[b].n ? contigs[a].n > contigs[b].n : a < b; });
    // one work item per tid (a tid split over several blocks is one map in the reference: blocks of a tid are chained in order)
    work = std::vector<ContigWork>(contigs.size());
    for (size_t c = 0; c < contigs.size(); c++) {
        work[c].in = &contigs[c];
        for (size_t d = 0; d < c; d++) if (contigs[d].tid == contigs[c].tid) throw std::runtime_error("findSplitSVSignatures: two blocks with the same tid");
    }

    // ---- phase 1: collect primaries (into the replayed map) and supplementaries (sv_caller.cpp:137-172), per contig ------------
    // Contigs whose names are known to be unique can get the map's iteration order from params.device_order (the device): for them
    // only the supplementary records are collected here and the primaries counted.
    auto on_device = [&](size_t c) { return params.device_order && contigs[c].unique_names; };
    // the device's share starts now: the part of it that needs no supplementary record runs beside the collection below
    std::vector<size_t> dev_contigs;
    for (size_t c = 0; c < contigs.size(); c++) if (on_device(c) && contigs[c].n) dev_contigs.push_back(c);
    size_t n_nonempty = 0;
    for (const SplitContig &C : contigs) n_nonempty += C.n != 0;
    if (!dev_contigs.empty()) params.device_order->begin(dev_contigs, params.min_mapq, dev_contigs.size() == n_nonempty);
    std::unique_ptr<csvhost::TraceScope> tr(new csvhost::TraceScope("split: collect"));
    parallel_over(contigs.size(), params.threads, [&](size_t k) {
        const size_t c = by_size[k];
        ContigWork &W = work[c];
        const SplitContig &C = *W.in;
        const bool dev = on_device(c);
        if (!dev) W.order.reserve((size_t)C.n);
        for (uint64_t i = 0; i < C.n; i++) {
            const uint16_t flag = C.flag[i];
            if ((flag & (FLAG_SECONDARY | FLAG_UNMAP | FLAG_DUP | FLAG_QCFAIL)) || C.mapq[i] < params.min_mapq) continue;
            if (flag & FLAG_SUPP) { W.supps.push_back(SuppRef{C.qhash[i], C.file_idx ? C.file_idx[i] : (((uint64_t)c << 40) | i), (uint32_t)c, (uint32_t)i}); continue; }
            if (dev) { W.n_primary++; continue; }
            const int64_t node = W.order.find(C.qhash[i], [&](uint32_t nd) { return same_name(C, W.first_rec[nd], C, i); });
            if (node >= 0) { W.last_rec[(size_t)node] = (uint32_t)i; continue; }        // operator[]: a later record of the name wins (:152)
            W.order.insert_new(C.qhash[i]);
            W.first_rec.push_back((uint32_t)i);
            W.last_rec.push_back((uint32_t)i);
        }
        if (!dev) W.n_primary = W.first_rec.size();
        std::sort(W.supps.begin(), W.supps.end(), [](const SuppRef &a, const SuppRef &b) { return a.hash != b.hash ? a.hash < b.hash : a.ord < b.ord; });
    });

    // ---- supp_map: every supplementary record of the run by name, file order within a name (:162-165) ---------------------------
    tr.reset(new csvhost::TraceScope("split: supp index"));
    {   // the contigs' lists are sorted (phase 1): merge them pairwise
        auto less = [](const SuppRef &a, const SuppRef &b) { return a.hash != b.hash ? a.hash < b.hash : a.ord < b.ord; };
        std::vector<size_t> cut{0};
        for (const ContigWork &W : work) { supp_index.insert(supp_index.end(), W.supps.begin(), W.supps.end()); cut.push_back(supp_index.size()); }
        while (cut.size() > 2) {
            std::vector<size_t> next{0};
            for (size_t k = 0; k + 2 < cut.size(); k += 2) {
                std::inplace_merge(supp_index.begin() + (std::ptrdiff_t)cut[k], supp_index.begin() + (std::ptrdiff_t)cut[k + 1], supp_index.begin() + (std::ptrdiff_t)cut[k + 2], less);
                next.push_back(cut[k + 2]);
            }
            if (cut.size() % 2 == 0) next.push_back(cut.back());
            cut.swap(next);
        }
    }

    // ---- the device's share: which primaries have a supplementary record's name hash, in the map's iteration order ----------------
    tr.reset(new csvhost::TraceScope("split: device order"));
    if (!dev_contigs.empty()) {
        std::vector<uint64_t> supp_hash;
        supp_hash.reserve(supp_index.size());
        for (const SuppRef &r : supp_index) if (supp_hash.empty() || supp_hash.back() != r.hash) supp_hash.push_back(r.hash);
        params.device_order->survivors(dev_contigs, params.min_mapq, supp_hash, dev_recs);
        for (size_t k = 0; k < dev_contigs.size(); k++) work[dev_contigs[k]].dev_order = &dev_recs[k];
    }

    // ---- survivors (primaries with a supplementary record, :183-202) in the map's iteration order, with their supplementary records ----
    tr.reset(new csvhost::TraceScope("split: survivors"));
    std::atomic<long> total_removed{0};
    parallel_over(contigs.size(), params.threads, [&](size_t k) {
        ContigWork &W = work[by_size[k]];
        const SplitContig &C = *W.in;
        auto visit = [&](uint32_t first_rec, uint32_t last_rec) {
            const uint64_t h = C.qhash[first_rec];
            auto lo = std::lower_bound(supp_index.begin(), supp_index.end(), h, [](const SuppRef &a, uint64_t x) { return a.hash < x; });
            const size_t before = W.member_supp_ref.size();
            for (; lo != supp_index.end() && lo->hash == h; ++lo)
                if (same_name(C, first_rec, *work[lo->contig].in, lo->rec)) W.member_supp_ref.push_back(std::make_pair(lo->contig, lo->rec));   // (equal hash, other name: skipped)
            if (W.member_supp_ref.size() == before) return;                             // erased: no supplementary record
            W.member_rec.push_back(last_rec);
            W.member_supp_off.push_back(W.member_supp_ref.size());
        };
        if (W.dev_order) for (uint32_t r : *W.dev_order) visit(r, r);
        else W.order.for_each([&](uint32_t node) { visit(W.first_rec[node], W.last_rec[node]); });
        total_removed += (long)(W.n_primary - W.member_rec.size());
        W.order = csvhost::UMapOrder(); W.first_rec = {}; W.last_rec = {};             // the map's storage is not needed any more
    });

    printMessage("Removed " + std::to_string(total_removed.load()) + " primary alignments without supplementary alignments");

    // ---- which records' alignment intervals finish() will ask the scan's outputs for (nothing here needs the scan itself) ----------
    tr.reset(new csvhost::TraceScope("split: need lists"));
    parallel_over(work.size(), params.threads, [&](size_t c) {
        ContigWork &W = work[c];
        if (W.in->ref_end) return;
        for (uint32_t r : W.member_rec) W.need.push_back(r);
        for (const SuppRef &sr : W.supps) W.need.push_back(sr.rec);
        std::sort(W.need.begin(), W.need.end());
        W.need.erase(std::unique(W.need.begin(), W.need.end()), W.need.end());
    });
    parallel_over(work.size(), params.threads, [&](size_t c) {                     // (every contig's list is complete by now)
        ContigWork &W = work[c];
        auto slot_of = [](const ContigWork &X, uint32_t rec) { return (uint32_t)(std::lower_bound(X.need.begin(), X.need.end(), rec) - X.need.begin()); };
        if (!W.in->ref_end) { W.member_slot.reserve(W.member_rec.size()); for (uint32_t r : W.member_rec) W.member_slot.push_back(slot_of(W, r)); }
        W.supp_slot.reserve(W.member_supp_ref.size());
        for (const auto &ref : W.member_supp_ref) W.supp_slot.push_back(work[ref.first].in->ref_end ? 0u : slot_of(work[ref.first], ref.second));
    });
    grouped.assign(work.size(), 0);
    called.assign(work.size(), 0);
    tr.reset();
}

// the alignment intervals of these contigs' records (ref_end / q_start / q_end of the scan kernel): straight from the arrays, or — contigs that
// carry an IntervalSource — gathered for just the records on prepare()'s need lists (a few per cent of the contig's)
void SplitPass::Impl::gatherFor(const std::vector<size_t> &ids)
{
    std::vector<size_t> which;
    std::vector<uint32_t> rec;
    std::vector<uint64_t> rec_off{0};
    for (size_t c : ids) {
        ContigWork &W = work[c];
        if (W.in->ref_end || W.need.empty()) continue;
        which.push_back(c);
        rec.insert(rec.end(), W.need.begin(), W.need.end());
        rec_off.push_back(rec.size());
    }
    if (which.empty()) return;
    std::vector<int32_t> a(rec.size()), b(rec.size()), d(rec.size());
    params.intervals->gather(which, rec, rec_off, a.data(), b.data(), d.data());
    for (size_t k = 0; k < which.size(); k++) {
        ContigWork &W = work[which[k]];
        W.got[0].assign(a.begin() + (std::ptrdiff_t)rec_off[k], a.begin() + (std::ptrdiff_t)rec_off[k + 1]);
        W.got[1].assign(b.begin() + (std::ptrdiff_t)rec_off[k], b.begin() + (std::ptrdiff_t)rec_off[k + 1]);
        W.got[2].assign(d.begin() + (std::ptrdiff_t)rec_off[k], d.begin() + (std::ptrdiff_t)rec_off[k + 1]);
    }
}

// phase 2 for one contig: interval tree, overlap groups, the six point sets (:215-347)
void SplitPass::Impl::groupsOf(size_t c, bool trace)
{
        ContigWork &W = work[c];
        const SplitContig &C = *W.in;
        const int primary_tid = C.tid;
        std::unique_ptr<csvhost::TraceScope> t2(trace ? new csvhost::TraceScope("split: groups[0] members") : nullptr);
        W.member.reserve(W.member_rec.size());
        W.member_supps.reserve(W.member_supp_ref.size());
        // (the records' places in the gathered arrays were looked up by prepare(): no searches on this side of the CIGAR pass)
        auto at = [](const ContigWork &X, uint32_t rec, uint32_t slot, int which) -> int32_t {
            const SplitContig &D = *X.in;
            if (D.ref_end) return which == 0 ? D.ref_end[rec] : (which == 1 ? D.q_start[rec] : D.q_end[rec]);
            return X.got[which][slot];
        };
        for (size_t m = 0; m < W.member_rec.size(); m++) {
            const uint32_t i = W.member_rec[m], si = C.ref_end ? 0u : W.member_slot[m];
            W.member.push_back(PrimaryAlignment{C.pos[i] + 1, at(W, i, si, 0), at(W, i, si, 1), at(W, i, si, 2), !(C.flag[i] & FLAG_REVERSE), 0});
            for (size_t q = m ? W.member_supp_off[m - 1] : 0; q < W.member_supp_off[m]; q++) {
                const ContigWork &SW = work[W.member_supp_ref[q].first];
                const SplitContig &S = *SW.in;
                const uint32_t r = W.member_supp_ref[q].second, sr = W.supp_slot[q];
                // (a record on another contig only ever answers the tid tests below, :352-354: its intervals are not looked at — and may not have been gathered yet)
                if (&SW != &W) { W.member_supps.push_back(SuppAlignment{S.tid, S.pos[r] + 1, 0, 0, 0, !(S.flag[r] & FLAG_REVERSE)}); continue; }
                W.member_supps.push_back(SuppAlignment{S.tid, S.pos[r] + 1, at(SW, r, sr, 0), at(SW, r, sr, 1), at(SW, r, sr, 2), !(S.flag[r] & FLAG_REVERSE)});
            }
        }

        if (trace) t2.reset(new csvhost::TraceScope("split: groups[0] tree"));
        // overlap groups (:215-238): direct overlaps of the first unprocessed read in iteration order, not transitive
        IntervalTree tree;
        tree.nodes.reserve(W.member.size());
        for (size_t m = 0; m < W.member.size(); m++) tree.add(W.member[m], (uint32_t)m);
        tree.build();
        if (trace) t2.reset(new csvhost::TraceScope("split: groups[0] seeds"));
        std::vector<std::vector<uint32_t>> primary_clusters;
        {
            std::vector<char> processed(W.member.size(), 0);
            std::vector<int32_t> walk;
            for (size_t m = 0; m < W.member.size(); m++) {
                if (processed[m]) continue;
                std::vector<uint32_t> group;
                tree.overlaps(W.member[m], group, walk);
                for (uint32_t q : group) processed[q] = 1;
                if (group.size() > 1) primary_clusters.push_back(std::move(group));
            }
        }
        if (trace) t2.reset(new csvhost::TraceScope("split: groups[0] sets"));
        W.groups.assign(primary_clusters.size(), Group());
        for (size_t g = 0; g < primary_clusters.size(); g++) {
            Group &G = W.groups[g];
            const auto &members = primary_clusters[g];
            int n_opposite = 0;
            for (uint32_t q : members) {
                const PrimaryAlignment &p = W.member[q];
                bool opposite = false;
                for (size_t z = q ? W.member_supp_off[q - 1] : 0; z < W.member_supp_off[q]; z++) {
                    const SuppAlignment &s = W.member_supps[z];
                    if (s.tid == primary_tid && s.strand != p.strand) opposite = true;
                }
                n_opposite += opposite;
                G.sets[0].push_back(p.start);
                G.sets[1].push_back(p.end);
            }
            G.inversion = (double)n_opposite / (double)(int)members.size() > 0.5;                   // :265
            for (uint32_t q : members) {
                const PrimaryAlignment &p = W.member[q];
                for (size_t z = q ? W.member_supp_off[q - 1] : 0; z < W.member_supp_off[q]; z++) {
                    const SuppAlignment &s = W.member_supps[z];
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
}

// Contigs whose scan outputs exist already (the caller knows): their intervals and groups now, the rest in finish()
void SplitPass::Impl::finishEarly(const std::vector<size_t> &ids)
{
    std::vector<size_t> todo;
    for (size_t c : ids) if (c < work.size() && !grouped[c]) todo.push_back(c);
    if (todo.empty()) return;
    csvhost::TraceScope tr("split: early gather + groups");
    gatherFor(todo);
    std::sort(todo.begin(), todo.end(), [&](size_t a, size_t b) { return work[a].member_rec.size() > work[b].member_rec.size(); });
    parallel_over(todo.size(), params.threads, [&](size_t k) { groupsOf(todo[k], false); });
    for (size_t c : todo) grouped[c] = 1;
}

void SplitPass::Impl::finish(std::unordered_map<std::string, std::vector<SVCall>> &sv_calls)
{
    std::vector<size_t> all;
    for (size_t k = 0; k < contigs.size(); k++) all.push_back(by_size[k]);                                  // (largest first)
    finishFor(all, sv_calls);
}

// Everything behind prepare() for these contigs (those not done yet): intervals, groups, the DBSCAN1D fits, the calls. A contig's
// calls depend on nothing outside the contig (records on other contigs only answer tid tests), so any partition of the contigs over
// calls of this function gives the same calls.
void SplitPass::Impl::finishFor(const std::vector<size_t> &ids, std::unordered_map<std::string, std::vector<SVCall>> &sv_calls)
{
    std::unique_ptr<csvhost::TraceScope> tr;
    std::vector<size_t> todo, rest;
    for (size_t c : ids) if (c < work.size() && !called[c]) todo.push_back(c);
    for (size_t c : todo) if (!grouped[c]) rest.push_back(c);
    tr.reset(new csvhost::TraceScope("split: interval gather"));
    gatherFor(rest);
    tr.reset(new csvhost::TraceScope("split: groups"));
    parallel_over(rest.size(), params.threads, [&](size_t k) { groupsOf(rest[k], k == 0); });
    for (size_t c : rest) grouped[c] = 1;

    // ---- the six DBSCAN1D(100, 5) fits of every group of every contig: ONE batched launch (:270-372) ------------------------------
    tr.reset(new csvhost::TraceScope("split: dbscan1d batch"));
    std::vector<int> flat_pts, flat_labels;
    std::vector<uint64_t> flat_off{0};
    for (size_t c : todo) {
        ContigWork &W = work[c];
        W.set_base = flat_off.size() - 1;
        for (Group &G : W.groups)
            for (int s = 0; s < 6; s++) { flat_pts.insert(flat_pts.end(), G.sets[s].begin(), G.sets[s].end()); flat_off.push_back(flat_pts.size()); }
    }
    if (!flat_pts.empty()) DBSCAN1D::fitBatchFlat(flat_pts, flat_off, params.eps, params.min_pts, flat_labels);

    // ---- phase 3: medians, SPLITDIST1 candidates, SPLIT dummies (:283-486), per contig --------------------------------------------
    tr.reset(new csvhost::TraceScope("split: calls"));
    parallel_over(todo.size(), params.threads, [&](size_t k) {
        ContigWork &W = work[todo[k]];
        if (W.n_primary == 0) return;                        // no entry in primary_map for this tid
        std::vector<SVCall> &chr_sv_calls = W.calls;
        chr_sv_calls.reserve(1000);
        for (size_t g = 0; g < W.groups.size(); g++) {
            Group &G = W.groups[g];
            std::vector<int> cl[6];
            for (int s = 0; s < 6; s++) cl[s] = largest_cluster(G.sets[s], flat_labels.data() + flat_off[W.set_base + g * 6 + (size_t)s]);
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
    });
    tr.reset();
    for (size_t c : todo) {
        ContigWork &W = work[c];
        called[c] = 1;
        if (W.n_primary == 0) continue;
        const std::string chr_name = target_names.at((size_t)W.in->tid);
        printMessage("Processing chromosome " + chr_name + " with " + std::to_string(W.member.size()) + " primary alignments");
        printMessage(chr_name + ": Found " + std::to_string(W.calls.size()) + " SV candidates");
        sv_calls[chr_name] = std::move(W.calls);
    }
}

void findSplitSVSignatures(const std::vector<SplitRecord> &records, const std::vector<std::string> &qnames,
                           const std::vector<std::string> &target_names, const SplitParams &params,
                           std::unordered_map<std::string, std::vector<SVCall>> &sv_calls)
{
    // regroup by tid (file order kept inside a tid; tids in order of first appearance, which is file order for a sorted BAM)
    struct Block { int32_t tid; std::vector<int32_t> pos, ref_end, q_start, q_end; std::vector<uint16_t> flag; std::vector<uint8_t> mapq;
                   std::vector<uint64_t> qhash, name_off, file_idx; std::string names; };
    std::vector<std::unique_ptr<Block>> blocks;
    std::unordered_map<int32_t, size_t> at;
    for (size_t i = 0; i < records.size(); i++) {
        const SplitRecord &r = records[i];
        auto it = at.find(r.tid);
        if (it == at.end()) { it = at.emplace(r.tid, blocks.size()).first; blocks.emplace_back(new Block()); blocks.back()->tid = r.tid; blocks.back()->name_off.push_back(0); }
        Block &b = *blocks[it->second];
        b.pos.push_back(r.pos); b.ref_end.push_back(r.ref_end); b.q_start.push_back(r.q_start); b.q_end.push_back(r.q_end);
        b.flag.push_back(r.flag); b.mapq.push_back(r.mapq);
        b.qhash.push_back(csvhost::std_string_hash(qnames[i].data(), qnames[i].size()));
        b.names += qnames[i];
        b.file_idx.push_back(i);
        b.name_off.push_back(b.names.size());
    }
    std::vector<SplitContig> contigs;
    for (auto &bp : blocks) {
        Block &b = *bp;
        SplitContig c;
        c.tid = b.tid; c.n = b.pos.size();
        c.pos = b.pos.data(); c.flag = b.flag.data(); c.mapq = b.mapq.data(); c.ref_end = b.ref_end.data(); c.q_start = b.q_start.data(); c.q_end = b.q_end.data();
        c.qhash = b.qhash.data(); c.name_bytes = b.names.data(); c.name_off = b.name_off.data(); c.file_idx = b.file_idx.data();
        contigs.push_back(c);
    }
    findSplitSVSignatures(contigs, target_names, params, sv_calls);
}
