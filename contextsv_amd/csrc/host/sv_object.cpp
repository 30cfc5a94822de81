// sv_object.cpp — host side of the clustering seam (reference: src/sv_object.cpp).
#include "sv_object.h"

#include <algorithm>
#include <numeric>
#include <stdexcept>
#include <tuple>

#include "dbscan.h"
#include "log.h"
#include "par.h"

// sorted insert by (start,end); an element equal to existing ones goes BEFORE them (std::lower_bound),
// which is what fixes DBSCAN's index order (reference sv_object.cpp:22-33)
void addSVCall(std::vector<SVCall> &sv_calls, SVCall &sv_call)
{
    if (sv_call.start > sv_call.end) {
        printError("ERROR: Invalid SV call at position " + std::to_string(sv_call.start) + "-" + std::to_string(sv_call.end) +
                   " from data type " + getSVAlignmentTypeString(sv_call.aln_type));
        return;
    }
    sv_calls.insert(std::lower_bound(sv_calls.begin(), sv_calls.end(), sv_call), sv_call);
}

uint32_t getSVCount(const std::vector<SVCall> &sv_calls) { return (uint32_t)sv_calls.size(); }

void concatenateSVCalls(std::vector<SVCall> &target, const std::vector<SVCall> &source)
{
    target.insert(target.end(), source.begin(), source.end());
}

// Representative choice for the calls of one type given their labels.
//  * buckets are visited in ascending label order, noise (-2) first — the reference's std::map walk (:98-121)
//  * buckets with < 2 members vanish (:126-128)
//  * noise with keep_noise: copied through; noise without keep_noise: merged like a cluster (:131-149)
//  * any member with hmm_likelihood != 0: std::sort by (cluster_size desc, length desc), first member with a
//    likelihood wins, cluster_size untouched (:155-181)
//  * otherwise: std::sort by length desc, top max(1, int(n*0.2)), its element [k/2], cluster_size = n (:187-244)
// The sorts run on index vectors with the reference's comparators: introsort's moves depend only on the
// comparison outcomes, so the permutation is the one the reference gets on its SVCall vector.
void mergeTypeWithLabels(std::vector<SVCall> &type_calls, const int32_t *labels, bool keep_noise, std::vector<SVCall> &merged)
{
    const size_t n = type_calls.size();
    if (n == 0) return;
    int32_t max_label = -2;
    for (size_t i = 0; i < n; i++) max_label = std::max(max_label, labels[i]);
    // counting sort of members by label (stable => members keep vector order inside a bucket); slot 0 = -2, slot 1 = -1
    const size_t n_slots = (size_t)(max_label + 3);
    std::vector<size_t> head(n_slots + 1, 0);
    for (size_t i = 0; i < n; i++) head[(size_t)(labels[i] + 2) + 1]++;
    for (size_t s = 0; s < n_slots; s++) head[s + 1] += head[s];
    std::vector<uint32_t> member(n);
    {
        std::vector<size_t> cur(head.begin(), head.end() - 1);
        for (size_t i = 0; i < n; i++) member[cur[(size_t)(labels[i] + 2)]++] = (uint32_t)i;
    }
    for (size_t s = 0; s < n_slots; s++) {
        const size_t b0 = head[s], b1 = head[s + 1], sz = b1 - b0;
        if (sz < 2) continue;
        const int cluster_id = (int)s - 2;
        uint32_t *m = member.data() + b0;
        if (cluster_id < 0 && keep_noise) {
            for (size_t k = 0; k < sz; k++) merged.push_back(type_calls[m[k]]);
            continue;
        }
        bool has_lh = false;
        for (size_t k = 0; k < sz && !has_lh; k++) has_lh = type_calls[m[k]].hmm_likelihood != 0.0;
        if (has_lh) {
            std::sort(m, m + sz, [&](uint32_t a, uint32_t b) {
                const SVCall &x = type_calls[a], &y = type_calls[b];
                return x.cluster_size > y.cluster_size || (x.cluster_size == y.cluster_size && x.end - x.start > y.end - y.start);
            });
            for (size_t k = 0; k < sz; k++)
                if (type_calls[m[k]].hmm_likelihood != 0.0) { merged.push_back(type_calls[m[k]]); break; }
        } else {
            std::sort(m, m + sz, [&](uint32_t a, uint32_t b) {
                const SVCall &x = type_calls[a], &y = type_calls[b];
                return (x.end - x.start) > (y.end - y.start);
            });
            const size_t top = (size_t)std::max(1, (int)(sz * 0.2));
            SVCall rep = type_calls[m[top / 2]];
            rep.cluster_size = (int)sz;
            merged.push_back(rep);
        }
    }
}

void mergeSVs(std::vector<SVCall> &sv_calls, double epsilon, int min_pts, bool keep_noise, const std::string &json_filepath)
{
    (void)json_filepath;   // cluster JSON dump (saveClustersToJSON) is debug tooling outside the hot path
    printMessage("Merging SVs with DBSCAN, eps=" + std::to_string(epsilon) + ", min_pts=" + std::to_string(min_pts));
    if (sv_calls.size() < 2) return;
    const size_t initial = sv_calls.size();
    std::vector<SVCall> merged;
    DBSCAN dbscan(epsilon, min_pts);
    for (SVType t : {SVType::DEL, SVType::DUP, SVType::INV, SVType::INS, SVType::BND}) {
        std::vector<SVCall> type_calls;
        for (const SVCall &c : sv_calls) if (c.sv_type == t) type_calls.push_back(c);
        if (type_calls.size() < 2) {            // passes through untouched
            merged.insert(merged.end(), type_calls.begin(), type_calls.end());
            continue;
        }
        dbscan.fit(type_calls);                 // HIP kernels
        mergeTypeWithLabels(type_calls, dbscan.getClusters().data(), keep_noise, merged);
    }
    sv_calls = std::move(merged);
    printMessage("Merged " + std::to_string(initial) + " SV calls into " + std::to_string(sv_calls.size()) + " SV calls");
}

void mergeSVsMany(const std::vector<std::vector<SVCall> *> &sets, double epsilon, int min_pts, bool keep_noise, int threads)
{
    static const SVType kTypes[5] = {SVType::DEL, SVType::DUP, SVType::INV, SVType::INS, SVType::BND};   // sv_object.cpp:62-68
    const size_t n = sets.size();
    std::vector<std::vector<SVCall>> type_calls(n * 5);
    csvhost::parallel_for(n, threads, [&](size_t k) {
        if (sets[k]->size() < 2) return;                                      // mergeSVs returns early (:49-51)
        for (int t = 0; t < 5; t++)
            for (const SVCall &c : *sets[k]) if (c.sv_type == kTypes[t]) type_calls[k * 5 + (size_t)t].push_back(c);
    });
    std::vector<const std::vector<SVCall> *> fits;
    std::vector<size_t> fit_of(n * 5, SIZE_MAX);
    for (size_t q = 0; q < n * 5; q++) if (type_calls[q].size() >= 2) { fit_of[q] = fits.size(); fits.push_back(&type_calls[q]); }
    std::vector<std::vector<int>> labels;
    DBSCAN::fitBatch(fits, epsilon, min_pts, labels);
    csvhost::parallel_for(n, threads, [&](size_t k) {
        if (sets[k]->size() < 2) return;
        std::vector<SVCall> merged;
        for (int t = 0; t < 5; t++) {
            std::vector<SVCall> &tc = type_calls[k * 5 + (size_t)t];
            if (tc.size() < 2) { merged.insert(merged.end(), tc.begin(), tc.end()); continue; }    // passes through untouched (:85-92)
            mergeTypeWithLabels(tc, labels[fit_of[k * 5 + (size_t)t]].data(), keep_noise, merged);
        }
        *sets[k] = std::move(merged);
    });
    printMessage("Merged " + std::to_string(n) + " call sets with DBSCAN, eps=" + std::to_string(epsilon) + ", min_pts=" + std::to_string(min_pts));
}

// equal (start,end) neighbours after a (start, sv_type) sort collapse into the later one with summed
// cluster_size (reference sv_object.cpp:324-350)
void mergeDuplicateSVs(std::vector<SVCall> &sv_calls)
{
    const size_t initial = sv_calls.size();
    std::sort(sv_calls.begin(), sv_calls.end(), [](const SVCall &a, const SVCall &b) {
        return std::tie(a.start, a.sv_type) < std::tie(b.start, b.sv_type);
    });
    std::vector<SVCall> out;
    for (size_t i = 0; i < sv_calls.size(); i++) {
        SVCall &c = sv_calls[i];
        if (i > 0 && c.start == sv_calls[i - 1].start && c.end == sv_calls[i - 1].end) {
            c.cluster_size += sv_calls[i - 1].cluster_size;
            out.back() = c;
        } else {
            out.push_back(c);
        }
    }
    const size_t dropped = initial - out.size();
    sv_calls = std::move(out);
    if (dropped > 0) printMessage("Merged " + std::to_string(dropped) + " SV candidates with identical start and end positions");
}
