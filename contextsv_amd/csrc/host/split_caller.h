// split_caller.h — host mirror of SVCaller::findSplitSVSignatures (src/sv_caller.cpp:68-504): split-alignment evidence ->
// SPLITDIST1 insertion / unknown candidates and SPLIT (or inversion) dummy calls for the copy-number pass.
//
// What moved to the GPU: the per-record CIGAR walk of the reference's third BAM pass (getAlignmentReadPositions +
// bam_endpos, :152, :162) — the scan kernel already produced ref_end / q_start / q_end for every record — and the six
// DBSCAN1D(100, 5) fits per overlap group (:270-372), which are collected for ALL groups of a chromosome and solved by
// one batched launch for the whole genome (DBSCAN1D::fitBatch -> csvgpu_dbscan_1d). What stays here: the order-defining
// steps (SURVEY §7 hard part 2) — the iteration order of the reference's qname-keyed unordered_map, replayed exactly by
// umap_order.h instead of building 6e5 string-keyed nodes per chromosome; the (unbalanced) interval tree that order builds; the
// overlap groups, medians and votes — one host thread per contig, contigs being independent.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "sv_object.h"

// One alignment record as the split pass sees it (file order).
struct SplitRecord {
    int32_t  tid;
    int32_t  pos;        // 0-based
    uint16_t flag;
    uint8_t  mapq;
    int32_t  ref_end;    // bam_endpos           } from the scan kernel
    int32_t  q_start;    // query_start          } (csv_chr_result.ref_end / q_start / q_end
    int32_t  q_end;      // query_end            }  or csvgpu_aln_intervals)
};

// Where the scan kernel's per-record intervals come from when they are not handed over as whole arrays: only the records that take
// part in a group (primaries with a supplementary record, and those records) are ever asked for — a few per cent of a contig — and
// for all contigs in one request.
struct IntervalSource {
    virtual ~IntervalSource() = default;
    // contig which[k] (index into the `contigs` vector): records rec[rec_off[k] .. rec_off[k+1]) (ascending, distinct) -> their
    // ref_end / q_start / q_end at the same positions of the three outputs
    virtual void gather(const std::vector<size_t> &which, const std::vector<uint32_t> &rec, const std::vector<uint64_t> &rec_off,
                        int32_t *ref_end, int32_t *q_start, int32_t *q_end) const = 0;
};

// The surviving primaries of several contigs, each in the iteration order of the reference's per-chromosome qname map, computed
// somewhere else than on this host thread (the device: csvgpu_split_order behind ShardOrderSource in sv_caller.cpp).
struct SplitOrderSource {
    virtual ~SplitOrderSource() = default;
    // which[k] indexes the `contigs` vector of findSplitSVSignatures; recs[k] = record indices of contig which[k]'s primaries whose name
    // hash is in supp_hash (sorted, distinct), in iteration order. Only asked for contigs with SplitContig::unique_names.
    virtual void survivors(const std::vector<size_t> &which, int min_mapq, const std::vector<uint64_t> &supp_hash, std::vector<std::vector<uint32_t>> &recs) const = 0;
    // Optional head start: called with the same `which` before the supplementary records have been collected, so that whatever does not
    // depend on them (on the device: the nodes and all but the last epochs of every map, csvgpu_split_order_begin) runs meanwhile.
    // `complete`: `which` holds every contig of the run that has records — every supplementary record of the run is one of theirs, so the
    // source may take the supplementary hashes from its own copy of the records (csvgpu_split_order_begin_self: nothing then waits for
    // the collection) and ignore the ones survivors() is given.
    virtual void begin(const std::vector<size_t> &which, int min_mapq, bool complete) const { (void)which; (void)min_mapq; (void)complete; }
};

struct SplitParams {
    int min_mapq = 20;        // sv_caller.h:72
    double eps = 100;         // DBSCAN1D(100, 5) at sv_caller.cpp:270
    int min_pts = 5;
    int min_length = 2000;    // :243
    int max_length = 1000000; // :244
    const IntervalSource *intervals = nullptr;        // for contigs given without ref_end / q_start / q_end arrays
    const SplitOrderSource *device_order = nullptr;   // where contigs with unique_names get their iteration order from (nullptr: replayed on the host)
    int threads = 0;          // host threads over contigs (0: one per contig, at most the hardware's); the result does not depend on it
};

// The records of one contig (one tid), struct of arrays, file order — what a decoded shard holds on the host plus the scan
// kernel's intervals. Query names are needed for two things only: their std::hash<std::string> value (qhash: it fixes where the
// reference's unordered_map puts a read, see umap_order.h) and equality. Equality comes from the names themselves (name_bytes /
// name_off) or from run-wide dictionary ids (name_id: equal id <=> equal name) — at least one of the two must be given.
struct SplitContig {
    int32_t tid = 0;
    uint64_t n = 0;
    const int32_t *pos = nullptr;
    const uint16_t *flag = nullptr;
    const uint8_t *mapq = nullptr;
    const int32_t *ref_end = nullptr, *q_start = nullptr, *q_end = nullptr;   // whole arrays, or null: SplitParams::intervals is asked for the few records that matter
    const uint64_t *qhash = nullptr;
    const uint64_t *name_id = nullptr;
    const char *name_bytes = nullptr;
    const uint64_t *name_off = nullptr;      // [n + 1]
    bool unique_names = false;               // established when the contig was staged: no two non-supplementary records share a name hash
    const uint64_t *file_idx = nullptr;      // optional: position of each record in the file (default: contigs in the order given, records in array order)
};

// Fills sv_calls[target_names[tid]] for every contig that has primary alignments, like the reference. Contigs must be given in
// file order (or carry file_idx): a read's supplementary records are visited in file order.
void findSplitSVSignatures(const std::vector<SplitContig> &contigs, const std::vector<std::string> &target_names, const SplitParams &params,
                           std::unordered_map<std::string, std::vector<SVCall>> &sv_calls);

// The same in two halves: prepare() needs no alignment intervals (primaries / supplementaries, the map's iteration order, survivors) and may
// run while the scan kernel's outputs are still being produced; finish() does the rest. `contigs` and `target_names` must outlive the object.
class SplitPass {
public:
    SplitPass(const std::vector<SplitContig> &contigs, const std::vector<std::string> &target_names, const SplitParams &params);
    ~SplitPass();
    void prepare();
    void finish(std::unordered_map<std::string, std::vector<SVCall>> &sv_calls);
    // between the two: contigs (indices into the constructor's list) whose alignment intervals exist already — their interval gather and
    // overlap groups now instead of inside finish(); any subset, any number of times, after prepare()
    void finishEarly(const std::vector<size_t> &contig_ids);
    // finish() for a subset of the contigs (those of it not done yet): a contig's calls depend on nothing outside the contig, so any
    // partition of the contigs over calls gives the calls one finish() gives; finish() afterwards does what is left
    void finishFor(const std::vector<size_t> &contig_ids, std::unordered_map<std::string, std::vector<SVCall>> &sv_calls);
private:
    struct Impl;
    std::unique_ptr<Impl> p;
    bool prepared = false;
};

// Same, from records in file order (any mix of tids) and their names as strings.
void findSplitSVSignatures(const std::vector<SplitRecord> &records, const std::vector<std::string> &qnames,
                           const std::vector<std::string> &target_names, const SplitParams &params,
                           std::unordered_map<std::string, std::vector<SVCall>> &sv_calls);
