// common.hpp — internals shared by the gfx950 kernels and the C-ABI glue (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/csvgpu.h"

namespace csv {

constexpr int WAVE = 64;   // gfx950 wavefront

// BAM CIGAR op codes and flags (SAM spec §4.2)
enum : uint32_t { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };
enum : uint32_t { F_UNMAP = 0x4, F_SECONDARY = 0x100, F_QCFAIL = 0x200, F_DUP = 0x400, F_SUPP = 0x800 };
constexpr uint32_t REF_OPS = (1u << OP_M) | (1u << OP_D) | (1u << OP_N) | (1u << OP_EQ) | (1u << OP_X);   // consume reference
constexpr uint32_t QRY_OPS = (1u << OP_M) | (1u << OP_I) | (1u << OP_S) | (1u << OP_EQ) | (1u << OP_X);   // consume query
constexpr uint32_t ALN_OPS = (1u << OP_M) | (1u << OP_EQ) | (1u << OP_X);                                  // count toward depth
// words allocated behind an uploaded shard's CIGAR array, so that the kernels' 1 KiB chunk loads never need a bounds test
constexpr uint32_t CIGAR_PAD_WORDS = 512;
constexpr uint32_t GAP_OPS = REF_OPS & ~ALN_OPS;                                                             // D, N: on the reference, not in the depth
constexpr uint32_t QST_OPS = (1u << OP_M) | (1u << OP_I) | (1u << OP_EQ) | (1u << OP_X);                   // open the query interval

// ---------------------------------------------------------------------------------------------
// context

struct Timer {
    hipEvent_t a, b;
    int id;
    hipStream_t s;
    bool own_a = true, own_b = true;   // returned to the event pool when the timer is folded (an event may close one group and open the next)
};

struct Arena {                     // grow-only bump allocator over one device buffer; reset per entry point
    char  *base = nullptr;
    size_t cap = 0, used = 0;
};

}  // namespace csv

// turn-taking of the bandwidth-bound phases of several contexts on one device (csvgpu_gate_*)
struct csv_gate {
    std::mutex  mu;                     // held while a context queues its scan + depth pair
    hipStream_t stream = nullptr;       // the ONE stream the attached contexts' big kernels run on, back to back
    int         device = -1;
};

struct csv_ctx {
    int          device = 0;
    hipStream_t  stream = nullptr;
    hipStream_t  side = nullptr;            // carries the counters to the host beside the depth pass (jobs)
    bool         own_stream = false;
    csv::Arena   arena;                     // staged inputs / outputs of the host-pointer entry points
    csv::Arena   work;                      // workspace sized after the signature count is known
    void        *pinned = nullptr;          // small pinned host block for scalar read-backs
    size_t       pinned_cap = 0;
    int          timing = 0;                // 0 off, 1 every kernel group, 2 only the two bandwidth-bound groups (scan, depth)
    uint32_t     timer_tick = 0;            // jobs seen by the pair timers (level 2 times every fourth pair on a gate's stream)
    std::vector<csv::Timer> timers;         // recorded, not yet folded
    std::vector<hipEvent_t> event_pool;
    double       t_ms[CSV_K_COUNT] = {0};
    uint64_t     t_n[CSV_K_COUNT] = {0};
    std::string  err;
    int          n_cu = 256;
    // csvgpu_host_alloc / csvgpu_host_free: page-locking is slow (hundreds of microseconds), so freed blocks are kept for reuse
    std::vector<std::pair<void *, size_t>> host_live, host_pool;
    csv_gate    *gate = nullptr;
    char        *job_pin = nullptr;         // page-locked counter slots of the jobs in flight (csvgpu_chr_job_*)
    size_t       job_pin_next = 0;
    uint32_t     job_pin_busy = 0;          // bit i: slot i belongs to a job between begin and end / abort
    struct csv_split_state *split_state = nullptr;   // between csvgpu_split_order_begin and _finish
};

struct csv_shard {
    csv_reads d;                   // device pointers
    uint32_t  depth_len = 0;
    bool      owned = false;       // true: arrays hipMalloc'd by csvgpu_shard_upload
    uint32_t  cigar_pad = 0;       // allocated (zeroed) words behind d.cigar[n_cigar]: CIGAR_PAD_WORDS for the library's own uploads, 0 for wrapped arrays
    int       unsorted = -1;       // pos[] not non-decreasing: 1 / 0, or -1 while unknown (wrapped device arrays before their first scan)
    // per-read side arrays + per-chromosome outputs (device, owned by the shard)
    int32_t  *ref_end = nullptr, *q_start = nullptr, *q_end = nullptr, *pmax_end = nullptr;
    uint32_t *ckpt = nullptr;      // per-256-word reference checkpoints (scan -> depth)
    uint32_t *depth = nullptr;
    csv_sig  *sig_raw = nullptr;   uint64_t sig_cap = 0;
    csv_sig  *sig_sorted = nullptr;
    int32_t  *labels = nullptr;
    uint32_t *ord = nullptr;       // read permutation by pos when the shard is not coordinate-sorted
    char     *scratch = nullptr;   size_t scratch_cap = 0;   // sort / dbscan workspace
    uint64_t *counters = nullptr;  // device scalars (see ScanCounters) + bucket tables + the depth tiles' candidate ranges, zeroed together per chromosome
    uint64_t *tile_range = nullptr;   // inside `counters`
    uint64_t *scan_split = nullptr;   // the scan's work split for this device's grid (launch_scan_split, once per shard)
    int       form = 0;               // SCAN_FORM_*: chosen from the mean CIGAR words per read when the shard is created
    void     *depth_items = nullptr;  // depth_items_bytes(depth_len): the depth tiles' work lists (depth.hip)
    uint64_t *qhash = nullptr;        // [n_reads] std::hash<std::string> of every record's query name (csvgpu_shard_set_qname_hash), or null
    size_t    counters_bytes = 0;
};

namespace csv {

// device scalars written by the scan / depth kernels, read back once per chromosome
struct ScanCounters {
    // The three counters that every workgroup of a big kernel adds to sit in cache lines of their own: same-address device atomics
    // retire one after the other, and a workgroup's epilogue waits for its slot reservation on n_sig (returning) behind whatever
    // else is queued on that line.
    unsigned long long n_sig;        // all emitted signatures
    char               pad0[56];
    unsigned long long n_del;        // kind == DEL
    char               pad1[56];
    unsigned long long depth_sum;
    char               pad2[56];
    unsigned int       depth_nonzero;
    unsigned int       max_start;    // 0xffffffff if a signature start exceeded scan_start_limit(depth_len), else 0
    unsigned int       max_len;      // largest bucket of the ordering pass's most-significant-digit split if above BK_LOCAL_MAX, else 0 (scan epilogue)
    unsigned int       unsorted;     // != 0 if pos[] is not non-decreasing
    int                min_pts;      // written by the min_pts kernel
    int                pad;
    double             mean_cov;
    char               pad3[32];
};
static_assert(sizeof(ScanCounters) == 256, "the counters head the 256-byte block in front of the bucket tables");


#define CSV_HIP(ctx, call)                                                                       \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
            return CSV_EHIP;                                                                     \
        }                                                                                        \
    } while (0)

int   arena_reserve(csv_ctx *ctx, Arena &a, size_t bytes);       // grow (sync + realloc) if needed, then reset
void *arena_alloc(Arena &a, size_t bytes);                       // 256-B aligned slice, nullptr if exhausted
bool  timer_begin(csv_ctx *ctx, int id, hipStream_t s = nullptr);   // false: not recorded (timing off, or a group outside the selected level); s: the stream the group runs on (default: the context's)
void  timer_end(csv_ctx *ctx);
int   ensure_pinned(csv_ctx *ctx, size_t bytes);

struct TimerScope {
    csv_ctx *c;
    bool on;
    TimerScope(csv_ctx *ctx, int id, hipStream_t s = nullptr) : c(ctx), on(timer_begin(c, id, s)) {}
    ~TimerScope() { if (on) timer_end(c); }
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int bits_of(uint64_t x) { int b = 0; while (x) { b++; x >>= 1; } return b; }

// ---------------------------------------------------------------------------------------------
// launchers implemented in kernels/*.hip (all asynchronous on `s`)

// scan.hip
// What the scan can produce on the side for the passes behind it, so that nothing small has to run between the big kernels:
struct ScanExtras {
    uint64_t *tile_range = nullptr;   // [2 * n_tiles], zeroed: candidate read range of every depth tile (see DEPTH_TILE below); coordinate-sorted shards only
    uint32_t  n_tiles = 0;
    uint32_t *bucket_hist = nullptr;  // [BK_N], zeroed: counts of the ordering pass's most-significant-digit buckets; a bucket above BK_LOCAL_MAX is reported in cnt->max_len
    int       type_pos = -1, bucket_shift = 0;
};
// Forms of the scan (and of the depth pass's walk): a wave per read (long reads), or groups of 16 / 8 lanes per read (short reads)
enum { SCAN_FORM_WAVE = 0, SCAN_FORM_ROWS16 = 1, SCAN_FORM_ROWS8 = 2, SCAN_FORM_LANES = 3 };      // (LANES: the scan walks a read per lane; the depth walk takes groups of 16)
int scan_form_for(uint64_t n_reads, uint64_t n_cigar);          // by mean CIGAR words per read
void launch_cigar_scan(hipStream_t s, int n_cu, const csv_reads &d, uint32_t depth_len, uint32_t min_oplen,
                       uint32_t min_mapq, int emit, csv_sig *sig_out, uint64_t sig_cap,
                       int32_t *ref_end, int32_t *q_start, int32_t *q_end, uint32_t *ckpt, ScanCounters *cnt, const ScanExtras &extras = ScanExtras(),
                       const uint64_t *split = nullptr /* launch_scan_split's table for the same n_cu, shard and form */, int form = SCAN_FORM_WAVE,
                       uint32_t cigar_pad_words = 0 /* allocated words behind d.cigar[n_cigar] */);
size_t scan_split_bytes(int n_cu, uint64_t n_reads, int form);
void launch_scan_split(hipStream_t s, int n_cu, const csv_reads &d, uint64_t *split, int form);      // once per resident shard
void launch_validate_offsets(hipStream_t s, const uint64_t *cigar_off, uint64_t n_reads, uint64_t n_cigar, uint64_t max_words, uint32_t *bad /* zeroed; set to 1 */);
uint32_t scan_start_limit(uint32_t depth_len);   // exclusive bound on signature starts that the ordering keys are sized for
// depth.hip
void launch_prefix_max(hipStream_t s, const int32_t *in, int32_t *out, uint64_t n, void *tmp /* >= 4 KiB + n/1024*4 */);
size_t prefix_max_tmp_bytes(uint64_t n);
// One workgroup of the depth pass owns DEPTH_TILE positions. tile_range[2t] = ~(first candidate read), tile_range[2t + 1] = one past
// the last: all-zero means "no read" and both halves are maintained with atomicMax, by the scan itself (ScanExtras) or by
// launch_depth_ranges.
constexpr int DEPTH_TILE_SHIFT = 14, DEPTH_TILE = 1 << DEPTH_TILE_SHIFT;
static inline uint32_t depth_n_tiles(uint32_t depth_len) { return (uint32_t)(((uint64_t)depth_len + DEPTH_TILE - 1) / DEPTH_TILE); }
size_t depth_tiles_tmp_bytes(uint32_t depth_len);
// ranges from a prefix maximum of the read ends, for shards whose ranges the scan did not produce (unsorted: in `ord` order)
void launch_depth_ranges(hipStream_t s, const int32_t *pos_s, const int32_t *pmax_end, uint64_t n_reads, uint32_t depth_len, uint64_t *tile_range);
// ord == nullptr: reads are coordinate-sorted; otherwise the tile ranges index `ord`.
// ckpt: reference offset of the owning read at every CKPT_WORDS-word CIGAR boundary (written by launch_cigar_scan).
size_t depth_items_bytes(uint32_t depth_len);
void launch_depth_tiles(hipStream_t s, const csv_reads &d, const uint32_t *ord, const int32_t *ref_end, const uint32_t *ckpt,
                        uint32_t depth_len, uint32_t *depth, ScanCounters *cnt, const uint64_t *tile_range, uint32_t cigar_pad_words = 0,
                        void *items = nullptr, int form = SCAN_FORM_WAVE /* how the tiles walk their items: SCAN_FORM_* */);
// reference-offset checkpoints every CKPT_WORDS CIGAR words (scan.hip writes them, depth.hip starts its walks from them)
constexpr int CKPT_SHIFT = 6, CKPT_WORDS = 1 << CKPT_SHIFT;
static inline size_t ckpt_bytes(uint64_t n_cigar) { return ((n_cigar >> CKPT_SHIFT) + 2) * sizeof(uint32_t); }
void launch_min_pts(hipStream_t s, ScanCounters *cnt, double min_pts_pct);
void launch_depth_lookup(hipStream_t s, const uint32_t *depth, uint32_t depth_len, const uint32_t *pos, uint64_t n, int32_t *out);
// sort.hip
size_t radix_sort_tmp_bytes(uint64_t n);
// stable LSD sort of (key,val) pairs by key bits [0,key_bits). Both buffer pairs are clobbered; returns 1 when the
// result is in keys_out/vals_out, 0 when it is in keys_in/vals_in (even number of passes).
int  launch_radix_sort_u64(hipStream_t s, uint64_t *keys_in, uint32_t *vals_in, uint64_t *keys_out, uint32_t *vals_out,
                           uint64_t n, int key_bits, void *tmp);
int  launch_radix_sort_u64_devn(hipStream_t s, uint64_t *keys_in, uint32_t *vals_in, uint64_t *keys_out, uint32_t *vals_out, uint64_t n_bound,
                                const uint32_t *n_dev, int key_bits, void *tmp);
// bucket ordering: BK_N most-significant-digit buckets + one wave ranking each bucket; see sort.hip
constexpr uint32_t BK_BITS = 14, BK_N = 1u << BK_BITS, BK_LOCAL_MAX = 2048;
// (the bucket counts come from the scan: ScanExtras::bucket_hist)
void launch_bucket_sort(hipStream_t s, const csv_sig *sig_raw, uint64_t n, int type_pos, int shift, uint32_t *off /* the scan's bucket counts */, uint32_t *cur,
                        csv_sig *tmp, csv_sig *sig_sorted, uint32_t *start_out, uint32_t *end_out);
void launch_sig_make_keys(hipStream_t s, const csv_sig *sig, uint64_t n, int len_bits, int type_bit_pos,
                          uint64_t *keys, uint32_t *vals);
void launch_sig_fix_ties_gather(hipStream_t s, const csv_sig *sig_raw, const uint64_t *keys, const uint32_t *vals,
                                uint64_t n, csv_sig *sig_sorted, uint32_t *start_out, uint32_t *end_out);
void launch_iota_keys_u32(hipStream_t s, const uint32_t *k32, uint64_t n, uint64_t *keys, uint32_t *vals);
void launch_iota_keys_i32(hipStream_t s, const int32_t *k32, uint64_t n, uint64_t *keys, uint32_t *vals);   // order-preserving bias
void launch_check_sorted_u32(hipStream_t s, const uint32_t *k, uint64_t n, unsigned int *flag);
void launch_gather_u32(hipStream_t s, const uint32_t *src, const uint32_t *idx, uint64_t n, uint32_t *dst);
void launch_gather3_u32(hipStream_t s, const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *idx, uint64_t n, uint32_t *da, uint32_t *db, uint32_t *dc);
void launch_exclusive_sum_u32(hipStream_t s, uint32_t *data, uint64_t n, void *tmp);   // in place
size_t exclusive_sum_tmp_bytes(uint64_t n);
// dbscan.hip
size_t dbscan_tmp_bytes(uint64_t n);
// s,e sorted by start; oid = original index of each sorted position (nullptr: identity).
// min_pts read from *d_min_pts when d_min_pts != nullptr, else the immediate.
// split: positions [0,split) and [split,n) are two independent sets clustered side by side (split == n: one set;
// only honoured when oid == nullptr).
void launch_dbscan_iv_sorted(hipStream_t s, const uint32_t *start, const uint32_t *end, const uint32_t *oid,
                             uint64_t n, uint64_t split, double eps, int min_pts, const int *d_min_pts, int32_t *labels, void *tmp);
// many small sets (caller order, at most DBSCAN_IV_SMALL_MAX points each: larger segments are skipped) in one launch, one workgroup per set
constexpr uint32_t DBSCAN_IV_SMALL_MAX = 2048;
void launch_dbscan_iv_small_batched(hipStream_t s, const uint32_t *start, const uint32_t *end, const uint64_t *seg_off, uint64_t n_seg, double eps,
                                    int min_pts, int32_t *labels);
// splitorder.hip — the node order of the reference's per-chromosome qname hash map, all contigs of a batch per launch
constexpr uint32_t SO_MAX_CONTIGS = 32;
struct SplitOrderTab {                       // passed to the kernels by value
    uint32_t A = 0;                          // contigs in this table
    uint64_t blk_off[SO_MAX_CONTIGS + 1];    // compaction: first 1024-record block of each contig
    uint64_t work_off[SO_MAX_CONTIGS + 1];   // epoch: first work item (node present in the epoch) of each active contig
    uint32_t nbase[SO_MAX_CONTIGS + 1];      // global index of each contig's first node
    uint64_t rev_off[SO_MAX_CONTIGS];        // epoch: work items of the active contigs BEHIND this one (the sort key's contig part)
    uint32_t m_old[SO_MAX_CONTIGS];          // epoch: nodes that were in the list when the epoch began (the rest were inserted during it)
    uint64_t n_reads[SO_MAX_CONTIGS];
    const uint16_t *flag[SO_MAX_CONTIGS];
    const uint8_t *mapq[SO_MAX_CONTIGS];
    const uint64_t *qhash[SO_MAX_CONTIGS];
};
struct csv_split_survivor { uint32_t contig, pos, rec; };
void launch_so_count(hipStream_t s, const SplitOrderTab &tab, uint32_t n_blocks, uint32_t min_mapq, uint32_t *blk_cnt);
void launch_so_scatter(hipStream_t s, const SplitOrderTab &tab, uint32_t n_blocks, uint32_t min_mapq, const uint32_t *blk_off, uint64_t *node_hash,
                       uint32_t *node_rec);
void launch_so_mint(hipStream_t s, const SplitOrderTab &tab, uint64_t M, uint32_t B, const uint64_t *node_hash, const uint32_t *list, uint32_t *minT);
void launch_so_keys(hipStream_t s, const SplitOrderTab &tab, uint64_t M, uint32_t B, int w, const uint64_t *node_hash, const uint32_t *list,
                    const uint32_t *minT, uint64_t *keys, uint32_t *vals);
void launch_so_setlist(hipStream_t s, const SplitOrderTab &tab, uint64_t M, const uint32_t *vals, uint32_t *list);
void launch_so_survivors(hipStream_t s, const SplitOrderTab &tab, uint64_t n_nodes, const uint64_t *node_hash, const uint32_t *node_rec, const uint32_t *list,
                         const uint64_t *supp_hash, uint64_t n_supp, csv_split_survivor *out, uint64_t cap, unsigned long long *count);
// the first epochs (while nodes and buckets fit the LDS) in one launch, one workgroup per contig
constexpr uint32_t SO_SMALL_B = 5087, SO_SMALL_EPOCHS = 12;
struct SplitSmallHost {
    uint32_t A = 0, n_epochs = 0;
    uint32_t nbase[SO_MAX_CONTIGS + 1] = {0};
    int32_t  k_last[SO_MAX_CONTIGS] = {0};
    uint32_t first[SO_SMALL_EPOCHS + 1] = {0};
    uint32_t B[SO_SMALL_EPOCHS] = {0};
};
void launch_so_small_epochs(hipStream_t s, const SplitSmallHost &h, const uint64_t *node_hash, uint32_t *list);
// the last epochs for the survivors only (splitorder.hip): level j = the contig's last epoch minus j
constexpr uint32_t SO_TAIL_MAX = 3;
struct SplitTailHost {
    uint32_t A = 0, D = 0;
    int      wv = 0, wa = 0;
    uint32_t nbase[SO_MAX_CONTIGS + 1] = {0};
    uint32_t B[SO_TAIL_MAX][SO_MAX_CONTIGS] = {{0}};
    uint32_t F[SO_TAIL_MAX][SO_MAX_CONTIGS] = {{0}};
    uint32_t boff[SO_TAIL_MAX][SO_MAX_CONTIGS + 1] = {{0}};
};
size_t st_filter_bytes();
void launch_st_survivors(hipStream_t s, const SplitTailHost &h, uint32_t n_nodes, const uint64_t *node_hash, const uint64_t *supp_hash, uint64_t n_supp, uint32_t *filter /* st_filter_bytes(), zeroed */,
                         uint8_t *is_surv, uint32_t *bitmap);
void launch_st_member(hipStream_t s, const SplitTailHost &h, uint32_t n_nodes, int j, const uint64_t *node_hash, const uint32_t *bitmap_prev, uint32_t *bitmap_next,
                      uint32_t *set, unsigned int *count);
void launch_st_inverse(hipStream_t s, const SplitTailHost &h, uint32_t n_nodes, int j_last, const uint32_t *list, uint32_t *prevrank);
void launch_st_mint(hipStream_t s, const SplitTailHost &h, int j, const uint32_t *set, uint32_t n, const uint32_t *n_dev, const uint64_t *node_hash, const uint32_t *prevrank,
                    uint32_t *minT);
void launch_st_keys(hipStream_t s, const SplitTailHost &h, int j, const uint32_t *set, uint32_t n, const uint32_t *n_dev, const uint64_t *node_hash, const uint32_t *prevrank,
                    const uint32_t *minT, uint64_t *keys, uint32_t *vals);
void launch_st_rank(hipStream_t s, const uint32_t *vals, uint32_t n, const uint32_t *n_dev, uint32_t *prevrank);
void launch_st_emit(hipStream_t s, const SplitTailHost &h, const uint32_t *vals, uint32_t n, const uint32_t *n_dev, const uint8_t *is_surv, const uint32_t *node_rec,
                    csv_split_survivor *out, uint64_t cap, unsigned long long *count);
void launch_so_supp(hipStream_t s, const SplitOrderTab &tab, uint32_t n_blocks, uint32_t min_mapq, uint64_t *supp_hash, unsigned int *count);
// dbscan1d.hip
void launch_dbscan_1d_batched(hipStream_t s, const int32_t *pts, const uint64_t *seg_off, uint64_t n_seg,
                              double eps, int min_pts, int32_t *labels, unsigned int *too_large_flag);
constexpr uint32_t DBSCAN1D_MAX_SEG = 512;   // larger segments take the generic sorted path
size_t dbscan1d_big_tmp_bytes(uint64_t n);
// pts_sorted ascending (int order); oid = original index per sorted position
void launch_dbscan_1d_big(hipStream_t s, const int32_t *pts_sorted, const uint32_t *oid, uint64_t n, double eps,
                          int min_pts, int32_t *labels, void *tmp);
// hmm.hip
void launch_window_log2(hipStream_t s, const uint32_t *depth, uint32_t depth_len, const uint32_t *rs, const uint32_t *re,
                        const int32_t *ss, const uint64_t *win_off, uint64_t n_regions, uint64_t n_windows,
                        double mean_cov, double *log2_cov, uint32_t *ws, uint32_t *we);
size_t viterbi_tmp_bytes(uint64_t n_obs, uint64_t n_seq);
void launch_viterbi(hipStream_t s, const csv_hmm &hmm, const double *o1, const double *o2, const double *pfb,
                    const uint64_t *seq_off, uint64_t n_seq, uint64_t n_obs, int32_t *states, double *loglik, void *tmp);

}  // namespace csv
