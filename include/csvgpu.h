/*
 * csvgpu.h — C-ABI of the MI355X (gfx950) hot path of ContextSV.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types,
 * no exceptions across it. Every entry point names the reference seam
 * (WGLab/ContextSV, file:line relative to the reference root) it replaces.
 *
 * Conventions
 *   - every call returns CSV_OK (0) or a negative csv_status; nothing throws.
 *   - the caller owns every buffer; the library owns only the opaque context
 *     (device workspace, stream, per-kernel timers).
 *   - one csv_ctx per GPU, used from one host thread at a time.
 *   - entry points without a suffix take HOST pointers (they stage H2D, run the
 *     kernels, copy the result back). Entry points ending in `_dev` take DEVICE
 *     pointers (hipMalloc'd / torch CUDA tensors), run asynchronously on the
 *     context's stream and are what the benchmark times with inputs resident
 *     in HBM.
 *   - there is no CPU fallback anywhere behind this ABI: without a usable HIP
 *     device csvgpu_create() fails with CSV_ENODEV and nothing else can be called.
 */
#ifndef CSVGPU_H
#define CSVGPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CSVGPU_ABI_VERSION 3   /* 3: csvgpu_split_order_begin_self; 2: CSV_K_SPLIT_ORDER, csvgpu_split_order_begin / _finish, the job / gate / batch entry points; the test hook left the product library */

typedef struct csv_ctx csv_ctx;

typedef enum csv_status {
    CSV_OK        =  0,
    CSV_EINVAL    = -1,  /* bad argument (null pointer, eps outside [0,1), min_pts < 1, ...) */
    CSV_ENODEV    = -2,  /* no usable HIP device / device ordinal out of range */
    CSV_ENOMEM    = -3,  /* device or host allocation failed */
    CSV_EHIP      = -4,  /* a HIP runtime call failed; see csvgpu_last_error() */
    CSV_ECAPACITY = -5   /* caller's output capacity too small; required count is returned */
} csv_status;

/* ---- alignment records of one shard (normally: one chromosome), struct of arrays ----
 * Mirrors exactly the fields of htslib's bam1_core_t that the reference reads on this
 * path (sv_caller.cpp:526,541-545; cnv_caller.cpp:491-502): core.pos (0-based), core.flag,
 * core.qual, core.tid, core.n_cigar and the packed CIGAR words (len<<4 | op, BAM op codes
 * M0 I1 D2 N3 S4 H5 P6 =7 X8). Records are in file (coordinate-sorted) order. */
typedef struct csv_reads {
    uint64_t        n_reads;
    uint64_t        n_cigar;    /* total CIGAR words == cigar_off[n_reads] */
    const int32_t  *pos;        /* [n_reads] 0-based leftmost reference position */
    const uint16_t *flag;       /* [n_reads] BAM FLAG */
    const uint8_t  *mapq;       /* [n_reads] MAPQ */
    const int32_t  *tid;        /* [n_reads] reference id (carried, not interpreted by kernels; may be NULL) */
    const uint64_t *cigar_off;  /* [n_reads+1] first CIGAR word of each read; non-decreasing, cigar_off[n_reads] <= n_cigar, < 2^31 words
                                 * per read — entry points return CSV_EINVAL otherwise (nothing reaches the device) */
    const uint32_t *cigar;      /* [n_cigar] packed CIGAR words */
} csv_reads;

/* One SV signature = one SVCall emitted by SVCaller::processCIGARRecord
 * (sv_caller.cpp:566-643). 16 bytes, written with a single store on device.
 *   kind: 0 = CIGARINS (I op), 1 = CIGARDEL (D op), 2 = CIGARCLIP (S op) — the SVDataType
 *         bit the reference sets (sv_types.h:60-63). SVType is DEL for kind 1, INS otherwise.
 *   qpos: query offset of the op (reference `query_pos`, sv_caller.cpp:546,653) — the host
 *         needs it to cut the inserted sequence when op_len == 50 (sv_caller.cpp:589,624). */
typedef struct csv_sig {
    uint32_t start;      /* 1-based, = pos+1 (sv_caller.cpp:584,619,636) */
    uint32_t end;        /* = start + op_len - 1 (sv_caller.cpp:585,620,637) */
    uint32_t read;       /* index of the emitting read inside the shard */
    uint32_t qpos_kind;  /* (qpos << 2) | kind */
} csv_sig;
#define CSV_SIG_KIND(s) ((s).qpos_kind & 3u)
#define CSV_SIG_QPOS(s) ((s).qpos_kind >> 2)
enum { CSV_KIND_INS = 0, CSV_KIND_DEL = 1, CSV_KIND_CLIP = 2 };

/* 6-state PennCNV-style HMM parameters = the fields ReadCHMM fills (khmm.cpp:395-553,
 * struct CHMM khmm.h:14-32). Row-major A[i*6+j] = P(i -> j). */
typedef struct csv_hmm {
    double A[36];
    double pi[6];
    double B1_mean[6];
    double B1_sd[6];
    double B1_uf;
    double B2_mean[5];
    double B2_sd[5];
    double B2_uf;
} csv_hmm;

/* Kernel ids for csvgpu_timing_get(). */
typedef enum csv_kernel_id {
    CSV_K_CIGAR_SCAN = 0,   /* signature emission + alignment intervals */
    CSV_K_DEPTH      = 1,   /* tile-owner depth map (+ sum / non-zero count) */
    CSV_K_SORT       = 2,   /* all radix-sort passes + tie fix-up */
    CSV_K_DBSCAN     = 3,   /* neighbour count + union + rank + label (interval metric) */
    CSV_K_DBSCAN1D   = 4,   /* batched 1-D DBSCAN */
    CSV_K_WINDOW     = 5,   /* window log2 coverage */
    CSV_K_VITERBI    = 6,   /* emissions + Viterbi DP + backtrack */
    CSV_K_MISC       = 7,   /* memsets, small scans, partition */
    CSV_K_SPLIT_ORDER = 8,  /* hash-map node order of the split-read pass (compaction + per-epoch sorts + survivors) */
    CSV_K_COUNT      = 9
} csv_kernel_id;

/* ------------------------------------------------------------------------------------------ */
/* context                                                                                      */

/* Create a context on HIP device `device_ordinal`. `stream` may be NULL (the context creates
 * its own non-blocking stream) or an existing hipStream_t (e.g. torch's current stream) that
 * every kernel of this context is then launched on. Returns NULL on failure (no device, ...);
 * csvgpu_last_error(NULL) then describes why. */
csv_ctx    *csvgpu_create(int device_ordinal, void *stream);
/* The same with the context's own stream at the device's LOWEST priority: for work that should fill the gaps other contexts leave (the
 * split-read pass's ordering kernels beside the lanes' bandwidth-bound pairs) instead of sharing the compute units with them. */
csv_ctx    *csvgpu_create_background(int device_ordinal);
void        csvgpu_destroy(csv_ctx *ctx);
int         csvgpu_abi_version(void);
const char *csvgpu_last_error(const csv_ctx *ctx);
/* Block until everything queued on the context's stream has finished. */
int         csvgpu_synchronize(csv_ctx *ctx);
/* Per-kernel timers: when enabled, each launch group is bracketed by HIP events recorded on
 * the stream the group runs on (the context's; for the scan + depth pair of a job behind a gate, the gate's — there the
 * pair shares three events, and at on = 2 only every fourth pair is timed); csvgpu_timing_get() synchronises and returns the accumulated device
 * time and the number of launch groups since the last reset. on = 1: every group; on = 2: only CSV_K_CIGAR_SCAN and
 * CSV_K_DEPTH (an event is a barrier packet in the queue, ~5 us of idle device each); on = 3: those two groups, every pair timed also
 * behind a gate (for runs whose launches differ in size, where a sampled quarter would not match the bytes); on = 0: off. */
int         csvgpu_timing_enable(csv_ctx *ctx, int on);
int         csvgpu_timing_reset(csv_ctx *ctx);
int         csvgpu_timing_get(csv_ctx *ctx, int kernel_id, double *total_ms, uint64_t *launches);

/* ------------------------------------------------------------------------------------------ */
/* host-pointer entry points (the seams of SURVEY.md §8b)                                       */

/* Replaces SVCaller::findCIGARSVs + processCIGARRecord + addSVCall for one chromosome
 * (sv_caller.cpp:506-537, 539-661; sv_object.cpp:22-33).
 *   filter: flag & (SECONDARY|UNMAP|DUP|QCFAIL|SUPPLEMENTARY) or mapq < min_mapq => skipped (:526).
 *   depth_len = pos_depth_map.size() (chromosome length + 1); soft clips with pos+1 >= depth_len are
 *   skipped together with their cursor update (:602-604 `continue`).
 *   min_oplen = 50 (:566), min_mapq = 20 (sv_caller.h:72) in the reference.
 * Output order == order of the reference's chr_sv_calls vector after all addSVCall()s: ascending
 * (start,end); equal (start,end) in REVERSE emission order (std::lower_bound insert, sv_object.cpp:31).
 * *n_out: in = capacity of `out` in records, out = number of signatures. If the capacity is too
 * small CSV_ECAPACITY is returned and *n_out holds the required count. */
int csvgpu_cigar_scan(csv_ctx *ctx, const csv_reads *reads, uint32_t depth_len,
                      uint32_t min_oplen, uint8_t min_mapq, csv_sig *out, uint64_t *n_out);

/* Replaces SVCaller::getAlignmentReadPositions (sv_caller.cpp:663-690) and htslib bam_endpos for
 * every record of the shard (used by findSplitSVSignatures, sv_caller.cpp:152,162):
 *   q_start = query offset of the first M/I/=/X op (0 if none), q_end = sum of M,I,S,=,X lengths,
 *   ref_end = pos + reference length of the CIGAR (pos+1 when that length is 0 or the read is unmapped). */
int csvgpu_aln_intervals(csv_ctx *ctx, const csv_reads *reads,
                         int32_t *ref_end, int32_t *q_start, int32_t *q_end);

/* Replaces the per-chromosome body of CNVCaller::calculateMeanChromosomeCoverage
 * (cnv_caller.cpp:488-543): depth[p] (1-based p, depth_len = chr_len+1 entries, entry 0 stays 0)
 * counts M/=/X bases of every record without UNMAP|SECONDARY|QCFAIL|DUP (no mapq filter,
 * supplementary included); bases at p >= depth_len are skipped. *sum = sum of depth, *nonzero =
 * #positions with depth > 0 (mean coverage = sum/nonzero, :534-538). `depth` may be NULL when only
 * the two scalars are wanted. */
int csvgpu_depth(csv_ctx *ctx, const csv_reads *reads, uint32_t depth_len,
                 uint32_t *depth, uint64_t *sum, uint32_t *nonzero);

/* Replaces DBSCAN::fit + getClusters (dbscan.cpp:9-81, dbscan.h:13-18): labels[i] for the i-th
 * interval in CALLER order: -2 noise, 0..k-1 cluster id, identical to the reference's sequential
 * visit-order labels. Metric = 1 - min reciprocal overlap evaluated in the same IEEE double
 * expression (dbscan.cpp:74-79). Requires 0 <= eps < 1 and min_pts >= 1 (CSV_EINVAL otherwise). */
int csvgpu_dbscan_iv(csv_ctx *ctx, const uint32_t *start, const uint32_t *end, uint64_t n,
                     double eps, int32_t min_pts, int32_t *labels);

/* DBSCAN::fit for a batch of independent interval sets in one call — the per-type fits of mergeSVs (sv_object.cpp:62-94) for every
 * contig of a run at once (the two final merges, sv_caller.cpp:907, :925): set s = intervals [seg_off[s], seg_off[s+1]) in CALLER
 * order, labels per set exactly as csvgpu_dbscan_iv gives them. Sets of up to 2048 intervals share one launch (one workgroup each,
 * all pairs in LDS); larger ones take the windowed path one by one. */
int csvgpu_dbscan_iv_batch(csv_ctx *ctx, const uint32_t *start, const uint32_t *end, const uint64_t *seg_off, uint64_t n_seg,
                           double eps, int32_t min_pts, int32_t *labels);

/* Replaces DBSCAN1D::fit + getClusters for a batch of independent point sets
 * (dbscan1d.cpp:8-70; the six fits per overlap group of sv_caller.cpp:270-372 become one call):
 * segment s = pts[seg_off[s] .. seg_off[s+1]); labels in caller order per segment. Metric
 * |a-b| (int) <= eps (double). Requires eps >= 0, min_pts >= 1. */
int csvgpu_dbscan_1d(csv_ctx *ctx, const int32_t *pts, const uint64_t *seg_off, uint64_t n_seg,
                     double eps, int32_t min_pts, int32_t *labels);

/* Replaces the window loop of CNVCaller::querySNPRegion (cnv_caller.cpp:76-113) for a batch of
 * regions: region r has sample_size[r] windows; window i sums depth[(uint32)(start + i*step + j)]
 * for integer j < step (step = (end-start+1)/sample_size as double), stops past `end`, skips
 * positions >= depth_len, and yields log2((sum/count)/mean_cov) (sum==0 -> 1e-9; count==0 -> 0.0).
 * win_off[r] (n_regions+1 entries) = first output window of region r. Also returns the window
 * bounds the reference uses as map keys (:80-81). */
int csvgpu_window_log2(csv_ctx *ctx, const uint32_t *depth, uint32_t depth_len,
                       const uint32_t *region_start, const uint32_t *region_end,
                       const int32_t *sample_size, const uint64_t *win_off, uint64_t n_regions,
                       double mean_cov, double *log2_cov, uint32_t *win_start, uint32_t *win_end);

/* Replaces testVit_CHMM / ViterbiLogNP_CHMM (khmm.cpp:28-56, 225-393) for a batch of observation
 * sequences: sequence s = observations [seq_off[s], seq_off[s+1]); o2 == -1 means "no BAF".
 * states[] are 1..6 (the reference's returned vector, dummy 0th element already dropped),
 * loglik[s] = max final delta. Sequences of length 0 get loglik -1e11 (VITHUGE) and no states. */
int csvgpu_viterbi(csv_ctx *ctx, const csv_hmm *hmm, const double *o1, const double *o2,
                   const double *pfb, const uint64_t *seq_off, uint64_t n_seq,
                   int32_t *states, double *loglik);

/* ------------------------------------------------------------------------------------------ */
/* device-pointer entry points (inputs resident in HBM; asynchronous on the context's stream)   */

/* Upload a shard once; the returned handle keeps pos/flag/mapq/cigar_off/cigar resident in HBM
 * together with the per-read side arrays the kernels produce (ref_end, q_start, q_end). */
typedef struct csv_shard csv_shard;
csv_shard *csvgpu_shard_upload(csv_ctx *ctx, const csv_reads *host_reads, uint32_t depth_len);
/* Wrap arrays that already live in HBM (e.g. torch tensors); nothing is copied or owned. */
csv_shard *csvgpu_shard_wrap_dev(csv_ctx *ctx, const csv_reads *dev_reads, uint32_t depth_len);
void       csvgpu_shard_free(csv_ctx *ctx, csv_shard *shard);

/* Result of one chromosome's device pipeline; pointers are device memory owned by the shard
 * and stay valid until the next csvgpu_chr_pipeline_dev() on the same shard or its free. */
typedef struct csv_chr_result {
    uint64_t        n_sig;        /* signatures emitted (all kinds) */
    uint64_t        n_del;        /* of which SVType DEL */
    uint64_t        n_ins;        /* of which SVType INS (CIGARINS + CIGARCLIP) */
    uint64_t        depth_sum;
    uint32_t        depth_nonzero;
    int32_t         min_pts;      /* ceil(mean_cov * min_pts_pct) or 5 (sv_caller.cpp:723-728) */
    double          mean_cov;
    const csv_sig  *sig_del;      /* [n_del] the DEL calls of chr_sv_calls, in vector order (std::copy_if, sv_object.cpp:80) */
    const csv_sig  *sig_ins;      /* [n_ins] the INS calls (CIGARINS + CIGARCLIP), in vector order */
    const int32_t  *label_del;    /* [n_del] DBSCAN labels of sig_del */
    const int32_t  *label_ins;    /* [n_ins] DBSCAN labels of sig_ins */
    const uint32_t *depth;        /* [depth_len] */
    const int32_t  *ref_end;      /* [n_reads] */
    const int32_t  *q_start;      /* [n_reads] */
    const int32_t  *q_end;        /* [n_reads] */
} csv_chr_result;

/* The whole per-chromosome device path of SVCaller::processChromosome up to cluster labels
 * (sv_caller.cpp:692-745 with cnv_caller.cpp:488-543 in front): CIGAR scan -> depth map and mean
 * coverage -> min_pts -> ordering -> per-type interval DBSCAN. Representative selection of
 * mergeSVs (sv_object.cpp:121-244) is host code above this ABI.
 * min_pts_pct <= 0 selects the fixed min_pts = 5 (sv_caller.cpp:724-727). */
int csvgpu_chr_pipeline_dev(csv_ctx *ctx, csv_shard *shard, uint32_t min_oplen, uint8_t min_mapq,
                            double eps, double min_pts_pct, csv_chr_result *result);

/* Copy the signatures (sig_del followed by sig_ins: n_sig records) and their labels (label_del followed by
 * label_ins) of the last csvgpu_chr_pipeline_dev() on `shard` to host memory with one synchronisation. */
int csvgpu_chr_fetch(csv_ctx *ctx, csv_shard *shard, const csv_chr_result *result, csv_sig *host_sig, int32_t *host_labels);

/* csvgpu_chr_pipeline_dev + csvgpu_chr_fetch in one call with a single final synchronisation: when the chromosome's
 * n_sig <= capacity the signatures and labels are copied to host_sig / host_labels behind the last kernel and the scalars
 * ride in the same wait. With n_sig > capacity nothing is copied and CSV_ECAPACITY is returned; *result is valid (n_sig
 * says how much room is needed) and csvgpu_chr_fetch() can still fetch. Pass buffers from csvgpu_host_alloc() so that the
 * copies are true asynchronous DMA. */
int csvgpu_chr_pipeline_fetch(csv_ctx *ctx, csv_shard *shard, uint32_t min_oplen, uint8_t min_mapq, double eps, double min_pts_pct,
                              csv_chr_result *result, csv_sig *host_sig, int32_t *host_labels, uint64_t capacity);

/* The same per-chromosome path as a job in three steps, for callers that keep the device busy across their own turn-around:
 *   begin   queues the CIGAR scan, the bucket counts and (for shards whose sortedness is known) the depth pass;
 *   cluster waits for the signature count, queues ordering + DBSCAN and the copies into host_sig / host_labels (capacity records
 *           each; page-locked memory from csvgpu_host_alloc) — and returns without waiting;
 *   end     waits for the job, fills *result, frees the job; CSV_ECAPACITY when the host buffers were too small (the results
 *           are then on the device: csvgpu_chr_fetch).
 * Between cluster(i) and end(i) the caller may begin(i + 1) on the same context — also on the same shard, everything being
 * ordered by the context's stream: the next chromosome's scan then starts on the device the moment this one's last copy is done.
 * csvgpu_chr_pipeline_fetch is begin + cluster + end. */
typedef struct csv_job csv_job;
/* At most CSV_MAX_JOBS jobs may be open (between begin and end / abort) on one context: each holds one of the context's page-locked
 * counter slots. One more csvgpu_chr_job_begin returns NULL ("more than CSV_MAX_JOBS jobs open"). */
#define CSV_MAX_JOBS 16
csv_job *csvgpu_chr_job_begin(csv_ctx *ctx, csv_shard *shard, uint32_t min_oplen, uint8_t min_mapq, double min_pts_pct);
int csvgpu_chr_job_cluster(csv_ctx *ctx, csv_job *job, double eps, csv_sig *host_sig, int32_t *host_labels, uint64_t capacity);
int csvgpu_chr_job_end(csv_ctx *ctx, csv_job *job, csv_chr_result *result);
/* Give up a job after csvgpu_chr_job_cluster (or anything between begin and end) failed: waits for what the job queued, frees it,
 * and leaves csvgpu_last_error() as the failure left it (csvgpu_chr_job_end on an unclustered job would overwrite it). */
int csvgpu_chr_job_abort(csv_ctx *ctx, csv_job *job);

/* Several chromosomes in flight on one GPU: several contexts (one per host thread, each with its own stream) that share a gate.
 * They queue the scan + depth pair of their jobs onto the gate's ONE stream — so the bandwidth-bound kernels of all lanes run back
 * to back in queue order, with no hand-over between queues — while each context's own stream carries that lane's small kernels
 * (ordering, clustering, copies), which then run beside another lane's pair, and the host merge of one chromosome hides behind
 * the device work of the others. The reference gets its overlap from a thread pool over chromosomes (sv_caller.cpp:827-863);
 * this is the device-side counterpart. Attach the same gate to every context of ONE GPU before running pipelines concurrently;
 * detach with gate == NULL. Jobs on shards that are not coordinate-sorted ignore the gate. */
typedef struct csv_gate csv_gate;
csv_gate *csvgpu_gate_create(void);
/* Creates the gate's stream now rather than at the first job. Streams get the runtime's hardware queues in creation order and a waiting
 * stream holds up the others of its queue: open the gate before creating the lanes' contexts. Optional. */
int csvgpu_gate_open(csv_gate *gate, int device_ordinal);
void csvgpu_gate_destroy(csv_gate *gate);                 /* after every attached context is destroyed or detached */
int csvgpu_set_gate(csv_ctx *ctx, csv_gate *gate);

/* Page-locked host memory for result buffers (hipHostMalloc); NULL on failure. Freed blocks are kept by the context for reuse
 * and released by csvgpu_destroy(), which also releases blocks never freed: do not use them after the context is gone. */
void *csvgpu_host_alloc(csv_ctx *ctx, size_t bytes);
void csvgpu_host_free(csv_ctx *ctx, void *p);

/* The alignment intervals that the last csvgpu_chr_pipeline_dev() computed for every record of `shard`, copied to host. */
int csvgpu_aln_intervals_resident(csv_ctx *ctx, csv_shard *shard, int32_t *ref_end, int32_t *q_start, int32_t *q_end);

/* The same for selected records only: ref_end[i] / q_start[i] / q_end[i] of record rec[i] (the split-read pass needs the intervals of
 * the primaries that have a supplementary record and of those records — a few per cent of a contig; sv_caller.cpp:152, :162). */
int csvgpu_aln_intervals_gather_resident(csv_ctx *ctx, csv_shard *shard, const uint32_t *rec, uint64_t n, int32_t *ref_end, int32_t *q_start, int32_t *q_end);
/* ... of several shards in one call: the records of shard c are rec[rec_off[c] .. rec_off[c+1]), outputs in the same layout. */
int csvgpu_aln_intervals_gather_batch(csv_ctx *ctx, int n_shards, csv_shard *const *shards, const uint32_t *rec, const uint64_t *rec_off,
                                      int32_t *ref_end, int32_t *q_start, int32_t *q_end);

/* The query-name column of a resident shard: qname_hash[i] = std::hash<std::string> (libstdc++) of record i's query name — the value that
 * decides where the reference's unordered_map<std::string, PrimaryAlignment> puts the read (sv_caller.cpp:152). Copied to HBM, owned by the shard. */
int csvgpu_shard_set_qname_hash(csv_ctx *ctx, csv_shard *shard, const uint64_t *qname_hash);

/* §8f-4: which primary alignments survive, and in which order the reference iterates them — for up to 32 contigs in one call.
 * For every contig the reference fills an unordered_map keyed by query name with every record that passes the filter of sv_caller.cpp:145
 * (not SECONDARY/UNMAP/DUP/QCFAIL, mapq >= min_mapq) and is not supplementary, in file order (:137-172); erases the names without a
 * supplementary record (:183-202); then iterates the map (:216, :224). Given the (sorted, distinct) name hashes of the run's supplementary
 * records, this returns per contig the record indices of the surviving primaries in that iteration order: out_rec[out_off[c] .. out_off[c+1]).
 * A record "survives" here when its name HASH is in supp_hash — the caller confirms the names (a 64-bit collision is the only difference).
 * Precondition (caller's to check when it stages the shard): within a contig no two non-supplementary records share a name hash
 * (no repeated query names, no 64-bit collisions) — otherwise the map would hold one node for two records and the host form
 * (host/umap_order.h) has to be used for that contig. CSV_ECAPACITY: out_rec too small, out_off[n_contigs] holds the required count. */
int csvgpu_split_order(csv_ctx *ctx, int n_contigs, csv_shard *const *shards, uint8_t min_mapq, const uint64_t *supp_hash, uint64_t n_supp,
                       uint32_t *out_rec, uint64_t capacity, uint64_t *out_off);
/* The same in two calls, so that the part that needs no supplementary record — the nodes and all but the last epochs of every contig's map:
 * most of the device's work — runs while the caller is still collecting the supplementary records (sv_caller.cpp:146-165 does both in one loop
 * over the file). _begin queues that work on the context's stream and returns; _finish takes the hashes and returns what csvgpu_split_order
 * returns. No other entry point may be called on this context between the two (they share its workspaces); one pending order per context;
 * after CSV_ECAPACITY _finish may be called again with a larger out_rec. */
int csvgpu_split_order_begin(csv_ctx *ctx, int n_contigs, csv_shard *const *shards, uint8_t min_mapq);
/* _begin for a call that holds EVERY contig of the run (sv_caller.cpp:137-172 fills supp_map from the same records): the supplementary
 * records' name hashes are then taken from these shards on the device (same filter, flag 0x800 set), and the whole order — nodes, epochs,
 * survivors — is queued by this call; _finish(ctx, NULL, 0, ...) only waits for it. A caller whose run has supplementary records on contigs
 * that are not in this call must use _begin and pass all hashes to _finish. */
int csvgpu_split_order_begin_self(csv_ctx *ctx, int n_contigs, csv_shard *const *shards, uint8_t min_mapq);
int csvgpu_split_order_finish(csv_ctx *ctx, const uint64_t *supp_hash, uint64_t n_supp, uint32_t *out_rec, uint64_t capacity, uint64_t *out_off);

/* csvgpu_window_log2 on the depth map that the last csvgpu_chr_pipeline_dev() left resident in `shard`
 * (region tables and outputs are host memory; the depth map never leaves HBM). */
int csvgpu_window_log2_resident(csv_ctx *ctx, csv_shard *shard, const uint32_t *region_start, const uint32_t *region_end,
                                const int32_t *sample_size, const uint64_t *win_off, uint64_t n_regions, double mean_cov,
                                double *log2_cov, uint32_t *win_start, uint32_t *win_end);

/* csvgpu_window_log2_resident for several shards in one call (the copy-number pass of a whole run: one region table per contig, each
 * evaluated on its own resident depth map; cnv_caller.cpp:76-113): table c has n_regions[c] regions, its win_off[c] starts at 0, its
 * outputs go to log2_cov[c] / win_start[c] / win_end[c]. One transfer in, one launch per shard, one transfer out, one wait. */
int csvgpu_window_log2_resident_many(csv_ctx *ctx, int n_shards, csv_shard *const *shards, const uint32_t *const *region_start,
                                     const uint32_t *const *region_end, const int32_t *const *sample_size, const uint64_t *const *win_off,
                                     const uint64_t *n_regions, const double *mean_cov, double *const *log2_cov, uint32_t *const *win_start,
                                     uint32_t *const *win_end);

/* depth_out[i] = depth[pos[i]] on the depth map resident in `shard`, or -1 where pos[i] >= depth_len: the VCF writer's
 * SUPPORT / DP lookups (SVCaller::getReadDepth, sv_caller.cpp:1332-1344, called at :1306) without moving the map. */
int csvgpu_depth_lookup_resident(csv_ctx *ctx, csv_shard *shard, const uint32_t *pos, uint64_t n, int32_t *depth_out);

/* Copy `bytes` from device memory returned by this library (csv_chr_result pointers) to host memory;
 * synchronous with respect to the context's stream. For host code above the ABI that does not link HIP. */
int csvgpu_download(csv_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);

/* Device-pointer twins of the clustering / HMM entry points (same semantics as above). */
int csvgpu_dbscan_iv_dev(csv_ctx *ctx, const uint32_t *d_start, const uint32_t *d_end, uint64_t n,
                         double eps, int32_t min_pts, int32_t *d_labels);
int csvgpu_dbscan_1d_dev(csv_ctx *ctx, const int32_t *d_pts, const uint64_t *d_seg_off,
                         uint64_t n_seg, uint64_t n_pts, uint32_t max_seg_len,
                         double eps, int32_t min_pts, int32_t *d_labels);
int csvgpu_window_log2_dev(csv_ctx *ctx, const uint32_t *d_depth, uint32_t depth_len,
                           const uint32_t *d_region_start, const uint32_t *d_region_end,
                           const int32_t *d_sample_size, const uint64_t *d_win_off, uint64_t n_regions,
                           uint64_t n_windows, double mean_cov, double *d_log2_cov,
                           uint32_t *d_win_start, uint32_t *d_win_end);
int csvgpu_viterbi_dev(csv_ctx *ctx, const csv_hmm *hmm, const double *d_o1, const double *d_o2,
                       const double *d_pfb, const uint64_t *d_seq_off, uint64_t n_seq,
                       uint64_t n_obs, int32_t *d_states, double *d_loglik);

#ifdef CSV_TEST_HOOKS
/* Test build only (libcsvgpu_testhooks.so, -DCSV_TEST_HOOKS; libcsvgpu.so does not export it): the next n guarded device allocations
 * inside the library fail as if HBM were exhausted (error-path tests of the signature-buffer growth in csvgpu_chr_job_cluster). */
void csvgpu_test_fail_next_alloc(int n);
#endif

#ifdef __cplusplus
}
#endif
#endif /* CSVGPU_H */
