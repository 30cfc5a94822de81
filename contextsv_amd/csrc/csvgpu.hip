// csvgpu.hip — implementation of the C-ABI in include/csvgpu.h: context, staging, kernel chains.
// No CPU fallback lives here: every result is produced by the kernels under kernels/.
#include <atomic>
#include <string.h>

#include <algorithm>
#include <new>

#include "common.hpp"
#include <chrono>
#include <unordered_map>

namespace csv {

static std::string g_create_err;
#ifdef CSV_TEST_HOOKS
// csvgpu_test_fail_next_alloc — only in the test build of the library (libcsvgpu_testhooks.so: Makefile), never in libcsvgpu.so
static std::atomic<int> g_fail_alloc{0};
static inline bool csv_test_fail_alloc()
{
    int n = g_fail_alloc.load();
    while (n > 0) if (g_fail_alloc.compare_exchange_weak(n, n - 1)) return true;
    return false;
}
#else
static inline bool csv_test_fail_alloc() { return false; }
#endif

int arena_reserve(csv_ctx *ctx, Arena &a, size_t bytes)
{
    bytes = align_up(bytes + 4096, 4096);
    if (bytes > a.cap) {
        CSV_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (a.base) CSV_HIP(ctx, hipFree(a.base));
        a.base = nullptr; a.cap = 0;
        size_t want = bytes + bytes / 4;
        if (hipMalloc((void **)&a.base, want) != hipSuccess) {
            (void)hipGetLastError();
            if (hipMalloc((void **)&a.base, bytes) != hipSuccess) { (void)hipGetLastError(); a.base = nullptr; ctx->err = "hipMalloc failed (arena)"; return CSV_ENOMEM; }
            want = bytes;
        }
        a.cap = want;
    }
    a.used = 0;
    return CSV_OK;
}

void *arena_alloc(Arena &a, size_t bytes)
{
    const size_t off = align_up(a.used, 256);
    if (off + bytes > a.cap) return nullptr;
    a.used = off + bytes;
    return a.base + off;
}

int ensure_pinned(csv_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->pinned_cap) return CSV_OK;
    if (ctx->pinned) CSV_HIP(ctx, hipHostFree(ctx->pinned));
    ctx->pinned = nullptr; ctx->pinned_cap = 0;
    CSV_HIP(ctx, hipHostMalloc(&ctx->pinned, bytes, hipHostMallocDefault));
    ctx->pinned_cap = bytes;
    return CSV_OK;
}

// Host arrays of the host-pointer entry points travel through the context's page-locked block: the runtime stages a pageable
// hipMemcpyAsync itself, in chunks and under a lock that the other contexts' launches also take (seen as millisecond gaps in the lanes'
// big kernels whenever the caller's context copied its observation vectors). in(): bytes copied into the block, the block's address
// returned for the async copy; out(): a slot of the block the device writes to, copied to the caller's array by finish() after the wait.
struct PinStage {
    csv_ctx *ctx;
    size_t used = 0;
    struct Out { void *dst; const void *src; size_t bytes; };
    std::vector<Out> outs;
    explicit PinStage(csv_ctx *c) : ctx(c) {}
    static size_t need(size_t bytes) { return (bytes + 255) / 256 * 256; }
    const void *in(const void *src, size_t bytes) { void *p = (char *)ctx->pinned + used; if (bytes) memcpy(p, src, bytes); used += need(bytes); return p; }
    void *out(void *dst, size_t bytes) { void *p = (char *)ctx->pinned + used; used += need(bytes); outs.push_back(Out{dst, p, bytes}); return p; }
    void finish() { for (const Out &o : outs) if (o.bytes) memcpy(o.dst, o.src, o.bytes); outs.clear(); }
};

static hipEvent_t get_event(csv_ctx *ctx)
{
    if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

bool timer_begin(csv_ctx *ctx, int id, hipStream_t s)
{
    if (!ctx->timing) return false;
    // every recorded event is a barrier packet in the queue (~5 us of idle device each): level 2 keeps them to the two groups a
    // roofline is quoted for
    if (ctx->timing >= 2 && id != CSV_K_CIGAR_SCAN && id != CSV_K_DEPTH) return false;
    Timer t; t.id = id; t.a = get_event(ctx); t.b = get_event(ctx); t.s = s ? s : ctx->stream;
    (void)hipEventRecord(t.a, t.s);
    ctx->timers.push_back(t);
    return true;
}

void timer_end(csv_ctx *ctx)
{
    if (!ctx->timing || ctx->timers.empty()) return;
    (void)hipEventRecord(ctx->timers.back().b, ctx->timers.back().s);
}

static void fold_timers(csv_ctx *ctx)
{
    for (Timer &t : ctx->timers) {
        float ms = 0.f;
        if (hipEventSynchronize(t.b) == hipSuccess && hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) {
            ctx->t_ms[t.id] += ms; ctx->t_n[t.id]++;
        }
        if (t.own_a) ctx->event_pool.push_back(t.a);
        if (t.own_b) ctx->event_pool.push_back(t.b);
    }
    ctx->timers.clear();
}

// ---------------------------------------------------------------------------------------------
// device-side chains shared by the host- and device-pointer entry points

struct DevReads {
    csv_reads d;
    int32_t *ref_end, *q_start, *q_end;
    uint32_t *ckpt;
    ScanCounters *cnt;
};

// device scalars + the ordering pass's bucket tables, zeroed together before every scan
static constexpr size_t kCntBytes = 256 + 2 * (size_t)BK_N * 4;
static constexpr uint64_t kMaxReadWords = 0x7ffff000ull;       // exclusive bound on one read's CIGAR words
static inline uint32_t *bucket_off(ScanCounters *cnt) { return (uint32_t *)((char *)cnt + 256); }
static inline uint32_t *bucket_cur(ScanCounters *cnt) { return bucket_off(cnt) + BK_N; }

static size_t reads_bytes(const csv_reads *r)
{
    const uint64_t n = r->n_reads, m = r->n_cigar;
    return align_up(n * 4, 256) + align_up(n * 2, 256) + align_up(n, 256) + align_up((n + 1) * 8, 256) + align_up(m * 4 + 16, 256) +
           3 * align_up(n * 4, 256) + align_up(ckpt_bytes(m), 256) + align_up(kCntBytes, 256) + 512;
}

// copy a host shard into the arena; returns device views
static int stage_reads(csv_ctx *ctx, const csv_reads *r, DevReads &o)
{
    Arena &a = ctx->arena;
    const uint64_t n = r->n_reads, m = r->n_cigar;
    int32_t *pos = (int32_t *)arena_alloc(a, n * 4);
    uint16_t *flag = (uint16_t *)arena_alloc(a, n * 2);
    uint8_t *mapq = (uint8_t *)arena_alloc(a, n);
    uint64_t *coff = (uint64_t *)arena_alloc(a, (n + 1) * 8);
    uint32_t *cig = (uint32_t *)arena_alloc(a, m * 4 + 16);
    o.ref_end = (int32_t *)arena_alloc(a, n * 4);
    o.q_start = (int32_t *)arena_alloc(a, n * 4);
    o.q_end = (int32_t *)arena_alloc(a, n * 4);
    o.cnt = (ScanCounters *)arena_alloc(a, kCntBytes);
    o.ckpt = (uint32_t *)arena_alloc(a, ckpt_bytes(m));
    if (!o.ckpt || !pos || !flag || !mapq || !coff || !cig || !o.ref_end || !o.q_start || !o.q_end || !o.cnt) { ctx->err = "arena exhausted"; return CSV_ENOMEM; }
    hipStream_t s = ctx->stream;
    if (n) {
        CSV_HIP(ctx, hipMemcpyAsync(pos, r->pos, n * 4, hipMemcpyHostToDevice, s));
        CSV_HIP(ctx, hipMemcpyAsync(flag, r->flag, n * 2, hipMemcpyHostToDevice, s));
        CSV_HIP(ctx, hipMemcpyAsync(mapq, r->mapq, n, hipMemcpyHostToDevice, s));
    }
    CSV_HIP(ctx, hipMemcpyAsync(coff, r->cigar_off, (n + 1) * 8, hipMemcpyHostToDevice, s));
    if (m) CSV_HIP(ctx, hipMemcpyAsync(cig, r->cigar, m * 4, hipMemcpyHostToDevice, s));
    CSV_HIP(ctx, hipMemsetAsync(o.cnt, 0, kCntBytes, s));
    o.d = *r;
    o.d.pos = pos; o.d.flag = flag; o.d.mapq = mapq; o.d.tid = nullptr; o.d.cigar_off = coff; o.d.cigar = cig;
    return CSV_OK;
}

static int check_reads_ptrs(csv_ctx *ctx, const csv_reads *r)
{
    if (!ctx) return CSV_EINVAL;
    if (!r || !r->cigar_off || (r->n_reads && (!r->pos || !r->flag || !r->mapq)) || (r->n_cigar && !r->cigar)) {
        ctx->err = "csv_reads: null array"; return CSV_EINVAL;
    }
    if (r->n_reads >= 0xffffffffull) { ctx->err = "csv_reads: more than 2^32-2 reads in one shard"; return CSV_EINVAL; }
    return CSV_OK;
}

// Host arrays. The kernels index the word array with cigar_off: nothing reaches the device unless the offsets are monotone and
// inside it (a read's own word count stays far below 2^31: the scan works in 32-bit read-relative indices).
static int check_reads(csv_ctx *ctx, const csv_reads *r)
{
    int rc = check_reads_ptrs(ctx, r);
    if (rc) return rc;
    for (uint64_t i = 0; i < r->n_reads; i++) {
        if (r->cigar_off[i + 1] < r->cigar_off[i]) { ctx->err = "csv_reads: cigar_off not monotone"; return CSV_EINVAL; }
        if (r->cigar_off[i + 1] - r->cigar_off[i] >= kMaxReadWords) { ctx->err = "csv_reads: a read with 2^31 CIGAR words"; return CSV_EINVAL; }
    }
    if (r->cigar_off[r->n_reads] > r->n_cigar) { ctx->err = "csv_reads: cigar_off beyond n_cigar"; return CSV_EINVAL; }
    return CSV_OK;
}

// The same test for arrays that already live in HBM (csvgpu_shard_wrap_dev): one small kernel, once per wrapped shard.
static int check_reads_dev(csv_ctx *ctx, const csv_reads *r)
{
    int rc = check_reads_ptrs(ctx, r);
    if (rc) return rc;
    if ((rc = ensure_pinned(ctx, 4096))) return rc;
    uint32_t *d_bad = nullptr;
    CSV_HIP(ctx, hipMalloc((void **)&d_bad, 256));
    hipError_t e = hipMemsetAsync(d_bad, 0, 4, ctx->stream);
    if (e == hipSuccess) { launch_validate_offsets(ctx->stream, r->cigar_off, r->n_reads, r->n_cigar, kMaxReadWords, d_bad); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->pinned, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_bad);
    if (e != hipSuccess) { ctx->err = std::string("csv_reads: offset check failed: ") + hipGetErrorString(e); return CSV_EHIP; }
    if (*(const uint32_t *)ctx->pinned) { ctx->err = "csv_reads: cigar_off not monotone or beyond n_cigar"; return CSV_EINVAL; }
    return CSV_OK;
}

// The waits of the per-chromosome pipeline last a fraction of a millisecond: polling for up to a millisecond before blocking
// saves the tens of microseconds a blocked thread takes to be woken, during which the device has nothing queued.
static std::chrono::microseconds spin_limit()
{
    static const int us = [] { const char *e = getenv("CSV_SPIN_US"); return e && *e ? atoi(e) : 100; }();
    return std::chrono::microseconds(us);
}
static hipError_t wait_stream(hipStream_t s)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > spin_limit()) return hipStreamSynchronize(s);
    }
}
static hipError_t wait_event(hipEvent_t ev)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::steady_clock::now() - t0 > spin_limit()) return hipEventSynchronize(ev);
    }
}

static int read_counters(csv_ctx *ctx, const ScanCounters *d_cnt, ScanCounters &h)
{
    int rc = ensure_pinned(ctx, 4096);
    if (rc) return rc;
    CSV_HIP(ctx, hipMemcpyAsync(ctx->pinned, d_cnt, sizeof(ScanCounters), hipMemcpyDeviceToHost, ctx->stream));
    CSV_HIP(ctx, wait_stream(ctx->stream));
    memcpy(&h, ctx->pinned, sizeof(ScanCounters));
    return CSV_OK;
}

// ordering workspace for n signatures
struct SortWs {
    uint64_t *k0, *k1;
    uint32_t *v0, *v1;
    void *tmp;
    csv_sig *sig_tmp;        // bucketed copy of the signatures (bucket ordering); aliases the radix key arrays, which that path does not use
};
static size_t sortws_bytes(uint64_t n) { return 2 * align_up(n * 8, 256) + 2 * align_up(n * 4, 256) + radix_sort_tmp_bytes(n) + 256; }
static bool sortws_carve(Arena &a, uint64_t n, SortWs &w)
{
    w.k0 = (uint64_t *)arena_alloc(a, n * 8); w.k1 = (uint64_t *)arena_alloc(a, n * 8);
    w.v0 = (uint32_t *)arena_alloc(a, n * 4); w.v1 = (uint32_t *)arena_alloc(a, n * 4);
    w.tmp = arena_alloc(a, radix_sort_tmp_bytes(n));
    w.sig_tmp = (csv_sig *)w.k0;                         // k0 and k1 are carved back to back: 2 x align_up(8n, 256) >= 16n bytes
    return w.k0 && w.k1 && w.v0 && w.v1 && w.tmp && (char *)w.k1 == (char *)w.k0 + align_up(n * 8, 256);
}

// sig_raw[0..n) (arbitrary order) -> sig_sorted in the reference's vector order; optional SoA start/end.
// with_type: DEL calls first, then INS calls (per-type subsequences of the vector).
// key layout of the ordering pass: start in the low bits (width from the contig length), the type bit above it
struct KeyLayout { int start_bits, type_pos, key_bits, bucket_shift; };
static KeyLayout key_layout(uint32_t depth_len, bool overflow, bool with_type)
{
    KeyLayout k;
    // starts are < scan_start_limit(depth_len) unless the scan flagged an overflow (then the full 32 bits are sorted)
    k.start_bits = overflow ? 32 : std::max(1, bits_of((uint64_t)scan_start_limit(depth_len) - 1));
    k.type_pos = with_type ? k.start_bits : -1;
    k.key_bits = k.start_bits + (with_type ? 1 : 0);
    k.bucket_shift = std::max(0, k.key_bits - (int)BK_BITS);
    return k;
}

// The ordering pass's bucket counts (and, for shards known to be coordinate-sorted, the depth tiles' candidate ranges) are taken
// by the scan itself: nothing small runs between the scan and the depth pass.
static ScanExtras scan_extras(ScanCounters *cnt, uint32_t depth_len, bool with_type, uint64_t *tile_range)
{
    ScanExtras x;
    const KeyLayout k = key_layout(depth_len, false, with_type);
    x.bucket_hist = bucket_off(cnt); x.type_pos = k.type_pos; x.bucket_shift = k.bucket_shift;
    x.tile_range = tile_range; x.n_tiles = tile_range ? depth_n_tiles(depth_len) : 0;
    return x;
}

static void order_signatures(csv_ctx *ctx, const csv_sig *sig_raw, uint64_t n, uint32_t depth_len, uint32_t overflow, uint32_t max_bucket,
                             ScanCounters *cnt, bool with_type, SortWs &w, csv_sig *sig_sorted, uint32_t *start_out, uint32_t *end_out)
{
    if (!n) return;
    TimerScope ts(ctx, CSV_K_SORT);
    if (!overflow && max_bucket <= BK_LOCAL_MAX) {
        const KeyLayout k = key_layout(depth_len, false, with_type);
        launch_bucket_sort(ctx->stream, sig_raw, n, k.type_pos, k.bucket_shift, bucket_off(cnt), bucket_cur(cnt), w.sig_tmp, sig_sorted, start_out, end_out);
        return;
    }
    const KeyLayout kl = key_layout(depth_len, overflow != 0, with_type);
    const int type_pos = kl.type_pos, key_bits = kl.key_bits;
    launch_sig_make_keys(ctx->stream, sig_raw, n, 0, type_pos, w.k0, w.v0);
    const int in_out = launch_radix_sort_u64(ctx->stream, w.k0, w.v0, w.k1, w.v1, n, key_bits, w.tmp);
    launch_sig_fix_ties_gather(ctx->stream, sig_raw, in_out ? w.k1 : w.k0, in_out ? w.v1 : w.v0, n, sig_sorted, start_out, end_out);
}

// depth chain on device arrays. pmax / ord / range scratch comes from `a`; `ranges` != nullptr: the scan already produced the
// tiles' candidate ranges (coordinate-sorted shard) and only the tile kernel remains.
static int depth_chain(csv_ctx *ctx, Arena &a, const csv_reads &d, const int32_t *ref_end, const uint32_t *ckpt, bool unsorted, uint32_t depth_len,
                       uint32_t *depth, ScanCounters *cnt, const uint64_t *ranges = nullptr, uint32_t cigar_pad = 0, void *items = nullptr,
                       int form = SCAN_FORM_WAVE)
{
    const uint64_t n = d.n_reads;
    TimerScope ts(ctx, CSV_K_DEPTH);
    if (n == 0 || depth_len == 0) {
        if (depth && depth_len) CSV_HIP(ctx, hipMemsetAsync(depth, 0, (size_t)depth_len * 4, ctx->stream));
        return CSV_OK;
    }
    if (ranges && !unsorted) {
        launch_depth_tiles(ctx->stream, d, nullptr, ref_end, ckpt, depth_len, depth, cnt, ranges, cigar_pad, items, form);
        return CSV_OK;
    }
    int32_t *pmax = (int32_t *)arena_alloc(a, n * 4);
    void *ptmp = arena_alloc(a, prefix_max_tmp_bytes(n));
    uint64_t *ttmp = (uint64_t *)arena_alloc(a, depth_tiles_tmp_bytes(depth_len));
    if (!pmax || !ptmp || !ttmp) { ctx->err = "arena exhausted (depth)"; return CSV_ENOMEM; }
    const uint32_t *ord = nullptr;
    const int32_t *pos_s = d.pos;
    const int32_t *end_s = ref_end;
    if (unsorted) {
        // shard not coordinate-sorted: sort the read indices by pos on device and feed the tile search through `ord`
        SortWs w;
        uint32_t *pos_g = (uint32_t *)arena_alloc(a, n * 4), *end_g = (uint32_t *)arena_alloc(a, n * 4);
        if (!sortws_carve(a, n, w) || !pos_g || !end_g) { ctx->err = "arena exhausted (depth/unsorted)"; return CSV_ENOMEM; }
        launch_iota_keys_i32(ctx->stream, d.pos, n, w.k0, w.v0);
        const int io = launch_radix_sort_u64(ctx->stream, w.k0, w.v0, w.k1, w.v1, n, 32, w.tmp);
        const uint32_t *perm = io ? w.v1 : w.v0;
        launch_gather_u32(ctx->stream, (const uint32_t *)d.pos, perm, n, pos_g);
        launch_gather_u32(ctx->stream, (const uint32_t *)ref_end, perm, n, end_g);
        ord = perm; pos_s = (const int32_t *)pos_g; end_s = (const int32_t *)end_g;
    }
    launch_prefix_max(ctx->stream, end_s, pmax, n, ptmp);
    launch_depth_ranges(ctx->stream, pos_s, pmax, n, depth_len, ttmp);
    launch_depth_tiles(ctx->stream, d, ord, ref_end, ckpt, depth_len, depth, cnt, ttmp, cigar_pad, items, form);
    return CSV_OK;
}
static size_t depth_chain_bytes(uint64_t n, uint32_t depth_len = 0xffffffffu)
{
    return depth_tiles_tmp_bytes(depth_len) + align_up(n * 4, 256) + prefix_max_tmp_bytes(n) + sortws_bytes(n) + 2 * align_up(n * 4, 256) + 1024;
}

// interval DBSCAN on device arrays in caller order
static int dbscan_iv_chain(csv_ctx *ctx, Arena &a, const uint32_t *d_start, const uint32_t *d_end, uint64_t n, double eps,
                           int min_pts, int32_t *d_labels)
{
    if (n == 0) return CSV_OK;
    unsigned int *flag = (unsigned int *)arena_alloc(a, 256);
    void *tmp = arena_alloc(a, dbscan_tmp_bytes(n));
    if (!flag || !tmp) { ctx->err = "arena exhausted (dbscan)"; return CSV_ENOMEM; }
    CSV_HIP(ctx, hipMemsetAsync(flag, 0, 4, ctx->stream));
    launch_check_sorted_u32(ctx->stream, d_start, n, flag);
    int rc = ensure_pinned(ctx, 4096);
    if (rc) return rc;
    CSV_HIP(ctx, hipMemcpyAsync(ctx->pinned, flag, 4, hipMemcpyDeviceToHost, ctx->stream));
    CSV_HIP(ctx, wait_stream(ctx->stream));
    const bool unsorted = *(unsigned int *)ctx->pinned != 0;
    if (!unsorted) {
        TimerScope ts(ctx, CSV_K_DBSCAN);
        launch_dbscan_iv_sorted(ctx->stream, d_start, d_end, nullptr, n, n, eps, min_pts, nullptr, d_labels, tmp);
        return CSV_OK;
    }
    SortWs w;
    uint32_t *s_s = (uint32_t *)arena_alloc(a, n * 4), *e_s = (uint32_t *)arena_alloc(a, n * 4);
    if (!sortws_carve(a, n, w) || !s_s || !e_s) { ctx->err = "arena exhausted (dbscan sort)"; return CSV_ENOMEM; }
    const uint32_t *perm;
    {
        TimerScope ts(ctx, CSV_K_SORT);
        launch_iota_keys_u32(ctx->stream, d_start, n, w.k0, w.v0);
        const int io = launch_radix_sort_u64(ctx->stream, w.k0, w.v0, w.k1, w.v1, n, 32, w.tmp);
        perm = io ? w.v1 : w.v0;
        launch_gather_u32(ctx->stream, d_start, perm, n, s_s);
        launch_gather_u32(ctx->stream, d_end, perm, n, e_s);
    }
    TimerScope ts(ctx, CSV_K_DBSCAN);
    launch_dbscan_iv_sorted(ctx->stream, s_s, e_s, perm, n, n, eps, min_pts, nullptr, d_labels, tmp);
    return CSV_OK;
}
static size_t dbscan_iv_chain_bytes(uint64_t n) { return 512 + dbscan_tmp_bytes(n) + sortws_bytes(n) + 2 * align_up(n * 4, 256) + 1024; }

}  // namespace csv

using namespace csv;

// =============================================================================================
extern "C" {

int csvgpu_abi_version(void) { return CSVGPU_ABI_VERSION; }

const char *csvgpu_last_error(const csv_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

static csv_ctx *create_ctx(int device_ordinal, void *stream, int low_priority);
csv_ctx *csvgpu_create(int device_ordinal, void *stream) { return create_ctx(device_ordinal, stream, 0); }
csv_ctx *csvgpu_create_background(int device_ordinal) { return create_ctx(device_ordinal, nullptr, 1); }

static csv_ctx *create_ctx(int device_ordinal, void *stream, int low_priority)
{
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        (void)hipGetLastError();
        g_create_err = std::string("no usable HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return nullptr;
    }
    if (device_ordinal < 0 || device_ordinal >= n_dev) { g_create_err = "device ordinal out of range"; return nullptr; }
    if (hipSetDevice(device_ordinal) != hipSuccess) { g_create_err = "hipSetDevice failed"; return nullptr; }
    csv_ctx *ctx = new (std::nothrow) csv_ctx();
    if (!ctx) { g_create_err = "out of host memory"; return nullptr; }
    ctx->device = device_ordinal;
    if (stream) { ctx->stream = (hipStream_t)stream; ctx->own_stream = false; }
    else {
        int least = 0, greatest = 0;
        if (low_priority) (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const char *pe = getenv("CSV_BG_PRIORITY");                    // experiments: "high" turns the background context into a foreground one
        const int prio = (pe && pe[0] == 'h') ? greatest : least;
        const hipError_t se = low_priority ? hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio) : hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (se != hipSuccess) { g_create_err = "hipStreamCreate failed"; delete ctx; return nullptr; }
        ctx->own_stream = true;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess) ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    return ctx;
}

static void split_state_free(csv_ctx *ctx);
void csvgpu_destroy(csv_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    fold_timers(ctx);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->arena.base) (void)hipFree(ctx->arena.base);
    if (ctx->work.base) (void)hipFree(ctx->work.base);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->job_pin) (void)hipHostFree(ctx->job_pin);
    for (auto &b : ctx->host_pool) (void)hipHostFree(b.first);
    for (auto &b : ctx->host_live) (void)hipHostFree(b.first);       // blocks the caller never returned
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    split_state_free(ctx);
    delete ctx;
}

int csvgpu_synchronize(csv_ctx *ctx)
{
    if (!ctx) return CSV_EINVAL;
    CSV_HIP(ctx, wait_stream(ctx->stream));
    return CSV_OK;
}

int csvgpu_timing_enable(csv_ctx *ctx, int on) { if (!ctx) return CSV_EINVAL; ctx->timing = on < 0 ? 0 : (on > 3 ? 1 : on); return CSV_OK; }

int csvgpu_timing_reset(csv_ctx *ctx)
{
    if (!ctx) return CSV_EINVAL;
    CSV_HIP(ctx, wait_stream(ctx->stream));
    fold_timers(ctx);
    for (int i = 0; i < CSV_K_COUNT; i++) { ctx->t_ms[i] = 0; ctx->t_n[i] = 0; }
    ctx->timer_tick = 0;
    return CSV_OK;
}

int csvgpu_timing_get(csv_ctx *ctx, int kernel_id, double *total_ms, uint64_t *launches)
{
    if (!ctx || kernel_id < 0 || kernel_id >= CSV_K_COUNT) return CSV_EINVAL;
    CSV_HIP(ctx, wait_stream(ctx->stream));
    fold_timers(ctx);
    if (total_ms) *total_ms = ctx->t_ms[kernel_id];
    if (launches) *launches = ctx->t_n[kernel_id];
    return CSV_OK;
}

// ---------------------------------------------------------------------------------------------
int csvgpu_cigar_scan(csv_ctx *ctx, const csv_reads *reads, uint32_t depth_len, uint32_t min_oplen, uint8_t min_mapq,
                      csv_sig *out, uint64_t *n_out)
{
    int rc = check_reads(ctx, reads);
    if (rc) return rc;
    if (!n_out || (*n_out && !out)) { ctx->err = "cigar_scan: null output"; return CSV_EINVAL; }
    (void)hipSetDevice(ctx->device);
    const uint64_t cap = std::min<uint64_t>(*n_out, reads->n_cigar);
    rc = arena_reserve(ctx, ctx->arena, reads_bytes(reads) + align_up(cap * sizeof(csv_sig), 256) + 1024);
    if (rc) return rc;
    DevReads dr;
    if ((rc = stage_reads(ctx, reads, dr))) return rc;
    csv_sig *sig_raw = (csv_sig *)arena_alloc(ctx->arena, cap * sizeof(csv_sig) + 16);
    if (!sig_raw) { ctx->err = "arena exhausted"; return CSV_ENOMEM; }
    {
        TimerScope ts(ctx, CSV_K_CIGAR_SCAN);
        launch_cigar_scan(ctx->stream, ctx->n_cu, dr.d, depth_len, min_oplen, min_mapq, 1, sig_raw, cap, dr.ref_end, dr.q_start, dr.q_end, dr.ckpt, dr.cnt,
                          scan_extras(dr.cnt, depth_len, false, nullptr), nullptr, scan_form_for(reads->n_reads, reads->n_cigar));
    }
    ScanCounters h;
    if ((rc = read_counters(ctx, dr.cnt, h))) return rc;
    const uint64_t n = h.n_sig;
    *n_out = n;
    if (n > cap) { ctx->err = "cigar_scan: output capacity too small"; return CSV_ECAPACITY; }
    if (n == 0) return CSV_OK;
    if ((rc = arena_reserve(ctx, ctx->work, sortws_bytes(n) + align_up(n * sizeof(csv_sig), 256) + 1024))) return rc;
    SortWs w;
    csv_sig *sig_sorted = (csv_sig *)arena_alloc(ctx->work, n * sizeof(csv_sig));
    if (!sortws_carve(ctx->work, n, w) || !sig_sorted) { ctx->err = "arena exhausted (sort)"; return CSV_ENOMEM; }
    order_signatures(ctx, sig_raw, n, depth_len, h.max_start, h.max_len, dr.cnt, false, w, sig_sorted, nullptr, nullptr);
    CSV_HIP(ctx, hipMemcpyAsync(out, sig_sorted, n * sizeof(csv_sig), hipMemcpyDeviceToHost, ctx->stream));
    CSV_HIP(ctx, wait_stream(ctx->stream));
    return CSV_OK;
}

int csvgpu_aln_intervals(csv_ctx *ctx, const csv_reads *reads, int32_t *ref_end, int32_t *q_start, int32_t *q_end)
{
    int rc = check_reads(ctx, reads);
    if (rc) return rc;
    if (reads->n_reads && (!ref_end || !q_start || !q_end)) { ctx->err = "aln_intervals: null output"; return CSV_EINVAL; }
    (void)hipSetDevice(ctx->device);
    if ((rc = arena_reserve(ctx, ctx->arena, reads_bytes(reads) + 1024))) return rc;
    DevReads dr;
    if ((rc = stage_reads(ctx, reads, dr))) return rc;
    {
        TimerScope ts(ctx, CSV_K_CIGAR_SCAN);
        launch_cigar_scan(ctx->stream, ctx->n_cu, dr.d, 0, 0, 0, 0, nullptr, 0, dr.ref_end, dr.q_start, dr.q_end, dr.ckpt, dr.cnt, ScanExtras(), nullptr,
                          scan_form_for(reads->n_reads, reads->n_cigar));
    }
    const uint64_t n = reads->n_reads;
    if (n) {
        CSV_HIP(ctx, hipMemcpyAsync(ref_end, dr.ref_end, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        CSV_HIP(ctx, hipMemcpyAsync(q_start, dr.q_start, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        CSV_HIP(ctx, hipMemcpyAsync(q_end, dr.q_end, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    CSV_HIP(ctx, wait_stream(ctx->stream));
    return CSV_OK;
}

int csvgpu_depth(csv_ctx *ctx, const csv_reads *reads, uint32_t depth_len, uint32_t *depth, uint64_t *sum, uint32_t *nonzero)
{
    int rc = check_reads(ctx, reads);
    if (rc) return rc;
    (void)hipSetDevice(ctx->device);
    if ((rc = arena_reserve(ctx, ctx->arena, reads_bytes(reads) + align_up((size_t)depth_len * 4 + 16, 256) + 1024))) return rc;
    DevReads dr;
    if ((rc = stage_reads(ctx, reads, dr))) return rc;
    uint32_t *d_depth = (uint32_t *)arena_alloc(ctx->arena, (size_t)depth_len * 4 + 16);
    if (!d_depth) { ctx->err = "arena exhausted"; return CSV_ENOMEM; }
    {
        TimerScope ts(ctx, CSV_K_CIGAR_SCAN);
        launch_cigar_scan(ctx->stream, ctx->n_cu, dr.d, depth_len, 0, 0, 0, nullptr, 0, dr.ref_end, dr.q_start, dr.q_end, dr.ckpt, dr.cnt, ScanExtras(), nullptr,
                          scan_form_for(reads->n_reads, reads->n_cigar));
    }
    ScanCounters h;
    if ((rc = read_counters(ctx, dr.cnt, h))) return rc;
    if ((rc = arena_reserve(ctx, ctx->work, depth_chain_bytes(reads->n_reads, depth_len)))) return rc;
    if ((rc = depth_chain(ctx, ctx->work, dr.d, dr.ref_end, dr.ckpt, h.unsorted != 0, depth_len, d_depth, dr.cnt, nullptr, 0, nullptr,
                          scan_form_for(reads->n_reads, reads->n_cigar)))) return rc;
    if (depth && depth_len) CSV_HIP(ctx, hipMemcpyAsync(depth, d_depth, (size_t)depth_len * 4, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = read_counters(ctx, dr.cnt, h))) return rc;
    if (sum) *sum = h.depth_sum;
    if (nonzero) *nonzero = h.depth_nonzero;
    return CSV_OK;
}

static int check_dbscan_args(csv_ctx *ctx, double eps, int32_t min_pts, bool interval)
{
    if (!ctx) return CSV_EINVAL;
    if (!(eps >= 0.0) || (interval && !(eps < 1.0))) { ctx->err = interval ? "dbscan: eps must be in [0,1)" : "dbscan1d: eps must be >= 0"; return CSV_EINVAL; }
    if (min_pts < 1) { ctx->err = "dbscan: min_pts must be >= 1"; return CSV_EINVAL; }
    return CSV_OK;
}

int csvgpu_dbscan_iv_dev(csv_ctx *ctx, const uint32_t *d_start, const uint32_t *d_end, uint64_t n, double eps,
                         int32_t min_pts, int32_t *d_labels)
{
    int rc = check_dbscan_args(ctx, eps, min_pts, true);
    if (rc) return rc;
    if (n && (!d_start || !d_end || !d_labels)) { ctx->err = "dbscan: null array"; return CSV_EINVAL; }
    if (n >= 0xffffffffull) { ctx->err = "dbscan: n too large"; return CSV_EINVAL; }
    (void)hipSetDevice(ctx->device);
    if ((rc = arena_reserve(ctx, ctx->work, dbscan_iv_chain_bytes(n)))) return rc;
    return dbscan_iv_chain(ctx, ctx->work, d_start, d_end, n, eps, min_pts, d_labels);
}

int csvgpu_dbscan_iv(csv_ctx *ctx, const uint32_t *start, const uint32_t *end, uint64_t n, double eps, int32_t min_pts,
                     int32_t *labels)
{
    int rc = check_dbscan_args(ctx, eps, min_pts, true);
    if (rc) return rc;
    if (n == 0) return CSV_OK;
    if (!start || !end || !labels) { ctx->err = "dbscan: null array"; return CSV_EINVAL; }
    (void)hipSetDevice(ctx->device);
    if ((rc = arena_reserve(ctx, ctx->arena, 3 * align_up(n * 4, 256) + 1024))) return rc;
    uint32_t *ds = (uint32_t *)arena_alloc(ctx->arena, n * 4), *de = (uint32_t *)arena_alloc(ctx->arena, n * 4);
    int32_t *dl = (int32_t *)arena_alloc(ctx->arena, n * 4);
    if (!ds || !de || !dl) { ctx->err = "arena exhausted"; return CSV_ENOMEM; }
    CSV_HIP(ctx, hipMemcpyAsync(ds, start, n * 4, hipMemcpyHostToDevice, ctx->stream));
    CSV_HIP(ctx, hipMemcpyAsync(de, end, n * 4, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = csvgpu_dbscan_iv_dev(ctx, ds, de, n, eps, min_pts, dl))) return rc;
    CSV_HIP(ctx, hipMemcpyAsync(labels, dl, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    CSV_HIP(ctx, wait_stream(ctx->stream));
    return CSV_OK;
}

int csvgpu_dbscan_iv_batch(csv_ctx *ctx, const uint32_t *start, const uint32_t *end, const uint64_t *seg_off, uint64_t n_seg, double eps,
                           int32_t min_pts, int32_t *labels)
{
    int rc = check_dbscan_args(ctx, eps, min_pts, true);
    if (rc) return rc;
    if (n_seg == 0) return CSV_OK;
    if (!seg_off) { ctx->err = "dbscan batch: null seg_off"; return CSV_EINVAL; }
    uint64_t max_len = 0;
    for (uint64_t s = 0; s < n_seg; s++) {
        if (seg_off[s + 1] < seg_off[s]) { ctx->err = "dbscan batch: seg_off not monotone"; return CSV_EINVAL; }
        max_len = std::max(max_len, seg_off[s + 1] - seg_off[s]);
    }
    const uint64_t n = seg_off[n_seg];
    if (n == 0) return CSV_OK;
    if (!start || !end || !labels) { ctx->err = "dbscan batch: null array"; return CSV_EINVAL; }
    if (n >= 0xffffffffull) { ctx->err = "dbscan batch: n too large"; return CSV_EINVAL; }
    (void)hipSetDevice(ctx->device);
    if ((rc = arena_reserve(ctx, ctx->arena, 3 * align_up(n * 4, 256) + align_up((n_seg + 1) * 8, 256) + 1024))) return rc;
    uint32_t *ds = (uint32_t *)arena_alloc(ctx->arena, n * 4), *de = (uint32_t *)arena_alloc(ctx->arena, n * 4);
    int32_t *dl = (int32_t *)arena_alloc(ctx->arena, n * 4);
    uint64_t *doff = (uint64_t *)arena_alloc(ctx->arena, (n_seg + 1) * 8);
    if (!ds || !de || !dl || !doff) { ctx->err = "arena exhausted"; return CSV_ENOMEM; }
    hipStream_t st = ctx->stream;
    if ((rc = ensure_pinned(ctx, 3 * PinStage::need(n * 4) + PinStage::need((n_seg + 1) * 8) + 4096))) return rc;
    PinStage pin(ctx);
    CSV_HIP(ctx, hipMemcpyAsync(ds, pin.in(start, n * 4), n * 4, hipMemcpyHostToDevice, st));
    CSV_HIP(ctx, hipMemcpyAsync(de, pin.in(end, n * 4), n * 4, hipMemcpyHostToDevice, st));
    CSV_HIP(ctx, hipMemcpyAsync(doff, pin.in(seg_off, (n_seg + 1) * 8), (n_seg + 1) * 8, hipMemcpyHostToDevice, st));
    {
        TimerScope ts(ctx, CSV_K_DBSCAN);
        launch_dbscan_iv_small_batched(st, ds, de, doff, n_seg, eps, min_pts, dl);
    }
    if (max_len > DBSCAN_IV_SMALL_MAX) {                     // the few sets that do not fit a workgroup's LDS: windowed path, one at a time
        CSV_HIP(ctx, wait_stream(st));                       // (that path reads its sortedness flag back through the same page-locked block)
        const size_t keep = pin.used;
        for (uint64_t s = 0; s < n_seg; s++) {
            const uint64_t len = seg_off[s + 1] - seg_off[s];
            if (len <= DBSCAN_IV_SMALL_MAX) continue;
            if ((rc = csvgpu_dbscan_iv_dev(ctx, ds + seg_off[s], de + seg_off[s], len, eps, min_pts, dl + seg_off[s]))) return rc;
        }
        pin.used = keep;
        if ((rc = ensure_pinned(ctx, keep + PinStage::need(n * 4) + 4096))) return rc;      // (a no-op: sized above)
    }
    CSV_HIP(ctx, hipMemcpyAsync(pin.out(labels, n * 4), dl, n * 4, hipMemcpyDeviceToHost, st));
    CSV_HIP(ctx, wait_stream(st));
    pin.finish();
    return CSV_OK;
}

int csvgpu_dbscan_1d_dev(csv_ctx *ctx, const int32_t *d_pts, const uint64_t *d_seg_off, uint64_t n_seg, uint64_t n_pts,
                         uint32_t max_seg_len, double eps, int32_t min_pts, int32_t *d_labels)
{
    int rc = check_dbscan_args(ctx, eps, min_pts, false);
    if (rc) return rc;
    if (n_seg == 0) return CSV_OK;
    if (!d_seg_off || (n_pts && (!d_pts || !d_labels))) { ctx->err = "dbscan1d: null array"; return CSV_EINVAL; }
    (void)hipSetDevice(ctx->device);
    const bool has_big = max_seg_len > DBSCAN1D_MAX_SEG;
    size_t need = 1024 + (has_big ? (n_seg + 1) * 8 + dbscan1d_big_tmp_bytes(max_seg_len) + sortws_bytes(max_seg_len) + align_up((size_t)max_seg_len * 4, 256) : 0);
    if ((rc = arena_reserve(ctx, ctx->work, need))) return rc;
    unsigned int *flag = (unsigned int *)arena_alloc(ctx->work, 256);
    CSV_HIP(ctx, hipMemsetAsync(flag, 0, 4, ctx->stream));
    {
        TimerScope ts(ctx, CSV_K_DBSCAN1D);
        launch_dbscan_1d_batched(ctx->stream, d_pts, d_seg_off, n_seg, eps, min_pts, d_labels, flag);
    }
    if (!has_big) return CSV_OK;
    // segments longer than the LDS kernel's limit: generic sorted-window path, one segment at a time
    std::vector<uint64_t> off(n_seg + 1);
    CSV_HIP(ctx, hipMemcpyAsync(off.data(), d_seg_off, (n_seg + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    CSV_HIP(ctx, wait_stream(ctx->stream));
    SortWs w;
    uint32_t *ks = (uint32_t *)arena_alloc(ctx->work, (size_t)max_seg_len * 4);
    void *tmp = arena_alloc(ctx->work, dbscan1d_big_tmp_bytes(max_seg_len));
    if (!sortws_carve(ctx->work, max_seg_len, w) || !ks || !tmp) { ctx->err = "arena exhausted (dbscan1d big)"; return CSV_ENOMEM; }
    for (uint64_t s = 0; s < n_seg; s++) {
        const uint64_t n = off[s + 1] - off[s];
        if (n <= DBSCAN1D_MAX_SEG) continue;
        if (n > max_seg_len) { ctx->err = "dbscan1d: max_seg_len smaller than a segment"; return CSV_EINVAL; }
        TimerScope ts(ctx, CSV_K_DBSCAN1D);
        launch_iota_keys_i32(ctx->stream, d_pts + off[s], n, w.k0, w.v0);
        const int io = launch_radix_sort_u64(ctx->stream, w.k0, w.v0, w.k1, w.v1, n, 32, w.tmp);
        const uint32_t *perm = io ? w.v1 : w.v0;
        launch_gather_u32(ctx->stream, (const uint32_t *)(d_pts + off[s]), perm, n, ks);   // points in sorted order
        launch_dbscan_1d_big(ctx->stream, (const int32_t *)ks, perm, n, eps, min_pts, d_labels + off[s], tmp);
    }
    return CSV_OK;
}

int csvgpu_dbscan_1d(csv_ctx *ctx, const int32_t *pts, const uint64_t *seg_off, uint64_t n_seg, double eps, int32_t min_pts,
                     int32_t *labels)
{
    int rc = check_dbscan_args(ctx, eps, min_pts, false);
    if (rc) return rc;
    if (n_seg == 0) return CSV_OK;
    if (!seg_off) { ctx->err = "dbscan1d: null seg_off"; return CSV_EINVAL; }
    const uint64_t n = seg_off[n_seg];
    uint64_t max_len = 0;
    for (uint64_t s = 0; s < n_seg; s++) {
        if (seg_off[s + 1] < seg_off[s]) { ctx->err = "dbscan1d: seg_off not monotone"; return CSV_EINVAL; }
        max_len = std::max(max_len, seg_off[s + 1] - seg_off[s]);
    }
    if (n == 0) return CSV_OK;
    if (!pts || !labels) { ctx->err = "dbscan1d: null array"; return CSV_EINVAL; }
    if (max_len >= 0xffffffffull) { ctx->err = "dbscan1d: segment too large"; return CSV_EINVAL; }
    (void)hipSetDevice(ctx->device);
    if ((rc = arena_reserve(ctx, ctx->arena, 2 * align_up(n * 4, 256) + align_up((n_seg + 1) * 8, 256) + 1024))) return rc;
    int32_t *dp = (int32_t *)arena_alloc(ctx->arena, n * 4), *dl = (int32_t *)arena_alloc(ctx->arena, n * 4);
    uint64_t *doff = (uint64_t *)arena_alloc(ctx->arena, (n_seg + 1) * 8);
    if (!dp || !dl || !doff) { ctx->err = "arena exhausted"; return CSV_ENOMEM; }
    if ((rc = ensure_pinned(ctx, 2 * PinStage::need(n * 4) + PinStage::need((n_seg + 1) * 8) + 4096))) return rc;
    PinStage pin(ctx);
    CSV_HIP(ctx, hipMemcpyAsync(dp, pin.in(pts, n * 4), n * 4, hipMemcpyHostToDevice, ctx->stream));
    CSV_HIP(ctx, hipMemcpyAsync(doff, pin.in(seg_off, (n_seg + 1) * 8), (n_seg + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    if (max_len > DBSCAN1D_MAX_SEG) CSV_HIP(ctx, wait_stream(ctx->stream));          // (the large-segment path may use the page-locked block itself)
    if ((rc = csvgpu_dbscan_1d_dev(ctx, dp, doff, n_seg, n, (uint32_t)max_len, eps, min_pts, dl))) return rc;
    if (max_len > DBSCAN1D_MAX_SEG && (rc = ensure_pinned(ctx, 2 * PinStage::need(n * 4) + PinStage::need((n_seg + 1) * 8) + 4096))) return rc;
    CSV_HIP(ctx, hipMemcpyAsync(pin.out(labels, n * 4), dl, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    CSV_HIP(ctx, wait_stream(ctx->stream));
    pin.finish();
    return CSV_OK;
}

int csvgpu_window_log2_dev(csv_ctx *ctx, const uint32_t *d_depth, uint32_t depth_len, const uint32_t *d_rs, const uint32_t *d_re,
                           const int32_t *d_ss, const uint64_t *d_win_off, uint64_t n_regions, uint64_t n_windows,
                           double mean_cov, double *d_log2, uint32_t *d_ws, uint32_t *d_we)
{
    if (!ctx) return CSV_EINVAL;
    if (n_regions == 0 || n_windows == 0) return CSV_OK;
    if (!d_depth || !d_rs || !d_re || !d_ss || !d_win_off || !d_log2 || !d_ws || !d_we) { ctx->err = "window_log2: null array"; return CSV_EINVAL; }
    (void)hipSetDevice(ctx->device);
    TimerScope ts(ctx, CSV_K_WINDOW);
    launch_window_log2(ctx->stream, d_depth, depth_len, d_rs, d_re, d_ss, d_win_off, n_regions, n_windows, mean_cov, d_log2, d_ws, d_we);
    return CSV_OK;
}

int csvgpu_window_log2(csv_ctx *ctx, const uint32_t *depth, uint32_t depth_len, const uint32_t *region_start,
                       const uint32_t *region_end, const int32_t *sample_size, const uint64_t *win_off, uint64_t n_regions,
                       double mean_cov, double *log2_cov, uint32_t *win_start, uint32_t *win_end)
{
    if (!ctx) return CSV_EINVAL;
    if (n_regions == 0) return CSV_OK;
    if (!depth || !region_start || !region_end || !sample_size || !win_off) { ctx->err = "window_log2: null array"; return CSV_EINVAL; }
    for (uint64_t r = 0; r < n_regions; r++) {
        if (sample_size[r] <= 0 || win_off[r + 1] - win_off[r] != (uint64_t)sample_size[r] || region_start[r] > region_end[r]) {
            ctx->err = "window_log2: bad region table"; return CSV_EINVAL;
        }
    }
    const uint64_t nw = win_off[n_regions];
    if (nw == 0) return CSV_OK;
    if (!log2_cov || !win_start || !win_end) { ctx->err = "window_log2: null output"; return CSV_EINVAL; }
    (void)hipSetDevice(ctx->device);
    int rc = arena_reserve(ctx, ctx->arena, align_up((size_t)depth_len * 4, 256) + 3 * align_up(n_regions * 4, 256) + align_up((n_regions + 1) * 8, 256) +
                                                align_up(nw * 8, 256) + 2 * align_up(nw * 4, 256) + 4096);
    if (rc) return rc;
    Arena &a = ctx->arena;
    uint32_t *dd = (uint32_t *)arena_alloc(a, (size_t)depth_len * 4), *drs = (uint32_t *)arena_alloc(a, n_regions * 4), *dre = (uint32_t *)arena_alloc(a, n_regions * 4);
    int32_t *dss = (int32_t *)arena_alloc(a, n_regions * 4);
    uint64_t *dwo = (uint64_t *)arena_alloc(a, (n_regions + 1) * 8);
    double *dl2 = (double *)arena_alloc(a, nw * 8);
    uint32_t *dws = (uint32_t *)arena_alloc(a, nw * 4), *dwe = (uint32_t *)arena_alloc(a, nw * 4);
    if (!dd || !drs || !dre || !dss || !dwo || !dl2 || !dws || !dwe) { ctx->err = "arena exhausted"; return CSV_ENOMEM; }
    hipStream_t s = ctx->stream;
    CSV_HIP(ctx, hipMemcpyAsync(dd, depth, (size_t)depth_len * 4, hipMemcpyHostToDevice, s));
    CSV_HIP(ctx, hipMemcpyAsync(drs, region_start, n_regions * 4, hipMemcpyHostToDevice, s));
    CSV_HIP(ctx, hipMemcpyAsync(dre, region_end, n_regions * 4, hipMemcpyHostToDevice, s));
    CSV_HIP(ctx, hipMemcpyAsync(dss, sample_size, n_regions * 4, hipMemcpyHostToDevice, s));
    CSV_HIP(ctx, hipMemcpyAsync(dwo, win_off, (n_regions + 1) * 8, hipMemcpyHostToDevice, s));
    if ((rc = csvgpu_window_log2_dev(ctx, dd, depth_len, drs, dre, dss, dwo, n_regions, nw, mean_cov, dl2, dws, dwe))) return rc;
    CSV_HIP(ctx, hipMemcpyAsync(log2_cov, dl2, nw * 8, hipMemcpyDeviceToHost, s));
    CSV_HIP(ctx, hipMemcpyAsync(win_start, dws, nw * 4, hipMemcpyDeviceToHost, s));
    CSV_HIP(ctx, hipMemcpyAsync(win_end, dwe, nw * 4, hipMemcpyDeviceToHost, s));
    CSV_HIP(ctx, wait_stream(s));
    return CSV_OK;
}

int csvgpu_viterbi_dev(csv_ctx *ctx, const csv_hmm *hmm, const double *d_o1, const double *d_o2, const double *d_pfb,
                       const uint64_t *d_seq_off, uint64_t n_seq, uint64_t n_obs, int32_t *d_states, double *d_loglik)
{
    if (!ctx) return CSV_EINVAL;
    if (!hmm) { ctx->err = "viterbi: null hmm"; return CSV_EINVAL; }
    if (n_seq == 0) return CSV_OK;
    if (!d_seq_off || !d_loglik || (n_obs && (!d_o1 || !d_o2 || !d_pfb || !d_states))) { ctx->err = "viterbi: null array"; return CSV_EINVAL; }
    (void)hipSetDevice(ctx->device);
    int rc = arena_reserve(ctx, ctx->work, viterbi_tmp_bytes(n_obs, n_seq) + 1024);
    if (rc) return rc;
    void *tmp = arena_alloc(ctx->work, viterbi_tmp_bytes(n_obs, n_seq));
    if (!tmp) { ctx->err = "arena exhausted"; return CSV_ENOMEM; }
    TimerScope ts(ctx, CSV_K_VITERBI);
    launch_viterbi(ctx->stream, *hmm, d_o1, d_o2, d_pfb, d_seq_off, n_seq, n_obs, d_states, d_loglik, tmp);
    return CSV_OK;
}

int csvgpu_viterbi(csv_ctx *ctx, const csv_hmm *hmm, const double *o1, const double *o2, const double *pfb,
                   const uint64_t *seq_off, uint64_t n_seq, int32_t *states, double *loglik)
{
    if (!ctx) return CSV_EINVAL;
    if (!hmm) { ctx->err = "viterbi: null hmm"; return CSV_EINVAL; }
    if (n_seq == 0) return CSV_OK;
    if (!seq_off || !loglik) { ctx->err = "viterbi: null array"; return CSV_EINVAL; }
    for (uint64_t s = 0; s < n_seq; s++) if (seq_off[s + 1] < seq_off[s]) { ctx->err = "viterbi: seq_off not monotone"; return CSV_EINVAL; }
    const uint64_t n = seq_off[n_seq];
    if (n && (!o1 || !o2 || !pfb || !states)) { ctx->err = "viterbi: null array"; return CSV_EINVAL; }
    (void)hipSetDevice(ctx->device);
    int rc = arena_reserve(ctx, ctx->arena, 3 * align_up(n * 8, 256) + align_up((n_seq + 1) * 8, 256) + align_up(n * 4, 256) + align_up(n_seq * 8, 256) + 4096);
    if (rc) return rc;
    Arena &a = ctx->arena;
    double *d1 = (double *)arena_alloc(a, n * 8 + 8), *d2 = (double *)arena_alloc(a, n * 8 + 8), *dp = (double *)arena_alloc(a, n * 8 + 8);
    uint64_t *doff = (uint64_t *)arena_alloc(a, (n_seq + 1) * 8);
    int32_t *dst = (int32_t *)arena_alloc(a, n * 4 + 8);
    double *dll = (double *)arena_alloc(a, n_seq * 8);
    if (!d1 || !d2 || !dp || !doff || !dst || !dll) { ctx->err = "arena exhausted"; return CSV_ENOMEM; }
    hipStream_t s = ctx->stream;
    if ((rc = ensure_pinned(ctx, 3 * PinStage::need(n * 8) + PinStage::need((n_seq + 1) * 8) + PinStage::need(n * 4) + PinStage::need(n_seq * 8) + 4096))) return rc;
    PinStage pin(ctx);
    if (n) {
        CSV_HIP(ctx, hipMemcpyAsync(d1, pin.in(o1, n * 8), n * 8, hipMemcpyHostToDevice, s));
        CSV_HIP(ctx, hipMemcpyAsync(d2, pin.in(o2, n * 8), n * 8, hipMemcpyHostToDevice, s));
        CSV_HIP(ctx, hipMemcpyAsync(dp, pin.in(pfb, n * 8), n * 8, hipMemcpyHostToDevice, s));
    }
    CSV_HIP(ctx, hipMemcpyAsync(doff, pin.in(seq_off, (n_seq + 1) * 8), (n_seq + 1) * 8, hipMemcpyHostToDevice, s));
    if ((rc = csvgpu_viterbi_dev(ctx, hmm, d1, d2, dp, doff, n_seq, n, dst, dll))) return rc;
    if (n) CSV_HIP(ctx, hipMemcpyAsync(pin.out(states, n * 4), dst, n * 4, hipMemcpyDeviceToHost, s));
    CSV_HIP(ctx, hipMemcpyAsync(pin.out(loglik, n_seq * 8), dll, n_seq * 8, hipMemcpyDeviceToHost, s));
    CSV_HIP(ctx, wait_stream(s));
    pin.finish();
    return CSV_OK;
}

// ---------------------------------------------------------------------------------------------
// resident shards + the per-chromosome pipeline

static void shard_release(csv_shard *sh)
{
    if (!sh) return;
    if (sh->owned) {
        (void)hipFree((void *)sh->d.pos); (void)hipFree((void *)sh->d.flag); (void)hipFree((void *)sh->d.mapq);
        (void)hipFree((void *)sh->d.cigar_off); (void)hipFree((void *)sh->d.cigar);
    }
    (void)hipFree(sh->ref_end); (void)hipFree(sh->q_start); (void)hipFree(sh->q_end);
    (void)hipFree(sh->ckpt);
    (void)hipFree(sh->depth_items);
    (void)hipFree(sh->scan_split);
    (void)hipFree(sh->qhash);
    (void)hipFree(sh->depth); (void)hipFree(sh->sig_raw); (void)hipFree(sh->scratch); (void)hipFree(sh->counters);
    delete sh;
}

static csv_shard *shard_common(csv_ctx *ctx, csv_shard *sh)
{
    const uint64_t n = sh->d.n_reads;
    bool ok = true;
    ok &= hipMalloc((void **)&sh->ref_end, n * 4 + 16) == hipSuccess;
    ok &= hipMalloc((void **)&sh->q_start, n * 4 + 16) == hipSuccess;
    ok &= hipMalloc((void **)&sh->q_end, n * 4 + 16) == hipSuccess;
    ok &= hipMalloc((void **)&sh->depth, (size_t)sh->depth_len * 4 + 16) == hipSuccess;
    sh->counters_bytes = align_up(kCntBytes, 256) + depth_tiles_tmp_bytes(sh->depth_len);
    ok &= hipMalloc((void **)&sh->counters, sh->counters_bytes) == hipSuccess;
    sh->tile_range = (uint64_t *)((char *)sh->counters + align_up(kCntBytes, 256));
    ok &= hipMalloc((void **)&sh->ckpt, ckpt_bytes(sh->d.n_cigar)) == hipSuccess;
    ok &= hipMalloc(&sh->depth_items, depth_items_bytes(sh->depth_len) + 16) == hipSuccess;
    sh->form = scan_form_for(sh->d.n_reads, sh->d.n_cigar);
    ok &= hipMalloc((void **)&sh->scan_split, scan_split_bytes(ctx->n_cu, sh->d.n_reads, sh->form) + 16) == hipSuccess;
    sh->sig_cap = std::max<uint64_t>(1u << 18, n * 2);
    ok &= hipMalloc((void **)&sh->sig_raw, sh->sig_cap * sizeof(csv_sig)) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); ctx->err = "hipMalloc failed (shard)"; shard_release(sh); return nullptr; }
    // the scan's work split for this device's grid, once per shard (the offsets are on the device by now)
    launch_scan_split(ctx->stream, ctx->n_cu, sh->d, sh->scan_split, sh->form);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "scan split failed (shard)"; shard_release(sh); return nullptr; }
    return sh;
}

csv_shard *csvgpu_shard_upload(csv_ctx *ctx, const csv_reads *r, uint32_t depth_len)
{
    if (check_reads(ctx, r)) return nullptr;
    (void)hipSetDevice(ctx->device);
    csv_shard *sh = new (std::nothrow) csv_shard();
    if (!sh) { ctx->err = "out of host memory"; return nullptr; }
    sh->owned = true; sh->depth_len = depth_len; sh->d = *r; sh->d.tid = nullptr;
    sh->d.pos = nullptr; sh->d.flag = nullptr; sh->d.mapq = nullptr; sh->d.cigar_off = nullptr; sh->d.cigar = nullptr;
    const uint64_t n = r->n_reads, m = r->n_cigar;
    bool ok = true;
    ok &= hipMalloc((void **)&sh->d.pos, n * 4 + 16) == hipSuccess;
    ok &= hipMalloc((void **)&sh->d.flag, n * 2 + 16) == hipSuccess;
    ok &= hipMalloc((void **)&sh->d.mapq, n + 16) == hipSuccess;
    ok &= hipMalloc((void **)&sh->d.cigar_off, (n + 1) * 8) == hipSuccess;
    ok &= hipMalloc((void **)&sh->d.cigar, (m + CIGAR_PAD_WORDS) * 4) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); ctx->err = "hipMalloc failed (shard upload)"; shard_release(sh); return nullptr; }
    sh->cigar_pad = CIGAR_PAD_WORDS;
    hipStream_t s = ctx->stream;
    sh->unsorted = 0;                                    // known before the first scan: lets the pipeline queue the depth pass without waiting
    for (uint64_t i = 1; i < n; i++) if (r->pos[i] < r->pos[i - 1]) { sh->unsorted = 1; break; }
    bool cp = true;
    if (n) {
        cp &= hipMemcpyAsync((void *)sh->d.pos, r->pos, n * 4, hipMemcpyHostToDevice, s) == hipSuccess;
        cp &= hipMemcpyAsync((void *)sh->d.flag, r->flag, n * 2, hipMemcpyHostToDevice, s) == hipSuccess;
        cp &= hipMemcpyAsync((void *)sh->d.mapq, r->mapq, n, hipMemcpyHostToDevice, s) == hipSuccess;
    }
    cp &= hipMemcpyAsync((void *)sh->d.cigar_off, r->cigar_off, (n + 1) * 8, hipMemcpyHostToDevice, s) == hipSuccess;
    if (m) cp &= hipMemcpyAsync((void *)sh->d.cigar, r->cigar, m * 4, hipMemcpyHostToDevice, s) == hipSuccess;
    cp &= hipMemsetAsync((void *)(sh->d.cigar + m), 0, (size_t)CIGAR_PAD_WORDS * 4, s) == hipSuccess;
    cp &= hipStreamSynchronize(s) == hipSuccess;
    if (!cp) { ctx->err = "H2D copy failed (shard upload)"; shard_release(sh); return nullptr; }
    return shard_common(ctx, sh);
}

csv_shard *csvgpu_shard_wrap_dev(csv_ctx *ctx, const csv_reads *r, uint32_t depth_len)
{
    if (!ctx) return nullptr;
    (void)hipSetDevice(ctx->device);
    if (check_reads_dev(ctx, r)) return nullptr;
    csv_shard *sh = new (std::nothrow) csv_shard();
    if (!sh) { ctx->err = "out of host memory"; return nullptr; }
    sh->owned = false; sh->depth_len = depth_len; sh->d = *r;
    return shard_common(ctx, sh);
}

void csvgpu_shard_free(csv_ctx *ctx, csv_shard *sh)
{
    if (!ctx || !sh) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    shard_release(sh);
}

int csvgpu_aln_intervals_resident(csv_ctx *ctx, csv_shard *sh, int32_t *ref_end, int32_t *q_start, int32_t *q_end)
{
    if (!ctx || !sh) return CSV_EINVAL;
    const uint64_t n = sh->d.n_reads;
    if (n == 0) return CSV_OK;
    if (!ref_end || !q_start || !q_end) { ctx->err = "aln_intervals: null output"; return CSV_EINVAL; }
    (void)hipSetDevice(ctx->device);
    CSV_HIP(ctx, hipMemcpyAsync(ref_end, sh->ref_end, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    CSV_HIP(ctx, hipMemcpyAsync(q_start, sh->q_start, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    CSV_HIP(ctx, hipMemcpyAsync(q_end, sh->q_end, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    CSV_HIP(ctx, wait_stream(ctx->stream));
    return CSV_OK;
}

int csvgpu_window_log2_resident_many(csv_ctx *ctx, int n_shards, csv_shard *const *shards, const uint32_t *const *region_start,
                                     const uint32_t *const *region_end, const int32_t *const *sample_size, const uint64_t *const *win_off,
                                     const uint64_t *n_regions, const double *mean_cov, double *const *log2_cov, uint32_t *const *win_start,
                                     uint32_t *const *win_end)
{
    if (!ctx || n_shards < 0) return CSV_EINVAL;
    if (n_shards == 0) return CSV_OK;
    if (!shards || !region_start || !region_end || !sample_size || !win_off || !n_regions || !mean_cov || !log2_cov || !win_start || !win_end) {
        ctx->err = "window_log2_many: null table"; return CSV_EINVAL;
    }
    uint64_t R = 0, W = 0;
    for (int c = 0; c < n_shards; c++) {
        const uint64_t nr = n_regions[c];
        if (!nr) continue;
        if (!shards[c] || !region_start[c] || !region_end[c] || !sample_size[c] || !win_off[c]) { ctx->err = "window_log2_many: null array"; return CSV_EINVAL; }
        for (uint64_t r = 0; r < nr; r++)
            if (sample_size[c][r] <= 0 || win_off[c][r + 1] - win_off[c][r] != (uint64_t)sample_size[c][r] || region_start[c][r] > region_end[c][r] || win_off[c][0] != 0) {
                ctx->err = "window_log2_many: bad region table"; return CSV_EINVAL;
            }
        if (win_off[c][nr] && (!log2_cov[c] || !win_start[c] || !win_end[c])) { ctx->err = "window_log2_many: null output"; return CSV_EINVAL; }
        R += nr; W += win_off[c][nr];
    }
    if (W == 0) return CSV_OK;
    (void)hipSetDevice(ctx->device);
    // one page-locked block carries every shard's tables to the device and every shard's windows back
    const size_t in_bytes = align_up(R * 4, 8) * 3 + (R + (size_t)n_shards) * 8, out_bytes = W * 8 + 2 * align_up(W * 4, 8);
    int rc = ensure_pinned(ctx, in_bytes + out_bytes + 4096);
    if (rc) return rc;
    if ((rc = arena_reserve(ctx, ctx->arena, in_bytes + out_bytes + 8192))) return rc;
    char *h_in = (char *)ctx->pinned, *h_out = h_in + align_up(in_bytes, 256);
    char *d_in = (char *)arena_alloc(ctx->arena, in_bytes), *d_out = (char *)arena_alloc(ctx->arena, out_bytes);
    if (!d_in || !d_out) { ctx->err = "arena exhausted"; return CSV_ENOMEM; }
    const size_t o_rs = 0, o_re = align_up(R * 4, 8), o_ss = 2 * align_up(R * 4, 8), o_wo = 3 * align_up(R * 4, 8);
    const size_t o_l2 = 0, o_ws = W * 8, o_we = W * 8 + align_up(W * 4, 8);
    uint64_t r0 = 0, w0 = 0;
    for (int c = 0; c < n_shards; c++) {
        const uint64_t nr = n_regions[c];
        if (!nr) continue;
        memcpy(h_in + o_rs + r0 * 4, region_start[c], nr * 4);
        memcpy(h_in + o_re + r0 * 4, region_end[c], nr * 4);
        memcpy(h_in + o_ss + r0 * 4, sample_size[c], nr * 4);
        memcpy(h_in + o_wo + (r0 + (uint64_t)c) * 8, win_off[c], (nr + 1) * 8);
        r0 += nr;
    }
    hipStream_t s = ctx->stream;
    CSV_HIP(ctx, hipMemcpyAsync(d_in, h_in, in_bytes, hipMemcpyHostToDevice, s));
    {
        TimerScope ts(ctx, CSV_K_WINDOW);
        r0 = 0;
        for (int c = 0; c < n_shards; c++) {
            const uint64_t nr = n_regions[c];
            if (!nr) continue;
            const uint64_t nw = win_off[c][nr];
            launch_window_log2(s, shards[c]->depth, shards[c]->depth_len, (const uint32_t *)(d_in + o_rs) + r0, (const uint32_t *)(d_in + o_re) + r0,
                               (const int32_t *)(d_in + o_ss) + r0, (const uint64_t *)(d_in + o_wo) + r0 + c, nr, nw, mean_cov[c],
                               (double *)(d_out + o_l2) + w0, (uint32_t *)(d_out + o_ws) + w0, (uint32_t *)(d_out + o_we) + w0);
            r0 += nr; w0 += nw;
        }
    }
    CSV_HIP(ctx, hipMemcpyAsync(h_out, d_out, out_bytes, hipMemcpyDeviceToHost, s));
    CSV_HIP(ctx, wait_stream(s));
    w0 = 0;
    for (int c = 0; c < n_shards; c++) {
        const uint64_t nr = n_regions[c];
        if (!nr) continue;
        const uint64_t nw = win_off[c][nr];
        memcpy(log2_cov[c], h_out + o_l2 + w0 * 8, nw * 8);
        memcpy(win_start[c], h_out + o_ws + w0 * 4, nw * 4);
        memcpy(win_end[c], h_out + o_we + w0 * 4, nw * 4);
        w0 += nw;
    }
    return CSV_OK;
}

int csvgpu_aln_intervals_gather_batch(csv_ctx *ctx, int n_shards, csv_shard *const *shards, const uint32_t *rec, const uint64_t *rec_off,
                                      int32_t *ref_end, int32_t *q_start, int32_t *q_end)
{
    if (!ctx || n_shards < 0) return CSV_EINVAL;
    if (n_shards == 0) return CSV_OK;
    if (!shards || !rec_off) { ctx->err = "aln_intervals_gather: null array"; return CSV_EINVAL; }
    const uint64_t n = rec_off[n_shards];
    if (n == 0) return CSV_OK;
    if (!rec || !ref_end || !q_start || !q_end) { ctx->err = "aln_intervals_gather: null array"; return CSV_EINVAL; }
    for (int c = 0; c < n_shards; c++) {
        if (!shards[c] || rec_off[c + 1] < rec_off[c]) { ctx->err = "aln_intervals_gather: bad shard table"; return CSV_EINVAL; }
        for (uint64_t i = rec_off[c]; i < rec_off[c + 1]; i++) if (rec[i] >= shards[c]->d.n_reads) { ctx->err = "aln_intervals_gather: record index beyond the shard"; return CSV_EINVAL; }
    }
    (void)hipSetDevice(ctx->device);
    int rc = arena_reserve(ctx, ctx->arena, 4 * align_up(n * 4, 256) + 4096);
    if (rc) return rc;
    if ((rc = ensure_pinned(ctx, 4 * align_up(n * 4, 256) + 4096))) return rc;      // the index list goes out and the three arrays come back through one page-locked block
    uint32_t *didx = (uint32_t *)arena_alloc(ctx->arena, n * 4);
    uint32_t *dout = (uint32_t *)arena_alloc(ctx->arena, 3 * n * 4);
    if (!didx || !dout) { ctx->err = "arena exhausted"; return CSV_ENOMEM; }
    uint32_t *h_idx = (uint32_t *)ctx->pinned, *h_out = h_idx + align_up(n * 4, 256) / 4;
    memcpy(h_idx, rec, n * 4);
    hipStream_t s = ctx->stream;
    CSV_HIP(ctx, hipMemcpyAsync(didx, h_idx, n * 4, hipMemcpyHostToDevice, s));
    for (int c = 0; c < n_shards; c++) {
        const uint64_t o = rec_off[c], m = rec_off[c + 1] - o;
        if (!m) continue;
        const csv_shard *sh = shards[c];
        launch_gather3_u32(s, (const uint32_t *)sh->ref_end, (const uint32_t *)sh->q_start, (const uint32_t *)sh->q_end, didx + o, m, dout + o, dout + n + o, dout + 2 * n + o);
    }
    CSV_HIP(ctx, hipMemcpyAsync(h_out, dout, 3 * n * 4, hipMemcpyDeviceToHost, s));
    CSV_HIP(ctx, wait_stream(s));
    memcpy(ref_end, h_out, n * 4); memcpy(q_start, h_out + n, n * 4); memcpy(q_end, h_out + 2 * n, n * 4);
    return CSV_OK;
}

int csvgpu_aln_intervals_gather_resident(csv_ctx *ctx, csv_shard *sh, const uint32_t *rec, uint64_t n, int32_t *ref_end, int32_t *q_start, int32_t *q_end)
{
    const uint64_t off[2] = {0, n};
    return csvgpu_aln_intervals_gather_batch(ctx, 1, &sh, rec, off, ref_end, q_start, q_end);
}

int csvgpu_shard_set_qname_hash(csv_ctx *ctx, csv_shard *sh, const uint64_t *qname_hash)
{
    if (!ctx || !sh) return CSV_EINVAL;
    const uint64_t n = sh->d.n_reads;
    if (n && !qname_hash) { ctx->err = "set_qname_hash: null array"; return CSV_EINVAL; }
    (void)hipSetDevice(ctx->device);
    if (!sh->qhash) {
        if (hipMalloc((void **)&sh->qhash, n * 8 + 16) != hipSuccess) { (void)hipGetLastError(); sh->qhash = nullptr; ctx->err = "hipMalloc failed (qname hashes)"; return CSV_ENOMEM; }
    }
    if (n) CSV_HIP(ctx, hipMemcpyAsync(sh->qhash, qname_hash, n * 8, hipMemcpyHostToDevice, ctx->stream));
    CSV_HIP(ctx, wait_stream(ctx->stream));
    return CSV_OK;
}

// the epochs of a libstdc++ hash table that grows by single insertions: node index at which each rehash happens, and the bucket
// count from there on — asked of the library's own policy object (what std::unordered_map itself consults)
static void split_order_epochs(uint64_t n_max, std::vector<uint64_t> &first_node, std::vector<uint64_t> &buckets)
{
    static std::mutex mu;
    static std::vector<uint64_t> c_first, c_bkt;
    static uint64_t covered = 0;                              // the plan is known for tables of up to `covered` nodes
    std::lock_guard<std::mutex> l(mu);
    if (n_max > covered) {
        c_first.clear(); c_bkt.clear();
        std::__detail::_Prime_rehash_policy pol;
        std::size_t nb = 1;
        const uint64_t want = std::max<uint64_t>(n_max, 1u << 20);
        for (uint64_t i = 0; i < want;) {
            const std::pair<bool, std::size_t> g = pol._M_need_rehash(nb, i, 1);
            if (g.first) { nb = g.second; c_first.push_back(i); c_bkt.push_back(nb); }
            // nothing can happen before the table is full again (max_load_factor 1): jump there
            i = (g.first || i + 1 >= nb) ? i + 1 : std::min<uint64_t>(want, (uint64_t)nb);
        }
        covered = want;
    }
    first_node = c_first; buckets = c_bkt;
}

}  // extern "C"

// What csvgpu_split_order_begin leaves for csvgpu_split_order_finish (one pending order per context; device pointers into ctx->arena / ctx->work).
struct csv_split_state {
    int n_contigs = 0;
    SplitOrderTab tab;
    std::vector<uint64_t> N;
    uint64_t n_nodes = 0, n_max = 0, total_reads = 0;
    int D = 0;
    SplitTailHost th;
    uint64_t *node_hash = nullptr, *d_supp = nullptr;
    uint32_t *node_rec = nullptr, *list = nullptr, *minT = nullptr;
    SortWs w;
    csv_split_survivor *d_out = nullptr;
    unsigned long long *d_count = nullptr;
    uint32_t *bitmap[SO_TAIL_MAX] = {nullptr, nullptr, nullptr}, *set[SO_TAIL_MAX + 1] = {nullptr, nullptr, nullptr, nullptr}, *prevrank = nullptr, *filter = nullptr;
    uint8_t *is_surv = nullptr;
    size_t bm_words = 0;
    bool finished = false;
    bool self = false;                 // the supplementary hashes are taken from the same shards: the whole order was queued by _begin
    uint64_t self_bound = 0;           // survivors the page-locked block has room for (self)
    std::vector<csv_split_survivor> surv;
    std::vector<uint64_t> off;
};

static void split_state_free(csv_ctx *ctx) { if (ctx) { delete ctx->split_state; ctx->split_state = nullptr; } }

// nodes + every epoch that does not depend on the supplementary records: queued, not waited for (beyond the node counts)
static int split_order_tail(csv_ctx *ctx, csv_split_state *st, const uint64_t *d_supp, uint64_t n_supp, bool devn);

static int split_order_begin(csv_ctx *ctx, int n_contigs, csv_shard *const *shards, uint8_t min_mapq, int64_t n_supp_hint /* < 0: unknown */, bool self = false)
{
    if (!ctx) return CSV_EINVAL;
    delete ctx->split_state; ctx->split_state = nullptr;
    if (n_contigs < 0 || (uint32_t)n_contigs > SO_MAX_CONTIGS) { ctx->err = "split_order: at most 32 contigs per call"; return CSV_EINVAL; }
    if (n_contigs && !shards) { ctx->err = "split_order: null array"; return CSV_EINVAL; }
    std::unique_ptr<csv_split_state> st(new csv_split_state());
    st->n_contigs = n_contigs;
    st->off.assign((size_t)n_contigs + 1, 0);
    if (n_contigs == 0) { st->finished = true; ctx->split_state = st.release(); return CSV_OK; }
    (void)hipSetDevice(ctx->device);
    hipStream_t s = ctx->stream;
    SplitOrderTab &tab = st->tab;
    tab.A = (uint32_t)n_contigs;
    uint64_t total_reads = 0, n_blocks = 0;
    for (int c = 0; c < n_contigs; c++) {
        const csv_shard *sh = shards[c];
        if (!sh || (sh->d.n_reads && !sh->qhash)) { ctx->err = "split_order: a shard without query-name hashes (csvgpu_shard_set_qname_hash)"; return CSV_EINVAL; }
        tab.blk_off[c] = n_blocks;
        tab.n_reads[c] = sh->d.n_reads; tab.flag[c] = sh->d.flag; tab.mapq[c] = sh->d.mapq; tab.qhash[c] = sh->qhash;
        n_blocks += (sh->d.n_reads + 1023) / 1024;
        total_reads += sh->d.n_reads;
    }
    tab.blk_off[n_contigs] = n_blocks;
    st->total_reads = total_reads;
    if (total_reads == 0 || n_supp_hint == 0) { st->finished = true; ctx->split_state = st.release(); return CSV_OK; }
    if (total_reads >= 0xfffffff0ull) { ctx->err = "split_order: too many records in one call"; return CSV_EINVAL; }
    TimerScope ts(ctx, CSV_K_SPLIT_ORDER);

    // ---- nodes: the filter-passing primaries of every contig, file order ----
    // (the supplementary hashes arrive with _finish: at most one per record)
    int rc = arena_reserve(ctx, ctx->arena, align_up((n_blocks + 1) * 4, 256) + exclusive_sum_tmp_bytes(n_blocks + 1) + align_up(total_reads * 8, 256) +
                                                2 * align_up(total_reads * 4, 256) + align_up(total_reads * 8, 256) + 4096);
    if (rc) return rc;
    unsigned int *d_nsupp = (unsigned int *)arena_alloc(ctx->arena, 256);
    uint32_t *blk = (uint32_t *)arena_alloc(ctx->arena, (n_blocks + 1) * 4);
    void *es_tmp = arena_alloc(ctx->arena, exclusive_sum_tmp_bytes(n_blocks + 1));
    uint64_t *node_hash = (uint64_t *)arena_alloc(ctx->arena, total_reads * 8);
    uint32_t *node_rec = (uint32_t *)arena_alloc(ctx->arena, total_reads * 4), *list = (uint32_t *)arena_alloc(ctx->arena, total_reads * 4);
    st->d_supp = (uint64_t *)arena_alloc(ctx->arena, total_reads * 8);
    if (!d_nsupp || !blk || !es_tmp || !node_hash || !node_rec || !list || !st->d_supp) { ctx->err = "arena exhausted (split order)"; return CSV_ENOMEM; }
    st->node_hash = node_hash; st->node_rec = node_rec; st->list = list;
    CSV_HIP(ctx, hipMemsetAsync(blk + n_blocks, 0, 4, s));
    launch_so_count(s, tab, (uint32_t)n_blocks, min_mapq, blk);
    launch_exclusive_sum_u32(s, blk, n_blocks + 1, es_tmp);
    launch_so_scatter(s, tab, (uint32_t)n_blocks, min_mapq, blk, node_hash, node_rec);
    if ((rc = ensure_pinned(ctx, (n_blocks + 1) * 4 + 64 + 256))) return rc;
    CSV_HIP(ctx, hipMemcpyAsync(ctx->pinned, blk, (n_blocks + 1) * 4, hipMemcpyDeviceToHost, s));
    const size_t nsupp_at = align_up((n_blocks + 1) * 4, 64);
    if (self) {                                  // the supplementary records' hashes of the same shards (their count comes back with the node counts)
        CSV_HIP(ctx, hipMemsetAsync(d_nsupp, 0, 4, s));
        launch_so_supp(s, tab, (uint32_t)n_blocks, min_mapq, st->d_supp, d_nsupp);
        CSV_HIP(ctx, hipMemcpyAsync((char *)ctx->pinned + nsupp_at, d_nsupp, 4, hipMemcpyDeviceToHost, s));
    }
    CSV_HIP(ctx, wait_stream(s));
    const uint32_t *h_blk = (const uint32_t *)ctx->pinned;
    uint64_t n_supp_self = 0;
    if (self) {
        n_supp_self = *(const uint32_t *)((const char *)ctx->pinned + nsupp_at);
        n_supp_hint = (int64_t)n_supp_self;
        st->self = true;
        if (n_supp_self == 0) { st->finished = true; ctx->split_state = st.release(); return CSV_OK; }          // nothing survives
    }
    std::vector<uint64_t> &N = st->N;
    N.assign((size_t)n_contigs, 0);
    uint64_t n_nodes = h_blk[n_blocks], n_max = 0;
    for (int c = 0; c < n_contigs; c++) {
        tab.nbase[c] = h_blk[tab.blk_off[c]];
        N[(size_t)c] = (uint64_t)h_blk[tab.blk_off[c + 1]] - h_blk[tab.blk_off[c]];
        n_max = std::max(n_max, N[(size_t)c]);
    }
    tab.nbase[n_contigs] = (uint32_t)n_nodes;
    st->n_nodes = n_nodes; st->n_max = n_max;
    if (n_nodes == 0) { st->finished = true; ctx->split_state = st.release(); return CSV_OK; }

    // ---- the chain of epochs: a contig takes part in epoch k while it still has nodes inserted at or after the epoch's first node ----
    std::vector<uint64_t> first_node, buckets;
    split_order_epochs(n_max, first_node, buckets);
    // The last D epochs of every contig are ordered for the survivors (and the nodes their order depends on) only: splitorder.hip.
    // The sets double per level and the epochs halve, so D levels pay while 4^D <= nodes per supplementary record (about a hundred
    // in a long-read run: D = 3, also taken when the caller has not counted its supplementary records yet).
    int D = 0;
    {
        const uint64_t ratio = n_supp_hint > 0 ? n_nodes / (uint64_t)n_supp_hint : (n_nodes >= 4096 ? 64 : 1);
        while (D < (int)SO_TAIL_MAX && (ratio >> (2 * (D + 1))) >= 1) D++;
        const char *e = getenv("CSV_SPLIT_TAIL");
        if (e && *e) D = std::min<int>(std::max(atoi(e), 0), (int)SO_TAIL_MAX);
    }
    SplitTailHost &th = st->th;
    th.A = (uint32_t)n_contigs; th.wv = std::max(1, bits_of(n_nodes)); th.wa = std::max(1, bits_of((uint64_t)n_contigs - 1));
    if (n_nodes >= (1ull << 31) || th.wa + 2 * (th.wv + 1) > 64 || (self && n_nodes >= (1ull << 30))) D = 0;      // (self: the queued sorts count in 30 bits)
    th.D = (uint32_t)D;
    std::vector<int> K((size_t)n_contigs, -1);                       // a contig's last epoch
    for (int c = 0; c < n_contigs; c++) {
        for (size_t k = 0; k < first_node.size() && N[(size_t)c] > first_node[k]; k++) K[(size_t)c] = (int)k;
        th.nbase[c] = tab.nbase[c];
    }
    th.nbase[n_contigs] = (uint32_t)n_nodes;
    uint64_t tail_buckets = 0;
    for (int j = 0; j < D; j++) {
        uint64_t off = 0;
        for (int c = 0; c < n_contigs; c++) {
            const int e = K[(size_t)c] - j;
            th.B[j][c] = e >= 0 ? (uint32_t)buckets[(size_t)e] : 1u;
            th.F[j][c] = e >= 0 ? (uint32_t)first_node[(size_t)e] : 0u;
            th.boff[j][c] = (uint32_t)off;
            off += th.B[j][c];
        }
        if (off >= 0xffffffe0ull) { D = 0; th.D = 0; break; }
        for (int c = n_contigs; c <= (int)SO_MAX_CONTIGS; c++) th.boff[j][c] = (uint32_t)off;
        tail_buckets = std::max(tail_buckets, off);
    }
    st->D = D;
    auto in_chain = [&](int c, size_t k) { return N[(size_t)c] > first_node[k] && (int)k <= K[(size_t)c] - D; };
    uint64_t scratch = tail_buckets * 4;
    for (size_t k = 0; k < first_node.size(); k++) {
        uint64_t A = 0;
        for (int c = 0; c < n_contigs; c++) A += in_chain(c, k);
        scratch = std::max(scratch, A * buckets[k] * 4);
    }
    const size_t bm_words = st->bm_words = (size_t)((tail_buckets + 31) / 32 + 8);
    const uint64_t n_sort = std::max(n_nodes, n_supp_self);
    if ((rc = arena_reserve(ctx, ctx->work, align_up(scratch + 16, 256) + sortws_bytes(n_sort) + align_up(n_nodes * sizeof(csv_split_survivor), 256) +
                                                (size_t)D * (align_up(bm_words * 4, 256) + align_up(n_nodes * 4, 256)) + align_up(n_nodes, 256) + align_up(n_nodes * 4, 256) +
                                                align_up(st_filter_bytes(), 256) + 8192))) return rc;
    uint32_t *minT = st->minT = (uint32_t *)arena_alloc(ctx->work, scratch + 16);
    SortWs &w = st->w;
    if (!minT || !sortws_carve(ctx->work, n_sort, w)) { ctx->err = "arena exhausted (split order epochs)"; return CSV_ENOMEM; }
    st->d_out = (csv_split_survivor *)arena_alloc(ctx->work, n_nodes * sizeof(csv_split_survivor));
    st->d_count = (unsigned long long *)arena_alloc(ctx->work, 256);      // [0] survivors; 32-bit set sizes from byte 64 on
    if (!st->d_out || !st->d_count) { ctx->err = "arena exhausted (split order survivors)"; return CSV_ENOMEM; }
    CSV_HIP(ctx, hipMemsetAsync(st->d_count, 0, 256, s));
    if (D > 0) {
        for (int j = 0; j < D; j++) {
            st->bitmap[j] = (uint32_t *)arena_alloc(ctx->work, bm_words * 4);
            st->set[j + 1] = (uint32_t *)arena_alloc(ctx->work, n_nodes * 4);
            if (!st->bitmap[j] || !st->set[j + 1]) { ctx->err = "arena exhausted (split order tail)"; return CSV_ENOMEM; }
        }
        st->is_surv = (uint8_t *)arena_alloc(ctx->work, n_nodes);
        st->prevrank = (uint32_t *)arena_alloc(ctx->work, n_nodes * 4);
        st->filter = (uint32_t *)arena_alloc(ctx->work, st_filter_bytes());
        if (!st->is_surv || !st->prevrank || !st->filter) { ctx->err = "arena exhausted (split order tail)"; return CSV_ENOMEM; }
    }

    // the first epochs (nodes and buckets in LDS) in one launch, one workgroup per contig
    size_t n_small = 0;
    {
        SplitSmallHost sm;
        while (n_small < first_node.size() && n_small < SO_SMALL_EPOCHS && buckets[n_small] <= SO_SMALL_B) n_small++;
        const char *e = getenv("CSV_SPLIT_SMALL");
        if (e && *e && atoi(e) == 0) n_small = 0;                               // (A/B and tests: every epoch through the chain's sorts)
        sm.A = (uint32_t)n_contigs; sm.n_epochs = (uint32_t)n_small;
        bool any = false;
        for (int c = 0; c <= n_contigs; c++) sm.nbase[c] = th.nbase[c];
        for (int c = 0; c < n_contigs; c++) {
            int kl = -1;
            for (size_t k = 0; k < n_small && in_chain(c, k); k++) kl = (int)k;
            sm.k_last[c] = kl; any |= kl >= 0;
        }
        for (size_t k = 0; k <= n_small && k < first_node.size(); k++) sm.first[k] = (uint32_t)std::min<uint64_t>(first_node[k], 0xffffffffu);
        if (n_small >= first_node.size()) sm.first[n_small] = 0xffffffffu;
        for (size_t k = 0; k < n_small; k++) sm.B[k] = (uint32_t)buckets[k];
        if (n_small && any) launch_so_small_epochs(s, sm, node_hash, list);
    }
    for (size_t k = n_small; k < first_node.size(); k++) {
        SplitOrderTab e;
        e.A = 0;
        uint64_t M = 0, m_max = 0;
        const uint64_t next_first = k + 1 < first_node.size() ? first_node[k + 1] : ~0ull;
        for (int c = 0; c < n_contigs; c++) {
            if (!in_chain(c, k)) continue;
            const uint64_t m = std::min(N[(size_t)c], next_first);                 // nodes present at the end of this epoch
            e.work_off[e.A] = M; e.nbase[e.A] = tab.nbase[c]; e.m_old[e.A] = (uint32_t)first_node[k];
            e.A++; M += m; m_max = std::max(m_max, m);
        }
        if (e.A == 0) break;
        e.work_off[e.A] = M;
        if (m_max <= 1) continue;                                                   // a single node: nothing to order
        const uint32_t B = (uint32_t)buckets[k];
        const int wbits = std::max(1, bits_of(m_max - 1));
        for (uint32_t a = 0; a < e.A; a++) e.rev_off[a] = M - e.work_off[a + 1];
        const int key_bits = std::max(1, bits_of(M - 1));
        CSV_HIP(ctx, hipMemsetAsync(minT, 0xff, (size_t)e.A * B * 4, s));
        launch_so_mint(s, e, M, B, node_hash, list, minT);
        launch_so_keys(s, e, M, B, wbits, node_hash, list, minT, w.k0, w.v0);
        const int io = launch_radix_sort_u64(s, w.k0, w.v0, w.k1, w.v1, M, key_bits, w.tmp);
        launch_so_setlist(s, e, M, io ? w.v1 : w.v0, list);
    }
    if (D > 0) launch_st_inverse(s, th, (uint32_t)n_nodes, D - 1, list, st->prevrank);
    if (self) {
        // everything else too: the hashes sorted (64-bit keys, the values are not used), the survivors-only levels with the set sizes read on
        // the device, the survivors copied to the page-locked block — _finish only waits
        const int io = launch_radix_sort_u64(s, st->d_supp, w.v0, w.k1, w.v1, n_supp_self, 64, w.tmp);
        if (io != 0) { ctx->err = "split_order: unexpected sort parity"; return CSV_EHIP; }
        if ((rc = split_order_tail(ctx, st.get(), st->d_supp, n_supp_self, true))) return rc;
        st->self_bound = std::min<uint64_t>(n_supp_self, n_nodes);
        if ((rc = ensure_pinned(ctx, 64 + st->self_bound * sizeof(csv_split_survivor) + 64))) return rc;
        CSV_HIP(ctx, hipMemcpyAsync(ctx->pinned, st->d_count, 8, hipMemcpyDeviceToHost, s));
        CSV_HIP(ctx, hipMemcpyAsync((char *)ctx->pinned + 64, st->d_out, st->self_bound * sizeof(csv_split_survivor), hipMemcpyDeviceToHost, s));
    }
    if (hipGetLastError() != hipSuccess) { ctx->err = "split_order: launch failed"; return CSV_EHIP; }
    ctx->split_state = st.release();
    return CSV_OK;
}

// the survivors in their final order: the last D epochs for them and the nodes their order depends on (or, D = 0, the chain's final
// positions). devn: the set sizes stay on the device (everything is queued, nothing waited for).
static int split_order_tail(csv_ctx *ctx, csv_split_state *st, const uint64_t *d_supp, uint64_t n_supp, bool devn)
{
    hipStream_t s = ctx->stream;
    const int D = st->D;
    const int n_contigs = st->n_contigs;
    SplitTailHost &th = st->th;
    SortWs &w = st->w;
    const uint64_t n_nodes = st->n_nodes, cap = n_nodes;
    if (D == 0) {
        // ---- survivors: nodes whose name hash is a supplementary record's; their final position orders them ----
        launch_so_survivors(s, st->tab, n_nodes, st->node_hash, st->node_rec, st->list, d_supp, n_supp, st->d_out, cap, st->d_count);
        return CSV_OK;
    }
    // ---- top-down: who takes part in the last D epochs (hashes only) ----
    unsigned int *d_setn = (unsigned int *)((char *)st->d_count + 64);
    uint32_t set_n[SO_TAIL_MAX + 1] = {0, 0, 0, 0};
    for (int j = 0; j < D; j++) CSV_HIP(ctx, hipMemsetAsync(st->bitmap[j], 0, st->bm_words * 4, s));
    CSV_HIP(ctx, hipMemsetAsync(st->filter, 0, st_filter_bytes(), s));
    launch_st_survivors(s, th, (uint32_t)n_nodes, st->node_hash, d_supp, n_supp, st->filter, st->is_surv, st->bitmap[0]);
    for (int j = 1; j <= D; j++)
        launch_st_member(s, th, (uint32_t)n_nodes, j, st->node_hash, st->bitmap[j - 1], j < D ? st->bitmap[j] : nullptr, st->set[j], d_setn + j);
    if (!devn) {
        CSV_HIP(ctx, hipMemcpyAsync(ctx->pinned, d_setn, 16, hipMemcpyDeviceToHost, s));
        CSV_HIP(ctx, wait_stream(s));
        for (int j = 1; j <= D; j++) set_n[j] = ((const uint32_t *)ctx->pinned)[j];
        // a t value is a list position or an insertion index (below the largest contig's node count) or a rank in a level's order (below the set's size)
        uint64_t t_max = st->n_max;
        for (int j = 1; j <= D; j++) t_max = std::max<uint64_t>(t_max, set_n[j]);
        th.wv = std::max(1, bits_of(t_max));
    } else {
        for (int j = 1; j <= D; j++) set_n[j] = (uint32_t)n_nodes;           // (bounds: the kernels read the sizes)
        th.wv = std::max(1, bits_of(n_nodes));
    }
    // ---- bottom-up: order S_D with the chain's positions, then each smaller set with the ranks of the order before ----
    const int key_bits = th.wa + 2 * (th.wv + 1);
    for (int j = D - 1; j >= 0; j--) {
        const uint32_t n = set_n[j + 1];
        const uint32_t *n_dev = devn ? d_setn + (j + 1) : nullptr;
        if (n == 0) continue;
        CSV_HIP(ctx, hipMemsetAsync(st->minT, 0xff, (size_t)th.boff[j][n_contigs] * 4, s));
        launch_st_mint(s, th, j, st->set[j + 1], n, n_dev, st->node_hash, st->prevrank, st->minT);
        launch_st_keys(s, th, j, st->set[j + 1], n, n_dev, st->node_hash, st->prevrank, st->minT, w.k0, w.v0);
        const int io = devn ? launch_radix_sort_u64_devn(s, w.k0, w.v0, w.k1, w.v1, n, n_dev, key_bits, w.tmp)
                            : launch_radix_sort_u64(s, w.k0, w.v0, w.k1, w.v1, n, key_bits, w.tmp);
        if (io < 0) { ctx->err = "split_order: set too large for the queued sort"; return CSV_EINVAL; }
        const uint32_t *sorted = (devn || n > 1) ? (io ? w.v1 : w.v0) : w.v0;
        if (j > 0) launch_st_rank(s, sorted, n, n_dev, st->prevrank);
        else launch_st_emit(s, th, sorted, n, n_dev, st->is_surv, st->node_rec, st->d_out, cap, st->d_count);
    }
    return CSV_OK;
}

static int split_order_finish(csv_ctx *ctx, const uint64_t *supp_hash, uint64_t n_supp, uint32_t *out_rec, uint64_t capacity, uint64_t *out_off)
{
    if (!ctx) return CSV_EINVAL;
    csv_split_state *st = ctx->split_state;
    if (!st) { ctx->err = "split_order_finish without split_order_begin"; return CSV_EINVAL; }
    const int n_contigs = st->n_contigs;
    if (!out_off || (!st->self && n_supp && !supp_hash) || (capacity && !out_rec)) { ctx->err = "split_order: null array"; return CSV_EINVAL; }
    if (!st->self)
        for (uint64_t i = 1; i < n_supp; i++) if (supp_hash[i] <= supp_hash[i - 1]) { ctx->err = "split_order: supp_hash must be sorted and distinct"; return CSV_EINVAL; }
    for (int c = 0; c <= n_contigs; c++) out_off[c] = 0;
    (void)hipSetDevice(ctx->device);
    hipStream_t s = ctx->stream;
    if (!st->finished && !st->self) {
        if (n_supp == 0) {               // nothing survives
            CSV_HIP(ctx, wait_stream(s));
            st->finished = true;
        }
    }
    if (!st->finished) {
        TimerScope ts(ctx, CSV_K_SPLIT_ORDER);
        std::vector<csv_split_survivor> &surv = st->surv;
        int prc;
        if (st->self) {
            // everything was queued by _begin: the count and the survivors are in (or on their way to) the page-locked block
            CSV_HIP(ctx, wait_stream(s));
            const uint64_t n_surv = *(const unsigned long long *)ctx->pinned;
            if (n_surv > st->self_bound) { ctx->err = "split_order: more survivors than supplementary records"; return CSV_EHIP; }
            surv.resize(n_surv);
            if (n_surv) memcpy(surv.data(), (const char *)ctx->pinned + 64, n_surv * sizeof(csv_split_survivor));
        } else {
            const uint64_t n_nodes = st->n_nodes;
            // (room for one hash per record of these contigs was set aside by _begin; a run's other contigs can add more)
            struct TmpBuf { void *p = nullptr; ~TmpBuf() { if (p) (void)hipFree(p); } } big_supp;
            uint64_t *d_supp = st->d_supp;
            if (n_supp > st->total_reads) {
                if (hipMalloc(&big_supp.p, n_supp * 8) != hipSuccess) { (void)hipGetLastError(); big_supp.p = nullptr; ctx->err = "hipMalloc failed (supplementary hashes)"; return CSV_ENOMEM; }
                d_supp = (uint64_t *)big_supp.p;
            }
            // (through the context's page-locked block: a pageable copy is staged by the runtime under a lock the lanes' launches also take)
            prc = ensure_pinned(ctx, std::max<size_t>(n_supp * 8, 4096) + 64);
            if (prc) return prc;
            memcpy(ctx->pinned, supp_hash, n_supp * 8);
            CSV_HIP(ctx, hipMemcpyAsync(d_supp, ctx->pinned, n_supp * 8, hipMemcpyHostToDevice, s));
            CSV_HIP(ctx, wait_stream(s));                    // (the block is reused for the set sizes below)
            if ((prc = split_order_tail(ctx, st, d_supp, n_supp, false))) return prc;
            CSV_HIP(ctx, hipMemcpyAsync(ctx->pinned, st->d_count, 8, hipMemcpyDeviceToHost, s));
            CSV_HIP(ctx, wait_stream(s));
            const uint64_t n_surv = *(const unsigned long long *)ctx->pinned;
            if (n_surv > n_nodes) { ctx->err = "split_order: survivor count out of range"; return CSV_EHIP; }
            surv.resize(n_surv);
            if (n_surv) {
                if ((prc = ensure_pinned(ctx, n_surv * sizeof(csv_split_survivor) + 64))) return prc;
                CSV_HIP(ctx, hipMemcpyAsync(ctx->pinned, st->d_out, n_surv * sizeof(csv_split_survivor), hipMemcpyDeviceToHost, s));
                CSV_HIP(ctx, wait_stream(s));
                memcpy(surv.data(), ctx->pinned, n_surv * sizeof(csv_split_survivor));
            }
        }
        std::sort(surv.begin(), surv.end(), [](const csv_split_survivor &a, const csv_split_survivor &b) { return a.contig != b.contig ? a.contig < b.contig : a.pos < b.pos; });
        for (const csv_split_survivor &v : surv) st->off[v.contig + 1]++;
        for (int c = 0; c < n_contigs; c++) st->off[(size_t)c + 1] += st->off[(size_t)c];
        st->finished = true;
    }
    for (int c = 0; c <= n_contigs; c++) out_off[c] = st->off[(size_t)c];
    if (st->surv.size() > capacity) { ctx->err = "split_order: output capacity too small"; return CSV_ECAPACITY; }      // (the state stays: _finish again with more room)
    for (size_t i = 0; i < st->surv.size(); i++) out_rec[i] = st->surv[i].rec;
    delete st; ctx->split_state = nullptr;
    return CSV_OK;
}

extern "C" {

int csvgpu_split_order_begin(csv_ctx *ctx, int n_contigs, csv_shard *const *shards, uint8_t min_mapq)
{
    return split_order_begin(ctx, n_contigs, shards, min_mapq, -1);
}

int csvgpu_split_order_begin_self(csv_ctx *ctx, int n_contigs, csv_shard *const *shards, uint8_t min_mapq)
{
    return split_order_begin(ctx, n_contigs, shards, min_mapq, -1, true);
}

int csvgpu_split_order_finish(csv_ctx *ctx, const uint64_t *supp_hash, uint64_t n_supp, uint32_t *out_rec, uint64_t capacity, uint64_t *out_off)
{
    return split_order_finish(ctx, supp_hash, n_supp, out_rec, capacity, out_off);
}

int csvgpu_split_order(csv_ctx *ctx, int n_contigs, csv_shard *const *shards, uint8_t min_mapq, const uint64_t *supp_hash, uint64_t n_supp,
                       uint32_t *out_rec, uint64_t capacity, uint64_t *out_off)
{
    if (!ctx) return CSV_EINVAL;
    if (!out_off || (n_contigs > 0 && !shards) || (n_supp && !supp_hash) || (capacity && !out_rec)) { ctx->err = "split_order: null array"; return CSV_EINVAL; }
    if (n_contigs >= 0 && (uint32_t)n_contigs <= SO_MAX_CONTIGS) for (int c = 0; c <= n_contigs; c++) out_off[c] = 0;
    for (uint64_t i = 1; i < n_supp; i++) if (supp_hash[i] <= supp_hash[i - 1]) { ctx->err = "split_order: supp_hash must be sorted and distinct"; return CSV_EINVAL; }
    int rc = split_order_begin(ctx, n_contigs, shards, min_mapq, (int64_t)n_supp);
    if (rc) return rc;
    rc = split_order_finish(ctx, supp_hash, n_supp, out_rec, capacity, out_off);
    if (rc) { delete ctx->split_state; ctx->split_state = nullptr; }         // (one call: nothing is kept for a retry, the caller repeats it with the size from out_off)
    return rc;
}

int csvgpu_window_log2_resident(csv_ctx *ctx, csv_shard *sh, const uint32_t *region_start, const uint32_t *region_end,
                                const int32_t *sample_size, const uint64_t *win_off, uint64_t n_regions, double mean_cov,
                                double *log2_cov, uint32_t *win_start, uint32_t *win_end)
{
    if (!ctx || !sh) return CSV_EINVAL;
    if (n_regions == 0) return CSV_OK;
    if (!region_start || !region_end || !sample_size || !win_off) { ctx->err = "window_log2: null array"; return CSV_EINVAL; }
    for (uint64_t r = 0; r < n_regions; r++) {
        if (sample_size[r] <= 0 || win_off[r + 1] - win_off[r] != (uint64_t)sample_size[r] || region_start[r] > region_end[r]) {
            ctx->err = "window_log2: bad region table"; return CSV_EINVAL;
        }
    }
    const uint64_t nw = win_off[n_regions];
    if (nw == 0) return CSV_OK;
    if (!log2_cov || !win_start || !win_end) { ctx->err = "window_log2: null output"; return CSV_EINVAL; }
    (void)hipSetDevice(ctx->device);
    int rc = arena_reserve(ctx, ctx->arena, 3 * align_up(n_regions * 4, 256) + align_up((n_regions + 1) * 8, 256) + align_up(nw * 8, 256) +
                                                2 * align_up(nw * 4, 256) + 4096);
    if (rc) return rc;
    Arena &a = ctx->arena;
    uint32_t *drs = (uint32_t *)arena_alloc(a, n_regions * 4), *dre = (uint32_t *)arena_alloc(a, n_regions * 4);
    int32_t *dss = (int32_t *)arena_alloc(a, n_regions * 4);
    uint64_t *dwo = (uint64_t *)arena_alloc(a, (n_regions + 1) * 8);
    double *dl2 = (double *)arena_alloc(a, nw * 8);
    uint32_t *dws = (uint32_t *)arena_alloc(a, nw * 4), *dwe = (uint32_t *)arena_alloc(a, nw * 4);
    if (!drs || !dre || !dss || !dwo || !dl2 || !dws || !dwe) { ctx->err = "arena exhausted"; return CSV_ENOMEM; }
    hipStream_t s = ctx->stream;
    CSV_HIP(ctx, hipMemcpyAsync(drs, region_start, n_regions * 4, hipMemcpyHostToDevice, s));
    CSV_HIP(ctx, hipMemcpyAsync(dre, region_end, n_regions * 4, hipMemcpyHostToDevice, s));
    CSV_HIP(ctx, hipMemcpyAsync(dss, sample_size, n_regions * 4, hipMemcpyHostToDevice, s));
    CSV_HIP(ctx, hipMemcpyAsync(dwo, win_off, (n_regions + 1) * 8, hipMemcpyHostToDevice, s));
    if ((rc = csvgpu_window_log2_dev(ctx, sh->depth, sh->depth_len, drs, dre, dss, dwo, n_regions, nw, mean_cov, dl2, dws, dwe))) return rc;
    CSV_HIP(ctx, hipMemcpyAsync(log2_cov, dl2, nw * 8, hipMemcpyDeviceToHost, s));
    CSV_HIP(ctx, hipMemcpyAsync(win_start, dws, nw * 4, hipMemcpyDeviceToHost, s));
    CSV_HIP(ctx, hipMemcpyAsync(win_end, dwe, nw * 4, hipMemcpyDeviceToHost, s));
    CSV_HIP(ctx, wait_stream(s));
    return CSV_OK;
}

int csvgpu_depth_lookup_resident(csv_ctx *ctx, csv_shard *sh, const uint32_t *pos, uint64_t n, int32_t *depth_out)
{
    if (!ctx || !sh) return CSV_EINVAL;
    if (n == 0) return CSV_OK;
    if (!pos || !depth_out) { ctx->err = "depth_lookup: null array"; return CSV_EINVAL; }
    if (!sh->depth) { ctx->err = "depth_lookup: shard has no depth map (run csvgpu_chr_pipeline_dev first)"; return CSV_EINVAL; }
    (void)hipSetDevice(ctx->device);
    int rc = arena_reserve(ctx, ctx->arena, 2 * align_up(n * 4, 256) + 4096);
    if (rc) return rc;
    uint32_t *dpos = (uint32_t *)arena_alloc(ctx->arena, n * 4);
    int32_t *dout = (int32_t *)arena_alloc(ctx->arena, n * 4);
    if (!dpos || !dout) { ctx->err = "arena exhausted"; return CSV_ENOMEM; }
    hipStream_t s = ctx->stream;
    CSV_HIP(ctx, hipMemcpyAsync(dpos, pos, n * 4, hipMemcpyHostToDevice, s));
    csv::launch_depth_lookup(s, sh->depth, sh->depth_len, dpos, n, dout);
    CSV_HIP(ctx, hipMemcpyAsync(depth_out, dout, n * 4, hipMemcpyDeviceToHost, s));
    CSV_HIP(ctx, wait_stream(s));
    return CSV_OK;
}

int csvgpu_chr_fetch(csv_ctx *ctx, csv_shard *sh, const csv_chr_result *res, csv_sig *host_sig, int32_t *host_labels)
{
    if (!ctx || !sh || !res) return CSV_EINVAL;
    if (res->n_sig == 0) return CSV_OK;
    if (!host_sig || !host_labels) { ctx->err = "chr_fetch: null output"; return CSV_EINVAL; }
    (void)hipSetDevice(ctx->device);
    CSV_HIP(ctx, hipMemcpyAsync(host_sig, res->sig_del, res->n_sig * sizeof(csv_sig), hipMemcpyDeviceToHost, ctx->stream));
    CSV_HIP(ctx, hipMemcpyAsync(host_labels, res->label_del, res->n_sig * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    CSV_HIP(ctx, wait_stream(ctx->stream));
    return CSV_OK;
}

int csvgpu_download(csv_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes)
{
    if (!ctx || (bytes && (!host_dst || !dev_src))) return CSV_EINVAL;
    if (!bytes) return CSV_OK;
    (void)hipSetDevice(ctx->device);
    CSV_HIP(ctx, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    CSV_HIP(ctx, wait_stream(ctx->stream));
    return CSV_OK;
}

csv_gate *csvgpu_gate_create(void) { return new (std::nothrow) csv_gate(); }

// Creates the gate's stream now instead of at the first job. The runtime deals its hardware queues (four by default) to streams in
// creation order, and a stream that waits for an event holds up every other stream of its hardware queue: a gate opened BEFORE the lanes'
// contexts are created shares its queue with none of the first lanes' streams.
int csvgpu_gate_open(csv_gate *gate, int device_ordinal)
{
    if (!gate) return CSV_EINVAL;
    if (gate->stream) return gate->device == device_ordinal ? CSV_OK : CSV_EINVAL;
    if (hipSetDevice(device_ordinal) != hipSuccess) { (void)hipGetLastError(); return CSV_ENODEV; }
    if (hipStreamCreateWithFlags(&gate->stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); gate->stream = nullptr; return CSV_EHIP; }
    gate->device = device_ordinal;
    return CSV_OK;
}

void csvgpu_gate_destroy(csv_gate *gate)
{
    if (!gate) return;
    if (gate->stream) { (void)hipSetDevice(gate->device); (void)hipStreamSynchronize(gate->stream); (void)hipStreamDestroy(gate->stream); }
    delete gate;
}

int csvgpu_set_gate(csv_ctx *ctx, csv_gate *gate)
{
    if (!ctx) return CSV_EINVAL;
    ctx->gate = gate;
    return CSV_OK;
}

void *csvgpu_host_alloc(csv_ctx *ctx, size_t bytes)
{
    if (!ctx || !bytes) return nullptr;
    for (size_t i = 0; i < ctx->host_pool.size(); i++) {
        if (ctx->host_pool[i].second >= bytes && ctx->host_pool[i].second <= 2 * bytes + 4096) {
            ctx->host_live.push_back(ctx->host_pool[i]);
            ctx->host_pool.erase(ctx->host_pool.begin() + (std::ptrdiff_t)i);
            return ctx->host_live.back().first;
        }
    }
    (void)hipSetDevice(ctx->device);
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx->err = "hipHostMalloc failed"; return nullptr; }
    ctx->host_live.emplace_back(p, bytes);
    return p;
}

void csvgpu_host_free(csv_ctx *ctx, void *p)
{
    if (!ctx || !p) return;
    for (size_t i = 0; i < ctx->host_live.size(); i++) {
        if (ctx->host_live[i].first != p) continue;
        ctx->host_pool.push_back(ctx->host_live[i]);
        ctx->host_live.erase(ctx->host_live.begin() + (std::ptrdiff_t)i);
        while (ctx->host_pool.size() > 8) {                // bounded: drop the oldest
            (void)hipHostFree(ctx->host_pool.front().first);
            ctx->host_pool.erase(ctx->host_pool.begin());
        }
        return;
    }
}

// ---- one chromosome as a job in three steps, so that the caller can queue the scan + depth pass of the next chromosome before it
// waits for this one's results (the device then never idles across the host's turn-around) ----
struct csv_job {
    csv_shard *sh = nullptr;
    uint32_t min_oplen = 50; uint8_t min_mapq = 20; double min_pts_pct = 0.1;
    hipEvent_t ev_zero = nullptr, ev_scan = nullptr, ev_depth = nullptr, ev_mid = nullptr, ev_done = nullptr, t0 = nullptr;
    bool on_gate = false;                // the pair runs on a gate's stream: this context's stream meets it only in job_cluster (ev_depth)
    char *pin = nullptr;                 // 512 B page-locked: [0,256) counters behind the scan, [256,512) counters at the end
    bool depth_queued = false, clustered = false, copied = false;
    uint64_t n = 0, n_del = 0, capacity = 0;
    csv_sig *sig_sorted = nullptr;
    int32_t *labels = nullptr;
};

// CSV_MAX_JOBS rotating 512-byte page-locked slots per context; a slot belongs to its job from begin to end / abort, so a caller
// that holds more than CSV_MAX_JOBS jobs open on one context is refused instead of aliasing another job's counters.
static char *job_pin_slot(csv_ctx *ctx)
{
    constexpr size_t kSlots = CSV_MAX_JOBS, kSlot = 512;
    if (!ctx->job_pin && hipHostMalloc((void **)&ctx->job_pin, kSlots * kSlot, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    for (size_t k = 0; k < kSlots; k++) {
        const size_t i = (ctx->job_pin_next + k) % kSlots;
        if (ctx->job_pin_busy & (1u << i)) continue;
        ctx->job_pin_busy |= 1u << i;
        ctx->job_pin_next = i + 1;
        return ctx->job_pin + i * kSlot;
    }
    return nullptr;
}
static void job_pin_release(csv_ctx *ctx, char *p)
{
    if (!p || !ctx->job_pin) return;
    ctx->job_pin_busy &= ~(1u << (size_t)((p - ctx->job_pin) / 512));
}

// scan (+ counters on their way to the host + depth pass, when the shard's sortedness is known). For a coordinate-sorted shard the
// device sees scan -> depth tiles back to back: the scan leaves the bucket counts and the tiles' candidate ranges behind, the
// counters travel beside the depth pass, and everything small (offsets, scatter, ranking, clustering) is queued behind it.
// With a gate, the scan + depth pairs of all attached contexts go onto the gate's one stream — back to back in queue order, no
// hand-over between queues — while each context's own (higher-priority) stream runs its small kernels beside the other lane's pair.
static int job_queue_front(csv_ctx *ctx, csv_job *job)
{
    csv_shard *sh = job->sh;
    hipStream_t s = ctx->stream;
    ScanCounters *cnt = (ScanCounters *)sh->counters;
    int rc;
    const bool sorted = sh->unsorted == 0;
    if (!job->ev_scan) job->ev_scan = get_event(ctx);          // (handed to the timers by an earlier pass of this job)
    if (!job->ev_depth) job->ev_depth = get_event(ctx);
    if (!job->ev_scan || !job->ev_depth) { ctx->err = "job: cannot allocate events"; return CSV_ENOMEM; }
    CSV_HIP(ctx, hipMemsetAsync(cnt, 0, sorted ? sh->counters_bytes : kCntBytes, s));
    csv_gate *gate = sorted ? ctx->gate : nullptr;
    hipStream_t big = s;
    std::unique_lock<std::mutex> turn;
    if (gate) {
        turn = std::unique_lock<std::mutex>(gate->mu);
        // (stream priorities — this stream low, the contexts' own high — measured 2 % slower: a small kernel waits for a whole
        // workgroup slot of the resident big kernel either way)
        if (!gate->stream) {
            if (hipStreamCreateWithFlags(&gate->stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); gate->stream = nullptr; }
            gate->device = ctx->device;
        }
        if (gate->stream && gate->device == ctx->device) {
            big = gate->stream;
            CSV_HIP(ctx, hipEventRecord(job->ev_zero, s));                 // the pair starts behind this context's memset (and whatever it was queued behind)
            CSV_HIP(ctx, hipStreamWaitEvent(big, job->ev_zero, 0));
        }
    }
    // On the gate's stream every recorded event is a barrier packet between the big kernels of ALL lanes (~5 us each): the pair is
    // timed with the two events the job records there anyway plus one in front (scan = ev_scan - t0, depth = ev_depth - ev_scan).
    // (at level 2 only every fourth pair: the extra event in front of the scan is a barrier packet on the stream all lanes share, 2.5 % of
    // the throughput when every pair has one; the averages are over the timed pairs)
    if (job->t0) { ctx->event_pool.push_back(job->t0); job->t0 = nullptr; }          // (a re-run after the signature buffer grew)
    const bool pair_timers = big != s && ctx->timing != 0 && (ctx->timing == 1 || ctx->timing == 3 || (ctx->timer_tick++ & 3u) == 0);
    hipEvent_t t0 = nullptr;
    if (pair_timers) {
        t0 = get_event(ctx);
        if (t0) CSV_HIP(ctx, hipEventRecord(t0, big));
        launch_cigar_scan(big, ctx->n_cu, sh->d, sh->depth_len, job->min_oplen, job->min_mapq, 1, sh->sig_raw, sh->sig_cap, sh->ref_end,
                          sh->q_start, sh->q_end, sh->ckpt, cnt, scan_extras(cnt, sh->depth_len, true, sh->tile_range), sh->owned ? sh->scan_split : nullptr, sh->form, sh->cigar_pad);
    } else if (big != s) {                  // on the gate's stream, not a timed pair: no events of its own
        launch_cigar_scan(big, ctx->n_cu, sh->d, sh->depth_len, job->min_oplen, job->min_mapq, 1, sh->sig_raw, sh->sig_cap, sh->ref_end,
                          sh->q_start, sh->q_end, sh->ckpt, cnt, scan_extras(cnt, sh->depth_len, true, sh->tile_range), sh->owned ? sh->scan_split : nullptr, sh->form, sh->cigar_pad);
    } else {
        TimerScope ts(ctx, CSV_K_CIGAR_SCAN, big);
        launch_cigar_scan(big, ctx->n_cu, sh->d, sh->depth_len, job->min_oplen, job->min_mapq, 1, sh->sig_raw, sh->sig_cap, sh->ref_end,
                          sh->q_start, sh->q_end, sh->ckpt, cnt, scan_extras(cnt, sh->depth_len, true, sorted ? sh->tile_range : nullptr), sh->owned ? sh->scan_split : nullptr, sh->form, sh->cigar_pad);
    }
    job->depth_queued = false;
    if (sh->unsorted >= 0) {
        // The depth pass does not depend on the signature count, so it is queued BEFORE the host waits for the counters: the
        // device works through it while the host wakes up, sizes the ordering and clustering launches and queues them.
        job->on_gate = big != s;
        if (job->on_gate) {
            // On a gate nothing of this job touches the context's own stream until job_cluster: a caller that queues several jobs ahead must
            // not find one job's clustering kernels behind a wait for a LATER job's scan. The counters leave from the gate's stream itself,
            // between the two big kernels (256 bytes to page-locked memory), ev_scan tells the host they have landed, and job_cluster makes
            // the context's stream wait for ev_depth before min_pts. (A relay through a side stream per context was tried: three more
            // streams whose only work is to wait share the four hardware queues with everything else and stalled the caller's context.)
            CSV_HIP(ctx, hipMemcpyAsync(job->pin, cnt, sizeof(ScanCounters), hipMemcpyDeviceToHost, big));
            CSV_HIP(ctx, hipEventRecord(job->ev_scan, big));
            launch_depth_tiles(big, sh->d, nullptr, sh->ref_end, sh->ckpt, sh->depth_len, sh->depth, cnt, sh->tile_range, sh->cigar_pad, sh->depth_items, sh->form);
            CSV_HIP(ctx, hipEventRecord(job->ev_depth, big));
            turn.unlock();
            job->t0 = pair_timers ? t0 : nullptr;            // (handed to the timers with ev_scan / ev_depth when the job ends)
        } else {
            // The counters leave on a side stream beside the depth pass.
            if (!ctx->side) CSV_HIP(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
            hipStream_t cs = ctx->side;
            CSV_HIP(ctx, hipEventRecord(job->ev_scan, big));
            CSV_HIP(ctx, hipStreamWaitEvent(cs, job->ev_scan, 0));
            CSV_HIP(ctx, hipMemcpyAsync(job->pin, cnt, sizeof(ScanCounters), hipMemcpyDeviceToHost, cs));
            CSV_HIP(ctx, hipEventRecord(job->ev_mid, cs));
            ctx->work.used = 0;
            if ((rc = depth_chain(ctx, ctx->work, sh->d, sh->ref_end, sh->ckpt, sh->unsorted != 0, sh->depth_len, sh->depth, cnt,
                                  sorted ? sh->tile_range : nullptr, sh->cigar_pad, sh->depth_items, sh->form))) return rc;
            launch_min_pts(s, cnt, job->min_pts_pct);
        }
        job->depth_queued = true;
    }
    return CSV_OK;
}

static void job_free(csv_ctx *ctx, csv_job *job)
{
    if (!job) return;
    job_pin_release(ctx, job->pin);
    if (job->t0 && job->ev_scan && job->ev_depth && job->on_gate && ctx->gate && ctx->gate->stream) {
        // a timed pair on the gate's stream: scan = ev_scan - t0 (the 256-byte counters copy included), depth = ev_depth - ev_scan; the
        // timers own the three events from here (folded when the times are read)
        Timer a; a.id = CSV_K_CIGAR_SCAN; a.a = job->t0; a.b = job->ev_scan; a.s = ctx->gate->stream;
        Timer b; b.id = CSV_K_DEPTH; b.a = job->ev_scan; b.b = job->ev_depth; b.s = ctx->gate->stream; b.own_a = false;
        ctx->timers.push_back(a); ctx->timers.push_back(b);
        job->t0 = nullptr; job->ev_scan = nullptr; job->ev_depth = nullptr;
    }
    if (job->t0) ctx->event_pool.push_back(job->t0);
    if (job->ev_zero) ctx->event_pool.push_back(job->ev_zero);
    if (job->ev_scan) ctx->event_pool.push_back(job->ev_scan);
    if (job->ev_depth) ctx->event_pool.push_back(job->ev_depth);
    if (job->ev_mid) ctx->event_pool.push_back(job->ev_mid);
    if (job->ev_done) ctx->event_pool.push_back(job->ev_done);
    delete job;
}

csv_job *csvgpu_chr_job_begin(csv_ctx *ctx, csv_shard *sh, uint32_t min_oplen, uint8_t min_mapq, double min_pts_pct)
{
    if (!ctx || !sh) return nullptr;
    (void)hipSetDevice(ctx->device);
    csv_job *job = new (std::nothrow) csv_job();
    if (!job) { ctx->err = "out of host memory"; return nullptr; }
    job->sh = sh; job->min_oplen = min_oplen; job->min_mapq = min_mapq; job->min_pts_pct = min_pts_pct;
    job->ev_zero = get_event(ctx); job->ev_scan = get_event(ctx); job->ev_depth = get_event(ctx); job->ev_mid = get_event(ctx); job->ev_done = get_event(ctx);
    job->pin = job_pin_slot(ctx);
    if (!job->pin && ctx->job_pin) { ctx->err = "job: more than CSV_MAX_JOBS jobs open on this context"; job_free(ctx, job); return nullptr; }
    if (!job->pin || !job->ev_zero || !job->ev_scan || !job->ev_depth || !job->ev_mid || !job->ev_done) { ctx->err = "job: cannot allocate events / page-locked memory"; job_free(ctx, job); return nullptr; }
    if (arena_reserve(ctx, ctx->work, depth_chain_bytes(sh->d.n_reads, sh->depth_len)) || job_queue_front(ctx, job)) { job_free(ctx, job); return nullptr; }
    return job;
}

int csvgpu_chr_job_cluster(csv_ctx *ctx, csv_job *job, double eps, csv_sig *host_sig, int32_t *host_labels, uint64_t capacity)
{
    if (!ctx || !job || job->clustered) return CSV_EINVAL;
    if (!(eps >= 0.0) || !(eps < 1.0)) { ctx->err = "pipeline: eps must be in [0,1)"; return CSV_EINVAL; }
    if (capacity && (!host_sig || !host_labels)) { ctx->err = "pipeline: null output"; return CSV_EINVAL; }
    (void)hipSetDevice(ctx->device);
    csv_shard *sh = job->sh;
    hipStream_t s = ctx->stream;
    ScanCounters *cnt = (ScanCounters *)sh->counters;
    ScanCounters h;
    int rc;
    for (int attempt = 0;; attempt++) {
        if (job->depth_queued) {
            CSV_HIP(ctx, wait_event(job->on_gate ? job->ev_scan : job->ev_mid));
            memcpy(&h, job->pin, sizeof(ScanCounters));
        } else {
            if ((rc = read_counters(ctx, cnt, h))) return rc;             // first scan of wrapped arrays: wait, then decide
            sh->unsorted = h.unsorted != 0;
        }
        if (h.n_sig <= sh->sig_cap) break;
        if (attempt) { ctx->err = "pipeline: signature buffer overflow twice"; return CSV_ENOMEM; }
        CSV_HIP(ctx, wait_stream(s));                            // the queued depth pass reads what the re-run scan rewrites
        // the larger buffer first: if it cannot be had, the shard keeps its old buffer AND its old capacity (a later job on this
        // shard must never see a capacity without a buffer behind it — the scan's `g < sig_cap` guard would write through null)
        const uint64_t new_cap = h.n_sig + h.n_sig / 8 + 1024;
        csv_sig *bigger = nullptr;
        if (csv_test_fail_alloc() || hipMalloc((void **)&bigger, new_cap * sizeof(csv_sig)) != hipSuccess) {
            (void)hipGetLastError();
            ctx->err = "hipMalloc failed (signature buffer)";
            return CSV_ENOMEM;
        }
        (void)hipFree(sh->sig_raw);
        sh->sig_raw = bigger; sh->sig_cap = new_cap;
        if ((rc = job_queue_front(ctx, job))) return rc;
    }
    const uint64_t n = h.n_sig, n_del = h.n_del;
    const uint32_t max_bucket = h.max_len;

    // shard scratch: sorted signatures, SoA start/end, labels, sort + dbscan workspace (grow-only)
    const size_t need = align_up(n * sizeof(csv_sig), 256) + 3 * align_up(n * 4 + 16, 256) + sortws_bytes(n) + dbscan_tmp_bytes(n) + 4096;
    if (need > sh->scratch_cap) {
        if (sh->scratch) CSV_HIP(ctx, hipFree(sh->scratch));
        sh->scratch = nullptr; sh->scratch_cap = 0;
        CSV_HIP(ctx, hipMalloc((void **)&sh->scratch, need + need / 4));
        sh->scratch_cap = need + need / 4;
    }
    Arena sa; sa.base = sh->scratch; sa.cap = sh->scratch_cap; sa.used = 0;
    csv_sig *sig_sorted = (csv_sig *)arena_alloc(sa, n * sizeof(csv_sig));
    uint32_t *st = (uint32_t *)arena_alloc(sa, n * 4 + 16), *en = (uint32_t *)arena_alloc(sa, n * 4 + 16);
    int32_t *labels = (int32_t *)arena_alloc(sa, n * 4 + 16);
    SortWs w;
    const bool ws_ok = sortws_carve(sa, n, w);
    void *db_tmp = arena_alloc(sa, dbscan_tmp_bytes(n));
    if (!sig_sorted || !st || !en || !labels || !ws_ok || !db_tmp) { ctx->err = "shard scratch exhausted"; return CSV_ENOMEM; }

    // depth map + mean coverage + min_pts (device scalar), unless already queued behind the scan
    if (!job->depth_queued) {
        ctx->work.used = 0;
        if ((rc = depth_chain(ctx, ctx->work, sh->d, sh->ref_end, sh->ckpt, sh->unsorted != 0, sh->depth_len, sh->depth, cnt, nullptr, sh->cigar_pad, sh->depth_items, sh->form))) return rc;
        launch_min_pts(s, cnt, job->min_pts_pct);
    }

    // ordering: DEL calls then INS calls, each in chr_sv_calls order
    order_signatures(ctx, sh->sig_raw, n, sh->depth_len, h.max_start, max_bucket, cnt, true, w, sig_sorted, st, en);

    // min_pts (and with it the clustering) reads what the depth pass leaves; the ordering above did not have to wait for it
    if (job->depth_queued && job->on_gate) {
        CSV_HIP(ctx, hipStreamWaitEvent(s, job->ev_depth, 0));
        launch_min_pts(s, cnt, job->min_pts_pct);
    }
    // per-type interval DBSCAN (mergeSVs walks DEL ... INS, sv_object.cpp:62-68)
    {
        TimerScope ts(ctx, CSV_K_DBSCAN);
        // DEL calls [0, n_del) and INS calls [n_del, n) are clustered side by side in the same five launches
        if (n) launch_dbscan_iv_sorted(s, st, en, nullptr, n, n_del, eps, 0, &cnt->min_pts, labels, db_tmp);
    }
    job->copied = capacity && n && n <= capacity;
    if (job->copied) {                                                   // results ride behind the last kernel, one wait for everything
        CSV_HIP(ctx, hipMemcpyAsync(host_sig, sig_sorted, n * sizeof(csv_sig), hipMemcpyDeviceToHost, s));
        CSV_HIP(ctx, hipMemcpyAsync(host_labels, labels, n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    }
    CSV_HIP(ctx, hipMemcpyAsync(job->pin + 256, cnt, sizeof(ScanCounters), hipMemcpyDeviceToHost, s));
    CSV_HIP(ctx, hipEventRecord(job->ev_done, s));
    job->n = n; job->n_del = n_del; job->capacity = capacity; job->sig_sorted = sig_sorted; job->labels = labels;
    job->clustered = true;
    return CSV_OK;
}

int csvgpu_chr_job_end(csv_ctx *ctx, csv_job *job, csv_chr_result *res)
{
    if (!ctx || !job) return CSV_EINVAL;
    (void)hipSetDevice(ctx->device);
    int rc = CSV_OK;
    if (!job->clustered) { ctx->err = "job_end before job_cluster"; rc = CSV_EINVAL; }
    else if (wait_event(job->ev_done) != hipSuccess) { (void)hipGetLastError(); ctx->err = "job: device error"; rc = CSV_EHIP; }
    else if (res) {
        ScanCounters h;
        memcpy(&h, job->pin + 256, sizeof(ScanCounters));
        csv_shard *sh = job->sh;
        res->n_sig = job->n; res->n_del = job->n_del; res->n_ins = job->n - job->n_del;
        res->depth_sum = h.depth_sum; res->depth_nonzero = h.depth_nonzero; res->min_pts = h.min_pts; res->mean_cov = h.mean_cov;
        res->sig_del = job->sig_sorted; res->sig_ins = job->sig_sorted + job->n_del;
        res->label_del = job->labels; res->label_ins = job->labels + job->n_del;
        res->depth = sh->depth; res->ref_end = sh->ref_end; res->q_start = sh->q_start; res->q_end = sh->q_end;
        if (job->capacity && job->n > job->capacity) { ctx->err = "pipeline_fetch: host buffers too small"; rc = CSV_ECAPACITY; }
    }
    job_free(ctx, job);
    return rc;
}

int csvgpu_chr_job_abort(csv_ctx *ctx, csv_job *job)
{
    if (!ctx || !job) return CSV_EINVAL;
    (void)hipSetDevice(ctx->device);
    // whatever the job queued reads the shard's buffers: let it drain before the caller reuses or frees them
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->gate && ctx->gate->stream) (void)hipStreamSynchronize(ctx->gate->stream);
    if (ctx->side) (void)hipStreamSynchronize(ctx->side);
    job_free(ctx, job);                                  // ctx->err keeps the failure that led here
    return CSV_OK;
}

#ifdef CSV_TEST_HOOKS
// Test hook (error-path tests): the next n device allocations guarded by csv_test_fail_alloc() fail.
void csvgpu_test_fail_next_alloc(int n) { csv::g_fail_alloc.store(n); }
#endif

static int chr_pipeline(csv_ctx *ctx, csv_shard *sh, uint32_t min_oplen, uint8_t min_mapq, double eps, double min_pts_pct, csv_chr_result *res,
                        csv_sig *host_sig, int32_t *host_labels, uint64_t capacity)
{
    if (!ctx || !sh || !res) return CSV_EINVAL;
    if (!(eps >= 0.0) || !(eps < 1.0)) { ctx->err = "pipeline: eps must be in [0,1)"; return CSV_EINVAL; }
    csv_job *job = csvgpu_chr_job_begin(ctx, sh, min_oplen, min_mapq, min_pts_pct);
    if (!job) return ctx->err.find("hipMalloc") != std::string::npos ? CSV_ENOMEM : CSV_EHIP;
    const int rc = csvgpu_chr_job_cluster(ctx, job, eps, host_sig, host_labels, capacity);
    if (rc) { csvgpu_chr_job_abort(ctx, job); return rc; }
    return csvgpu_chr_job_end(ctx, job, res);
}

int csvgpu_chr_pipeline_dev(csv_ctx *ctx, csv_shard *sh, uint32_t min_oplen, uint8_t min_mapq, double eps, double min_pts_pct,
                            csv_chr_result *res)
{
    return chr_pipeline(ctx, sh, min_oplen, min_mapq, eps, min_pts_pct, res, nullptr, nullptr, 0);
}

int csvgpu_chr_pipeline_fetch(csv_ctx *ctx, csv_shard *sh, uint32_t min_oplen, uint8_t min_mapq, double eps, double min_pts_pct,
                              csv_chr_result *res, csv_sig *host_sig, int32_t *host_labels, uint64_t capacity)
{
    if (capacity && (!host_sig || !host_labels)) { if (ctx) ctx->err = "pipeline_fetch: null output"; return CSV_EINVAL; }
    return chr_pipeline(ctx, sh, min_oplen, min_mapq, eps, min_pts_pct, res, host_sig, host_labels, capacity);
}

}  // extern "C"
