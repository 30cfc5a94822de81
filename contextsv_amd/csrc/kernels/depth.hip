// depth.hip — kernel #2: tile-owner depth map (gfx950).
//
// Replaces the per-chromosome body of CNVCaller::calculateMeanChromosomeCoverage
// (cnv_caller.cpp:488-543): depth[p]++ for every base of every M/=/X op, then sum and #non-zero.
//
// The reference does ~30 x chromosome-length scalar increments. A read-owner GPU version would
// need two scattered global atomics per M-run (~1.5e8 per chr22 at 30x, far below the HBM rate), so
// the ownership is turned around: one 1024-thread workgroup owns one 16 Ki-position tile of the
// chromosome and keeps the tile's DIFFERENCE array in LDS (64 KiB + a 12 KiB work list -> 2
// workgroups = 32 waves per CU). Per tile:
//   1. candidate reads = one precomputed range: for coordinate-sorted shards the CIGAR scan itself records, per tile, the
//      first and last read that reach it (ScanExtras::tile_range), so nothing runs between the scan and this kernel;
//      otherwise depth_ranges_kernel derives the ranges from a prefix maximum of the read ends;
//   2. all threads together build an LDS work list: thread t loads candidate t's metadata, drops reads
//      that end left of the tile or fail the depth filter, and binary-searches the read's checkpoints
//      (scan.hip records the reference offset of a read at every 64-word CIGAR boundary) for the last
//      boundary left of the tile;
//   3. waves pull items from an LDS counter and walk 1 KiB CIGAR chunks from that boundary (16-byte loads,
//      one DPP wave scan per chunk, four chunks in flight), applying +1/-1 with LDS atomics, until the
//      tile's right edge;
//   4. the tile is scanned in LDS and depth is written once, coalesced, 16 B per lane, while sum and
//      non-zero count are reduced.
// No global atomics on the depth array, no memset, no separate scan pass. HBM traffic = CIGAR stream x
// (1 + ~1 partial chunk per (tile, read) pair) + 4 B/base written.
#include "../common.hpp"
#include "../devutil.hpp"

namespace csv {

constexpr int DEPTH_THREADS = 1024;
constexpr int DEPTH_WAVES = DEPTH_THREADS / WAVE;          // 8
constexpr int DEPTH_PER_WAVE = DEPTH_TILE / DEPTH_WAVES;   // entries scanned per wave
constexpr int DEPTH_ROUNDS = DEPTH_PER_WAVE / (4 * WAVE);  // rounds of 256 entries
constexpr int WL_CAP = 512;                                // work-list items staged per batch (12 KiB of LDS)
constexpr int WL_SPEC = 256;                               // list entries a tile requests before it knows how many it has
constexpr int DEPTH_PF = 5;                                // a walk keeps DEPTH_PF + 1 chunks in flight ahead of the one being worked on (1 -> 4: -4 %)

// ------------------------------------------------------------------------------- prefix max
constexpr int PM_THREADS = 256;
constexpr int PM_ITEMS = 8;
constexpr int PM_TILE = PM_THREADS * PM_ITEMS;             // 2048

__global__ __launch_bounds__(PM_THREADS) void pmax_reduce_kernel(const int32_t *__restrict__ in, uint64_t n, int32_t *__restrict__ blk)
{
    __shared__ int32_t wm[PM_THREADS / WAVE];
    const uint64_t b0 = (uint64_t)blockIdx.x * PM_TILE;
    int32_t m = INT32_MIN;
    for (int k = 0; k < PM_ITEMS; k++) {
        uint64_t i = b0 + (uint64_t)k * PM_THREADS + threadIdx.x;
        if (i < n) m = max(m, in[i]);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
    if (lane_id() == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < PM_THREADS / WAVE; w++) m = max(m, wm[w]);
        blk[blockIdx.x] = m;
    }
}

// single workgroup: exclusive prefix max of the block maxima (in place)
__global__ __launch_bounds__(PM_THREADS) void pmax_spine_kernel(int32_t *blk, uint64_t nb)
{
    __shared__ int32_t wm[PM_THREADS / WAVE];
    __shared__ int32_t carry_s;
    if (threadIdx.x == 0) carry_s = INT32_MIN;
    __syncthreads();
    for (uint64_t b0 = 0; b0 < nb; b0 += PM_THREADS) {
        const uint64_t i = b0 + threadIdx.x;
        const int32_t v = i < nb ? blk[i] : INT32_MIN;
        int32_t inc = wave_incl_max(v);
        if (lane_id() == 63) wm[threadIdx.x >> 6] = inc;
        __syncthreads();
        int32_t pre = carry_s;
        for (int w = 0; w < (int)(threadIdx.x >> 6); w++) pre = max(pre, wm[w]);
        int32_t excl = __shfl_up(inc, 1, 64);
        if (lane_id() == 0) excl = INT32_MIN;
        excl = max(excl, pre);
        if (i < nb) blk[i] = excl;
        __syncthreads();
        if (threadIdx.x == PM_THREADS - 1) carry_s = max(pre, inc);
        __syncthreads();
    }
}

__global__ __launch_bounds__(PM_THREADS) void pmax_down_kernel(const int32_t *__restrict__ in, uint64_t n,
                                                              const int32_t *__restrict__ blk, int32_t *__restrict__ out)
{
    __shared__ int32_t wm[PM_THREADS / WAVE];
    const uint64_t b0 = (uint64_t)blockIdx.x * PM_TILE + (uint64_t)threadIdx.x * PM_ITEMS;   // blocked layout
    int32_t v[PM_ITEMS];
    int32_t m = INT32_MIN;
#pragma unroll
    for (int k = 0; k < PM_ITEMS; k++) {
        v[k] = (b0 + k < n) ? in[b0 + k] : INT32_MIN;
        m = max(m, v[k]);
        v[k] = m;
    }
    int32_t inc = wave_incl_max(m);
    if (lane_id() == 63) wm[threadIdx.x >> 6] = inc;
    __syncthreads();
    int32_t pre = blk[blockIdx.x];
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) pre = max(pre, wm[w]);
    int32_t excl = __shfl_up(inc, 1, 64);
    if (lane_id() == 0) excl = INT32_MIN;
    pre = max(pre, excl);
#pragma unroll
    for (int k = 0; k < PM_ITEMS; k++)
        if (b0 + k < n) out[b0 + k] = max(pre, v[k]);
}

size_t prefix_max_tmp_bytes(uint64_t n) { return align_up(((n + PM_TILE - 1) / PM_TILE + 1) * sizeof(int32_t), 256); }

void launch_prefix_max(hipStream_t s, const int32_t *in, int32_t *out, uint64_t n, void *tmp)
{
    if (n == 0) return;
    const uint64_t nb = (n + PM_TILE - 1) / PM_TILE;
    int32_t *blk = (int32_t *)tmp;
    hipLaunchKernelGGL(pmax_reduce_kernel, dim3((unsigned)nb), dim3(PM_THREADS), 0, s, in, n, blk);
    hipLaunchKernelGGL(pmax_spine_kernel, dim3(1), dim3(PM_THREADS), 0, s, blk, nb);
    hipLaunchKernelGGL(pmax_down_kernel, dim3((unsigned)nb), dim3(PM_THREADS), 0, s, in, n, blk, out);
}

// ------------------------------------------------------------------------------- depth tiles
// first index i in [0, n] with a[i] >= target (a non-decreasing); 64-ary search by one wave
__device__ __forceinline__ uint64_t wave_first_ge_i32(const int32_t *__restrict__ a, uint64_t n, int64_t target, int lane)
{
    uint64_t lo = 0, hi = n;
    while (hi > lo) {
        const uint64_t span = hi - lo;
        const uint64_t step = (span + 63) / 64;
        const uint64_t probe = lo + (uint64_t)lane * step;
        const bool ge = probe >= hi || (int64_t)a[probe] >= target;
        const uint64_t m = __ballot(ge);
        const int first_ge = m ? __ffsll((long long)m) - 1 : 64;
        if (first_ge == 0) return lo;
        const uint64_t new_hi = first_ge < 64 ? min(lo + (uint64_t)first_ge * step, hi) : hi;
        lo = lo + (uint64_t)(first_ge - 1) * step + 1;
        hi = new_hi;
        if (step == 1) return hi;
    }
    return lo;
}

__device__ __forceinline__ uint32_t depth_grab(unsigned int *counter, int lane)
{
    uint32_t v = 0;
    if (lane == 0) v = atomicAdd(counter, 1u);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

__device__ __forceinline__ void depth_load4(const uint32_t *__restrict__ cigar, uint64_t n_cigar, int vec_ok, uint64_t idx, uint32_t (&w)[4])
{
    if (vec_ok && idx + 4 <= n_cigar) { const uint4 v = *reinterpret_cast<const uint4 *>(cigar + idx); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
    else {
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = (idx + k < n_cigar) ? cigar[idx + k] : (uint32_t)OP_P;
    }
}

// Candidate reads of every tile: reads cover 1-based positions [pos+1, ref_end]; tile t = [t*TILE, (t+1)*TILE) is touched by
// reads k with pmax_end[k] >= T0 (prefix maximum of the read ends: first such k) and pos+1 < T1. One wave per tile, two
// 64-ary searches, three dependent loads each — done once for all tiles so that no tile waits for it.
__global__ __launch_bounds__(256) void depth_ranges_kernel(const int32_t *__restrict__ pos_s, const int32_t *__restrict__ pmax_end,
                                                          uint64_t n_reads, uint32_t depth_len, uint32_t n_tiles,
                                                          uint64_t *__restrict__ tile_range)
{
    const int lane = lane_id();
    const uint32_t t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= n_tiles) return;
    const uint64_t T0 = (uint64_t)t * DEPTH_TILE;
    const uint64_t T1 = min(T0 + (uint64_t)DEPTH_TILE, (uint64_t)depth_len);
    const uint64_t lo = wave_first_ge_i32(pmax_end, n_reads, (int64_t)T0, lane);
    const uint64_t hi = wave_first_ge_i32(pos_s, n_reads, (int64_t)T1 - 1, lane);
    if (lane == 0) { tile_range[2 * (uint64_t)t] = ~lo; tile_range[2 * (uint64_t)t + 1] = hi; }      // encoding: common.hpp
}

// One (tile, candidate read) pair -> the walk item of the tile's work list, or nothing. Drops reads that end left of the tile or fail
// the depth filter (cnv_caller.cpp:491-495), and searches the read's checkpoints (scan.hip records the read's reference offset at every
// 64-word CIGAR boundary) for the last boundary left of the tile AND for the first one at or right of the tile's right edge: where the
// walk starts and how many 1 KiB chunks it needs. Everything a candidate needs first is requested together (one round trip), the
// checkpoint probes are the second; NP probes per search (more close more brackets in that trip, and cost registers).
struct DepthItem {
    uint32_t b0;        // the walk's first word / CKPT_WORDS (a checkpoint boundary)
    uint32_t nrem;      // words from there to the end of what the walk needs (the read's end, or the first boundary right of the tile)
    uint32_t pack;      // chunks to walk << 6 | first valid word, relative to the first (< 64)
    int32_t  start;     // reference position of the first word, relative to the tile
};
template <int NP>
__device__ __forceinline__ bool depth_make_item(uint64_t r, uint64_t T0, uint64_t T1, const int32_t *__restrict__ pos, const uint16_t *__restrict__ flag,
                                                const uint64_t *__restrict__ cigar_off, const int32_t *__restrict__ ref_end,
                                                const uint32_t *__restrict__ ckpt, DepthItem &out)
{
    const uint64_t c0 = cigar_off[r], c1 = cigar_off[r + 1];
    const uint32_t fl = flag[r];
    const int32_t r_end = ref_end[r], r_pos = pos[r];
    if (!(((int64_t)r_end >= (int64_t)T0) && !(fl & (F_UNMAP | F_SECONDARY | F_QCFAIL | F_DUP)) && c1 > c0)) return false;
    const uint64_t p1 = (uint64_t)(uint32_t)((uint32_t)r_pos + 1u);     // 1-based first reference position, in uint32 as there (:498): pos -1 wraps to 0
    // the read's boundaries are ckr[1 .. nb] (32-bit indices: a read has fewer than 2^25 of them)
    const uint64_t g0 = c0 >> CKPT_SHIFT;
    const uint32_t nb = (uint32_t)(((c1 - 1) >> CKPT_SHIFT) - g0);
    const uint32_t *__restrict__ const ckr = ckpt + g0;
    // f(x) = (p1 + ckr[x] <= X) is true up to the answer and false from it on; two such searches over x in [1, nb]:
    // X = T0 -> `lo`, the first boundary NOT left of the tile (the walk starts at lo - 1), and X = T1 - 1 -> `e_lo`, the
    // first boundary at or right of the tile's right edge (the walk needs no chunk that starts there).
    uint32_t lo = 1, hi = nb + 1, e_lo = 1, e_hi = nb + 1;
    uint32_t lo_val = 0;                                                 // ckr[lo - 1] whenever lo has moved
    const uint64_t X1 = T1 - 1;
    const bool want0 = nb > 0 && p1 <= T0;
    const bool want1 = nb > 0 && p1 <= X1 && (int64_t)r_end >= (int64_t)T1;      // (a read that ends inside the tile needs all its chunks)
    // The reference offset grows almost linearly with the word index, so each boundary is guessed by interpolation and NP
    // independent probes around the guess usually close the bracket: one round trip (for both searches together) where a
    // bisection of the read's ~20 checkpoints is five dependent ones. f is monotone, so the probes that satisfy it are a prefix of them.
    const uint32_t rlen = (uint32_t)r_end - (uint32_t)p1 + 1u;          // (r_end >= T0 >= p1 wherever the guess is used)
    const float per_pos = (float)nb / (float)(rlen ? rlen : 1u);
    uint32_t v0[NP], v1[NP];
    uint32_t gs0 = 0, gs1 = 0;
    auto probe_at = [&](uint32_t guess, int i) { return min(max(guess + (uint32_t)i, (uint32_t)(1 + (NP / 2 - 1))) - (uint32_t)(NP / 2 - 1), nb); };
    if (want0) {
        gs0 = min(max(1u + (uint32_t)((float)(uint32_t)(T0 - p1) * per_pos), 1u), nb);
#pragma unroll
        for (int i = 0; i < NP; i++) v0[i] = ckr[probe_at(gs0, i)];
    }
    if (want1) {
        gs1 = min(max(1u + (uint32_t)((float)(uint32_t)(X1 - p1) * per_pos), 1u), nb);
#pragma unroll
        for (int i = 0; i < NP; i++) v1[i] = ckr[probe_at(gs1, i)];
    }
    if (want0) {
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const uint32_t q = probe_at(gs0, i);
            if (p1 + v0[i] <= T0) { lo = q + 1; lo_val = v0[i]; }
            else hi = min(hi, q);
        }
    }
    if (want1) {
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const uint32_t q = probe_at(gs1, i);
            if (p1 + v1[i] <= X1) e_lo = q + 1;
            else e_hi = min(e_hi, q);
        }
    } else e_lo = e_hi = (p1 <= X1 ? nb + 1 : 1u);
    while (lo < hi || e_lo < e_hi) {                                     // both brackets shrink in the same round trips
        const bool a0 = lo < hi, a1 = e_lo < e_hi;
        const uint32_t mid = (lo + hi) >> 1, mie = (e_lo + e_hi) >> 1;
        const uint32_t v = a0 ? ckr[mid] : 0u, u = a1 ? ckr[mie] : 0u;
        if (a0) { if (p1 + v <= T0) { lo = mid + 1; lo_val = v; } else hi = mid; }
        if (a1) { if (p1 + u <= X1) e_lo = mie + 1; else e_hi = mie; }
    }
    uint64_t chunk = c0 & ~(uint64_t)(CKPT_WORDS - 1); uint32_t carry = 0;     // walks start on a checkpoint: at most 63 words re-read
    if (lo > 1) { chunk = (g0 + lo - 1) << CKPT_SHIFT; carry = lo_val; }
    if (!(p1 + carry < T1)) return false;                                // the whole read lies right of the tile
    // Words the walk needs: up to the read's end, or up to the first boundary at or right of the tile's right edge (e_lo, relative to
    // g0 like b0) — the walk fetches and decodes nothing behind it, and the reference offset there is >= T1 by definition, so the span
    // the walk accumulates still reaches the tile's edge.
    const uint32_t b0 = lo > 1 ? lo - 1 : 0u;
    uint32_t nrem = (uint32_t)(c1 - chunk);
    if (e_lo <= nb) nrem = min(nrem, (e_lo - b0) << CKPT_SHIFT);
    const uint32_t n_chunks = (nrem + 4 * WAVE - 1) / (4 * WAVE);
    out.b0 = (uint32_t)(chunk >> CKPT_SHIFT);
    out.nrem = nrem;
    out.pack = (n_chunks << 6) | (uint32_t)(c0 > chunk ? c0 - chunk : 0);
    out.start = (int32_t)((uint32_t)p1 - (uint32_t)T0 + carry);
    return true;
}

// The work lists of all tiles, ahead of the tile kernel: one small workgroup per tile examines the tile's first WL_CAP candidates and
// leaves their items in HBM (items[tile][..], n_items[tile]). Inside the tile kernel the same work costs a tile 30 % of its time — four
// dependent round trips to HBM in front of the walk, every thread of the 1024 waiting for the slowest candidate, the tile's 64 KiB of
// LDS idle meanwhile; here nothing waits for it: 8 workgroups per CU hide each other's round trips. (Later batches of a tile with
// more than WL_CAP candidates — hundreds-fold coverage — are still built by the tile kernel itself.)
constexpr int ITEMS_THREADS = 128;
__global__ __launch_bounds__(ITEMS_THREADS) void depth_items_kernel(
    const int32_t *__restrict__ pos, const uint16_t *__restrict__ flag, const uint64_t *__restrict__ cigar_off, const uint32_t *__restrict__ ord,
    const int32_t *__restrict__ ref_end, const uint64_t *tile_range, const uint32_t *__restrict__ ckpt, uint32_t depth_len,
    DepthItem *__restrict__ items, uint32_t *__restrict__ n_items)
{
    __shared__ unsigned int wl_n;
    const uint64_t T0 = (uint64_t)blockIdx.x * DEPTH_TILE;
    const uint64_t T1 = min(T0 + (uint64_t)DEPTH_TILE, (uint64_t)depth_len);
    const uint64_t k_lo = ~tile_range[2 * (uint64_t)blockIdx.x];
    // (an empty tile's range is all-zero, k_lo = ~0: candidates are counted from k_lo, never compared with a wrapped-around k_lo + WL_CAP —
    // that comparison made an empty tile examine the shard's first reads, beyond n_reads in a shard of a few reads)
    const uint64_t k_top = (uint64_t)tile_range[2 * (uint64_t)blockIdx.x + 1];
    const uint32_t k_n = k_top > k_lo ? (uint32_t)min(k_top - k_lo, (uint64_t)WL_CAP) : 0u;
    if (threadIdx.x == 0) wl_n = 0;
    __syncthreads();
    DepthItem *__restrict__ const mine = items + (uint64_t)blockIdx.x * WL_CAP;
    for (uint32_t i = threadIdx.x; i < k_n; i += ITEMS_THREADS) {
        const uint64_t kk = k_lo + i;
        DepthItem it;
        if (depth_make_item<8>(ord ? (uint64_t)ord[kk] : kk, T0, T1, pos, flag, cigar_off, ref_end, ckpt, it))
            *reinterpret_cast<uint4 *>(&mine[atomicAdd(&wl_n, 1u)]) = *reinterpret_cast<const uint4 *>(&it);
    }
    __syncthreads();
    if (threadIdx.x == 0) n_items[blockIdx.x] = wl_n;
}

#ifdef DEPTH_PHASE_PROBE
__device__ unsigned long long g_depth_phase[8];
extern "C" void csvgpu_debug_depth_phase(unsigned long long *out, int reset)
{
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_depth_phase), z, sizeof z); }
    else (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_depth_phase), 64);
}
#define PHASE_MARK(i) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&g_depth_phase[i], t_ - t_prev); t_prev = t_; } } while (0)
#else
#define PHASE_MARK(i) do { } while (0)
#endif
// PADDED: the CIGAR array is 16-byte aligned and followed by at least 4 * WAVE allocated words (csvgpu_shard_upload's own arrays):
// every chunk load is one unconditional 16-byte load per lane.
// GL: lanes that walk one item together — 64 (a wave per item, the stream of 1 KiB chunks below: long reads) or 16 / 8 (short reads:
// a (tile, read) item of a HiFi shard is 40 words, and a whole wave per item left 5 lanes in 6 idle; 64-word windows, see scan.hip).
// WPL (wave-per-item walk only): CIGAR words per lane and chunk visit — 4 (1 KiB chunks, six in flight) or 8 (2 KiB chunks, three in flight: the
// per-visit bookkeeping, the wave scan and the span update once per 512 words instead of once per 256).
template <bool PADDED, int GL, int WPL = 4>
__global__ __launch_bounds__(DEPTH_THREADS, 8) void depth_tile_kernel(
    uint64_t n_reads, uint64_t n_cigar, const int32_t *__restrict__ pos, const uint16_t *__restrict__ flag,
    const uint64_t *__restrict__ cigar_off, const uint32_t *__restrict__ cigar, int vec_ok, int dvec_ok,
    const uint32_t *__restrict__ ord,        // nullptr: reads already sorted by pos; else the tile ranges index `ord`
    const int32_t *__restrict__ ref_end, const uint64_t *tile_range,        // (written by atomics of the previous kernel: no __restrict__ / read-only path)
    const uint32_t *__restrict__ ckpt,       // reference offset of the owning read at every CKPT_WORDS-word boundary (scan.hip)
    uint32_t depth_len, uint32_t *__restrict__ depth, ScanCounters *__restrict__ cnt,
    const DepthItem *__restrict__ items, const uint32_t *__restrict__ n_items_pre)      // depth_items_kernel's lists, or null
{
    __shared__ alignas(16) uint32_t diff[DEPTH_TILE + 4];
    __shared__ uint32_t wave_tot[DEPTH_WAVES];
    __shared__ unsigned long long blk_sum;
    __shared__ unsigned int blk_nz;
    __shared__ unsigned int next_item, wl_n;
    __shared__ alignas(16) DepthItem wl[WL_CAP];

#ifdef DEPTH_PHASE_PROBE
    unsigned long long t_prev = __builtin_readcyclecounter();
#endif
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const uint64_t T0 = (uint64_t)blockIdx.x * DEPTH_TILE;
    const uint64_t T1 = min(T0 + (uint64_t)DEPTH_TILE, (uint64_t)depth_len);

    // candidate range of this tile (left by the scan, or by depth_ranges_kernel): a workgroup-uniform address, so a scalar load that
    // is in flight while the tile's difference array is zeroed; the first barrier of the batch loop below covers both
    const uint64_t k_lo = ~tile_range[2 * (uint64_t)blockIdx.x];
    const uint64_t k_hi = max(k_lo, (uint64_t)tile_range[2 * (uint64_t)blockIdx.x + 1]);
    const uint32_t n_pre = items ? n_items_pre[blockIdx.x] : 0u;        // (scalar load, in flight with the zeroing like the range)
    // the first WL_SPEC list entries are requested before their count is known (most tiles of a 30x shard have fewer): one round trip
    // less in front of the walk
    uint4 spec = make_uint4(0u, 0u, 0u, 0u);
    if (items && threadIdx.x < WL_SPEC) spec = *reinterpret_cast<const uint4 *>(&items[(uint64_t)blockIdx.x * WL_CAP + threadIdx.x]);
    for (int i = threadIdx.x * 4; i < DEPTH_TILE + 4; i += DEPTH_THREADS * 4) *reinterpret_cast<uint4 *>(&diff[i]) = make_uint4(0u, 0u, 0u, 0u);
    if (threadIdx.x == 0) { blk_sum = 0; blk_nz = 0; }

    // Work list: the tile's (start chunk, start position, word range, chunk count) items, from depth_items_kernel's list (first batch)
    // or built here by all threads together — thread t takes candidate t of the batch; the waves then pull items from an LDS counter
    // and go straight to CIGAR chunk loads, with no per-read metadata or checkpoint latency on their critical path.
    for (uint64_t cb = k_lo; cb < k_hi; cb += WL_CAP) {
        if (threadIdx.x == 0) { wl_n = (items && cb == k_lo) ? n_pre : 0u; next_item = 0; }
        __syncthreads();
        PHASE_MARK(0);
        if (items && cb == k_lo) {
            if (threadIdx.x < WL_SPEC) *reinterpret_cast<uint4 *>(&wl[threadIdx.x]) = spec;
            else if (threadIdx.x < n_pre)
                *reinterpret_cast<uint4 *>(&wl[threadIdx.x]) = *reinterpret_cast<const uint4 *>(&items[(uint64_t)blockIdx.x * WL_CAP + threadIdx.x]);
        } else {
            const uint64_t kk = cb + threadIdx.x;
            if (threadIdx.x < WL_CAP && kk < k_hi) {
                DepthItem it;
                if (depth_make_item<3>(ord ? (uint64_t)ord[kk] : kk, T0, T1, pos, flag, cigar_off, ref_end, ckpt, it))
                    *reinterpret_cast<uint4 *>(&wl[atomicAdd(&wl_n, 1u)]) = *reinterpret_cast<const uint4 *>(&it);
            }
        }
        __syncthreads();
        PHASE_MARK(1);
        const uint32_t n_items = wl_n;
        const int32_t TW = (int32_t)(T1 - T0);                                      // tile width in positions
        // Positions are kept relative to the tile's left edge in 32-bit signed arithmetic (coordinates and run lengths are below 2^31, the
        // BAM limit).
        //
        // The walk. A (tile, read) item is short — three or four chunks on a 30x ONT shard, one on HiFi — so a wave that fetched an item's
        // chunks when it reached the item spent most of its time waiting for that first round trip (the kernel was bound by latency, not
        // by instructions: halving its LDS atomics changed nothing). The wave's items are therefore ONE stream of chunks: a fetch cursor
        // runs DEPTH_PF + 1 chunks ahead of the chunk being worked on, straight across item boundaries, into a ring of chunk buffers with
        // STATIC indices (the loop is unrolled by the ring's size; a buffer is refilled in place the moment its words have been decoded).
        // Every step issues exactly one load (a repeat of the last address once the stream has run dry), so the compiler's vmcnt counts
        // stay exact. What the consumer needs to know about a buffer's chunk travels in scalar registers next to it.
        if constexpr (GL != WAVE) {
            // Short reads: every group of GL lanes takes the items gid, gid + n_groups, ... of the list and walks each in 64-word windows from the
            // item's first valid word rounded down to 16 bytes (an item of a HiFi shard is one window); the next window's words — the next
            // item's, usually — are requested before the current one is decoded. Same arithmetic as the stream below, per group.
            constexpr int GW = WAVE / GL;       // words per lane and window
            constexpr uint32_t NGRP = DEPTH_THREADS / GL;
            const uint32_t sub = threadIdx.x & (uint32_t)(GL - 1), gid = threadIdx.x / (uint32_t)GL;
            const uint32_t upper = (lane & 8) ? 0xffffffffu : 0u;
            auto loadw = [&](uint32_t g, uint32_t (&ww)[GW]) {
                if (PADDED) {
#pragma unroll
                    for (int q = 0; q < GW; q += 4) {
                        const uint4 x = *reinterpret_cast<const uint4 *>(cigar + g + q);
                        ww[q] = x.x; ww[q + 1] = x.y; ww[q + 2] = x.z; ww[q + 3] = x.w;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < GW; k++) ww[k] = ((uint64_t)g + k < n_cigar) ? cigar[g + k] : (uint32_t)OP_P;
                }
            };
            uint32_t it = gid;
            bool act = it < n_items;
            uint4 cd = *reinterpret_cast<const uint4 *>(&wl[act ? it : 0u]);       // {b0, nrem, pack, start}
            uint32_t v = 0;
            int32_t base_rel = 0;
            uint32_t ww[GW];
            loadw(act ? (((cd.x << CKPT_SHIFT) + (cd.z & 63u)) & ~3u) + sub * GW : 0u, ww);
            while (__ballot(act)) {
                const uint32_t fv = (cd.x << CKPT_SHIFT) + (cd.z & 63u);            // first valid word (32-bit word indices: the short-read forms are only chosen below 2^32 words)
                const uint32_t nval = act ? cd.y - (cd.z & 63u) : 0u;             // valid words from there
                const uint32_t a = fv & ~3u;
                const uint32_t n_win = max(1u, (fv - a + nval + 63u) >> 6);
                const bool last = v + 1 >= n_win;
                const uint32_t nit = last ? it + NGRP : it;
                const bool nact = act && nit < n_items;
                uint4 nd = cd;
                if (last) nd = *reinterpret_cast<const uint4 *>(&wl[nact ? nit : 0u]);
                const uint32_t nv = last ? 0u : v + 1u;
                uint32_t wn[GW];
                loadw(nact ? (((nd.x << CKPT_SHIFT) + (nd.z & 63u)) & ~3u) + nv * 64u + sub * GW : 0u, wn);

                if (v == 0) base_rel = (int32_t)cd.w;
                const uint32_t t = a + v * 64u + sub * GW - fv;
                uint32_t rl[GW], gp[GW], lane_ref = 0;
#pragma unroll
                for (int k = 0; k < GW; k++) {
                    if (!((uint32_t)(t + k) < nval)) ww[k] = (uint32_t)OP_P;
                    const uint32_t len = ww[k] >> 4;
                    rl[k] = len & (uint32_t)__builtin_amdgcn_sbfe((int)(REF_OPS | (REF_OPS << 16)), ww[k], 1u);
                    gp[k] = (uint32_t)__builtin_amdgcn_sbfe((int)(GAP_OPS | (GAP_OPS << 16)), ww[k], 1u);
                    lane_ref += rl[k];
                }
                const uint32_t incl = grp_incl_sum<GL>(lane_ref, upper);
                const int32_t total = (int32_t)grp_last<GL>(incl);
                {   // the window's share of the read's span, clipped to the tile
                    const int32_t sa = clamp0_i32(base_rel, TW), sb = clamp0_i32(base_rel + total, TW);
                    if (act && sa != sb && sub < 2) atomicAdd(&diff[sub ? sb : sa], sub ? 0xffffffffu : 1u);
                }
                {
                    int32_t rel = base_rel + (int32_t)(incl - lane_ref), ca = clamp0_i32(rel, TW);
#pragma unroll
                    for (int k = 0; k < GW; k++) {
                        rel += (int32_t)rl[k];
                        const int32_t cb2 = clamp0_i32(rel, TW);
                        if (gp[k]) {
                            atomicAdd(&diff[ca], 0xffffffffu);
                            atomicAdd(&diff[cb2], 1u);
                        }
                        ca = cb2;
                    }
                }
                base_rel += total;
                if (last) { v = 0; it = nit; cd = nd; act = nact; } else v = nv;
#pragma unroll
                for (int k = 0; k < GW; k++) ww[k] = wn[k];
            }
        } else if (n_items) {
            constexpr int NS = WPL == 8 ? 3 : DEPTH_PF + 1;                            // ring slots (chunks in flight)
            constexpr uint32_t CW = (uint32_t)WPL * WAVE;                            // words per chunk
            uint32_t w[NS][WPL];
            uint32_t m_o0[NS], m_nrem[NS], m_c0rel[NS];
            int32_t m_start[NS];
            bool m_valid[NS], m_first[NS];
            uint64_t f_chunk0 = 0, f_addr = 0;
            uint32_t f_n = 0, f_c = 0, f_nrem = 0, f_c0rel = 0, f_off = 0;
            int32_t f_start = 0;
            bool f_done = false;
            uint32_t nxt_v = 0;                                                     // lane 0: the next item's index (an LDS atomic in flight)
            if (lane == 0) nxt_v = atomicAdd(&next_item, 1u);
            auto fetch = [&](int j) {
                if (f_c == f_n && !f_done) {
                    const uint32_t it = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt_v);
                    if (it < n_items) {
                        const uint4 d = *reinterpret_cast<const uint4 *>(&wl[it]);      // {b0, nrem, pack, start}
                        f_chunk0 = (uint64_t)uniform32(d.x) << CKPT_SHIFT;
                        f_nrem = uniform32(d.y);
                        f_start = (int32_t)uniform32(d.w);
                        const uint32_t pk = uniform32(d.z);
                        f_n = WPL == 4 ? pk >> 6 : (f_nrem + CW - 1) / CW; f_c0rel = pk & 63u; f_c = 0;
                        if (lane == 0) nxt_v = atomicAdd(&next_item, 1u);
                    } else f_done = true;
                }
                m_valid[j] = !f_done;
                if (!f_done) {
                    m_o0[j] = f_c * CW; m_nrem[j] = f_nrem; m_c0rel[j] = f_c0rel; m_start[j] = f_start; m_first[j] = f_c == 0;
                    f_off = f_c * CW;
                    f_addr = f_chunk0 + (uint64_t)f_off;
                    f_c++;
                }
                if (PADDED) {
                    // lanes whose words all lie behind the item's last word repeat the last lane that has one: no cache line is fetched
                    // for them, and the load stays unconditional (the consumer masks those words anyway)
                    const uint32_t lw = min((uint32_t)lane * WPL, (f_nrem - 1u - f_off) & ~(uint32_t)(WPL - 1));
#pragma unroll
                    for (int q = 0; q < WPL; q += 4) {
                        const uint4 v = *reinterpret_cast<const uint4 *>(cigar + f_addr + lw + q);
                        w[j][q] = v.x; w[j][q + 1] = v.y; w[j][q + 2] = v.z; w[j][q + 3] = v.w;
                    }
                } else if (vec_ok && f_addr + CW <= n_cigar) {
#pragma unroll
                    for (int q = 0; q < WPL; q += 4) {
                        const uint4 v = *reinterpret_cast<const uint4 *>(cigar + f_addr + (uint64_t)lane * WPL + q);
                        w[j][q] = v.x; w[j][q + 1] = v.y; w[j][q + 2] = v.z; w[j][q + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < WPL; q += 4) {
                        uint32_t t4[4];
                        depth_load4(cigar, n_cigar, vec_ok, f_addr + (uint64_t)lane * WPL + q, t4);
                        w[j][q] = t4[0]; w[j][q + 1] = t4[1]; w[j][q + 2] = t4[2]; w[j][q + 3] = t4[3];
                    }
                }
            };
#pragma unroll
            for (int j = 0; j < NS; j++) fetch(j);
            int32_t base_rel = 0;                                                    // wave-uniform: position of the chunk's first word
            bool walking = true;
            while (walking) {
#pragma unroll
                for (int j = 0; j < NS; j++) {
                    if (!walking) break;
                    if (!m_valid[j]) { walking = false; break; }                    // the stream has run dry (buffers are consumed in fetch order)
                    const uint32_t o0 = m_o0[j], nrem = m_nrem[j], c0rel = m_c0rel[j];      // word offset of this chunk from the item's first chunk
                    if (m_first[j]) base_rel = m_start[j];
                    uint32_t (&cur)[WPL] = w[j];
                    // only the first and the last chunk of an item can hold words of a neighbouring read
                    if (!(o0 >= c0rel && o0 + CW <= nrem)) {
                        const uint32_t t = o0 + (uint32_t)lane * WPL - c0rel, lim = nrem - c0rel;      // word o is the item's iff o - c0rel < nrem - c0rel (unsigned)
#pragma unroll
                        for (int k = 0; k < WPL; k++) if (!(t + k < lim)) cur[k] = (uint32_t)OP_P;
                    }
                    // depth = (reads whose reference span covers the position) - (their D / N gaps over it): the same number as counting the
                    // aligned bases of every M / = / X run (cnv_caller.cpp:498-520), with HALF the difference-array updates on an ONT CIGAR
                    // (a gap op is every fourth op, an aligned run every second) and none of the run-end arithmetic: a gap covers
                    // [cursor before it, cursor after it), so its two updates sit at cursor values the walk computes anyway.
                    uint32_t rl[WPL], gp[WPL], lane_ref = 0;
#pragma unroll
                    for (int k = 0; k < WPL; k++) {
                        // v_bfe_i32 takes its bit offset from the word's low five bits (op + the length's lowest bit): op masks repeated at bit 16
                        const uint32_t len = cur[k] >> 4;
                        rl[k] = len & (uint32_t)__builtin_amdgcn_sbfe((int)(REF_OPS | (REF_OPS << 16)), cur[k], 1u);    // all-ones when the op consumes the reference
                        gp[k] = (uint32_t)__builtin_amdgcn_sbfe((int)(GAP_OPS | (GAP_OPS << 16)), cur[k], 1u);          // ... when it does so without aligned bases
                        lane_ref += rl[k];
                    }
                    fetch(j);                                                       // the buffer's words are decoded: refill it
                    const uint32_t incl = wave_incl_sum_dpp(lane_ref);
                    const int32_t total = (int32_t)(uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                    const int32_t rel0 = base_rel + (int32_t)(incl - lane_ref);
                    {   // the chunk's share of the read's span, clipped to the tile (T1 <= depth_len: out-of-range bases dropped, :511-515);
                        // consecutive chunks cancel at their common boundary
                        const int32_t sa = min(max(base_rel, 0), TW), sb = min(max(base_rel + total, 0), TW);
                        if (sa != sb && lane < 2) atomicAdd(&diff[lane ? sb : sa], lane ? 0xffffffffu : 1u);
                    }
                    if (base_rel >= 0 && base_rel + total <= TW) {
                        // the whole chunk lies inside the tile (most do: a tile is seven chunks wide): cursors as byte offsets, no clipping
                        uint32_t a4 = (uint32_t)rel0 << 2;
#pragma unroll
                        for (int k = 0; k < WPL; k++) {
                            const uint32_t n4 = lshl2_add(rl[k], a4);
                            if (gp[k]) {
                                atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(diff) + a4), 0xffffffffu);
                                atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(diff) + n4), 1u);
                            }
                            a4 = n4;
                        }
                    } else {
                        int32_t rel = rel0, a = clamp0_i32(rel0, TW);
#pragma unroll
                        for (int k = 0; k < WPL; k++) {
                            rel += (int32_t)rl[k];
                            const int32_t b = clamp0_i32(rel, TW);
                            if (gp[k]) {                                       // (both ends clipped to the same edge cancel)
                                atomicAdd(&diff[a], 0xffffffffu);
                                atomicAdd(&diff[b], 1u);
                            }
                            a = b;
                        }
                    }
                    base_rel += total;
                }
            }
        }
        PHASE_MARK(2);
        __syncthreads();
        PHASE_MARK(3);
    }

    if (k_lo >= k_hi) __syncthreads();       // no batch ran (empty tile): the zeroing above has not met a barrier yet
    // scan the difference array: wave w owns entries [w*DEPTH_PER_WAVE, (w+1)*DEPTH_PER_WAVE)
    const int w_base = wave * DEPTH_PER_WAVE;
    uint32_t tot = 0;
#pragma unroll
    for (int rd = 0; rd < DEPTH_ROUNDS; rd++) {
        const uint4 v = *reinterpret_cast<const uint4 *>(&diff[w_base + rd * 4 * WAVE + lane * 4]);
        tot += v.x + v.y + v.z + v.w;
    }
    tot = wave_total_dpp(tot);
    if (lane == 0) wave_tot[wave] = tot;
    __syncthreads();
    // prefix of the waves' totals: lanes 0..15 scan them (one DPP row), the wave reads its predecessor's
    static_assert(DEPTH_WAVES == 16, "the waves' totals are scanned by one DPP row");
    uint32_t carry;
    {
        uint32_t wt = lane < DEPTH_WAVES ? wave_tot[lane] : 0u;
        wt += dpp_u32<0x111, 0xf>(0u, wt); wt += dpp_u32<0x112, 0xf>(0u, wt); wt += dpp_u32<0x114, 0xf>(0u, wt); wt += dpp_u32<0x118, 0xf>(0u, wt);
        const int wv = (int)uniform32((uint32_t)wave);
        carry = wv ? (uint32_t)__builtin_amdgcn_readlane((int)wt, wv - 1) : 0u;
    }

    uint64_t my_sum = 0;
    uint32_t my_nz = 0;
    // A whole tile with an aligned output: no per-position bound tests, one 16-byte store per lane and round at a constant offset from one
    // address, the tile's sum in 32-bit partial sums — a position's depth is at most the tile's candidate count (a read covers a position
    // once), so 16 values per lane and 64 lanes stay below 2^32 while there are fewer than 2^20 candidates —, non-zero counts on the scalar
    // unit (ballot + popcount). The scan + write-out was 400 of a wave's ~600 (HiFi) / ~2 500 (ONT) vector instructions per tile.
    const bool fast = depth && dvec_ok && (T1 - T0) == (uint64_t)DEPTH_TILE && (k_hi - k_lo) < (1ull << 20);
    if (fast) {
        uint32_t *__restrict__ const dptr = depth + T0 + (uint64_t)(w_base + lane * 4);
        uint32_t s32 = 0, nzs = 0;
#pragma unroll
        for (int rd = 0; rd < DEPTH_ROUNDS; rd++) {
            const uint4 v = *reinterpret_cast<const uint4 *>(&diff[w_base + rd * 4 * WAVE + lane * 4]);
            const uint32_t l0 = v.x, l1 = l0 + v.y, l2 = l1 + v.z, l3 = l2 + v.w;
            const uint32_t incl = wave_incl_sum_dpp(l3);
            const uint32_t pre = carry + (incl - l3);
            uint4 d;
            d.x = pre + l0; d.y = pre + l1; d.z = pre + l2; d.w = pre + l3;
            *reinterpret_cast<uint4 *>(dptr + rd * 4 * WAVE) = d;
            s32 += (d.x + d.y) + (d.z + d.w);
            nzs += (uint32_t)__popcll(__ballot(d.x != 0)) + (uint32_t)__popcll(__ballot(d.y != 0)) + (uint32_t)__popcll(__ballot(d.z != 0)) +
                   (uint32_t)__popcll(__ballot(d.w != 0));
            carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        const uint32_t tot32 = wave_total_dpp(s32);
        if (lane == 0) { atomicAdd(&blk_sum, (unsigned long long)tot32); atomicAdd(&blk_nz, nzs); }
    } else {
#pragma unroll
    for (int rd = 0; rd < DEPTH_ROUNDS; rd++) {
        const int off = w_base + rd * 4 * WAVE + lane * 4;
        const uint4 v = *reinterpret_cast<const uint4 *>(&diff[off]);
        const uint32_t l0 = v.x, l1 = l0 + v.y, l2 = l1 + v.z, l3 = l2 + v.w;
        const uint32_t incl = wave_incl_sum_dpp(l3);
        const uint32_t pre = carry + (incl - l3);
        uint4 d;
        d.x = pre + l0; d.y = pre + l1; d.z = pre + l2; d.w = pre + l3;
        const uint64_t g = T0 + (uint64_t)off;
        if (depth) {
            if (dvec_ok && g + 4 <= T1) {
                *reinterpret_cast<uint4 *>(depth + g) = d;
            } else {
                if (g + 0 < T1) depth[g + 0] = d.x;
                if (g + 1 < T1) depth[g + 1] = d.y;
                if (g + 2 < T1) depth[g + 2] = d.z;
                if (g + 3 < T1) depth[g + 3] = d.w;
            }
        }
        if (g + 0 < T1) { my_sum += d.x; my_nz += d.x > 0; }
        if (g + 1 < T1) { my_sum += d.y; my_nz += d.y > 0; }
        if (g + 2 < T1) { my_sum += d.z; my_nz += d.z > 0; }
        if (g + 3 < T1) { my_sum += d.w; my_nz += d.w > 0; }
        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    my_sum = wave_sum64(my_sum);
    my_nz = wave_sum(my_nz);
    if (lane == 0) { atomicAdd(&blk_sum, (unsigned long long)my_sum); atomicAdd(&blk_nz, my_nz); }
    }
    __syncthreads();
    PHASE_MARK(4);
    if (threadIdx.x == 0) {
        if (blk_sum) atomicAdd(&cnt->depth_sum, blk_sum);
        if (blk_nz) atomicAdd(&cnt->depth_nonzero, blk_nz);
    }
}

size_t depth_tiles_tmp_bytes(uint32_t depth_len) { return align_up(((size_t)depth_n_tiles(depth_len) + 1) * 16, 256); }

void launch_depth_ranges(hipStream_t s, const int32_t *pos_s, const int32_t *pmax_end, uint64_t n_reads, uint32_t depth_len, uint64_t *tile_range)
{
    const unsigned tiles = depth_n_tiles(depth_len);
    if (!tiles) return;
    hipLaunchKernelGGL(depth_ranges_kernel, dim3((tiles + 3) / 4), dim3(256), 0, s, pos_s, pmax_end, n_reads, depth_len, tiles, tile_range);
}

// items: depth_items_bytes(depth_len) bytes of scratch (work lists of all tiles, built by a kernel of its own ahead of the tile kernel), or
// null: every tile builds its own list
size_t depth_items_bytes(uint32_t depth_len) { return align_up((size_t)depth_n_tiles(depth_len) * (WL_CAP * sizeof(DepthItem) + sizeof(uint32_t)), 256); }

void launch_depth_tiles(hipStream_t s, const csv_reads &d, const uint32_t *ord, const int32_t *ref_end, const uint32_t *ckpt,
                        uint32_t depth_len, uint32_t *depth, ScanCounters *cnt, const uint64_t *tile_range, uint32_t cigar_pad_words, void *items, int form)
{
    if (depth_len == 0) return;
    const unsigned tiles = depth_n_tiles(depth_len);
    const int vec_ok = (((uintptr_t)d.cigar) & 15u) == 0;
    const int dvec_ok = (((uintptr_t)depth) & 15u) == 0;
    DepthItem *it = (DepthItem *)items;
    uint32_t *n_it = items ? (uint32_t *)(it + (size_t)tiles * WL_CAP) : nullptr;
    if (items)
        hipLaunchKernelGGL(depth_items_kernel, dim3(tiles), dim3(ITEMS_THREADS), 0, s, d.pos, d.flag, d.cigar_off, ord, ref_end, tile_range, ckpt, depth_len, it, n_it);
    if (form != SCAN_FORM_WAVE && d.n_cigar >= 0xffffffffull) form = SCAN_FORM_WAVE;
    const bool padded = vec_ok && cigar_pad_words >= 4 * WAVE;
#define CSV_TILE_LAUNCH(PAD, GL)                                                                                                                 \
    hipLaunchKernelGGL((depth_tile_kernel<PAD, GL>), dim3(tiles), dim3(DEPTH_THREADS), 0, s, d.n_reads, d.n_cigar, d.pos, d.flag, d.cigar_off, d.cigar, \
                       vec_ok, dvec_ok, ord, ref_end, tile_range, ckpt, depth_len, depth, cnt, it, n_it)
    if (form == SCAN_FORM_ROWS16 || form == SCAN_FORM_LANES) { if (padded) CSV_TILE_LAUNCH(true, 16); else CSV_TILE_LAUNCH(false, 16); }
    else if (form == SCAN_FORM_ROWS8) { if (padded) CSV_TILE_LAUNCH(true, 8); else CSV_TILE_LAUNCH(false, 8); }
    else {
        static const int wpl = [] { const char *e = getenv("CSV_DEPTH_WPL"); return e && atoi(e) == 8 ? 8 : 4; }();
        if (wpl == 8 && cigar_pad_words >= 8 * WAVE) {
            hipLaunchKernelGGL((depth_tile_kernel<true, WAVE, 8>), dim3(tiles), dim3(DEPTH_THREADS), 0, s, d.n_reads, d.n_cigar, d.pos, d.flag, d.cigar_off, d.cigar,
                               vec_ok, dvec_ok, ord, ref_end, tile_range, ckpt, depth_len, depth, cnt, it, n_it);
        } else if (padded) CSV_TILE_LAUNCH(true, WAVE); else CSV_TILE_LAUNCH(false, WAVE);
    }
#undef CSV_TILE_LAUNCH
}

// min_pts = (int)ceil(mean_cov * pct), or 5 when pct <= 0 (sv_caller.cpp:723-728); mean = sum / #non-zero
// (cnv_caller.cpp:534-538). Same IEEE double ops as the host expression.
__global__ void min_pts_kernel(ScanCounters *cnt, double pct)
{
    const double mean = cnt->depth_nonzero > 0 ? (double)cnt->depth_sum / (double)cnt->depth_nonzero : 0.0;
    cnt->mean_cov = mean;
    cnt->min_pts = pct > 0.0 ? (int)ceil(mean * pct) : 5;
}

void launch_min_pts(hipStream_t s, ScanCounters *cnt, double min_pts_pct)
{
    hipLaunchKernelGGL(min_pts_kernel, dim3(1), dim3(1), 0, s, cnt, min_pts_pct);
}

// depth[pos[i]] for a handful of positions (the VCF writer's SUPPORT / DP lookups, sv_caller.cpp:1306, :1332-1344);
// -1 marks a position outside the map, which std::vector::at reports as out_of_range in the reference.
__global__ void depth_lookup_kernel(const uint32_t *__restrict__ depth, uint32_t depth_len, const uint32_t *__restrict__ pos,
                                    uint64_t n, int32_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t p = pos[i];
    out[i] = p < depth_len ? (int32_t)depth[p] : -1;
}

void launch_depth_lookup(hipStream_t s, const uint32_t *depth, uint32_t depth_len, const uint32_t *pos, uint64_t n, int32_t *out)
{
    if (n == 0) return;
    hipLaunchKernelGGL(depth_lookup_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, depth, depth_len, pos, n, out);
}

}  // namespace csv
