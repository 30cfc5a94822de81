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
constexpr int DEPTH_PF = 3;                                // a walk keeps DEPTH_PF + 1 chunks in flight ahead of the one being worked on (1 -> 4: -4 %)

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

__global__ __launch_bounds__(DEPTH_THREADS, 8) void depth_tile_kernel(
    uint64_t n_reads, uint64_t n_cigar, const int32_t *__restrict__ pos, const uint16_t *__restrict__ flag,
    const uint64_t *__restrict__ cigar_off, const uint32_t *__restrict__ cigar, int vec_ok, int dvec_ok,
    const uint32_t *__restrict__ ord,        // nullptr: reads already sorted by pos; else the tile ranges index `ord`
    const int32_t *__restrict__ ref_end, const uint64_t *tile_range,        // (written by atomics of the previous kernel: no __restrict__ / read-only path)
    const uint32_t *__restrict__ ckpt,       // reference offset of the owning read at every CKPT_WORDS-word boundary (scan.hip)
    uint32_t depth_len, uint32_t *__restrict__ depth, ScanCounters *__restrict__ cnt)
{
    __shared__ alignas(16) uint32_t diff[DEPTH_TILE + 4];
    __shared__ uint32_t wave_tot[DEPTH_WAVES];
    __shared__ unsigned long long blk_sum;
    __shared__ unsigned int blk_nz;
    __shared__ unsigned int next_item, wl_n;
    __shared__ uint64_t wl_chunk[WL_CAP];
    __shared__ int32_t wl_c0rel[WL_CAP];
    __shared__ uint32_t wl_nrem[WL_CAP], wl_p1[WL_CAP], wl_carry[WL_CAP];

    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const uint64_t T0 = (uint64_t)blockIdx.x * DEPTH_TILE;
    const uint64_t T1 = min(T0 + (uint64_t)DEPTH_TILE, (uint64_t)depth_len);

    // candidate range of this tile (left by the scan, or by depth_ranges_kernel): a workgroup-uniform address, so a scalar load that
    // is in flight while the tile's difference array is zeroed; the first barrier of the batch loop below covers both
    const uint64_t k_lo = ~tile_range[2 * (uint64_t)blockIdx.x];
    const uint64_t k_hi = max(k_lo, (uint64_t)tile_range[2 * (uint64_t)blockIdx.x + 1]);
    for (int i = threadIdx.x * 4; i < DEPTH_TILE + 4; i += DEPTH_THREADS * 4) *reinterpret_cast<uint4 *>(&diff[i]) = make_uint4(0u, 0u, 0u, 0u);
    if (threadIdx.x == 0) { blk_sum = 0; blk_nz = 0; }

    // Work list: the candidates are examined ONCE per tile by all threads together — thread t takes candidate t of the
    // batch, loads its metadata (coalesced across threads), drops reads that end left of the tile or fail the depth filter
    // (cnv_caller.cpp:491-495), and searches the read's checkpoints for the last 64-word boundary left of the tile.
    // Surviving (start chunk, reference carry, word range) items go to LDS; the waves then pull items from an LDS counter
    // and go straight to CIGAR chunk loads, with no per-read metadata or checkpoint latency on their critical path.
    for (uint64_t cb = k_lo; cb < k_hi; cb += WL_CAP) {
        if (threadIdx.x == 0) { wl_n = 0; next_item = 0; }
        __syncthreads();
        const uint64_t kk = cb + threadIdx.x;
        if (threadIdx.x < WL_CAP && kk < k_hi) {
            const uint64_t r = ord ? (uint64_t)ord[kk] : kk;
            // everything a candidate needs first is requested together (one round trip), the checkpoint probes are the second, and the
            // reference offset of the start boundary is the value the last successful probe returned: two dependent round trips per tile
            const uint64_t c0 = cigar_off[r], c1 = cigar_off[r + 1];
            const uint32_t fl = flag[r];
            const int32_t r_end = ref_end[r], r_pos = pos[r];
            const bool ok = ((int64_t)r_end >= (int64_t)T0) && !(fl & (F_UNMAP | F_SECONDARY | F_QCFAIL | F_DUP)) && c1 > c0;
            if (ok) {
                const uint64_t p1 = (uint64_t)(uint32_t)((uint32_t)r_pos + 1u);     // 1-based first reference position, in uint32 as there (:498): pos -1 wraps to 0
                const uint64_t g0 = c0 >> CKPT_SHIFT, g1 = (c1 - 1) >> CKPT_SHIFT;
                uint64_t lo = g0 + 1, hi = g1 + 1;                                   // first boundary NOT left of the tile
                uint32_t lo_val = 0;                                                 // ckpt[lo - 1] whenever lo has moved
                // The reference offset grows almost linearly with the word index, so the boundary is guessed by interpolation and
                // three independent probes around the guess usually close the bracket: one round trip where a bisection of the
                // read's ~20 checkpoints is five dependent ones (this search sits on every tile's critical path).
                if (lo < hi && p1 <= T0) {
                    const uint64_t rlen = (uint64_t)((int64_t)r_end - (int64_t)p1 + 1);
                    uint64_t guess = lo + (uint64_t)((double)(T0 - p1) / (double)(rlen ? rlen : 1) * (double)(hi - lo));
                    guess = min(max(guess, lo), hi - 1);
                    const uint64_t ga = guess > lo ? guess - 1 : lo, gb = guess, gc = min(guess + 1, hi - 1);
                    const uint32_t ka = ckpt[ga], kb = ckpt[gb], kc = ckpt[gc];
                    // f(g) = (p1 + ckpt[g] <= T0) is true up to the answer and false from it on
                    if (p1 + kc <= T0) { lo = gc + 1; lo_val = kc; }
                    else {
                        hi = gc;
                        if (p1 + kb <= T0) { lo = gb + 1; lo_val = kb; }
                        else { hi = gb; if (p1 + ka <= T0) { lo = ga + 1; lo_val = ka; } else hi = ga; }
                    }
                }
                while (lo < hi) {
                    const uint64_t mid = (lo + hi) >> 1;
                    const uint32_t v = ckpt[mid];
                    if (p1 + v <= T0) { lo = mid + 1; lo_val = v; } else hi = mid;
                }
                uint64_t chunk = c0 & ~(uint64_t)(CKPT_WORDS - 1); uint32_t carry = 0;     // walks start on a checkpoint: at most 63 words re-read
                if (lo > g0 + 1) { chunk = (lo - 1) << CKPT_SHIFT; carry = lo_val; }
                if (p1 + carry < T1) {                                               // else the whole read lies right of the tile
                    const uint32_t slot = atomicAdd(&wl_n, 1u);
                    wl_chunk[slot] = chunk;
                    wl_c0rel[slot] = (int32_t)((int64_t)c0 - (int64_t)chunk);        // first valid word, relative to the start chunk
                    wl_nrem[slot] = (uint32_t)(c1 - chunk);                          // words from the start chunk to the read's end
                    wl_p1[slot] = (uint32_t)p1;
                    wl_carry[slot] = carry;
                }
            }
        }
        __syncthreads();
        const uint32_t n_items = wl_n;
        const int32_t TW = (int32_t)(T1 - T0);                                      // tile width in positions
        // How many chunks does a walk need? A chunk boundary is a checkpoint slot, so the checkpoints say exactly where a read's
        // chunks start on the reference: lane l looks up the start of chunk l + 1 of the item, one vector load — requested while the
        // PREVIOUS item of this wave is being walked, so nobody waits for it. The walk then fetches no chunk that starts right of the
        // tile's right edge (DEPTH_PF + 1 chunks are kept in flight — the walk is bound by the bytes its 32 waves per CU keep in
        // flight more than by anything else —: without the count every (tile, read) pair fetched that many chunks past the edge).
        auto look_ahead = [&](uint32_t item) -> uint32_t {
            const uint32_t off = (uint32_t)(lane + 1) * (4 * WAVE);
            return off < uniform32(wl_nrem[item]) ? ckpt[(uniform64(wl_chunk[item]) + off) >> CKPT_SHIFT] : 0xffffffffu;
        };
        uint32_t it = depth_grab(&next_item, lane);
        uint32_t la = it < n_items ? look_ahead(it) : 0u;
        while (it < n_items) {
            const uint64_t chunk0 = uniform64(wl_chunk[it]);                        // `it` is wave-uniform: keep the item in scalar registers
            const int32_t c0rel = (int32_t)uniform32((uint32_t)wl_c0rel[it]);
            const uint32_t nrem = uniform32(wl_nrem[it]);
            // Positions are kept relative to the tile's left edge in 32-bit signed arithmetic (coordinates and run lengths are
            // below 2^31, the BAM limit): a run [rel, rel + len) clips to [max(rel,0), min(rel+len, TW)) with one max and one min,
            // and a run with no aligned bases (len masked to 0) clips to nothing, so no separate op test is needed.
            const int32_t p1rel = (int32_t)uniform32(wl_p1[it] - (uint32_t)T0);     // the read's first position, relative to the tile
            int32_t base_rel = p1rel + (int32_t)uniform32(wl_carry[it]);            // wave-uniform: position of the chunk's first staged word
            const uint32_t it_next = depth_grab(&next_item, lane);
            auto load_chunk = [&](uint32_t off, uint32_t (&dst)[4]) {
                const uint64_t c = chunk0 + off;
                if (vec_ok && c + 4 * WAVE <= n_cigar) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(cigar + c + (uint64_t)lane * 4); dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
                } else depth_load4(cigar, n_cigar, vec_ok, c + (uint64_t)lane * 4, dst);
            };
            uint32_t la_next = 0u;
            {
                uint32_t n_chunks = (nrem + 4 * WAVE - 1) / (4 * WAVE);
                {   // chunks after the first that start left of the right edge: a prefix of the lanes (reference offsets do not decrease).
                    // All 64 looked-up chunks needed: the walk is longer than the look-up reaches and runs to the read's end as before.
                    const uint32_t n_more = (uint32_t)__popcll(__ballot(la != 0xffffffffu && p1rel + (int32_t)la < TW));
                    if (n_more < (uint32_t)WAVE) n_chunks = min(n_chunks, n_more + 1u);
                }
                // DEPTH_PF + 1 chunk buffers used as a ring with STATIC indices: the chunk loop is unrolled by the ring's size, so a buffer
                // is refilled in place the moment its words have been decoded (rotating the buffers cost 12 register moves per chunk,
                // a sixth of the loop's vector instructions).
                uint32_t w[DEPTH_PF + 1][4];
#pragma unroll
                for (int j = 0; j <= DEPTH_PF; j++) {
#pragma unroll
                    for (int k = 0; k < 4; k++) w[j][k] = 0;
                    if ((uint32_t)j < n_chunks) load_chunk((uint32_t)j * (4 * WAVE), w[j]);
                }
                if (it_next < n_items) la_next = look_ahead(it_next);                // in flight during this walk
                bool walking = true;
                for (uint32_t c0 = 0; walking; c0 += DEPTH_PF + 1) {
#pragma unroll
                    for (int j = 0; j <= DEPTH_PF; j++) {
                        if (!walking) break;
                        const uint32_t c = c0 + (uint32_t)j;
                        const uint32_t o0 = c * (4 * WAVE);                         // word offset of this chunk from chunk0
                        uint32_t (&cur)[4] = w[j];
                        // only the first and the last chunk of an item can hold words of a neighbouring read
                        if (!((int32_t)o0 >= c0rel && o0 + 4 * WAVE <= nrem)) {
                            const int32_t o = (int32_t)o0 + lane * 4;
#pragma unroll
                            for (int k = 0; k < 4; k++) if (!((o + k >= c0rel) && ((uint32_t)(o + k) < nrem))) cur[k] = (uint32_t)OP_P;
                        }
                        uint32_t rl[4], al[4], lane_ref = 0;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            // v_bfe_i32 takes its bit offset from the word's low five bits (op + the length's lowest bit): op masks repeated at bit 16
                            const uint32_t len = cur[k] >> 4;
                            rl[k] = len & (uint32_t)__builtin_amdgcn_sbfe((int)(REF_OPS | (REF_OPS << 16)), cur[k], 1u);    // all-ones when the op consumes the reference
                            al[k] = len & (uint32_t)__builtin_amdgcn_sbfe((int)(ALN_OPS | (ALN_OPS << 16)), cur[k], 1u);    // ... when its bases count toward depth
                            lane_ref += rl[k];
                        }
                        // the buffer's words are decoded: refill it with the chunk DEPTH_PF + 1 ahead
                        if (c + DEPTH_PF + 1 < n_chunks) load_chunk(o0 + (DEPTH_PF + 1) * (4 * WAVE), w[j]);
                        const uint32_t incl = wave_incl_sum_dpp(lane_ref);
                        int32_t rel = base_rel + (int32_t)(incl - lane_ref);
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int32_t a = max(rel, 0), b = min(rel + (int32_t)al[k], TW);   // T1 <= depth_len: out-of-range bases dropped (:511-515)
                            if (a < b) {
                                atomicAdd(&diff[a], 1u);
                                atomicAdd(&diff[b], 0xffffffffu);
                            }
                            rel += (int32_t)rl[k];
                        }
                        base_rel += (int32_t)(uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                        if (c + 1 >= n_chunks || base_rel >= TW) walking = false;   // the read ends here, or the rest of it lies right of the tile
                    }
                }
            }
            it = it_next; la = la_next;
        }
        __syncthreads();
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
    uint32_t carry = 0;
    for (int w = 0; w < wave; w++) carry += wave_tot[w];

    uint64_t my_sum = 0;
    uint32_t my_nz = 0;
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
    __syncthreads();
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

void launch_depth_tiles(hipStream_t s, const csv_reads &d, const uint32_t *ord, const int32_t *ref_end, const uint32_t *ckpt,
                        uint32_t depth_len, uint32_t *depth, ScanCounters *cnt, const uint64_t *tile_range)
{
    if (depth_len == 0) return;
    const unsigned tiles = depth_n_tiles(depth_len);
    const int vec_ok = (((uintptr_t)d.cigar) & 15u) == 0;
    const int dvec_ok = (((uintptr_t)depth) & 15u) == 0;
    hipLaunchKernelGGL(depth_tile_kernel, dim3(tiles), dim3(DEPTH_THREADS), 0, s, d.n_reads, d.n_cigar, d.pos, d.flag,
                       d.cigar_off, d.cigar, vec_ok, dvec_ok, ord, ref_end, tile_range, ckpt, depth_len, depth, cnt);
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
