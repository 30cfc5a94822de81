// sort.hip — ordering pass (gfx950): stable LSD radix sort + the tie rule of addSVCall.
//
// The reference keeps one vector per chromosome sorted by (start,end) with std::lower_bound inserts
// (sv_object.cpp:22-33), so equal (start,end) calls end up in REVERSE insertion order. That order is
// DBSCAN's index order and decides which member std::sort leaves at the representative's slot, so it
// is reproduced exactly: (1) radix-sort (type | start | end-start) keys, (2) inside each run of equal
// keys rank the members by (read, query offset) descending — the insertion order of the reference is
// (read ascending, CIGAR order), and the query offset grows along the CIGAR for ops at one position.
//
// Radix sort: 8-bit digits, one 64-lane wave owns a tile of 2048 keys and walks it in 32 rounds of
// 64 (round-major = index order). Ranks inside a round come from an 8-ballot match-any, per-digit
// running offsets live in the wave's private 1 KiB of LDS, so the scatter is stable without any
// workgroup barrier. Three small kernels per pass: histogram, single-workgroup exclusive scan of the
// (digit, tile) table, scatter. Only as many passes as the keys have bits (41-bit keys: 6 passes;
// 11-bit digits were measured slower: 2x the time in table traffic for 2 passes less).
#include "../common.hpp"
#include "../devutil.hpp"

namespace csv {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / WAVE;
constexpr int RS_ROUNDS = 32;

constexpr int RS_BITS = 8;                    // digit width
constexpr int RS_BINS = 1 << RS_BITS;         // 256 digit cursors per wave = 1 KiB of LDS

// ------------------------------------------------------------------------------- generic exclusive sum (u32)
constexpr int ES_THREADS = 256;
constexpr int ES_ITEMS = 8;
constexpr int ES_TILE = ES_THREADS * ES_ITEMS;

__global__ __launch_bounds__(ES_THREADS) void es_reduce_kernel(const uint32_t *__restrict__ in, uint64_t n, uint32_t *__restrict__ blk)
{
    __shared__ uint32_t ws[ES_THREADS / WAVE];
    const uint64_t b0 = (uint64_t)blockIdx.x * ES_TILE;
    uint32_t m = 0;
    for (int k = 0; k < ES_ITEMS; k++) {
        uint64_t i = b0 + (uint64_t)k * ES_THREADS + threadIdx.x;
        if (i < n) m += in[i];
    }
    m = wave_sum(m);
    if (lane_id() == 0) ws[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < ES_THREADS / WAVE; w++) m += ws[w];
        blk[blockIdx.x] = m;
    }
}

__global__ __launch_bounds__(ES_THREADS) void es_spine_kernel(uint32_t *blk, uint64_t nb)
{
    __shared__ uint32_t ws[ES_THREADS / WAVE];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint64_t b0 = 0; b0 < nb; b0 += ES_THREADS) {
        const uint64_t i = b0 + threadIdx.x;
        const uint32_t v = i < nb ? blk[i] : 0u;
        const uint32_t inc = wave_incl_sum(v);
        if (lane_id() == 63) ws[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t pre = carry_s;
        for (int w = 0; w < (int)(threadIdx.x >> 6); w++) pre += ws[w];
        if (i < nb) blk[i] = pre + inc - v;
        __syncthreads();
        if (threadIdx.x == ES_THREADS - 1) carry_s = pre + inc;
        __syncthreads();
    }
}

__global__ __launch_bounds__(ES_THREADS) void es_down_kernel(uint32_t *__restrict__ data, uint64_t n, const uint32_t *__restrict__ blk)
{
    __shared__ uint32_t ws[ES_THREADS / WAVE];
    const uint64_t b0 = (uint64_t)blockIdx.x * ES_TILE + (uint64_t)threadIdx.x * ES_ITEMS;
    uint32_t v[ES_ITEMS], m = 0;
#pragma unroll
    for (int k = 0; k < ES_ITEMS; k++) { v[k] = (b0 + k < n) ? data[b0 + k] : 0u; m += v[k]; }
    const uint32_t inc = wave_incl_sum(m);
    if (lane_id() == 63) ws[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t pre = blk[blockIdx.x] + inc - m;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) pre += ws[w];
#pragma unroll
    for (int k = 0; k < ES_ITEMS; k++) {
        if (b0 + k < n) data[b0 + k] = pre;
        pre += v[k];
    }
}

// one workgroup scans the whole array in 16 Ki-entry chunks staged through LDS (coalesced loads and stores; each
// thread scans 16 consecutive LDS entries, the 1024 run totals go through a DPP wave scan). One launch instead of
// three — these tables are a few thousand to a few hundred thousand entries, where latency, not bandwidth, is the cost.
constexpr int ES1_THREADS = 1024;
constexpr int ES1_PER = 16;                        // entries per thread per chunk
constexpr int ES1_CHUNK = ES1_THREADS * ES1_PER;   // 16384 entries (64 KiB of LDS) per chunk
constexpr uint64_t ES1_MAX = 1ull << 16;      // beyond four chunks the three-launch form wins (one workgroup takes ~0.1 us per Ki entries: 215 us for the 1.2 M-entry table of a 5 M-key radix pass)
__global__ __launch_bounds__(ES1_THREADS) void es_single_kernel(uint32_t *__restrict__ data, uint64_t n)
{
    __shared__ uint32_t buf[ES1_CHUNK + ES1_CHUNK / 32];   // +1 word of padding every 32: the per-thread runs stay conflict-light
    __shared__ uint32_t ws[ES1_THREADS / WAVE];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    for (uint64_t c0 = 0; c0 < n; c0 += ES1_CHUNK) {
        const uint32_t cn = (uint32_t)min((uint64_t)ES1_CHUNK, n - c0);
        for (uint32_t i = threadIdx.x; i < cn; i += ES1_THREADS) buf[i + (i >> 5)] = data[c0 + i];      // coalesced in
        __syncthreads();
        const uint32_t b0 = threadIdx.x * ES1_PER;
        uint32_t v[ES1_PER], m = 0;
#pragma unroll
        for (int k = 0; k < ES1_PER; k++) { const uint32_t i = b0 + k; v[k] = i < cn ? buf[i + (i >> 5)] : 0u; m += v[k]; }
        const uint32_t inc = wave_incl_sum_dpp(m);
        if (lane_id() == 63) ws[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t pre = carry_s + inc - m;
        for (int w = 0; w < (int)(threadIdx.x >> 6); w++) pre += ws[w];
#pragma unroll
        for (int k = 0; k < ES1_PER; k++) { const uint32_t i = b0 + k; if (i < cn) buf[i + (i >> 5)] = pre; pre += v[k]; }
        __syncthreads();
        if (threadIdx.x == ES1_THREADS - 1) carry_s = pre;
        for (uint32_t i = threadIdx.x; i < cn; i += ES1_THREADS) data[c0 + i] = buf[i + (i >> 5)];      // coalesced out
        __syncthreads();
    }
}

size_t exclusive_sum_tmp_bytes(uint64_t n) { return align_up(((n + ES_TILE - 1) / ES_TILE + 1) * sizeof(uint32_t), 256); }

void launch_exclusive_sum_u32(hipStream_t s, uint32_t *data, uint64_t n, void *tmp)
{
    if (n == 0) return;
    if (n <= ES1_MAX) { hipLaunchKernelGGL(es_single_kernel, dim3(1), dim3(ES1_THREADS), 0, s, data, n); return; }
    const uint64_t nb = (n + ES_TILE - 1) / ES_TILE;
    uint32_t *blk = (uint32_t *)tmp;
    hipLaunchKernelGGL(es_reduce_kernel, dim3((unsigned)nb), dim3(ES_THREADS), 0, s, data, n, blk);
    hipLaunchKernelGGL(es_spine_kernel, dim3(1), dim3(ES_THREADS), 0, s, blk, nb);
    hipLaunchKernelGGL(es_down_kernel, dim3((unsigned)nb), dim3(ES_THREADS), 0, s, data, n, blk);
}

// ------------------------------------------------------------------------------- radix sort passes
// keys per wave-tile: small inputs get small tiles so that the pass still fills the chip with waves
__host__ __device__ static inline int rs_rounds_for(uint64_t n)
{
    int rounds = RS_ROUNDS;
    // fewer keys per tile = more waves but a larger (digit, tile) table to scan: ~128 tiles is the measured sweet spot
    while (rounds > 2 && (n + (uint64_t)rounds * WAVE - 1) / ((uint64_t)rounds * WAVE) < 128) rounds >>= 1;
    return rounds;
}

__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const uint64_t *__restrict__ keys, uint64_t n, int shift,
                                                            uint32_t *__restrict__ table, uint32_t n_tiles, int rounds)
{
    __shared__ uint32_t hist[RS_WAVES][RS_BINS];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const uint64_t tile = (uint64_t)blockIdx.x * RS_WAVES + wave;
    for (int b = lane; b < RS_BINS; b += WAVE) hist[wave][b] = 0;
    __builtin_amdgcn_wave_barrier();
    if (tile < n_tiles) {
        const uint64_t t0 = tile * (uint64_t)rounds * WAVE;
        for (int rd = 0; rd < rounds; rd++) {
            const uint64_t i = t0 + (uint64_t)rd * WAVE + lane;
            if (i < n) atomicAdd(&hist[wave][(keys[i] >> shift) & (RS_BINS - 1)], 1u);
        }
        __builtin_amdgcn_wave_barrier();
        for (int b = lane; b < RS_BINS; b += WAVE) table[(uint64_t)b * n_tiles + tile] = hist[wave][b];
    }
}

__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                               uint64_t n, int shift, const uint32_t *__restrict__ table,
                                                               uint32_t n_tiles, int rounds, uint64_t *__restrict__ keys_out,
                                                               uint32_t *__restrict__ vals_out)
{
    __shared__ uint32_t offs[RS_WAVES][RS_BINS];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const uint64_t tile = (uint64_t)blockIdx.x * RS_WAVES + wave;
    if (tile >= n_tiles) return;
    for (int b = lane; b < RS_BINS; b += WAVE) offs[wave][b] = table[(uint64_t)b * n_tiles + tile];
    __builtin_amdgcn_wave_barrier();
    const uint64_t t0 = tile * (uint64_t)rounds * WAVE;
    const uint64_t lt = lanemask_lt();
    for (int rd = 0; rd < rounds; rd++) {
        const uint64_t i = t0 + (uint64_t)rd * WAVE + lane;
        const bool valid = i < n;
        uint64_t key = 0; uint32_t val = 0;
        if (valid) { key = keys_in[i]; val = vals_in[i]; }
        const uint32_t d = (uint32_t)(key >> shift) & (uint32_t)(RS_BINS - 1);
        uint64_t mask = __ballot(valid);
        if (mask == 0) break;
#pragma unroll
        for (int b = 0; b < RS_BITS; b++) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            mask &= bit ? bal : ~bal;
        }
        // mask = valid lanes of this round with my digit
        uint32_t base = 0;
        if (valid) base = offs[wave][d];
        __builtin_amdgcn_wave_barrier();
        const uint32_t rank = (uint32_t)__popcll(mask & lt);
        if (valid && rank == 0) offs[wave][d] = base + (uint32_t)__popcll(mask);   // leader advances the digit's cursor
        __builtin_amdgcn_wave_barrier();
        if (valid) { keys_out[base + rank] = key; vals_out[base + rank] = val; }
    }
}


// ---- the same passes in ONE launch each ("onesweep": Adinets & Merrill) ---------------------------------------------------------------
// The three launches of a pass exist to turn per-tile digit counts into global offsets. A digit's base (how many keys have a smaller
// digit) does not depend on the order of the keys: ONE kernel counts all passes' digits up front. What remains — how many keys of MY
// digit sit in the tiles before mine — a workgroup gets by looking back over its predecessors' published counts: a status word per
// (tile, digit) carries a 2-bit state (nothing yet / this tile's own count / the inclusive count up to this tile) and a 30-bit
// value in ONE 32-bit store, so no fence is needed; thread d walks digit d's column backwards until it meets an inclusive count.
// Tiles are handed out by an atomic counter in the order the workgroups START, so every predecessor of a running workgroup is itself
// running (or done): the look-back always terminates. (Belt and braces: a poll count bounds every wait — a pass can produce garbage,
// flagged in `err`, but never hang the device.) A workgroup = 4 waves = 4 consecutive wave tiles of the three-launch form, walked the
// same way (round-major, 8-ballot match-any ranks, per-wave cursors in LDS): the output is the same stable order, bit for bit.
// 24 contigs' split order: 33 passes = 99 launches (and a one-workgroup table scan of ~78 us beside the big kernels each) -> 33 + 7.
constexpr uint32_t OS_AGG = 1u << 30, OS_INC = 2u << 30, OS_VAL = (1u << 30) - 1;
constexpr int OS_MAX_PASSES = 8;
constexpr uint32_t OS_SPIN_LIMIT = 1u << 24;
constexpr int OS_LOOK = 8;

// (n_dev: the key count lives in device memory — a sort queued before the kernel that counts its keys has run; n is then only its bound)
__global__ __launch_bounds__(RS_THREADS) void os_hist_kernel(const uint64_t *__restrict__ keys, uint64_t n, const uint32_t *__restrict__ n_dev, int passes,
                                                            uint32_t *__restrict__ ghist)
{
    __shared__ uint32_t h[OS_MAX_PASSES][RS_BINS];
    if (n_dev) n = *n_dev;
    for (int p = 0; p < passes; p++) h[p][threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * RS_THREADS + threadIdx.x; i < n; i += (uint64_t)gridDim.x * RS_THREADS) {
        const uint64_t k = keys[i];
        for (int p = 0; p < passes; p++) atomicAdd(&h[p][(uint32_t)(k >> (p * RS_BITS)) & (RS_BINS - 1)], 1u);
    }
    __syncthreads();
    for (int p = 0; p < passes; p++) {
        const uint32_t v = h[p][threadIdx.x];
        if (v) atomicAdd(&ghist[p * RS_BINS + threadIdx.x], v);
    }
}

__global__ __launch_bounds__(RS_THREADS) void os_pass_kernel(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, uint64_t n, int shift,
                                                            const uint32_t *__restrict__ n_dev, const uint32_t *__restrict__ ghist, uint32_t *status,
                                                            uint32_t *counter, uint32_t *err, uint32_t n_tiles, int rounds, uint64_t *__restrict__ keys_out,
                                                            uint32_t *__restrict__ vals_out)
{
    static_assert(RS_THREADS == RS_BINS, "one thread per digit");
    if (n_dev) {                                           // every workgroup derives the same tiling from the same count
        n = *n_dev;
        rounds = rs_rounds_for(n);
        n_tiles = (uint32_t)((n + (uint64_t)rounds * WAVE - 1) / ((uint64_t)rounds * WAVE));
    }
    __shared__ uint32_t offs[RS_WAVES][RS_BINS];
    __shared__ uint32_t wsum[RS_WAVES];
    __shared__ uint32_t s_tile;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_tile = atomicAdd(counter, 1u);
    for (int b = lane; b < RS_BINS; b += WAVE) offs[wave][b] = 0;
    __syncthreads();
    const uint32_t wg_tile = s_tile;
    if ((uint64_t)wg_tile * RS_WAVES >= n_tiles) return;                 // (a grid sized for the bound: nobody looks back at a tile without keys)
    const uint64_t tile = (uint64_t)wg_tile * RS_WAVES + wave;
    const uint64_t t0 = tile * (uint64_t)rounds * WAVE;
    // The wave's whole tile goes into registers with ONE batch of loads: beside the big kernels a global load takes several microseconds,
    // and a round that waits for its own (twice: count, then scatter) made a 32-round pass ~150 us.
    uint64_t key[RS_ROUNDS];
    uint32_t val[RS_ROUNDS];
    const uint64_t n_here = tile < n_tiles ? n : 0;
#pragma unroll
    for (int rd = 0; rd < RS_ROUNDS; rd++) {
        const uint64_t i = t0 + (uint64_t)rd * WAVE + lane;
        const bool ok = rd < rounds && i < n_here;
        key[rd] = ok ? keys_in[i] : 0ull;
        val[rd] = ok ? vals_in[i] : 0u;
    }
#pragma unroll
    for (int rd = 0; rd < RS_ROUNDS; rd++) {
        const uint64_t i = t0 + (uint64_t)rd * WAVE + lane;
        if (rd < rounds && i < n_here) atomicAdd(&offs[wave][(uint32_t)(key[rd] >> shift) & (RS_BINS - 1)], 1u);
    }
    __syncthreads();
    {
        const uint32_t d = threadIdx.x;                                    // this thread's digit
        uint32_t c[RS_WAVES], tot = 0;
        for (int w = 0; w < RS_WAVES; w++) { c[w] = offs[w][d]; tot += c[w]; }
        uint32_t *st = status + (uint64_t)wg_tile * RS_BINS + d;
        __hip_atomic_store(st, (wg_tile == 0 ? OS_INC : OS_AGG) | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // keys with a smaller digit: exclusive sum over the 256 digit totals
        const uint32_t g = ghist[d];
        const uint32_t incl = wave_incl_sum_dpp(g);
        if (lane == WAVE - 1) wsum[wave] = incl;
        __syncthreads();
        uint32_t base = incl - g;
        for (int w = 0; w < wave; w++) base += wsum[w];
        // keys of this digit in the tiles before this one
        // (OS_LOOK predecessors per step, their loads in flight together: when every tile of a pass starts at once the walk is as long as
        // the tile's index, and one dependent global load per predecessor made a 300-tile pass ~250 us)
        uint32_t excl = 0;
        if (wg_tile > 0) {
            int64_t j = (int64_t)wg_tile - 1;
            uint32_t spins = 0;
            bool done = false;
            while (!done && j >= 0) {
                uint32_t v[OS_LOOK];
#pragma unroll
                for (int k = 0; k < OS_LOOK; k++)
                    v[k] = j - k >= 0 ? __hip_atomic_load(status + (uint64_t)(j - k) * RS_BINS + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : OS_INC;
                int used = 0;
#pragma unroll
                for (int k = 0; k < OS_LOOK; k++) {
                    if (done || used < k) continue;                   // (stopped at an earlier entry of this batch)
                    const uint32_t f = v[k] >> 30;
                    if (f == 0) continue;                              // not published yet: poll again from here
                    excl += v[k] & OS_VAL;
                    used = k + 1;
                    if (f == 2) done = true;
                }
                j -= used;
                if (used == 0) {
                    if (++spins > OS_SPIN_LIMIT) { *err = 1u; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __hip_atomic_store(st, OS_INC | ((excl + tot) & OS_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t o = base + excl;
        for (int w = 0; w < RS_WAVES; w++) { offs[w][d] = o; o += c[w]; }
    }
    __syncthreads();
    if (tile >= n_tiles) return;
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int rd = 0; rd < RS_ROUNDS; rd++) {
        const uint64_t i = t0 + (uint64_t)rd * WAVE + lane;
        const bool valid = rd < rounds && i < n;
        const uint32_t d = (uint32_t)(key[rd] >> shift) & (uint32_t)(RS_BINS - 1);
        uint64_t mask = __ballot(valid);
        if (mask != 0) {
#pragma unroll
            for (int b = 0; b < RS_BITS; b++) {
                const bool bit = (d >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                mask &= bit ? bal : ~bal;
            }
            uint32_t base = 0;
            if (valid) base = offs[wave][d];
            __builtin_amdgcn_wave_barrier();
            const uint32_t rank = (uint32_t)__popcll(mask & lt);
            if (valid && rank == 0) offs[wave][d] = base + (uint32_t)__popcll(mask);
            __builtin_amdgcn_wave_barrier();
            if (valid) { keys_out[base + rank] = key[rd]; vals_out[base + rank] = val[rd]; }
        }
    }
}

static uint64_t os_wg_tiles_max(uint64_t n)       // workgroup tiles of the largest table a sort of n keys can need (smallest tile above 128 wave tiles)
{
    const uint64_t big = (n + (uint64_t)RS_ROUNDS * WAVE * RS_WAVES - 1) / ((uint64_t)RS_ROUNDS * WAVE * RS_WAVES);
    return std::max<uint64_t>(big, 64) + 1;
}
static size_t os_tmp_bytes(uint64_t n) { return 256 + align_up((size_t)OS_MAX_PASSES * RS_BINS * 4, 256) + (size_t)OS_MAX_PASSES * os_wg_tiles_max(n) * RS_BINS * 4; }

static bool onesweep_on()
{
    const char *e = getenv("CSV_SORT_ONESWEEP");                                  // "0": the three-launch passes (A/B, tests)
    return !(e && *e == '0');
}


// [0, 256): tile counters (one per pass) and the error flag; digit totals of every pass; status words of every pass
static void launch_onesweep(hipStream_t s, uint64_t *ki, uint32_t *vi, uint64_t *ko, uint32_t *vo, uint64_t n, const uint32_t *n_dev, int passes, unsigned grid,
                            uint32_t n_tiles, int rounds, void *tmp)
{
    uint32_t *counters = (uint32_t *)tmp, *err = counters + OS_MAX_PASSES;
    uint32_t *ghist = (uint32_t *)((char *)tmp + 256);
    uint32_t *status = (uint32_t *)((char *)ghist + align_up((size_t)OS_MAX_PASSES * RS_BINS * 4, 256));
    const size_t used = 256 + align_up((size_t)OS_MAX_PASSES * RS_BINS * 4, 256) + (size_t)passes * grid * RS_BINS * 4;
    (void)hipMemsetAsync(tmp, 0, used, s);
    const unsigned hgrid = (unsigned)std::min<uint64_t>((n + 8 * RS_THREADS - 1) / (8 * RS_THREADS), 1024);
    hipLaunchKernelGGL(os_hist_kernel, dim3(hgrid), dim3(RS_THREADS), 0, s, ki, n, n_dev, passes, ghist);
    for (int p = 0; p < passes; p++) {
        hipLaunchKernelGGL(os_pass_kernel, dim3(grid), dim3(RS_THREADS), 0, s, ki, vi, n, p * RS_BITS, n_dev, ghist + p * RS_BINS, status + (size_t)p * grid * RS_BINS,
                           counters + p, err, n_tiles, rounds, ko, vo);
        uint64_t *tk = ki; ki = ko; ko = tk;
        uint32_t *tv = vi; vi = vo; vo = tv;
    }
}

// The same sort for a key count that is still being computed on the device when the sort is queued: n_bound sizes the grid and the
// workspace (radix_sort_tmp_bytes(n_bound)), *n_dev (<= n_bound, read by the kernels) is the count. Returns where the result ends up
// (as launch_radix_sort_u64) — the passes always run, also for a count of 0 or 1 — or -1 when n_bound is beyond the look-back's 30-bit counts.
int launch_radix_sort_u64_devn(hipStream_t s, uint64_t *keys_in, uint32_t *vals_in, uint64_t *keys_out, uint32_t *vals_out, uint64_t n_bound,
                               const uint32_t *n_dev, int key_bits, void *tmp)
{
    const int passes = (key_bits + RS_BITS - 1) / RS_BITS;
    if (n_bound == 0 || passes <= 0) return 0;
    if (n_bound >= (1ull << 30) || passes > OS_MAX_PASSES) return -1;
    const unsigned grid = (unsigned)(os_wg_tiles_max(n_bound) - 1);
    launch_onesweep(s, keys_in, vals_in, keys_out, vals_out, n_bound, n_dev, passes, grid, 0, 0, tmp);
    return (passes & 1) ? 1 : 0;
}

size_t radix_sort_tmp_bytes(uint64_t n)
{
    const uint64_t n_tiles = (n + 2 * WAVE - 1) / (2 * WAVE);         // upper bound (smallest tile)
    const uint64_t tab = (uint64_t)RS_BINS * (n_tiles ? n_tiles : 1);
    return std::max(align_up(tab * sizeof(uint32_t), 256) + exclusive_sum_tmp_bytes(tab), os_tmp_bytes(n));
}

int launch_radix_sort_u64(hipStream_t s, uint64_t *keys_in, uint32_t *vals_in, uint64_t *keys_out, uint32_t *vals_out,
                          uint64_t n, int key_bits, void *tmp)
{
    if (n <= 1 || key_bits <= 0) return 0;
    const int rounds = rs_rounds_for(n);
    const uint64_t tile_keys = (uint64_t)rounds * WAVE;
    const uint32_t n_tiles = (uint32_t)((n + tile_keys - 1) / tile_keys);
    const uint64_t tab = (uint64_t)RS_BINS * n_tiles;
    uint32_t *table = (uint32_t *)tmp;
    void *es_tmp = (char *)tmp + align_up(tab * sizeof(uint32_t), 256);
    const unsigned grid = (n_tiles + RS_WAVES - 1) / RS_WAVES;
    const int passes = (key_bits + RS_BITS - 1) / RS_BITS;
    uint64_t *ki = keys_in, *ko = keys_out;
    uint32_t *vi = vals_in, *vo = vals_out;
    if (onesweep_on() && n < (1ull << 30) && passes <= OS_MAX_PASSES && grid + 1 <= os_wg_tiles_max(n)) {
        launch_onesweep(s, ki, vi, ko, vo, n, nullptr, passes, grid, n_tiles, rounds, tmp);
        return (passes & 1) ? 1 : 0;
    }
    for (int p = 0; p < passes; p++) {
        const int shift = p * RS_BITS;
        hipLaunchKernelGGL(rs_hist_kernel, dim3(grid), dim3(RS_THREADS), 0, s, ki, n, shift, table, n_tiles, rounds);
        launch_exclusive_sum_u32(s, table, tab, es_tmp);
        hipLaunchKernelGGL(rs_scatter_kernel, dim3(grid), dim3(RS_THREADS), 0, s, ki, vi, n, shift, table, n_tiles, rounds, ko, vo);
        uint64_t *tk = ki; ki = ko; ko = tk;
        uint32_t *tv = vi; vi = vo; vo = tv;
    }
    return (passes & 1) ? 1 : 0;
}

// ------------------------------------------------------------------------------- bucket ordering (the common case)
// A chromosome yields 1e4..1e6 signatures whose starts spread over the contig, so one most-significant-digit split into
// BK_N buckets leaves a few dozen records per bucket; a wave then ranks its bucket with the full comparator
// ((type, start) asc, end asc, (read, query offset) desc = the reference's lower_bound insertion order) and writes the final
// records. The bucket counts come with the scan (its epilogue counts what it emits); three launches — offsets, scatter, rank —
// instead of the fourteen of the LSD radix path (which stays as the fallback for skewed input: a
// bucket above BK_LOCAL_MAX, or starts that overflow the key).
// one workgroup: the scan's bucket counts -> exclusive offsets (in place) + scatter cursors
__global__ void __launch_bounds__(1024) bk_prep_kernel(uint32_t *__restrict__ hist, uint32_t *__restrict__ cur)
{
    constexpr int PER = BK_N / 1024;
    __shared__ uint32_t wsum[16];
    uint32_t v[PER], tot = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) { v[k] = hist[threadIdx.x * PER + k]; tot += v[k]; }
    const uint32_t incl = wave_incl_sum_dpp(tot);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int k = 0; k < w; k++) base += wsum[k];
    uint32_t run = base + incl - tot;
#pragma unroll
    for (int k = 0; k < PER; k++) { hist[threadIdx.x * PER + k] = run; cur[threadIdx.x * PER + k] = run; run += v[k]; }
}

__global__ void bk_scatter_kernel(const csv_sig *__restrict__ sig, uint64_t n, int type_pos, int shift, uint32_t *__restrict__ cur,
                                  csv_sig *__restrict__ tmp)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const csv_sig sg = sig[i];
    tmp[atomicAdd(&cur[bk_bucket(sg, type_pos, shift)], 1u)] = sg;
}

// a precedes b in the reference's vector
__device__ __forceinline__ bool sig_before(const csv_sig &a, uint64_t ka, const csv_sig &b, uint64_t kb)
{
    if (ka != kb) return ka < kb;
    if (a.end != b.end) return a.end < b.end;
    if (a.read != b.read) return a.read > b.read;
    return a.qpos_kind > b.qpos_kind;
}

__global__ void __launch_bounds__(256) bk_local_kernel(const csv_sig *__restrict__ tmp, const uint32_t *__restrict__ off, uint64_t n, int type_pos,
                                                       csv_sig *__restrict__ sig_sorted, uint32_t *__restrict__ start_out, uint32_t *__restrict__ end_out)
{
    __shared__ csv_sig stage[4][64];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const uint32_t b = blockIdx.x * 4 + w;
    const uint32_t base = off[b];
    const uint32_t m = (b + 1 < BK_N ? off[b + 1] : (uint32_t)n) - base;
    for (uint32_t j0 = 0; j0 < m; j0 += 64) {
        const uint32_t j = j0 + l;
        csv_sig me{};
        uint64_t km = 0;
        if (j < m) {
            me = tmp[base + j];
            km = me.start;
            if (type_pos >= 0 && (me.qpos_kind & 3u) != CSV_KIND_DEL) km |= 1ull << type_pos;
        }
        uint32_t rank = 0;
        for (uint32_t c0 = 0; c0 < m; c0 += 64) {
            const uint32_t cn = min(64u, m - c0);
            if (c0 + l < m) stage[w][l] = tmp[base + c0 + l];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // this wave's staging stores before its broadcast reads
            __builtin_amdgcn_wave_barrier();
            for (uint32_t t = 0; t < cn; t++) {
                const csv_sig o = stage[w][t];
                uint64_t ko = o.start;
                if (type_pos >= 0 && (o.qpos_kind & 3u) != CSV_KIND_DEL) ko |= 1ull << type_pos;
                // identical records (never produced by the scan) keep their bucket order so that the result stays a permutation
                const bool same = ko == km && o.end == me.end && o.read == me.read && o.qpos_kind == me.qpos_kind;
                rank += (sig_before(o, ko, me, km) || (same && c0 + t < j)) ? 1u : 0u;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     // reads done before the next chunk overwrites the stage
            __builtin_amdgcn_wave_barrier();
        }
        if (j < m) {
            sig_sorted[base + rank] = me;
            if (start_out) { start_out[base + rank] = me.start; end_out[base + rank] = me.end; }
        }
    }
}

void launch_bucket_sort(hipStream_t s, const csv_sig *sig_raw, uint64_t n, int type_pos, int shift, uint32_t *off, uint32_t *cur,
                        csv_sig *tmp, csv_sig *sig_sorted, uint32_t *start_out, uint32_t *end_out)
{
    if (!n) return;
    hipLaunchKernelGGL(bk_prep_kernel, dim3(1), dim3(1024), 0, s, off, cur);
    hipLaunchKernelGGL(bk_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sig_raw, n, type_pos, shift, cur, tmp);
    hipLaunchKernelGGL(bk_local_kernel, dim3(BK_N / 4), dim3(256), 0, s, tmp, off, n, type_pos, sig_sorted, start_out, end_out);
}

// ------------------------------------------------------------------------------- signature ordering
// key = [type bit | start]; type bit 0 = DEL, 1 = INS so the DEL calls come first (mergeSVs walks DEL, DUP, INV,
// INS, BND — sv_object.cpp:62-68). type_bit_pos < 0: no type bit (the interleaved order of the reference's single
// chr_sv_calls vector). The END is not part of the radix key: calls sharing a start are few (the local coverage at
// most), so (end, reverse insertion) is settled by the rank pass below and the sort needs 2 passes less.
__global__ void sig_make_keys_kernel(const csv_sig *__restrict__ sig, uint64_t n, int len_bits, int type_bit_pos,
                                     uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const csv_sig s = sig[i];
    (void)len_bits;
    uint64_t k = (uint64_t)s.start;
    if (type_bit_pos >= 0 && (s.qpos_kind & 3u) != CSV_KIND_DEL) k |= 1ull << type_bit_pos;
    keys[i] = k;
    vals[i] = (uint32_t)i;
}

void launch_sig_make_keys(hipStream_t s, const csv_sig *sig, uint64_t n, int len_bits, int type_bit_pos,
                          uint64_t *keys, uint32_t *vals)
{
    if (!n) return;
    hipLaunchKernelGGL(sig_make_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sig, n, len_bits, type_bit_pos, keys, vals);
}

// Inside a run of equal (type, start) the reference order is: end ascending, then reverse insertion order — later
// read first, and within one read the later CIGAR op (larger query offset) first. Each thread ranks its element
// inside its run (runs are as long as the local coverage at most) and writes the final record.
__global__ void sig_fix_ties_gather_kernel(const csv_sig *__restrict__ sig_raw, const uint64_t *__restrict__ keys,
                                           const uint32_t *__restrict__ vals, uint64_t n, csv_sig *__restrict__ sig_sorted,
                                           uint32_t *__restrict__ start_out, uint32_t *__restrict__ end_out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = keys[i];
    const csv_sig me = sig_raw[vals[i]];
    uint64_t pos = i;
    const bool tie_l = i > 0 && keys[i - 1] == k, tie_r = i + 1 < n && keys[i + 1] == k;
    if (tie_l || tie_r) {
        uint64_t a = i, b = i + 1;
        while (a > 0 && keys[a - 1] == k) a--;
        while (b < n && keys[b] == k) b++;
        const uint64_t mine = ((uint64_t)me.read << 32) | me.qpos_kind;
        uint64_t rank = 0;
        for (uint64_t j = a; j < b; j++) {
            const csv_sig o = sig_raw[vals[j]];
            // o precedes me: smaller end, or equal end and inserted later
            rank += (o.end < me.end) || (o.end == me.end && (((uint64_t)o.read << 32) | o.qpos_kind) > mine);
        }
        pos = a + rank;
    }
    sig_sorted[pos] = me;
    if (start_out) { start_out[pos] = me.start; end_out[pos] = me.end; }
}

void launch_sig_fix_ties_gather(hipStream_t s, const csv_sig *sig_raw, const uint64_t *keys, const uint32_t *vals,
                                uint64_t n, csv_sig *sig_sorted, uint32_t *start_out, uint32_t *end_out)
{
    if (!n) return;
    hipLaunchKernelGGL(sig_fix_ties_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sig_raw, keys, vals, n,
                       sig_sorted, start_out, end_out);
}

// ------------------------------------------------------------------------------- small helpers
__global__ void iota_keys_u32_kernel(const uint32_t *__restrict__ k32, uint64_t n, uint64_t *keys, uint32_t *vals)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = k32[i]; vals[i] = (uint32_t)i; }
}
__global__ void iota_keys_i32_kernel(const int32_t *__restrict__ k32, uint64_t n, uint64_t *keys, uint32_t *vals)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = (uint32_t)k32[i] ^ 0x80000000u; vals[i] = (uint32_t)i; }
}
__global__ void check_sorted_u32_kernel(const uint32_t *__restrict__ k, uint64_t n, unsigned int *flag)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > 0 && i < n && k[i] < k[i - 1]) *flag = 1u;
}
__global__ void gather_u32_kernel(const uint32_t *__restrict__ src, const uint32_t *__restrict__ idx, uint64_t n, uint32_t *dst)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

#define GRID1D(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256)
void launch_iota_keys_u32(hipStream_t s, const uint32_t *k32, uint64_t n, uint64_t *keys, uint32_t *vals)
{ if (n) hipLaunchKernelGGL(iota_keys_u32_kernel, GRID1D(n), 0, s, k32, n, keys, vals); }
void launch_iota_keys_i32(hipStream_t s, const int32_t *k32, uint64_t n, uint64_t *keys, uint32_t *vals)
{ if (n) hipLaunchKernelGGL(iota_keys_i32_kernel, GRID1D(n), 0, s, k32, n, keys, vals); }
void launch_check_sorted_u32(hipStream_t s, const uint32_t *k, uint64_t n, unsigned int *flag)
{ if (n) hipLaunchKernelGGL(check_sorted_u32_kernel, GRID1D(n), 0, s, k, n, flag); }
void launch_gather_u32(hipStream_t s, const uint32_t *src, const uint32_t *idx, uint64_t n, uint32_t *dst)
{ if (n) hipLaunchKernelGGL(gather_u32_kernel, GRID1D(n), 0, s, src, idx, n, dst); }

// dst_k[i] = src_k[idx[i]] for three arrays that share the index list (the split-read pass's ref_end / q_start / q_end of selected records)
__global__ __launch_bounds__(256) void gather3_u32_kernel(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, const uint32_t *__restrict__ c,
                                                          const uint32_t *__restrict__ idx, uint64_t n, uint32_t *__restrict__ da, uint32_t *__restrict__ db,
                                                          uint32_t *__restrict__ dc)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = idx[i];
    da[i] = a[k]; db[i] = b[k]; dc[i] = c[k];
}
void launch_gather3_u32(hipStream_t s, const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *idx, uint64_t n, uint32_t *da, uint32_t *db, uint32_t *dc)
{
    if (n) hipLaunchKernelGGL(gather3_u32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, c, idx, n, da, db, dc);
}

}  // namespace csv
