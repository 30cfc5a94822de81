// dbscan1d.hip — kernel #3b: batched 1-D DBSCAN, one wavefront per point set (gfx950).
//
// Replaces DBSCAN1D::fit (dbscan1d.cpp:8-70). The reference runs it six times per overlap group of
// split reads on vectors of 5-200 ints (sv_caller.cpp:270-372): tiny sets, huge call count. Here
// every set of a batch is one segment and one 64-lane wave solves it entirely in its private LDS
// slice (no workgroup barrier), with the same order-free labelling as dbscan.hip:
//   1. neighbour counts and the sorted rank of every point by brute force (n <= 512: each lane
//      owns points lane, lane+64, ... and reads the others as LDS broadcasts);
//   2. in sorted order the components of core points are the runs whose consecutive gaps are
//      <= eps (1-D: two cores within eps have every core between them within eps of both), found
//      with two wave max-scans; LDS atomicMin gives each run its smallest ORIGINAL index (= start);
//   3. start flags are prefix-summed in original-index order to the reference's cluster ids;
//   4. cores take their run's id; borders scan their neighbours: largest id among neighbouring
//      start points, else smallest id among neighbouring cores, else -2 (see dbscan.hip).
// Segments longer than DBSCAN1D_MAX_SEG are flagged and solved by the generic sorted-window path.
#include "../common.hpp"
#include "../devutil.hpp"

namespace csv {

constexpr int D1_THREADS = 256;
constexpr int D1_WAVES = D1_THREADS / WAVE;
constexpr int D1_MAX = (int)DBSCAN1D_MAX_SEG;      // 512
constexpr uint32_t D1_NONE = 0xffffffffu;

struct D1Lds {
    int32_t  p[D1_MAX];        // points, original order
    uint32_t rank[D1_MAX];     // sorted position of original index i
    uint32_t sidx[D1_MAX];     // original index at sorted position k
    uint32_t core[D1_MAX];     // by original index
    uint32_t comp[D1_MAX];     // by sorted position: sorted position of the run head (cores only)
    uint32_t rootmin[D1_MAX];  // by run head position: smallest original index in the run
    uint32_t cid[D1_MAX + 1];  // by original index: start flag -> exclusive prefix sum
};

__global__ __launch_bounds__(D1_THREADS) void dbscan1d_kernel(const int32_t *__restrict__ pts, const uint64_t *__restrict__ seg_off,
                                                             uint64_t n_seg, double eps, int min_pts,
                                                             int32_t *__restrict__ labels, unsigned int *too_large)
{
    __shared__ D1Lds lds_all[D1_WAVES];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    D1Lds &L = lds_all[wave];
    const uint64_t wave_gid = (uint64_t)blockIdx.x * D1_WAVES + wave;
    const uint64_t wave_stride = (uint64_t)gridDim.x * D1_WAVES;

    for (uint64_t seg = wave_gid; seg < n_seg; seg += wave_stride) {
        const uint64_t o0 = seg_off[seg], o1 = seg_off[seg + 1];
        const uint64_t n64 = o1 - o0;
        if (n64 == 0) continue;
        if (n64 > (uint64_t)D1_MAX) { if (lane == 0) *too_large = 1u; continue; }
        const int n = (int)n64;
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < n; i += WAVE) { L.p[i] = pts[o0 + i]; L.rootmin[i] = D1_NONE; L.cid[i] = 0; }
        if (lane == 0) L.cid[n] = 0;
        __builtin_amdgcn_wave_barrier();

        // 1. neighbour count + sorted rank (stable by original index)
        for (int i = lane; i < n; i += WAVE) {
            const int32_t pi = L.p[i];
            int cnt = 0; uint32_t rk = 0;
            for (int j = 0; j < n; j++) {
                const int32_t pj = L.p[j];
                cnt += ((double)abs(pi - pj) <= eps);              // dbscan1d.cpp:68-70
                rk += (pj < pi) || (pj == pi && j < i);
            }
            L.core[i] = cnt >= min_pts;
            L.rank[i] = rk;
            L.sidx[rk] = (uint32_t)i;
        }
        __builtin_amdgcn_wave_barrier();

        // 2. runs of core points in sorted order
        int32_t carry_prev = -1;      // sorted position of the last core seen so far
        int32_t carry_head = -1;      // sorted position of the current run head
        for (int k0 = 0; k0 < n; k0 += WAVE) {
            const int k = k0 + lane;
            const bool in = k < n;
            const uint32_t oi = in ? L.sidx[k] : 0u;
            const bool is_core = in && L.core[oi];
            // previous core position (exclusive max-scan of core positions)
            const int32_t incl_c = wave_incl_max(is_core ? k : -1);
            int32_t prev = __shfl_up(incl_c, 1, 64);
            if (lane == 0) prev = -1;
            prev = max(prev, carry_prev);
            bool head = false;
            if (is_core) {
                head = prev < 0 || !((double)abs(L.p[oi] - L.p[L.sidx[prev]]) <= eps);
            }
            const int32_t incl_h = max(wave_incl_max(head ? k : -1), carry_head);
            if (is_core) {
                L.comp[k] = (uint32_t)incl_h;
                atomicMin(&L.rootmin[incl_h], oi);
            }
            carry_prev = max(carry_prev, __shfl(incl_c, 63, 64));
            carry_head = __shfl(incl_h, 63, 64);
        }
        __builtin_amdgcn_wave_barrier();

        // 3. start points ranked by original index
        for (int i = lane; i < n; i += WAVE)
            if (L.core[i] && L.rootmin[L.comp[L.rank[i]]] == (uint32_t)i) L.cid[i] = 1;
        __builtin_amdgcn_wave_barrier();
        uint32_t carry = 0;
        for (int i0 = 0; i0 <= n; i0 += WAVE) {
            const int i = i0 + lane;
            const uint32_t v = i <= n ? L.cid[i] : 0u;
            const uint32_t incl = wave_incl_sum(v);
            if (i <= n) L.cid[i] = carry + incl - v;
            carry += __shfl(incl, 63, 64);
        }
        __builtin_amdgcn_wave_barrier();

        // 4. labels
        for (int i = lane; i < n; i += WAVE) {
            int32_t lab;
            if (L.core[i]) {
                lab = (int32_t)L.cid[L.rootmin[L.comp[L.rank[i]]]];
            } else {
                const int32_t pi = L.p[i];
                int32_t max_start = -1, min_core = INT32_MAX;
                for (int j = 0; j < n; j++) {
                    if (L.core[j] && ((double)abs(pi - L.p[j]) <= eps)) {
                        const uint32_t rj = L.rootmin[L.comp[L.rank[j]]];
                        const int32_t c = (int32_t)L.cid[rj];
                        if (rj == (uint32_t)j) max_start = max(max_start, c); else min_core = min(min_core, c);
                    }
                }
                lab = max_start >= 0 ? max_start : (min_core != INT32_MAX ? min_core : -2);
            }
            labels[o0 + i] = lab;
        }
    }
}

void launch_dbscan_1d_batched(hipStream_t s, const int32_t *pts, const uint64_t *seg_off, uint64_t n_seg,
                              double eps, int min_pts, int32_t *labels, unsigned int *too_large_flag)
{
    if (n_seg == 0) return;
    uint64_t want = (n_seg + D1_WAVES - 1) / D1_WAVES;
    if (want > 4096) want = 4096;
    hipLaunchKernelGGL(dbscan1d_kernel, dim3((unsigned)want), dim3(D1_THREADS), 0, s, pts, seg_off, n_seg, eps, min_pts,
                       labels, too_large_flag);
}

}  // namespace csv
