// dbscan.hip — kernel #3a: order-free windowed DBSCAN on start-sorted intervals (gfx950).
//
// Replaces DBSCAN::fit (dbscan.cpp:9-81). The reference is sequential and visit-order dependent,
// with an O(n) regionQuery per point. Its labels are reproduced exactly by an order-free form
// (checked against the reference's own dbscan.cpp on thousands of random fits, tests/):
//   core(i)   <=> |N(i)| >= minPts                    (N(i) includes i when its length is > 0)
//   clusters   = connected components of core points under the eps-relation
//   start(c)   = the component's core point with the smallest ORIGINAL index
//   id(c)      = rank of start(c) among all start points, by original index
//   core i     -> id of its component
//   non-core b -> if N(b) contains start points: the LARGEST such id (expandCluster overwrites the
//                 labels of the start point's whole neighbourhood, dbscan.cpp:33-35);
//                 else if N(b) contains core points: the SMALLEST id among them; else -2.
// On intervals sorted by start the neighbours of i lie in a window:
//   forward  (s_j >= s_i): s_j <= s_i + eps*len_i            (overlap <= e_i - s_j must reach (1-eps) len_i)
//   backward (s_j <= s_i): s_j >= s_i - eps*len_i/(1-eps)    (len_j <= len_i/(1-eps) and s_i <= s_j + eps*len_j)
// padded by 2 bp because acceptance is decided by the ROUNDED double expression; inside the window
// the reference's exact expression is evaluated (devutil.hpp iv_neighbor). So O(n*w) instead of O(n^2).
//
// Five launches (for one set or for two sets side by side): neighbour count (+ union-find init) -> lock-free union-find over core pairs (agent-scope atomics; larger
// root linked under smaller, so a component's root is its minimum original index) -> roots flagged
// and ranked by an exclusive scan in original-index space -> labels.
// The same machinery, instantiated with the 1-D metric |a-b| <= eps, serves DBSCAN1D segments that
// are too large for the LDS kernel (dbscan1d.hip).
#include "../common.hpp"
#include "../devutil.hpp"

namespace csv {

constexpr uint32_t NONE = 0xffffffffu;

struct IntervalMetric {
    const uint32_t *s, *e;
    double eps;
    __device__ __forceinline__ uint32_t key(uint64_t i) const { return s[i]; }
    __device__ __forceinline__ void window(uint64_t i, uint64_t &lo, uint64_t &hi) const
    {
        const uint32_t si = s[i];
        const int li = (int)(e[i] - si);
        const double l = li > 0 ? (double)li : 0.0;
        const uint64_t wf = (uint64_t)(eps * l) + 2;
        const uint64_t wb = (uint64_t)(eps * l / (1.0 - eps)) + 2;
        hi = (uint64_t)si + wf;
        lo = (uint64_t)si > wb ? (uint64_t)si - wb : 0;
    }
    __device__ __forceinline__ bool nb(uint64_t i, uint64_t j) const { return iv_neighbor(s[i], e[i], s[j], e[j], eps); }
};

struct PointMetric {                   // p = points sorted ascending (int order); keys biased by 2^31 so unsigned order == int order
    const int32_t *p;
    double eps;
    __device__ __forceinline__ uint32_t key(uint64_t i) const { return (uint32_t)p[i] ^ 0x80000000u; }
    __device__ __forceinline__ void window(uint64_t i, uint64_t &lo, uint64_t &hi) const
    {
        const uint64_t w = (uint64_t)eps + 1, k = key(i);
        hi = k + w;
        lo = k > w ? k - w : 0;
    }
    __device__ __forceinline__ bool nb(uint64_t i, uint64_t j) const
    {   // dbscan1d.cpp:68-70: std::abs(int - int) converted to double, <= epsilon
        return (double)abs(p[i] - p[j]) <= eps;
    }
};

// Two independent point sets can share every launch: positions [0, split) are one set, [split, n) the other (the DEL
// and INS calls of a chromosome). A window never leaves its own set; cluster ids restart at 0 in the second set.
template <class M>
__global__ void db_count_kernel(M m, uint64_t n, uint64_t split, int min_pts_imm, const int *__restrict__ d_min_pts,
                                const uint32_t *__restrict__ oid, uint8_t *__restrict__ core, uint32_t *__restrict__ parent,
                                uint32_t *__restrict__ is_root)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == n) is_root[n] = 0;
    if (i >= n) return;
    const uint32_t me = oid ? oid[i] : (uint32_t)i;
    parent[me] = me;                                   // union-find + root flags start here (no separate init launch)
    is_root[me] = 0;
    const int min_pts = d_min_pts ? *d_min_pts : min_pts_imm;
    const uint64_t s0 = i < split ? 0 : split, s1 = i < split ? split : n;
    uint64_t lo, hi;
    m.window(i, lo, hi);
    int cnt = 0;
    for (uint64_t j = i; j < s1 && (uint64_t)m.key(j) <= hi; j++) cnt += m.nb(i, j);
    for (uint64_t j = i; j-- > s0 && (uint64_t)m.key(j) >= lo;) cnt += m.nb(i, j);
    core[i] = cnt >= min_pts;
}

__device__ __forceinline__ uint32_t uf_load(uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t x)
{
    for (;;) {
        const uint32_t p = uf_load(&parent[x]);
        if (p == x) return x;
        const uint32_t gp = uf_load(&parent[p]);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // path halving
        x = p;
    }
}

__device__ __forceinline__ void uf_union(uint32_t *parent, uint32_t a, uint32_t b)
{
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a > b) { const uint32_t t = a; a = b; b = t; }
        // link the larger root b under the smaller root a
        if (atomicCAS(&parent[b], b, a) == b) return;
    }
}

template <class M>
__global__ void db_union_kernel(M m, uint64_t n, uint64_t split, const uint8_t *__restrict__ core, const uint32_t *__restrict__ oid, uint32_t *parent)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !core[i]) return;
    uint64_t lo, hi;
    m.window(i, lo, hi);
    const uint64_t s1 = i < split ? split : n;
    const uint32_t me = oid ? oid[i] : (uint32_t)i;
    for (uint64_t j = i + 1; j < s1 && (uint64_t)m.key(j) <= hi; j++)
        if (core[j] && m.nb(i, j)) uf_union(parent, me, oid ? oid[j] : (uint32_t)j);
}

__global__ void db_roots_kernel(uint64_t n, const uint8_t *__restrict__ core, const uint32_t *__restrict__ oid,
                                uint32_t *parent, uint32_t *__restrict__ root_of, uint32_t *__restrict__ is_root)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!core[i]) { root_of[i] = NONE; return; }
    const uint32_t me = oid ? oid[i] : (uint32_t)i;
    uint32_t x = me;
    for (;;) { const uint32_t p = parent[x]; if (p == x) break; x = p; }   // union kernel finished: plain loads
    root_of[i] = x;
    if (x == me) is_root[me] = 1;
}

template <class M>
__global__ void db_label_kernel(M m, uint64_t n, uint64_t split, const uint32_t *__restrict__ oid, const uint32_t *__restrict__ root_of,
                                const uint32_t *__restrict__ cid_raw, int32_t *__restrict__ labels)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t me = oid ? oid[i] : (uint32_t)i;
    const uint32_t r = root_of[i];
    // ids of the second set restart at 0: subtract the number of roots of the first set (only used with oid == nullptr,
    // where original index == position, so cid_raw[split] is that count)
    const uint32_t id0 = (i >= split && split < n) ? cid_raw[split] : 0u;
    const uint64_t s0 = i < split ? 0 : split, s1 = i < split ? split : n;
    struct { const uint32_t *p; uint32_t off; __device__ uint32_t operator[](uint32_t k) const { return p[k] - off; } } cid{cid_raw, id0};
    if (r != NONE) { labels[me] = (int32_t)cid[r]; return; }
    uint64_t lo, hi;
    m.window(i, lo, hi);
    int32_t max_start = -1, min_core = INT32_MAX;
    for (uint64_t j = i + 1; j < s1 && (uint64_t)m.key(j) <= hi; j++) {
        const uint32_t rj = root_of[j];
        if (rj != NONE && m.nb(i, j)) {
            const int32_t c = (int32_t)cid[rj];
            if (rj == (oid ? oid[j] : (uint32_t)j)) max_start = max(max_start, c); else min_core = min(min_core, c);
        }
    }
    for (uint64_t j = i; j-- > s0 && (uint64_t)m.key(j) >= lo;) {
        const uint32_t rj = root_of[j];
        if (rj != NONE && m.nb(i, j)) {
            const int32_t c = (int32_t)cid[rj];
            if (rj == (oid ? oid[j] : (uint32_t)j)) max_start = max(max_start, c); else min_core = min(min_core, c);
        }
    }
    labels[me] = max_start >= 0 ? max_start : (min_core != INT32_MAX ? min_core : -2);
}

// tmp: core u8[n] | parent u32[n] | root_of u32[n] | is_root/cid u32[n+1] | scan tmp
size_t dbscan_tmp_bytes(uint64_t n)
{
    return align_up(n, 256) + 2 * align_up(n * 4, 256) + align_up((n + 1) * 4, 256) + exclusive_sum_tmp_bytes(n + 1);
}

template <class M>
static void run_dbscan(hipStream_t s, M m, const uint32_t *oid, uint64_t n, uint64_t split, int min_pts, const int *d_min_pts,
                       int32_t *labels, void *tmp)
{
    if (n == 0) return;
    char *p = (char *)tmp;
    uint8_t *core = (uint8_t *)p;      p += align_up(n, 256);
    uint32_t *parent = (uint32_t *)p;  p += align_up(n * 4, 256);
    uint32_t *root_of = (uint32_t *)p; p += align_up(n * 4, 256);
    uint32_t *cid = (uint32_t *)p;     p += align_up((n + 1) * 4, 256);
    void *es_tmp = p;
    const dim3 grid((unsigned)((n + 255) / 256)), grid1((unsigned)((n + 1 + 255) / 256)), blk(256);
    hipLaunchKernelGGL(db_count_kernel<M>, grid1, blk, 0, s, m, n, split, min_pts, d_min_pts, oid, core, parent, cid);
    hipLaunchKernelGGL(db_union_kernel<M>, grid, blk, 0, s, m, n, split, core, oid, parent);
    hipLaunchKernelGGL(db_roots_kernel, grid, blk, 0, s, n, core, oid, parent, root_of, cid);
    launch_exclusive_sum_u32(s, cid, n + 1, es_tmp);
    hipLaunchKernelGGL(db_label_kernel<M>, grid, blk, 0, s, m, n, split, oid, root_of, cid, labels);
}

void launch_dbscan_iv_sorted(hipStream_t s, const uint32_t *start, const uint32_t *end, const uint32_t *oid,
                             uint64_t n, uint64_t split, double eps, int min_pts, const int *d_min_pts, int32_t *labels, void *tmp)
{
    IntervalMetric m{start, end, eps};
    run_dbscan(s, m, oid, n, oid ? n : split, min_pts, d_min_pts, labels, tmp);
}

size_t dbscan1d_big_tmp_bytes(uint64_t n) { return dbscan_tmp_bytes(n); }

void launch_dbscan_1d_big(hipStream_t s, const int32_t *pts_sorted, const uint32_t *oid, uint64_t n, double eps,
                          int min_pts, int32_t *labels, void *tmp)
{
    PointMetric m{pts_sorted, eps};
    run_dbscan(s, m, oid, n, n, min_pts, nullptr, labels, tmp);
}

}  // namespace csv
