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
// Five launches (for one set or for two sets side by side): neighbour count (+ union-find init) -> union-find over core
// pairs inside 256-point tiles in LDS -> the pairs that cross a tile border through a lock-free global union-find (agent-scope
// atomics); in both the larger root is linked under the smaller, so a component's root is its minimum original index ->
// roots flagged and ranked in original-index space (per-tile ranks + a last-workgroup scan of the tile totals when positions
// are original indices, else an exclusive scan) -> labels.
// The same machinery, instantiated with the 1-D metric |a-b| <= eps, serves DBSCAN1D segments that
// are too large for the LDS kernel (dbscan1d.hip).
#include "../common.hpp"
#include "../devutil.hpp"

namespace csv {

constexpr uint32_t NONE = 0xffffffffu;

// A metric names the per-point record (Elem), how to fetch it, its sort key, the key window that can hold neighbours, and the
// reference's neighbour predicate. Kernels stage the Elems of a tile (plus a halo) in LDS: the window loops are chains of
// dependent loads, and an LDS hit costs a tenth of an L2 hit.
struct IntervalMetric {
    struct Elem { uint32_t s, e; };
    const uint32_t *s, *e;
    double eps;
    __device__ __forceinline__ Elem load(uint64_t i) const { return Elem{s[i], e[i]}; }
    __device__ __forceinline__ static uint32_t key(const Elem &a) { return a.s; }
    __device__ __forceinline__ void window(const Elem &a, uint64_t &lo, uint64_t &hi) const
    {
        const int li = (int)(a.e - a.s);
        const double l = li > 0 ? (double)li : 0.0;
        const uint64_t wf = (uint64_t)(eps * l) + 2;
        const uint64_t wb = (uint64_t)(eps * l / (1.0 - eps)) + 2;
        hi = (uint64_t)a.s + wf;
        lo = (uint64_t)a.s > wb ? (uint64_t)a.s - wb : 0;
    }
    __device__ __forceinline__ bool nb(const Elem &a, const Elem &b) const { return iv_neighbor(a.s, a.e, b.s, b.e, eps); }
};

struct PointMetric {                   // p = points sorted ascending (int order); keys biased by 2^31 so unsigned order == int order
    struct Elem { int32_t p; };
    const int32_t *p;
    double eps;
    __device__ __forceinline__ Elem load(uint64_t i) const { return Elem{p[i]}; }
    __device__ __forceinline__ static uint32_t key(const Elem &a) { return (uint32_t)a.p ^ 0x80000000u; }
    __device__ __forceinline__ void window(const Elem &a, uint64_t &lo, uint64_t &hi) const
    {
        const uint64_t w = (uint64_t)eps + 1, k = key(a);
        hi = k + w;
        lo = k > w ? k - w : 0;
    }
    __device__ __forceinline__ bool nb(const Elem &a, const Elem &b) const
    {   // dbscan1d.cpp:68-70: std::abs(int - int) converted to double, <= epsilon
        return (double)abs(a.p - b.p) <= eps;
    }
};

constexpr int UF_TILE = 256;           // points per workgroup
constexpr int DB_HALO = 128;           // staged on each side of the tile; windows that reach further read global memory
constexpr int DB_G = 16;               // lanes that share one point's window (a DPP row)
constexpr int DB_THREADS = 1024;       // UF_TILE points x DB_G lanes / 4 points per lane group
constexpr int DB_GROUPS = DB_THREADS / DB_G;

__device__ __forceinline__ int group_sum(int v)
{
#pragma unroll
    for (int d = DB_G / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, DB_G);
    return v;
}
__device__ __forceinline__ int group_max(int v)
{
#pragma unroll
    for (int d = DB_G / 2; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d, DB_G));
    return v;
}
__device__ __forceinline__ int group_min(int v)
{
#pragma unroll
    for (int d = DB_G / 2; d > 0; d >>= 1) v = min(v, __shfl_xor(v, d, DB_G));
    return v;
}

// Elems of positions [t0 - DB_HALO, t0 + UF_TILE + DB_HALO) in LDS
template <class M>
struct Staged {
    typename M::Elem *sh;
    int64_t first;                     // position of sh[0]
    const M &m;
    __device__ __forceinline__ typename M::Elem operator()(uint64_t j) const
    {
        const int64_t k = (int64_t)j - first;
        return (k >= 0 && k < UF_TILE + 2 * DB_HALO) ? sh[k] : m.load(j);
    }
};
template <class M>
__device__ __forceinline__ Staged<M> stage_tile(const M &m, typename M::Elem *sh, uint64_t t0, uint64_t n)
{
    const int64_t first = (int64_t)t0 - DB_HALO;
    for (int k = threadIdx.x; k < UF_TILE + 2 * DB_HALO; k += blockDim.x) {
        const int64_t idx = first + k;
        if (idx >= 0 && (uint64_t)idx < n) sh[k] = m.load((uint64_t)idx);
    }
    __syncthreads();
    return Staged<M>{sh, first, m};
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

__device__ __forceinline__ uint32_t lds_ld(uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

__device__ __forceinline__ uint32_t lds_find(uint32_t *lpar, uint32_t x)
{
    for (;;) {
        const uint32_t p = lds_ld(&lpar[x]);
        if (p == x) return x;
        const uint32_t gp = lds_ld(&lpar[p]);
        if (gp != p) __hip_atomic_store(&lpar[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // path halving
        x = p;
    }
}

// Two independent point sets can share every launch: positions [0, split) are one set, [split, n) the other (the DEL
// and INS calls of a chromosome). A window never leaves its own set; cluster ids restart at 0 in the second set.
// A window holds a few dozen candidates and its loop is a chain of dependent steps, so one thread per point leaves the chip
// idle (n / 64 waves for n ~ 1e5): DB_G lanes share a point and stride its window together.
template <class M>
__global__ void __launch_bounds__(DB_THREADS) db_count_union_kernel(M m, uint64_t n, uint64_t split, int min_pts_imm, const int *__restrict__ d_min_pts,
                                const uint32_t *__restrict__ oid, uint8_t *__restrict__ core, uint32_t *__restrict__ parent,
                                uint32_t *__restrict__ is_root, unsigned int *__restrict__ ticket)
{
    __shared__ typename M::Elem sh[UF_TILE + 2 * DB_HALO];
    __shared__ uint32_t lpar[UF_TILE], loid[UF_TILE];
    __shared__ uint8_t lcore[UF_TILE];
    const uint64_t t0 = (uint64_t)blockIdx.x * UF_TILE;
    if (threadIdx.x < UF_TILE) {
        const uint32_t li = threadIdx.x;
        const uint64_t i = t0 + li;
        lpar[li] = li;
        loid[li] = i < n ? (oid ? oid[i] : (uint32_t)i) : NONE;
        lcore[li] = 0;
    }
    const Staged<M> at = stage_tile(m, sh, t0, n);         // (ends with a barrier)
    if (t0 + threadIdx.x == n) { is_root[n] = 0; *ticket = 0; }
    const int min_pts = d_min_pts ? *d_min_pts : min_pts_imm;
    const uint32_t g = threadIdx.x / DB_G, lane = threadIdx.x % DB_G;
    // ---- core points: |window neighbours| >= min_pts ----
    for (uint32_t li = g; li < UF_TILE; li += DB_GROUPS) {
        const uint64_t i = t0 + li;
        if (i >= n) break;
        const uint64_t s0 = i < split ? 0 : split, s1 = i < split ? split : n;
        const typename M::Elem mine = at(i);
        uint64_t lo, hi;
        m.window(mine, lo, hi);
        int cnt = 0;
        for (uint64_t j0 = i; j0 < s1 && (uint64_t)M::key(at(j0)) <= hi; j0 += DB_G) {        // j0 == i: the point itself
            const uint64_t j = j0 + lane;
            if (j < s1) { const typename M::Elem o = at(j); if ((uint64_t)M::key(o) <= hi) cnt += m.nb(mine, o); }
        }
        for (uint64_t b0 = i; b0 > s0 && (uint64_t)M::key(at(b0 - 1)) >= lo; b0 = b0 > DB_G ? b0 - DB_G : 0) {
            if (b0 >= (uint64_t)lane + 1 && b0 - 1 - lane >= s0) { const typename M::Elem o = at(b0 - 1 - lane); if ((uint64_t)M::key(o) >= lo) cnt += m.nb(mine, o); }
            if (b0 <= DB_G) break;
        }
        cnt = group_sum(cnt);
        if (lane == 0) {
            const uint32_t me = loid[li];
            parent[me] = me;                           // union-find + root flags start here (no separate init launch)
            is_root[me] = 0;
            core[i] = cnt >= min_pts;
            lcore[li] = cnt >= min_pts;
        }
    }
    __syncthreads();
    // ---- pairs with both ends in this tile: union-find in LDS (a cluster's members are neighbours in the sorted order, so this is
    // nearly every pair), then each core point's parent is written flat: parent[me] = its tile root ----
    typename M::Elem *tile = sh + DB_HALO;                  // the tile's own points inside the staged window
    for (uint32_t li = g; li < UF_TILE; li += DB_GROUPS) {
        if (!lcore[li]) continue;
        const uint64_t i = t0 + li;
        const typename M::Elem mine = tile[li];
        uint64_t lo, hi;
        m.window(mine, lo, hi);
        const uint64_t s1 = i < split ? split : n;
        const uint32_t l_end = (uint32_t)(min(t0 + UF_TILE, s1) - t0);
        for (uint32_t j0 = li + 1; j0 < l_end && (uint64_t)M::key(tile[j0]) <= hi; j0 += DB_G) {
            const uint32_t lj = j0 + lane;
            if (lj >= l_end) continue;
            const typename M::Elem o = tile[lj];
            if ((uint64_t)M::key(o) > hi || !lcore[lj] || !m.nb(mine, o)) continue;
            uint32_t a = li, b = lj;
            for (;;) {
                a = lds_find(lpar, a);
                b = lds_find(lpar, b);
                if (a == b) break;
                if (loid[a] > loid[b]) { const uint32_t t = a; a = b; b = t; }      // the smaller original index stays root
                if (atomicCAS(&lpar[b], b, a) == b) break;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < UF_TILE && lcore[threadIdx.x]) parent[loid[threadIdx.x]] = loid[lds_find(lpar, threadIdx.x)];
}

// Pairs whose later end lies beyond the tile of the earlier one.
template <class M>
__global__ void db_union_cross_kernel(M m, uint64_t n, uint64_t split, const uint8_t *__restrict__ core, const uint32_t *__restrict__ oid, uint32_t *parent)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !core[i]) return;
    const uint64_t s1 = i < split ? split : n;
    const uint64_t tile_end = (i / UF_TILE + 1) * UF_TILE;
    if (tile_end >= s1) return;
    const typename M::Elem mine = m.load(i);
    uint64_t lo, hi;
    m.window(mine, lo, hi);
    if ((uint64_t)M::key(m.load(tile_end)) > hi) return;  // the window ends inside the tile
    const uint32_t me = oid ? oid[i] : (uint32_t)i;
    // After the local pass every point of a tile component carries the same parent, so consecutive neighbours that belong to one
    // component need a single union; each union is a chain of agent-scope round trips, which is what this kernel's time is made of.
    uint32_t last_parent = NONE;
    for (uint64_t j = tile_end; j < s1; j++) {
        const typename M::Elem o = m.load(j);
        if ((uint64_t)M::key(o) > hi) break;
        if (!core[j] || !m.nb(mine, o)) continue;
        const uint32_t oj = oid ? oid[j] : (uint32_t)j;
        const uint32_t pj = parent[oj];                   // the local pass's value is enough to tell tile components apart (a plain, cached load)
        if (pj == last_parent) continue;
        last_parent = pj;
        uf_union(parent, me, oj);
    }
}

// Positions are original indices (oid == nullptr): roots, their rank inside the tile, and — by the workgroup that finishes
// last — the exclusive scan of the tile totals. cid(k) = tile_prefix[k / UF_TILE] + rank[k] = number of roots below k.
__global__ void __launch_bounds__(UF_TILE) db_roots_rank_kernel(uint64_t n, const uint8_t *__restrict__ core, const uint32_t *__restrict__ parent,
                                                                uint32_t *__restrict__ root_of, uint32_t *__restrict__ rank,
                                                                uint32_t *__restrict__ tile_count, uint32_t *__restrict__ tile_prefix,
                                                                unsigned int *__restrict__ ticket, uint32_t n_tiles)
{
    __shared__ uint32_t wtot[UF_TILE / 64];
    __shared__ bool last;
    const uint64_t i = (uint64_t)blockIdx.x * UF_TILE + threadIdx.x;
    bool is_root = false;
    if (i < n) {
        uint32_t r = NONE;
        if (core[i]) {
            uint32_t x = (uint32_t)i;
            for (;;) { const uint32_t p = parent[x]; if (p == x) break; x = p; }   // unions finished: plain loads
            r = x;
            is_root = x == (uint32_t)i;
        }
        root_of[i] = r;
    }
    const uint64_t bal = __ballot(is_root);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) wtot[w] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t before = (uint32_t)__popcll(bal & lanemask_lt());
    uint32_t total = 0;
    for (int k = 0; k < UF_TILE / 64; k++) { if (k < w) before += wtot[k]; total += wtot[k]; }
    if (i <= n) rank[i] = before;                        // rank[n] closes the last tile (cid(n) is never asked for)
    if (threadIdx.x == 0) {
        tile_count[blockIdx.x] = total;
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    // exclusive scan of the tile totals, UF_TILE at a time
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t c0 = 0; c0 < n_tiles; c0 += UF_TILE) {
        const uint32_t k = c0 + threadIdx.x;
        const uint32_t v = k < n_tiles ? __hip_atomic_load(&tile_count[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const uint32_t inc = wave_incl_sum_dpp(v);
        if ((threadIdx.x & 63) == 63) wtot[w] = inc;
        __syncthreads();
        uint32_t base = carry_s;
        for (int q = 0; q < w; q++) base += wtot[q];
        if (k < n_tiles) tile_prefix[k] = base + inc - v;
        __syncthreads();
        if (threadIdx.x == UF_TILE - 1) carry_s = base + inc;
        __syncthreads();
    }
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

// number of roots with an original index below k: from the exclusive scan, or from tile prefix + rank inside the tile
struct ScanCid { const uint32_t *c; __device__ __forceinline__ uint32_t operator()(uint32_t k) const { return c[k]; } };
struct TileCid {
    const uint32_t *tile_prefix, *rank;
    __device__ __forceinline__ uint32_t operator()(uint32_t k) const { return tile_prefix[k / UF_TILE] + rank[k]; }
};

template <class M, class C>
__global__ void __launch_bounds__(DB_THREADS) db_label_kernel(M m, uint64_t n, uint64_t split, const uint32_t *__restrict__ oid, const uint32_t *__restrict__ root_of,
                                C cid_raw, int32_t *__restrict__ labels)
{
    __shared__ typename M::Elem sh[UF_TILE + 2 * DB_HALO];
    const uint64_t t0 = (uint64_t)blockIdx.x * UF_TILE;
    const Staged<M> at = stage_tile(m, sh, t0, n);
    const uint32_t g = threadIdx.x / DB_G, lane = threadIdx.x % DB_G;
    for (uint32_t li = g; li < UF_TILE; li += DB_GROUPS) {
        const uint64_t i = t0 + li;
        if (i >= n) break;
        const uint32_t me = oid ? oid[i] : (uint32_t)i;
        const uint32_t r = root_of[i];
        // ids of the second set restart at 0: subtract the number of roots of the first set (only used with oid == nullptr,
        // where original index == position, so cid_raw(split) is that count)
        const uint32_t id0 = (i >= split && split < n) ? cid_raw((uint32_t)split) : 0u;
        const uint64_t s0 = i < split ? 0 : split, s1 = i < split ? split : n;
        struct { C c; uint32_t off; __device__ uint32_t operator[](uint32_t k) const { return c(k) - off; } } cid{cid_raw, id0};
        if (r != NONE) { if (lane == 0) labels[me] = (int32_t)cid[r]; continue; }
        const typename M::Elem mine = at(i);
        uint64_t lo, hi;
        m.window(mine, lo, hi);
        int32_t max_start = -1, min_core = INT32_MAX;
        auto visit = [&](uint64_t j, const typename M::Elem &o) {
            if (!m.nb(mine, o)) return;
            const uint32_t rj = root_of[j];
            if (rj == NONE) return;
            const int32_t c = (int32_t)cid[rj];
            if (rj == (oid ? oid[j] : (uint32_t)j)) max_start = max(max_start, c); else min_core = min(min_core, c);
        };
        for (uint64_t j0 = i + 1; j0 < s1 && (uint64_t)M::key(at(j0)) <= hi; j0 += DB_G) {
            const uint64_t j = j0 + lane;
            if (j < s1) { const typename M::Elem o = at(j); if ((uint64_t)M::key(o) <= hi) visit(j, o); }
        }
        for (uint64_t b0 = i; b0 > s0 && (uint64_t)M::key(at(b0 - 1)) >= lo; b0 = b0 > DB_G ? b0 - DB_G : 0) {
            if (b0 >= (uint64_t)lane + 1 && b0 - 1 - lane >= s0) { const typename M::Elem o = at(b0 - 1 - lane); if ((uint64_t)M::key(o) >= lo) visit(b0 - 1 - lane, o); }
            if (b0 <= DB_G) break;
        }
        max_start = group_max(max_start);
        min_core = group_min(min_core);
        if (lane == 0) labels[me] = max_start >= 0 ? max_start : (min_core != INT32_MAX ? min_core : -2);
    }
}

// tmp: core u8[n] | parent u32[n] | root_of u32[n] | is_root/cid u32[n+1] | scan tmp
size_t dbscan_tmp_bytes(uint64_t n)
{
    const uint64_t nt = (n + UF_TILE) / UF_TILE + 1;
    return align_up(n, 256) + 2 * align_up(n * 4, 256) + align_up((n + 1) * 4, 256) + exclusive_sum_tmp_bytes(n + 1) + 2 * align_up(nt * 4, 256) + 256;
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
    void *es_tmp = p;                  p += exclusive_sum_tmp_bytes(n + 1);
    const uint32_t n_tiles = (uint32_t)((n + UF_TILE) / UF_TILE);          // tiles of positions 0..n (rank[n] included)
    uint32_t *tile_count = (uint32_t *)p;  p += align_up(((uint64_t)n_tiles + 1) * 4, 256);
    uint32_t *tile_prefix = (uint32_t *)p; p += align_up(((uint64_t)n_tiles + 1) * 4, 256);
    unsigned int *ticket = (unsigned int *)p;
    const dim3 grid((unsigned)((n + 255) / 256)), grid1((unsigned)((n + 1 + 255) / 256)), blk(256);
    const dim3 wide(DB_THREADS);
    hipLaunchKernelGGL(db_count_union_kernel<M>, grid1, wide, 0, s, m, n, split, min_pts, d_min_pts, oid, core, parent, cid, ticket);
    hipLaunchKernelGGL(db_union_cross_kernel<M>, grid, blk, 0, s, m, n, split, core, oid, parent);
    if (!oid) {
        hipLaunchKernelGGL(db_roots_rank_kernel, dim3(n_tiles), dim3(UF_TILE), 0, s, n, core, parent, root_of, cid, tile_count, tile_prefix, ticket, n_tiles);
        hipLaunchKernelGGL((db_label_kernel<M, TileCid>), grid, wide, 0, s, m, n, split, oid, root_of, TileCid{tile_prefix, cid}, labels);
    } else {
        hipLaunchKernelGGL(db_roots_kernel, grid, blk, 0, s, n, core, oid, parent, root_of, cid);
        launch_exclusive_sum_u32(s, cid, n + 1, es_tmp);
        hipLaunchKernelGGL((db_label_kernel<M, ScanCid>), grid, wide, 0, s, m, n, split, oid, root_of, ScanCid{cid}, labels);
    }
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

// ---- many small interval sets in ONE launch (the final mergeSVs of a run: one set per (contig, SV type), a few hundred to a few
// thousand calls each, in the caller's — not start-sorted — order) --------------------------------------------------------------------
// One workgroup per set, everything in LDS, brute force over all pairs with the reference's predicate: core flags, union-find of the
// core points (larger root under smaller: a component's root is its smallest original index = its start point), start points ranked
// in index order, then the border rule. Same order-free labelling as the windowed kernels above; no sort, no scratch in HBM.
// Intervals that share no position are never neighbours for eps < 1 (overlap 0 -> distance 1, or NaN for an empty interval): in a set
// that spans a contig that is almost every pair, and a wave whose 64 points all miss interval j skips the division path altogether.
__device__ __forceinline__ bool ivs_nb(uint32_t s1, uint32_t e1, uint32_t s2, uint32_t e2, double eps)
{
    if (min((int)e1, (int)e2) <= max((int)s1, (int)s2)) return false;
    return iv_neighbor(s1, e1, s2, e2, eps);
}

constexpr int IVS_THREADS = 1024;      // a launch lasts as long as its largest set: 16 waves share that set's n^2 pair tests
constexpr int IVS_MAX = (int)DBSCAN_IV_SMALL_MAX;

__global__ __launch_bounds__(IVS_THREADS) void dbscan_iv_small_kernel(const uint32_t *__restrict__ start, const uint32_t *__restrict__ end,
                                                                     const uint64_t *__restrict__ seg_off, uint64_t n_seg, double eps, int min_pts,
                                                                     int32_t *__restrict__ labels)
{
    __shared__ uint32_t ls[IVS_MAX], le[IVS_MAX], lpar[IVS_MAX], lcid[IVS_MAX + 1];
    __shared__ uint8_t lcore[IVS_MAX];
    __shared__ uint32_t wave_tot[IVS_THREADS / WAVE];
    for (uint64_t seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
        const uint64_t o0 = seg_off[seg];
        const uint64_t n64 = seg_off[seg + 1] - o0;
        if (n64 == 0 || n64 > (uint64_t)IVS_MAX) continue;           // larger sets: the windowed path (host decides)
        const int n = (int)n64;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += IVS_THREADS) { ls[i] = start[o0 + i]; le[i] = end[o0 + i]; lpar[i] = (uint32_t)i; }
        __syncthreads();
        // core points: |N(i)| >= min_pts, N(i) = { j : distance(i, j) <= eps } (regionQuery, dbscan.cpp:59-67; includes i itself)
        for (int i = threadIdx.x; i < n; i += IVS_THREADS) {
            const uint32_t si = ls[i], ei = le[i];
            int cnt = 0;
            for (int j = 0; j < n; j++) cnt += ivs_nb(si, ei, ls[j], le[j], eps);
            lcore[i] = cnt >= min_pts;
        }
        __syncthreads();
        // components of the core points
        for (int i = threadIdx.x; i < n; i += IVS_THREADS) {
            if (!lcore[i]) continue;
            const uint32_t si = ls[i], ei = le[i];
            for (int j = i + 1; j < n; j++) {
                if (!lcore[j] || !ivs_nb(si, ei, ls[j], le[j], eps)) continue;
                uint32_t a = (uint32_t)i, b = (uint32_t)j;
                for (;;) {
                    a = lds_find(lpar, a); b = lds_find(lpar, b);
                    if (a == b) break;
                    if (a > b) { const uint32_t t = a; a = b; b = t; }
                    if (atomicCAS(&lpar[b], b, a) == b) break;
                }
            }
        }
        __syncthreads();
        // start points (roots) ranked in index order: exclusive prefix sum of the root flags
        uint32_t carry = 0;
        for (int i0 = 0; i0 < n; i0 += IVS_THREADS) {
            const int i = i0 + (int)threadIdx.x;
            const uint32_t v = (i < n && lcore[i] && lds_ld(&lpar[i]) == (uint32_t)i) ? 1u : 0u;
            const uint32_t incl = wave_incl_sum(v);
            if (lane_id() == 63) wave_tot[threadIdx.x >> 6] = incl;
            __syncthreads();
            uint32_t before = carry;
            for (int w = 0; w < (int)(threadIdx.x >> 6); w++) before += wave_tot[w];
            if (i < n) lcid[i] = before + incl - v;
            uint32_t tot = 0;
            for (int w = 0; w < IVS_THREADS / WAVE; w++) tot += wave_tot[w];
            carry += tot;
            __syncthreads();
        }
        // labels
        for (int i = threadIdx.x; i < n; i += IVS_THREADS) {
            int32_t lab;
            if (lcore[i]) {
                lab = (int32_t)lcid[lds_find(lpar, (uint32_t)i)];
            } else {
                const uint32_t si = ls[i], ei = le[i];
                int32_t max_start = -1, min_core = INT32_MAX;
                for (int j = 0; j < n; j++) {
                    if (!lcore[j] || !ivs_nb(si, ei, ls[j], le[j], eps)) continue;
                    const uint32_t rj = lds_find(lpar, (uint32_t)j);
                    const int32_t c = (int32_t)lcid[rj];
                    if (rj == (uint32_t)j) max_start = max(max_start, c); else min_core = min(min_core, c);
                }
                lab = max_start >= 0 ? max_start : (min_core != INT32_MAX ? min_core : -2);
            }
            labels[o0 + i] = lab;
        }
    }
}

// The same labels without the n^2: two intervals can only be neighbours when they share a position, so with the set's points in start
// order (a bitonic sort of (start, index) pairs in LDS: 66 steps for 2 048 points) the neighbours of p behind it in that order are a
// stretch that ends at the first start at or beyond p's end. Every neighbour PAIR is met once, from its earlier member, three times over:
// counts (LDS atomicAdd on both ends), union of core pairs, and the border rule as LDS atomicMax / atomicMin on the non-core end — all
// results live in ORIGINAL-index space (roots = smallest original index, cluster ids = rank of the root in index order), so the order-free
// labelling is the brute-force kernel's bit for bit. A genome's final merge: 2 400 calls of one type on chr1 -> 5.8e6 pair tests per phase
// on one workgroup (350 us, the whole launch waits for it) against a few thousand.
__global__ __launch_bounds__(IVS_THREADS) void dbscan_iv_small_sorted_kernel(const uint32_t *__restrict__ start, const uint32_t *__restrict__ end,
                                                                            const uint64_t *__restrict__ seg_off, uint64_t n_seg, double eps, int min_pts,
                                                                            int32_t *__restrict__ labels)
{
    __shared__ uint32_t ss[IVS_MAX], se[IVS_MAX], oi[IVS_MAX];          // start order: start (sign-biased while sorting), end, original index
    __shared__ uint32_t lpar[IVS_MAX], lcid[IVS_MAX + 1];               // original-index space
    __shared__ int32_t acc[IVS_MAX], bmin[IVS_MAX];                     // neighbour count, then the border rule's maximum start-point id; its minimum core id
    __shared__ uint8_t lcore[IVS_MAX];
    __shared__ uint32_t wave_tot[IVS_THREADS / WAVE];
    for (uint64_t seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
        const uint64_t o0 = seg_off[seg];
        const uint64_t n64 = seg_off[seg + 1] - o0;
        if (n64 == 0 || n64 > (uint64_t)IVS_MAX) continue;           // larger sets: the windowed path (host decides)
        const int n = (int)n64;
        int np2 = 1;
        while (np2 < n) np2 <<= 1;
        __syncthreads();
        // (start, index) pairs in the order of (int)start — the predicate compares as int (sv_object / dbscan.cpp work in int) —, padded with +inf
        for (int i = threadIdx.x; i < np2; i += IVS_THREADS) {
            ss[i] = i < n ? (start[o0 + i] ^ 0x80000000u) : 0xffffffffu;
            oi[i] = (uint32_t)i;
            if (i < n) { lpar[i] = (uint32_t)i; acc[i] = 0; bmin[i] = INT32_MAX; }
        }
        __syncthreads();
        for (int k = 2; k <= np2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = threadIdx.x; t < np2 / 2; t += IVS_THREADS) {
                    const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
                    const bool up = (lo & k) == 0;
                    const uint32_t a = ss[lo], b = ss[hi], ia = oi[lo], ib = oi[hi];
                    const bool gt = a > b || (a == b && ia > ib);                  // (ties by index: a definite order, nothing depends on it)
                    if (gt == up) { ss[lo] = b; ss[hi] = a; oi[lo] = ib; oi[hi] = ia; }
                }
                __syncthreads();
            }
        for (int p = threadIdx.x; p < n; p += IVS_THREADS) { ss[p] ^= 0x80000000u; se[p] = end[o0 + oi[p]]; }
        __syncthreads();
        // ---- neighbour counts: |N(i)| includes i itself (regionQuery, dbscan.cpp:59-67)
        for (int p = threadIdx.x; p < n; p += IVS_THREADS) {
            const uint32_t sp = ss[p], ep = se[p];
            int c = ivs_nb(sp, ep, sp, ep, eps) ? 1 : 0;
            for (int q = p + 1; q < n && (int)ss[q] < (int)ep; q++)
                if (ivs_nb(sp, ep, ss[q], se[q], eps)) { c++; atomicAdd(&acc[oi[q]], 1); }
            atomicAdd(&acc[oi[p]], c);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += IVS_THREADS) lcore[i] = acc[i] >= min_pts;
        __syncthreads();
        // ---- components of the core points (larger root under smaller: a component's root is its smallest original index)
        for (int p = threadIdx.x; p < n; p += IVS_THREADS) {
            acc[oi[p]] = -1;                                             // (from here on: the border rule's maximum)
            if (!lcore[oi[p]]) continue;
            const uint32_t sp = ss[p], ep = se[p];
            for (int q = p + 1; q < n && (int)ss[q] < (int)ep; q++) {
                if (!lcore[oi[q]] || !ivs_nb(sp, ep, ss[q], se[q], eps)) continue;
                uint32_t a = oi[p], b = oi[q];
                for (;;) {
                    a = lds_find(lpar, a); b = lds_find(lpar, b);
                    if (a == b) break;
                    if (a > b) { const uint32_t t = a; a = b; b = t; }
                    if (atomicCAS(&lpar[b], b, a) == b) break;
                }
            }
        }
        __syncthreads();
        // start points (roots) ranked in index order: exclusive prefix sum of the root flags
        uint32_t carry = 0;
        for (int i0 = 0; i0 < n; i0 += IVS_THREADS) {
            const int i = i0 + (int)threadIdx.x;
            const uint32_t v = (i < n && lcore[i] && lds_ld(&lpar[i]) == (uint32_t)i) ? 1u : 0u;
            const uint32_t incl = wave_incl_sum(v);
            if (lane_id() == 63) wave_tot[threadIdx.x >> 6] = incl;
            __syncthreads();
            uint32_t before = carry;
            for (int w = 0; w < (int)(threadIdx.x >> 6); w++) before += wave_tot[w];
            if (i < n) lcid[i] = before + incl - v;
            uint32_t tot = 0;
            for (int w = 0; w < IVS_THREADS / WAVE; w++) tot += wave_tot[w];
            carry += tot;
            __syncthreads();
        }
        // ---- border rule: a non-core point takes the largest id among its neighbouring start points, else the smallest among its neighbouring cores
        for (int p = threadIdx.x; p < n; p += IVS_THREADS) {
            const uint32_t sp = ss[p], ep = se[p], a = oi[p];
            const bool ca = lcore[a];
            for (int q = p + 1; q < n && (int)ss[q] < (int)ep; q++) {
                const uint32_t b = oi[q];
                const bool cb = lcore[b];
                if (ca == cb || !ivs_nb(sp, ep, ss[q], se[q], eps)) continue;
                const uint32_t core = ca ? a : b, other = ca ? b : a;
                const uint32_t r = lds_find(lpar, core);
                const int32_t c = (int32_t)lcid[r];
                if (r == core) atomicMax(&acc[other], c); else atomicMin(&bmin[other], c);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += IVS_THREADS) {
            int32_t lab;
            if (lcore[i]) lab = (int32_t)lcid[lds_find(lpar, (uint32_t)i)];
            else lab = acc[i] >= 0 ? acc[i] : (bmin[i] != INT32_MAX ? bmin[i] : -2);
            labels[o0 + i] = lab;
        }
    }
}

void launch_dbscan_iv_small_batched(hipStream_t s, const uint32_t *start, const uint32_t *end, const uint64_t *seg_off, uint64_t n_seg, double eps,
                                    int min_pts, int32_t *labels)
{
    if (n_seg == 0) return;
    const unsigned grid = (unsigned)std::min<uint64_t>(n_seg, 8192);
    const char *be = getenv("CSV_DBSCAN_SMALL_BRUTE");                         // (A/B, tests: all pairs)
    const bool brute = be && *be && *be != '0';
    if (brute) hipLaunchKernelGGL(dbscan_iv_small_kernel, dim3(grid), dim3(IVS_THREADS), 0, s, start, end, seg_off, n_seg, eps, min_pts, labels);
    else hipLaunchKernelGGL(dbscan_iv_small_sorted_kernel, dim3(grid), dim3(IVS_THREADS), 0, s, start, end, seg_off, n_seg, eps, min_pts, labels);
}

}  // namespace csv
