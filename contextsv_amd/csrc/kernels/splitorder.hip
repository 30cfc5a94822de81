// splitorder.hip — §8f-4, the part of the split-read pass that scales with the read count: the ITERATION ORDER of the reference's
// `unordered_map<std::string, PrimaryAlignment>` per chromosome (src/sv_caller.cpp:137-172 fills it with every primary alignment,
// :183-202 erases those without a supplementary record, :216 / :224 iterate what is left: tree insertion order and group seeds).
//
// libstdc++'s _Hashtable keeps ONE singly linked node list; a node is always put at the front of its bucket, a bucket that becomes
// non-empty goes to the front of the list, and a rehash re-inserts the nodes in list order by the same two rules (host/umap_order.h
// replays this node by node). Between two rehashes ("epoch", bucket count B) that is a closed form: give every node present at the
// end of the epoch its processing time t — its position in the list when the epoch began, or its insertion index if it came later —
// then the list at the end of the epoch is the nodes sorted by (min t of the node's bucket, descending; t, descending).
// So the order of N keys is a chain of ~log2(N) stable sorts of geometrically growing size (~2N keys sorted in all), each of them:
//   so_mint   minT[bucket] = min t            (atomicMin; bucket = hash % B, B from the library's own _Prime_rehash_policy)
//   so_keys   key = one number for (contig, ~minT), items enumerated in descending t (all contigs of the genome share every launch)
//   stable radix sort (sort.hip), so_setlist    the new list
// and the survivors (nodes whose name hash is among the supplementary records' hashes) leave with their final position.
// Checked against umap_order.h (which is checked against the real container) in tests/test_gpu_split_order.py.
#include "../common.hpp"
#include "../devutil.hpp"

namespace csv {

constexpr int SO_THREADS = 256;
constexpr int SO_BLOCK = 1024;          // records per compaction block

// which contig owns block / work item x, given the exclusive prefix table off[0..A] (A <= SO_MAX_CONTIGS)
__device__ __forceinline__ uint32_t so_owner(const SplitOrderTab &t, uint64_t x, const uint64_t *off)
{
    uint32_t a = 0;
    for (uint32_t k = 1; k < t.A; k++) a += (x >= off[k]);       // A is small (<= 64): a linear scan of scalar loads
    return a;
}

__device__ __forceinline__ bool so_pass(uint16_t flag, uint8_t mapq, uint32_t min_mapq)
{
    // sv_caller.cpp:145-150 (filter) and :151 (primary = not supplementary)
    return !(flag & (F_SECONDARY | F_UNMAP | F_DUP | F_QCFAIL | F_SUPP)) && mapq >= min_mapq;
}

// h % B for a 32-bit B, exactly, without the 64-bit integer division (some 150 instructions with branches, and every node of every epoch
// and level pays it): the quotient in two 32-bit steps estimated in double precision (error far below 1), remainder in integers, fixed up.
__device__ __forceinline__ uint32_t so_mod(uint64_t h, uint32_t B, double inv /* 1.0 / B */)
{
    const uint32_t hi = (uint32_t)(h >> 32), lo = (uint32_t)h;
    int64_t r = (int64_t)hi - (int64_t)((uint64_t)(uint32_t)((double)hi * inv) * B);
    while (r < 0) r += B;
    while (r >= (int64_t)B) r -= B;
    const uint64_t x = ((uint64_t)(uint32_t)r << 32) | lo;                          // < B * 2^32: the quotient fits 32 bits
    const double xf = fma((double)(uint32_t)r, 4294967296.0, (double)lo);
    int64_t r2 = (int64_t)(x - (uint64_t)(uint32_t)(xf * inv) * B);
    while (r2 < 0) r2 += B;
    while (r2 >= (int64_t)B) r2 -= B;
    return (uint32_t)r2;
}

// ---- nodes = the filter-passing primary records in file order (the insertion order of the map) ------------------------------------
__global__ __launch_bounds__(SO_THREADS) void so_count_kernel(SplitOrderTab tab, uint32_t min_mapq, uint32_t *__restrict__ blk_cnt)
{
    const uint32_t a = so_owner(tab, blockIdx.x, tab.blk_off);
    const uint64_t r0 = (uint64_t)(blockIdx.x - tab.blk_off[a]) * SO_BLOCK;
    const uint64_t n = tab.n_reads[a];
    uint32_t c = 0;
    for (int k = threadIdx.x; k < SO_BLOCK; k += SO_THREADS) {
        const uint64_t r = r0 + k;
        if (r < n) c += so_pass(tab.flag[a][r], tab.mapq[a][r], min_mapq);
    }
    c = wave_sum(c);
    __shared__ uint32_t ws[SO_THREADS / WAVE];
    if (lane_id() == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) { uint32_t t = 0; for (int w = 0; w < SO_THREADS / WAVE; w++) t += ws[w]; blk_cnt[blockIdx.x] = t; }
}

// blk_off = exclusive sum of blk_cnt over ALL blocks: a contig's first block holds its first node's global index
__global__ __launch_bounds__(SO_THREADS) void so_scatter_kernel(SplitOrderTab tab, uint32_t min_mapq, const uint32_t *__restrict__ blk_off,
                                                                uint64_t *__restrict__ node_hash, uint32_t *__restrict__ node_rec)
{
    const uint32_t a = so_owner(tab, blockIdx.x, tab.blk_off);
    const uint64_t r0 = (uint64_t)(blockIdx.x - tab.blk_off[a]) * SO_BLOCK;
    const uint64_t n = tab.n_reads[a];
    __shared__ uint32_t ws[SO_THREADS / WAVE];
    uint32_t base = blk_off[blockIdx.x];
    for (int k0 = 0; k0 < SO_BLOCK; k0 += SO_THREADS) {
        const uint64_t r = r0 + k0 + threadIdx.x;
        const bool p = r < n && so_pass(tab.flag[a][r], tab.mapq[a][r], min_mapq);
        const uint64_t bal = __ballot(p);
        const uint32_t before = (uint32_t)__popcll(bal & lanemask_lt());
        if (lane_id() == 0) ws[threadIdx.x >> 6] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t wbase = 0, tot = 0;
        for (int w = 0; w < SO_THREADS / WAVE; w++) { if (w < (int)(threadIdx.x >> 6)) wbase += ws[w]; tot += ws[w]; }
        if (p) {
            const uint32_t g = base + wbase + before;
            node_hash[g] = tab.qhash[a][r];
            node_rec[g] = (uint32_t)r;
        }
        base += tot;
        __syncthreads();
    }
}

// ---- the supplementary records of the same shards: name hashes, in no particular order (the caller sorts them) ----------------------
// sv_caller.cpp:145-165: a record that passes the same filter and IS supplementary goes to supp_map[qname]; a primary survives the erase
// (:183-202) when its name has an entry there. When the call holds every contig of the run, the set of those names' hashes is computed
// here instead of waiting for the caller to collect them.
__device__ __forceinline__ bool so_pass_supp(uint16_t flag, uint8_t mapq, uint32_t min_mapq)
{
    return !(flag & (F_SECONDARY | F_UNMAP | F_DUP | F_QCFAIL)) && (flag & F_SUPP) && mapq >= min_mapq;
}
__device__ __forceinline__ uint32_t wave_append(bool p, unsigned int *counter);
__global__ __launch_bounds__(SO_THREADS) void so_supp_kernel(SplitOrderTab tab, uint32_t min_mapq, uint64_t *__restrict__ supp_hash, unsigned int *__restrict__ count)
{
    const uint32_t a = so_owner(tab, blockIdx.x, tab.blk_off);
    const uint64_t r0 = (uint64_t)(blockIdx.x - tab.blk_off[a]) * SO_BLOCK;
    const uint64_t n = tab.n_reads[a];
    for (int k0 = 0; k0 < SO_BLOCK; k0 += SO_THREADS) {
        const uint64_t r = r0 + k0 + threadIdx.x;
        const bool p = r < n && so_pass_supp(tab.flag[a][r], tab.mapq[a][r], min_mapq);
        if (!__ballot(p)) continue;
        const uint32_t slot = wave_append(p, count);
        if (p) supp_hash[slot] = tab.qhash[a][r];
    }
}

// ---- one epoch ------------------------------------------------------------------------------------------------------------------------
// Work item j of active contig a enumerates the contig's present nodes in DESCENDING processing time t: the nodes that were in the list
// when the epoch began sit at t = their list position (list[]), the ones inserted during the epoch at t = their insertion index (which is
// also their node index). A STABLE sort of the items by (contig, bucket time descending) then leaves every bucket's nodes in descending
// own time without t being part of the key: three 8-bit passes instead of six.
__device__ __forceinline__ uint32_t so_node_at(const SplitOrderTab &tab, uint32_t a, uint32_t t, const uint32_t *__restrict__ list)
{
    return tab.nbase[a] + (t < tab.m_old[a] ? list[tab.nbase[a] + t] : t);
}

__global__ __launch_bounds__(SO_THREADS) void so_mint_kernel(SplitOrderTab tab, uint64_t M, uint32_t B, const uint64_t *__restrict__ node_hash,
                                                             const uint32_t *__restrict__ list, uint32_t *__restrict__ minT)
{
    const uint64_t j = (uint64_t)blockIdx.x * SO_THREADS + threadIdx.x;
    if (j >= M) return;
    const uint32_t a = so_owner(tab, j, tab.work_off);
    const uint32_t m = (uint32_t)(tab.work_off[a + 1] - tab.work_off[a]);
    const uint32_t t = m - 1u - (uint32_t)(j - tab.work_off[a]);
    const uint32_t g = so_node_at(tab, a, t, list);
    const uint32_t b = so_mod(node_hash[g], B, 1.0 / (double)B);
    atomicMin(&minT[(uint64_t)a * B + b], t);
}

__global__ __launch_bounds__(SO_THREADS) void so_keys_kernel(SplitOrderTab tab, uint64_t M, uint32_t B, int w, const uint64_t *__restrict__ node_hash,
                                                             const uint32_t *__restrict__ list, const uint32_t *__restrict__ minT,
                                                             uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const uint64_t j = (uint64_t)blockIdx.x * SO_THREADS + threadIdx.x;
    if (j >= M) return;
    const uint32_t a = so_owner(tab, j, tab.work_off);
    const uint32_t m = (uint32_t)(tab.work_off[a + 1] - tab.work_off[a]);
    const uint32_t t = m - 1u - (uint32_t)(j - tab.work_off[a]);
    const uint32_t g = so_node_at(tab, a, t, list);
    const uint32_t b = so_mod(node_hash[g], B, 1.0 / (double)B);
    // One number orders contig and bucket time together: rev_off[a] = items of the contigs behind a, so contig 0 owns the largest
    // values and, sorted descending (ascending in M - 1 - x), comes first, its buckets with the LARGER time first. bits(M - 1) <= 23
    // for a genome: three 8-bit passes.
    keys[j] = (M - 1ull) - (tab.rev_off[a] + (uint64_t)minT[(uint64_t)a * B + b]);
    vals[j] = g - tab.nbase[a];                                                        // node index inside its contig
}

// the sorted items are the new lists, contig after contig
__global__ __launch_bounds__(SO_THREADS) void so_setlist_kernel(SplitOrderTab tab, uint64_t M, const uint32_t *__restrict__ vals, uint32_t *__restrict__ list)
{
    const uint64_t j = (uint64_t)blockIdx.x * SO_THREADS + threadIdx.x;
    if (j >= M) return;
    const uint32_t a = so_owner(tab, j, tab.work_off);
    list[tab.nbase[a] + (uint32_t)(j - tab.work_off[a])] = vals[j];
}

// ---- the nodes that survive the erase: their name hash is among the supplementary records' (sorted, distinct) hashes ----------------
// item = (contig a, list position p): node list[p] of the contig's final list (a contig with a single node never went through an epoch:
// its list is the node itself)
__global__ __launch_bounds__(SO_THREADS) void so_survivors_kernel(SplitOrderTab tab, uint64_t n_nodes, const uint64_t *__restrict__ node_hash,
                                                                  const uint32_t *__restrict__ node_rec, const uint32_t *__restrict__ list,
                                                                  const uint64_t *__restrict__ supp_hash, uint64_t n_supp, csv_split_survivor *__restrict__ out,
                                                                  uint64_t cap, unsigned long long *__restrict__ count)
{
    const uint64_t x = (uint64_t)blockIdx.x * SO_THREADS + threadIdx.x;
    if (x >= n_nodes) return;
    uint32_t a = 0;
    for (uint32_t k = 1; k < tab.A; k++) a += (x >= tab.nbase[k]);
    const uint32_t p = (uint32_t)x - tab.nbase[a];
    const uint32_t n_a = tab.nbase[a + 1] - tab.nbase[a];
    const uint32_t g = tab.nbase[a] + (n_a > 1 ? list[x] : 0u);
    const uint64_t h = node_hash[g];
    uint64_t lo = 0, hi = n_supp;
    while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (supp_hash[mid] < h) lo = mid + 1; else hi = mid; }
    if (lo >= n_supp || supp_hash[lo] != h) return;
    const unsigned long long slot = atomicAdd(count, 1ull);
    if (slot < cap) out[slot] = csv_split_survivor{a, p, node_rec[g]};
}

// ---- the first epochs, one launch ---------------------------------------------------------------------------------------------------------
// The epochs double in size from 13 buckets on, and each costs the chain sixteen launches whatever its size. While an epoch's nodes and
// buckets fit the LDS (B <= SO_SMALL_B: epochs 0..8, up to 5 087 nodes) ONE workgroup per contig runs through all of them: bucket times by
// LDS atomicMin, a histogram of the nodes' bucket times and its suffix sums (a bucket's first slot = the nodes in buckets with a later time),
// slots claimed by LDS atomics and every bucket's few nodes put in descending own time by the bucket's leader — the same order as a stable
// sort by (bucket time descending) of the nodes enumerated in descending time, which is what the chain's epochs compute.
struct SplitSmallTab {                       // by value
    uint32_t A = 0, n_epochs = 0;
    uint32_t nbase[SO_MAX_CONTIGS + 1];
    int32_t  k_last[SO_MAX_CONTIGS];         // last epoch this contig runs here (-1: none)
    uint32_t first[SO_SMALL_EPOCHS + 1];     // first node of epoch k (k = n_epochs: first node of the epoch behind the last one here)
    uint32_t B[SO_SMALL_EPOCHS];
};
__global__ __launch_bounds__(1024) void so_small_epochs_kernel(SplitSmallTab tab, const uint64_t *__restrict__ node_hash, uint32_t *__restrict__ list)
{
    constexpr int T = 1024, PER = (SO_SMALL_B + T - 1) / T;
    __shared__ uint16_t P[SO_SMALL_B];                  // node -> position in the list (its own index until its first epoch)
    __shared__ uint32_t A1[SO_SMALL_B], A2[SO_SMALL_B], A3[SO_SMALL_B], A4[SO_SMALL_B];
    __shared__ uint32_t wsum[T / WAVE];
    const uint32_t a = blockIdx.x;
    const int k_last = tab.k_last[a];
    if (k_last < 0) return;
    const uint32_t N = tab.nbase[a + 1] - tab.nbase[a];
    const uint64_t *__restrict__ const hash = node_hash + tab.nbase[a];
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    uint32_t m = 0;
    for (int k = 0; k <= k_last; k++) {
        const uint32_t m_old = tab.first[k], B = tab.B[k];
        m = min(N, tab.first[k + 1]);
        const double inv = 1.0 / (double)B;
        for (uint32_t x = m_old + tid; x < m; x += T) P[x] = (uint16_t)x;
        for (uint32_t b = tid; b < B; b += T) A1[b] = 0xffffffffu;
        for (uint32_t t = tid; t < m; t += T) A2[t] = 0;
        __syncthreads();
        uint32_t bq[PER];
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const uint32_t x = (uint32_t)tid + (uint32_t)i * T;
            if (x < m) { bq[i] = so_mod(hash[x], B, inv); atomicMin(&A1[bq[i]], (uint32_t)P[x]); }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const uint32_t x = (uint32_t)tid + (uint32_t)i * T;
            if (x < m) { const uint32_t w = A1[bq[i]]; atomicAdd(&A2[w], 1u); A3[x] = (w << 16) | (uint32_t)P[x]; }
        }
        __syncthreads();
        {   // A2[w] <- number of nodes with a bucket time above w (blocked layout over the reversed index), A1 <- 0 (the slot counters)
            uint32_t v[PER], tot = 0;
#pragma unroll
            for (int i = 0; i < PER; i++) { const uint32_t r = (uint32_t)tid * PER + i; v[i] = r < m ? A2[m - 1 - r] : 0u; tot += v[i]; }
            const uint32_t incl = wave_incl_sum_dpp(tot);
            if (lane == 63) wsum[wave] = incl;
            for (uint32_t b = tid; b < m; b += T) A1[b] = 0;
            __syncthreads();
            uint32_t run = incl - tot;
            for (int w2 = 0; w2 < wave; w2++) run += wsum[w2];
#pragma unroll
            for (int i = 0; i < PER; i++) { const uint32_t r = (uint32_t)tid * PER + i; if (r < m) A2[m - 1 - r] = run; run += v[i]; }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const uint32_t x = (uint32_t)tid + (uint32_t)i * T;
            if (x < m) { const uint32_t wt = A3[x], w = wt >> 16; A4[A2[w] + atomicAdd(&A1[w], 1u)] = (wt << 16) | x; }      // (t << 16) | node
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PER; i++) {
            const uint32_t x = (uint32_t)tid + (uint32_t)i * T;
            if (x < m) {
                const uint32_t wt = A3[x];
                if ((wt >> 16) == (wt & 0xffffu)) {                     // the bucket's leader orders the bucket: descending own time
                    const uint32_t s0 = A2[wt >> 16], n = A1[wt >> 16];
                    for (uint32_t i1 = 1; i1 < n; i1++) {
                        const uint32_t key = A4[s0 + i1];
                        uint32_t j = i1;
                        while (j > 0 && A4[s0 + j - 1] < key) { A4[s0 + j] = A4[s0 + j - 1]; j--; }
                        A4[s0 + j] = key;
                    }
                }
            }
        }
        __syncthreads();
        for (uint32_t p = tid; p < m; p += T) P[A4[p] & 0xffffu] = (uint16_t)p;
        __syncthreads();
    }
    for (uint32_t p = tid; p < m; p += T) list[tab.nbase[a] + p] = m > 1 ? (A4[p] & 0xffffu) : 0u;
}

// ---- the last epochs, survivors only ---------------------------------------------------------------------------------------------------
// Half of the chain's work is its last epoch, three quarters its last two — and all the caller wants is the relative order of the ~1 % of
// nodes that survive the erase. The order at the end of an epoch is by (min t of the node's bucket, t), so the order among a SET of nodes
// needs the t of the set's members and of every node that shares a bucket with one of them, nothing else; and t is only ever COMPARED, so
// for the nodes that were in the list when the epoch began any order-preserving number will do in place of the list position. Hence, for
// the last D epochs e = K, K-1, ... of every contig (K its last epoch; level j = K - e):
//   top-down (hashes only):  S_1 = the nodes that share an epoch-K bucket with a survivor; S_2 = the nodes that were in the list when epoch K
//            began and share an epoch-(K-1) bucket with such a member of S_1; ... Each step streams the epoch's nodes past a bitmap of marked
//            buckets (2 N bits: resident in L2) — sequential reads and a 64-bit modulo, no sort, no scattered atomics. The sets double per
//            level while the epochs halve: D = 3 for 1 % survivors.
//   bottom-up:  the chain of full sorts runs only up to epoch K - D (an eighth of its keys); then level D-1 orders S_D with t = list position
//            (old nodes: the chain's result) or insertion index (new nodes, flag bit above), level D-2 orders S_(D-1) with t = rank in the
//            order just computed, ..., level 0 orders S_1 and the survivors leave in that order.
struct SplitTailTab {                        // passed by value
    uint32_t A = 0, D = 0;
    int      wv = 0;                         // bits of a t value (the new-node flag sits above them)
    int      wa = 0;                         // bits of a contig index
    uint32_t nbase[SO_MAX_CONTIGS + 1];
    uint32_t B[SO_TAIL_MAX][SO_MAX_CONTIGS];           // bucket count of epoch K_c - j (1 where the contig has no such epoch)
    uint32_t F[SO_TAIL_MAX][SO_MAX_CONTIGS];           // first node inserted in that epoch = nodes in the list when it began (0 where no such epoch)
    uint32_t boff[SO_TAIL_MAX][SO_MAX_CONTIGS + 1];    // first bucket of contig c in level j's bucket tables (bitmap bits; minT entries)
    double   invB[SO_TAIL_MAX][SO_MAX_CONTIGS];        // 1.0 / B
};

__device__ __forceinline__ uint32_t tail_owner(const SplitTailTab &t, uint32_t g)
{
    uint32_t a = 0;
    for (uint32_t k = 1; k < t.A; k++) a += (g >= t.nbase[k]);
    return a;
}

// one atomic per wave reserves the slots of its appending lanes
__device__ __forceinline__ uint32_t wave_append(bool p, unsigned int *counter)
{
    const uint64_t m = __ballot(p);
    if (!m) return 0;
    uint32_t base = 0;
    if (lane_id() == (int)__builtin_ctzll(m)) base = atomicAdd(counter, (unsigned int)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(m));
    return base + (uint32_t)__popcll(m & lanemask_lt());
}

// a bit per value of a supplementary hash's top bits: 1 node in ~20 passes it, the binary search (17 dependent reads) is for those
constexpr int ST_FILTER_BITS = 21;
__global__ __launch_bounds__(SO_THREADS) void st_filter_kernel(const uint64_t *__restrict__ supp_hash, uint64_t n_supp, uint32_t *__restrict__ filter)
{
    const uint64_t i = (uint64_t)blockIdx.x * SO_THREADS + threadIdx.x;
    if (i >= n_supp) return;
    const uint32_t f = (uint32_t)(supp_hash[i] >> (64 - ST_FILTER_BITS));
    atomicOr(&filter[f >> 5], 1u << (f & 31u));
}

// level 0 of the top-down pass: the survivors (name hash among the supplementary records') mark their epoch-K buckets
__global__ __launch_bounds__(SO_THREADS) void st_survivors_kernel(SplitTailTab tab, uint32_t n_nodes, const uint64_t *__restrict__ node_hash,
                                                                  const uint64_t *__restrict__ supp_hash, uint64_t n_supp, const uint32_t *__restrict__ filter,
                                                                  uint8_t *__restrict__ is_surv, uint32_t *__restrict__ bitmap)
{
    const uint32_t g = blockIdx.x * SO_THREADS + threadIdx.x;
    if (g >= n_nodes) return;
    const uint64_t h = node_hash[g];
    const uint32_t f = (uint32_t)(h >> (64 - ST_FILTER_BITS));
    bool sv = (filter[f >> 5] >> (f & 31u)) & 1u;
    if (sv) {
        uint64_t lo = 0, hi = n_supp;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (supp_hash[mid] < h) lo = mid + 1; else hi = mid; }
        sv = lo < n_supp && supp_hash[lo] == h;
    }
    is_surv[g] = sv ? 1 : 0;
    if (sv) {
        const uint32_t a = tail_owner(tab, g);
        const uint32_t b = tab.boff[0][a] + so_mod(h, tab.B[0][a], tab.invB[0][a]);
        atomicOr(&bitmap[b >> 5], 1u << (b & 31u));
    }
}

// level j >= 1: the nodes present at the end of epoch K - (j - 1) whose bucket there is marked form S_j; those of them that were already in
// the list when that epoch began mark their bucket of the epoch before (unless this is the last level). A workgroup takes ST_PER * 256
// nodes and reserves its members' slots with ONE global atomic (a wave each: 1.6e5 atomics on one address, 0.76 ms of a 1e7-node pass).
constexpr int ST_PER = 8;
__global__ __launch_bounds__(SO_THREADS) void st_member_kernel(SplitTailTab tab, uint32_t n_nodes, int j, const uint64_t *__restrict__ node_hash,
                                                               const uint32_t *__restrict__ bitmap_prev, uint32_t *__restrict__ bitmap_next,
                                                               uint32_t *__restrict__ set, unsigned int *__restrict__ count)
{
    __shared__ unsigned int n_local, base_s;
    if (threadIdx.x == 0) n_local = 0;
    __syncthreads();
    uint32_t slot[ST_PER], mem = 0;
#pragma unroll
    for (int i = 0; i < ST_PER; i++) {
        const uint32_t g = (blockIdx.x * ST_PER + i) * SO_THREADS + threadIdx.x;
        bool in = false;
        if (g < n_nodes) {
            const uint32_t a = tail_owner(tab, g);
            const uint32_t x = g - tab.nbase[a];
            const uint32_t lim = j == 1 ? tab.nbase[a + 1] - tab.nbase[a] : tab.F[j - 2][a];        // nodes present at the end of epoch K - (j - 1)
            if (x < lim) {
                const uint64_t h = node_hash[g];
                const uint32_t b = tab.boff[j - 1][a] + so_mod(h, tab.B[j - 1][a], tab.invB[j - 1][a]);
                in = (bitmap_prev[b >> 5] >> (b & 31u)) & 1u;
                if (in && bitmap_next && x < tab.F[j - 1][a]) {
                    const uint32_t b2 = tab.boff[j][a] + so_mod(h, tab.B[j][a], tab.invB[j][a]);
                    atomicOr(&bitmap_next[b2 >> 5], 1u << (b2 & 31u));
                }
            }
        }
        slot[i] = wave_append(in, &n_local);
        mem |= (in ? 1u : 0u) << i;
    }
    __syncthreads();
    if (threadIdx.x == 0) base_s = n_local ? atomicAdd(count, n_local) : 0u;
    __syncthreads();
    const uint32_t base = base_s;
#pragma unroll
    for (int i = 0; i < ST_PER; i++)
        if ((mem >> i) & 1u) set[base + slot[i]] = (blockIdx.x * ST_PER + i) * SO_THREADS + threadIdx.x;
}

// the chain's result as node -> list position
__global__ __launch_bounds__(SO_THREADS) void st_inverse_kernel(SplitTailTab tab, uint32_t n_nodes, int j_last, const uint32_t *__restrict__ list,
                                                                uint32_t *__restrict__ prevrank)
{
    const uint32_t g = blockIdx.x * SO_THREADS + threadIdx.x;
    if (g >= n_nodes) return;
    const uint32_t a = tail_owner(tab, g);
    const uint32_t p = g - tab.nbase[a];
    const uint32_t m = tab.F[j_last][a];                              // nodes the chain ordered for this contig
    if (p >= m) return;
    prevrank[tab.nbase[a] + (m > 1 ? list[g] : 0u)] = p;
}

// level j, bottom-up: t of every member of S_(j+1) at epoch K - j and the minimum per bucket
__device__ __forceinline__ uint32_t st_t(const SplitTailTab &tab, int j, uint32_t a, uint32_t x, uint32_t g, const uint32_t *__restrict__ prevrank)
{
    return x < tab.F[j][a] ? prevrank[g] : ((1u << tab.wv) | x);
}
// (n_dev: the set's size is still on the device when the level is queued — csvgpu_split_order_begin_self —; n is then its bound and the
// launch a fixed grid that strides over the set)
__global__ __launch_bounds__(SO_THREADS) void st_mint_kernel(SplitTailTab tab, int j, const uint32_t *__restrict__ set, uint32_t n, const uint32_t *__restrict__ n_dev,
                                                             const uint64_t *__restrict__ node_hash, const uint32_t *__restrict__ prevrank, uint32_t *__restrict__ minT)
{
    if (n_dev) n = *n_dev;
    for (uint32_t i = blockIdx.x * SO_THREADS + threadIdx.x; i < n; i += gridDim.x * SO_THREADS) {
        const uint32_t g = set[i];
        const uint32_t a = tail_owner(tab, g);
        const uint32_t b = tab.boff[j][a] + so_mod(node_hash[g], tab.B[j][a], tab.invB[j][a]);
        atomicMin(&minT[b], st_t(tab, j, a, g - tab.nbase[a], g, prevrank));
    }
}
__global__ __launch_bounds__(SO_THREADS) void st_keys_kernel(SplitTailTab tab, int j, const uint32_t *__restrict__ set, uint32_t n, const uint32_t *__restrict__ n_dev,
                                                             const uint64_t *__restrict__ node_hash, const uint32_t *__restrict__ prevrank, const uint32_t *__restrict__ minT,
                                                             uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    if (n_dev) n = *n_dev;
    for (uint32_t i = blockIdx.x * SO_THREADS + threadIdx.x; i < n; i += gridDim.x * SO_THREADS) {
        const uint32_t g = set[i];
        const uint32_t a = tail_owner(tab, g);
        const uint32_t b = tab.boff[j][a] + so_mod(node_hash[g], tab.B[j][a], tab.invB[j][a]);
        const uint32_t t = st_t(tab, j, a, g - tab.nbase[a], g, prevrank);
        const int w = tab.wv + 1;
        const uint64_t mask = (1ull << w) - 1ull;
        // ascending in the key = contig ascending, bucket time descending, own time descending: the list order
        keys[i] = ((uint64_t)a << (2 * w)) | ((~(uint64_t)minT[b] & mask) << w) | (~(uint64_t)t & mask);
        vals[i] = g;
    }
}
__global__ __launch_bounds__(SO_THREADS) void st_rank_kernel(const uint32_t *__restrict__ vals, uint32_t n, const uint32_t *__restrict__ n_dev, uint32_t *__restrict__ prevrank)
{
    if (n_dev) n = *n_dev;
    for (uint32_t i = blockIdx.x * SO_THREADS + threadIdx.x; i < n; i += gridDim.x * SO_THREADS) prevrank[vals[i]] = i;
}
// the survivors of the ordered S_1 with their rank as the position
__global__ __launch_bounds__(SO_THREADS) void st_emit_kernel(SplitTailTab tab, const uint32_t *__restrict__ vals, uint32_t n, const uint32_t *__restrict__ n_dev,
                                                             const uint8_t *__restrict__ is_surv, const uint32_t *__restrict__ node_rec, csv_split_survivor *__restrict__ out,
                                                             uint64_t cap, unsigned long long *__restrict__ count)
{
    if (n_dev) n = *n_dev;
    for (uint32_t i = blockIdx.x * SO_THREADS + threadIdx.x; i < n; i += gridDim.x * SO_THREADS) {
        const uint32_t g = vals[i];
        if (!is_surv[g]) continue;
        const unsigned long long slot = atomicAdd(count, 1ull);
        if (slot < cap) out[slot] = csv_split_survivor{tail_owner(tab, g), i, node_rec[g]};
    }
}

void launch_so_count(hipStream_t s, const SplitOrderTab &tab, uint32_t n_blocks, uint32_t min_mapq, uint32_t *blk_cnt)
{
    if (n_blocks) hipLaunchKernelGGL(so_count_kernel, dim3(n_blocks), dim3(SO_THREADS), 0, s, tab, min_mapq, blk_cnt);
}
void launch_so_scatter(hipStream_t s, const SplitOrderTab &tab, uint32_t n_blocks, uint32_t min_mapq, const uint32_t *blk_off, uint64_t *node_hash,
                       uint32_t *node_rec)
{
    if (n_blocks) hipLaunchKernelGGL(so_scatter_kernel, dim3(n_blocks), dim3(SO_THREADS), 0, s, tab, min_mapq, blk_off, node_hash, node_rec);
}
static inline unsigned so_grid(uint64_t n) { return (unsigned)((n + SO_THREADS - 1) / SO_THREADS); }
void launch_so_mint(hipStream_t s, const SplitOrderTab &tab, uint64_t M, uint32_t B, const uint64_t *node_hash, const uint32_t *list, uint32_t *minT)
{
    if (M) hipLaunchKernelGGL(so_mint_kernel, dim3(so_grid(M)), dim3(SO_THREADS), 0, s, tab, M, B, node_hash, list, minT);
}
void launch_so_keys(hipStream_t s, const SplitOrderTab &tab, uint64_t M, uint32_t B, int w, const uint64_t *node_hash, const uint32_t *list,
                    const uint32_t *minT, uint64_t *keys, uint32_t *vals)
{
    if (M) hipLaunchKernelGGL(so_keys_kernel, dim3(so_grid(M)), dim3(SO_THREADS), 0, s, tab, M, B, w, node_hash, list, minT, keys, vals);
}
void launch_so_setlist(hipStream_t s, const SplitOrderTab &tab, uint64_t M, const uint32_t *vals, uint32_t *list)
{
    if (M) hipLaunchKernelGGL(so_setlist_kernel, dim3(so_grid(M)), dim3(SO_THREADS), 0, s, tab, M, vals, list);
}
void launch_so_survivors(hipStream_t s, const SplitOrderTab &tab, uint64_t n_nodes, const uint64_t *node_hash, const uint32_t *node_rec, const uint32_t *list,
                         const uint64_t *supp_hash, uint64_t n_supp, csv_split_survivor *out, uint64_t cap, unsigned long long *count)
{
    if (n_nodes) hipLaunchKernelGGL(so_survivors_kernel, dim3(so_grid(n_nodes)), dim3(SO_THREADS), 0, s, tab, n_nodes, node_hash, node_rec, list, supp_hash, n_supp, out, cap, count);
}

void launch_so_small_epochs(hipStream_t s, const SplitSmallHost &h, const uint64_t *node_hash, uint32_t *list)
{
    if (!h.A || !h.n_epochs) return;
    SplitSmallTab t;
    t.A = h.A; t.n_epochs = h.n_epochs;
    for (uint32_t c = 0; c <= SO_MAX_CONTIGS; c++) t.nbase[c] = h.nbase[c <= h.A ? c : h.A];
    for (uint32_t c = 0; c < SO_MAX_CONTIGS; c++) t.k_last[c] = c < h.A ? h.k_last[c] : -1;
    for (uint32_t k = 0; k <= SO_SMALL_EPOCHS; k++) t.first[k] = h.first[k];
    for (uint32_t k = 0; k < SO_SMALL_EPOCHS; k++) t.B[k] = h.B[k];
    hipLaunchKernelGGL(so_small_epochs_kernel, dim3(h.A), dim3(1024), 0, s, t, node_hash, list);
}

// ---- launchers of the survivors-only tail
static inline SplitTailTab to_tab(const SplitTailHost &h)
{
    SplitTailTab t;
    t.A = h.A; t.D = h.D; t.wv = h.wv; t.wa = h.wa;
    for (uint32_t c = 0; c <= h.A; c++) t.nbase[c] = h.nbase[c];
    for (uint32_t j = 0; j < SO_TAIL_MAX; j++) {
        for (uint32_t c = 0; c < SO_MAX_CONTIGS; c++) { t.B[j][c] = c < h.A ? h.B[j][c] : 1u; t.F[j][c] = c < h.A ? h.F[j][c] : 0u; t.invB[j][c] = 1.0 / (double)t.B[j][c]; }
        for (uint32_t c = 0; c <= SO_MAX_CONTIGS; c++) t.boff[j][c] = h.boff[j][c <= h.A ? c : h.A];
    }
    return t;
}
size_t st_filter_bytes() { return ((size_t)1 << ST_FILTER_BITS) / 8; }
void launch_st_survivors(hipStream_t s, const SplitTailHost &h, uint32_t n_nodes, const uint64_t *node_hash, const uint64_t *supp_hash, uint64_t n_supp, uint32_t *filter /* st_filter_bytes(), zeroed */,
                         uint8_t *is_surv, uint32_t *bitmap)
{
    if (!n_nodes) return;
    if (n_supp) hipLaunchKernelGGL(st_filter_kernel, dim3(so_grid(n_supp)), dim3(SO_THREADS), 0, s, supp_hash, n_supp, filter);
    hipLaunchKernelGGL(st_survivors_kernel, dim3(so_grid(n_nodes)), dim3(SO_THREADS), 0, s, to_tab(h), n_nodes, node_hash, supp_hash, n_supp, filter, is_surv, bitmap);
}
void launch_st_member(hipStream_t s, const SplitTailHost &h, uint32_t n_nodes, int j, const uint64_t *node_hash, const uint32_t *bitmap_prev, uint32_t *bitmap_next,
                      uint32_t *set, unsigned int *count)
{
    if (n_nodes) hipLaunchKernelGGL(st_member_kernel, dim3((n_nodes + ST_PER * SO_THREADS - 1) / (ST_PER * SO_THREADS)), dim3(SO_THREADS), 0, s, to_tab(h), n_nodes, j, node_hash, bitmap_prev, bitmap_next, set, count);
}
void launch_st_inverse(hipStream_t s, const SplitTailHost &h, uint32_t n_nodes, int j_last, const uint32_t *list, uint32_t *prevrank)
{
    if (n_nodes) hipLaunchKernelGGL(st_inverse_kernel, dim3(so_grid(n_nodes)), dim3(SO_THREADS), 0, s, to_tab(h), n_nodes, j_last, list, prevrank);
}
static inline unsigned st_grid(uint32_t n, const uint32_t *n_dev) { return n_dev ? std::min(so_grid(n), 2048u) : so_grid(n); }
void launch_st_mint(hipStream_t s, const SplitTailHost &h, int j, const uint32_t *set, uint32_t n, const uint32_t *n_dev, const uint64_t *node_hash, const uint32_t *prevrank,
                    uint32_t *minT)
{
    if (n) hipLaunchKernelGGL(st_mint_kernel, dim3(st_grid(n, n_dev)), dim3(SO_THREADS), 0, s, to_tab(h), j, set, n, n_dev, node_hash, prevrank, minT);
}
void launch_st_keys(hipStream_t s, const SplitTailHost &h, int j, const uint32_t *set, uint32_t n, const uint32_t *n_dev, const uint64_t *node_hash, const uint32_t *prevrank,
                    const uint32_t *minT, uint64_t *keys, uint32_t *vals)
{
    if (n) hipLaunchKernelGGL(st_keys_kernel, dim3(st_grid(n, n_dev)), dim3(SO_THREADS), 0, s, to_tab(h), j, set, n, n_dev, node_hash, prevrank, minT, keys, vals);
}
void launch_st_rank(hipStream_t s, const uint32_t *vals, uint32_t n, const uint32_t *n_dev, uint32_t *prevrank)
{
    if (n) hipLaunchKernelGGL(st_rank_kernel, dim3(st_grid(n, n_dev)), dim3(SO_THREADS), 0, s, vals, n, n_dev, prevrank);
}
void launch_st_emit(hipStream_t s, const SplitTailHost &h, const uint32_t *vals, uint32_t n, const uint32_t *n_dev, const uint8_t *is_surv, const uint32_t *node_rec,
                    csv_split_survivor *out, uint64_t cap, unsigned long long *count)
{
    if (n) hipLaunchKernelGGL(st_emit_kernel, dim3(st_grid(n, n_dev)), dim3(SO_THREADS), 0, s, to_tab(h), vals, n, n_dev, is_surv, node_rec, out, cap, count);
}
void launch_so_supp(hipStream_t s, const SplitOrderTab &tab, uint32_t n_blocks, uint32_t min_mapq, uint64_t *supp_hash, unsigned int *count)
{
    if (n_blocks) hipLaunchKernelGGL(so_supp_kernel, dim3(n_blocks), dim3(SO_THREADS), 0, s, tab, min_mapq, supp_hash, count);
}

}  // namespace csv

