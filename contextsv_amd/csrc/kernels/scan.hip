// scan.hip — kernel #1: wavefront-per-read CIGAR scan (gfx950).
//
// Replaces SVCaller::findCIGARSVs / processCIGARRecord (sv_caller.cpp:506-661) and
// getAlignmentReadPositions + bam_endpos (sv_caller.cpp:663-690) for one shard of reads.
//
// One 64-lane wave owns one read at a time. Each lane takes 4 consecutive packed CIGAR words
// (one 16-byte load, 1 KiB per wave instruction, aligned by starting at cigar_off & ~255 and
// masking the words that belong to the neighbouring reads), a wave prefix sum turns op lengths
// into reference / query cursors, and ops with len >= min_oplen of kind I / S / D become 16-byte
// signatures. Signatures are rare (~1e-3 of ops) so they are staged in a per-workgroup LDS buffer
// (slots reserved with an LDS compare-and-swap) and flushed with ONE global atomic per workgroup;
// a single hot global counter would otherwise serialise the chip. Emission order is arbitrary —
// the ordering pass (sort.hip) reproduces the reference's addSVCall order afterwards.
//
// HBM traffic per read: 4*n_cigar + 23 B in (pos 4, flag 2, mapq 1, cigar_off 2x8 shared) and
// 12 B out (ref_end, q_start, q_end) + 16 B per signature.
#include "../common.hpp"
#include "../devutil.hpp"

namespace csv {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_WAVES = SCAN_THREADS / WAVE;
constexpr uint32_t CAND_OPS = (1u << OP_I) | (1u << OP_D) | (1u << OP_S);
constexpr uint32_t SIG_BUF = 1024;           // signatures staged per workgroup (16 KiB LDS)

struct Chunk {
    uint32_t w[4];
};

// `base` = first word of the 1 KiB chunk (wave-uniform), lane l takes words base + 4l .. base + 4l + 3. Only the last chunk of
// the whole array can be partial, so the bounds test is done once per wave, not per lane.
__device__ __forceinline__ Chunk load_chunk(const uint32_t *__restrict__ cigar, uint64_t base, int lane, uint64_t n_cigar, int vec_ok)
{
    Chunk c;
    const uint64_t idx = base + (uint64_t)lane * 4;
    if (vec_ok && base + 4 * WAVE <= n_cigar) {
        uint4 v = *reinterpret_cast<const uint4 *>(cigar + idx);
        c.w[0] = v.x; c.w[1] = v.y; c.w[2] = v.z; c.w[3] = v.w;
    } else if (vec_ok && idx + 4 <= n_cigar) {
        uint4 v = *reinterpret_cast<const uint4 *>(cigar + idx);
        c.w[0] = v.x; c.w[1] = v.y; c.w[2] = v.z; c.w[3] = v.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) c.w[k] = (idx + k < n_cigar) ? cigar[idx + k] : (uint32_t)OP_P;
    }
    return c;
}

// first read index r in [0, n] with cigar_off[r] >= target (cigar_off is non-decreasing); 64-ary search by one wave
__device__ __forceinline__ uint64_t wave_lower_bound(const uint64_t *__restrict__ cigar_off, uint64_t n, uint64_t target, int lane)
{
    uint64_t lo = 0, hi = n;                 // answer in [lo, hi]
    while (hi - lo > 0) {
        const uint64_t span = hi - lo;
        const uint64_t step = (span + 63) / 64;
        const uint64_t probe = lo + (uint64_t)lane * step;          // probes lo, lo+step, ...
        const bool ge = probe >= hi || cigar_off[probe] >= target;  // monotone in lane
        const uint64_t m = __ballot(ge);
        const int first_ge = m ? __ffsll((long long)m) - 1 : 64;
        // answer lies in (probe[first_ge-1], probe[first_ge]]
        const uint64_t new_hi = first_ge < 64 ? min(lo + (uint64_t)first_ge * step, hi) : hi;
        const uint64_t new_lo = first_ge > 0 ? lo + (uint64_t)(first_ge - 1) * step + 1 : lo;
        if (first_ge == 0) return lo;
        lo = new_lo; hi = new_hi;
        if (step == 1) return hi;
    }
    return lo;
}

// lane `i` (wave-uniform) of a per-lane value, as a scalar
__device__ __forceinline__ uint32_t bcast32(uint32_t v, uint32_t i) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)i); }
__device__ __forceinline__ uint64_t bcast64(uint64_t v, uint32_t i)
{
    return ((uint64_t)bcast32((uint32_t)(v >> 32), i) << 32) | bcast32((uint32_t)v, i);
}

struct ScanMd {
    uint64_t c0, c1;
    uint32_t p0, fl, mq, unsorted;
};
__device__ __forceinline__ ScanMd scan_load_md(uint64_t r, uint64_t n_reads, const int32_t *__restrict__ pos,
                                               const uint16_t *__restrict__ flag, const uint8_t *__restrict__ mapq,
                                               const uint64_t *__restrict__ cigar_off)
{
    ScanMd m; m.c0 = 0; m.c1 = 0; m.p0 = 0; m.fl = 0; m.mq = 0; m.unsorted = 0;
    if (r < n_reads) {
        m.c0 = cigar_off[r]; m.c1 = cigar_off[r + 1];
        const int32_t p = pos[r];
        m.p0 = (uint32_t)p; m.fl = flag[r]; m.mq = mapq[r];
        m.unsorted = (r > 0 && p < pos[r - 1]) ? 1u : 0u;
    }
    return m;
}

// 6 waves/SIMD (80 VGPRs, a dozen spilled dwords on the signature path) instead of 5 at the natural 95: the walk is bound by
// occupancy
__global__ __launch_bounds__(SCAN_THREADS, 6) void cigar_scan_kernel(
    uint64_t n_reads, uint64_t n_cigar, const int32_t *__restrict__ pos, const uint16_t *__restrict__ flag,
    const uint8_t *__restrict__ mapq, const uint64_t *__restrict__ cigar_off, const uint32_t *__restrict__ cigar,
    int vec_ok, uint32_t depth_len, uint32_t start_limit, uint32_t min_oplen, uint32_t min_mapq, int emit,
    csv_sig *__restrict__ sig_out, uint64_t sig_cap, int32_t *__restrict__ ref_end, int32_t *__restrict__ q_start,
    int32_t *__restrict__ q_end, uint32_t *__restrict__ ckpt, ScanCounters *__restrict__ cnt)
{
    __shared__ csv_sig buf[SIG_BUF];
    __shared__ uint32_t buf_n, blk_n_del, blk_direct, blk_overflow;
    __shared__ unsigned long long blk_gbase;

    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { buf_n = 0; blk_n_del = 0; blk_direct = 0; blk_overflow = 0; }
    // (the __syncthreads() after the split-point search below also publishes these)

    uint32_t my_n_del = 0, my_overflow = 0;

    // Work split: read lengths are log-normal, so handing out reads round-robin leaves the slowest wave with ~1.5x the
    // mean work (measured: waves alive 65 % of the kernel). Instead every wave takes the CONTIGUOUS run of reads whose
    // first CIGAR word falls into its equal share of the word array — cigar_off is the cumulative word count, so the
    // split points are found with a 64-ary search done by the wave itself. Balanced to within one read, and the wave's
    // reads are contiguous in memory.
    const uint64_t n_waves = (uint64_t)gridDim.x * SCAN_WAVES;
    const uint64_t wave_gid = (uint64_t)blockIdx.x * SCAN_WAVES + wave;
    const uint64_t share = (n_cigar + n_waves - 1) / n_waves;
    // one search per wave (its own start); the end is the next wave's start, handed over through LDS
    __shared__ uint64_t split_s[SCAN_WAVES + 1];
    {
        const uint64_t b = wave_lower_bound(cigar_off, n_reads, wave_gid * share, lane);
        if (lane == 0) split_s[wave] = b;
        if (wave == SCAN_WAVES - 1) {
            const uint64_t e = (wave_gid + 1 >= n_waves) ? n_reads : wave_lower_bound(cigar_off, n_reads, (wave_gid + 1) * share, lane);
            if (lane == 0) split_s[SCAN_WAVES] = e;
        }
    }
    __syncthreads();
    const uint64_t r_begin = split_s[wave];
    const uint64_t r_end = split_s[wave + 1];

    // Per-read metadata is loaded for 64 reads at a time, lane-parallel and coalesced (lane l <-> read rb + l), and handed to
    // the wave with readlane; a wave's whole share is usually one batch. Inside the read loop only CIGAR chunk loads remain:
    // the next read's first chunk is requested when the current read reaches its last chunk.
    for (uint64_t rb = r_begin; rb < r_end; rb += WAVE) {
    const uint64_t nb_reads = min((uint64_t)WAVE, r_end - rb);
    uint64_t l_c0 = 0; uint32_t l_p0 = 0, l_flmq = 0, l_uns = 0;     // lane l: first CIGAR word, pos, flag | mapq << 16 of read rb + l
    if ((uint64_t)lane < nb_reads) {
        const uint64_t rr = rb + lane;
        l_c0 = cigar_off[rr];
        const int32_t p = pos[rr];
        l_p0 = (uint32_t)p; l_flmq = (uint32_t)flag[rr] | ((uint32_t)mapq[rr] << 16);
        l_uns = (rr > 0 && p < pos[rr - 1]) ? 1u : 0u;
    }
    const uint64_t batch_end_word = cigar_off[rb + nb_reads];        // wave-uniform
    if (__ballot(l_uns != 0) && lane == 0) cnt->unsorted = 1u;
    Chunk first = load_chunk(cigar, bcast64(l_c0, 0) & ~255ull, lane, n_cigar, vec_ok);
    for (uint32_t ri = 0; ri < (uint32_t)nb_reads; ri++) {
        const uint64_t r = rb + ri;
        const bool have_next = ri + 1 < (uint32_t)nb_reads;
        const uint64_t c0 = bcast64(l_c0, ri);
        const uint64_t c1 = have_next ? bcast64(l_c0, ri + 1) : batch_end_word;     // cigar_off[r + 1]
        const uint32_t p0 = bcast32(l_p0, ri);
        const uint32_t flmq = bcast32(l_flmq, ri);
        const uint32_t fl = flmq & 0xffffu, mq = flmq >> 16;
        const uint64_t next_base = c1 & ~255ull;
        bool next_first_issued = false;
        // sv_caller.cpp:526
        const bool emit_ok = emit && !(fl & (F_SECONDARY | F_UNMAP | F_DUP | F_QCFAIL | F_SUPP)) && mq >= min_mapq;

        // The common chunk does the minimum: decode, ONE DPP scan (reference
        // cursor: needed for the checkpoint and ref_end), lane-local query sums. Query cursors, skipped-clip
        // bookkeeping and signature assembly run only in chunks that contain an op >= min_oplen or that still
        // have to find query_start.
        uint32_t ref_carry = 0;      // reference bases consumed before this chunk (wave-uniform)
        uint32_t acc_q = 0;          // this lane's share of the query bases consumed before this chunk
        uint32_t acc_skip = 0;       // this lane's share of soft clips skipped by the `continue` at sv_caller.cpp:602-604
        int32_t  qs = -1;            // query_start

        const uint32_t n_words = (uint32_t)(c1 - c0);
        const uint64_t base = c0 & ~255ull;       // chunks are aligned to 256 words (1 KiB): a chunk boundary is a checkpoint slot
        int32_t rel = (int32_t)(base - c0) + lane * 4;   // index of this lane's first word relative to the read's first word
        Chunk cur = first;
        for (uint64_t chunk = base; chunk < c1; chunk += 4 * WAVE, rel += 4 * WAVE) {
            Chunk nxt;
            const bool more = chunk + 4 * WAVE < c1;
            if (more) nxt = load_chunk(cigar, chunk + 4 * WAVE, lane, n_cigar, vec_ok);   // prefetch next 1 KiB
            else if (have_next) {                                                                        // last chunk: next read's first
                first = load_chunk(cigar, next_base, lane, n_cigar, vec_ok);
                next_first_issued = true;
            }

            // only the first and the last chunk of a read can hold words of its neighbours: mask those to a no-op (P, length 0)
            if (!(chunk >= c0 && chunk + 4 * WAVE <= c1)) {
#pragma unroll
                for (int k = 0; k < 4; k++) if (!((uint32_t)(rel + k) < n_words)) cur.w[k] = (uint32_t)OP_P;   // also catches rel + k < 0
            }
            uint32_t len[4], op[4], rl[4], ql[4];
            uint32_t lane_ref = 0, lane_q = 0, max_cand = 0, qst = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                op[k] = cur.w[k] & 15u;
                len[k] = cur.w[k] >> 4;
                rl[k] = len[k] & (uint32_t)__builtin_amdgcn_sbfe((int)REF_OPS, op[k], 1u);     // all-ones mask when the op consumes the reference
                ql[k] = len[k] & (uint32_t)__builtin_amdgcn_sbfe((int)QRY_OPS, op[k], 1u);
                lane_ref += rl[k];
                lane_q += ql[k];
                max_cand = max(max_cand, len[k] & (uint32_t)__builtin_amdgcn_sbfe((int)CAND_OPS, op[k], 1u));   // only I / D / S ops can become signatures
            }
            // a conservative trigger for the slow path, which re-tests every op exactly
            const uint32_t big = max_cand >= min_oplen ? 1u : 0u;
            if (qs < 0) {
#pragma unroll
                for (int k = 0; k < 4; k++) qst |= (QST_OPS >> op[k]) & 1u;
            }
            const uint32_t incl_ref = wave_incl_sum_dpp(lane_ref);
            {   // checkpoints: the reference offset of this read at every CKPT_WORDS-th word strictly inside it (depth.hip starts its
                // walks there); one store instruction per chunk, lanes 0, 16, 32, 48
                const uint64_t w = chunk + (uint64_t)lane * 4;
                if ((lane & (CKPT_WORDS / 4 - 1)) == 0 && w > c0 && w < c1) ckpt[w >> CKPT_SHIFT] = ref_carry + (incl_ref - lane_ref);
            }
            const bool need_q = (qs < 0) || (emit_ok && __ballot(big != 0) != 0);
            if (need_q) {
                // ---------------- slow path: query cursors --------------------------------------------------
                const uint32_t q_carry = wave_total_dpp(acc_q);
                const uint32_t incl_q = wave_incl_sum_dpp(lane_q);
                const uint32_t rp = p0 + ref_carry + (incl_ref - lane_ref);   // reference `pos` before this lane's first op
                const uint32_t qp = q_carry + (incl_q - lane_q);              // plain query cursor before this lane's first op

                // query_start = cursor at the first M/I/=/X op (sv_caller.cpp:674-676)
                if (qs < 0) {
                    const uint64_t m = __ballot(qst != 0);
                    if (m) {
                        bool found = false;
                        uint32_t q_at = 0, acc = qp;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (!found && ((QST_OPS >> op[k]) & 1u)) { found = true; q_at = acc; }
                            acc += ql[k];
                        }
                        qs = (int32_t)__shfl(q_at, __ffsll((long long)m) - 1, 64);
                    }
                }

                if (emit_ok && __ballot(big != 0) != 0) {
                    // candidate ops: len >= min_oplen and I / S / D (sv_caller.cpp:566-643)
                    uint32_t cand = 0, skipped = 0, lane_skip = 0;
                    {
                        uint32_t rpk = rp;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (len[k] >= min_oplen) {
                                if (op[k] == OP_I || op[k] == OP_D) cand |= 1u << k;
                                else if (op[k] == OP_S) {
                                    if ((uint32_t)(rpk + 1u) >= depth_len) { skipped |= 1u << k; lane_skip += len[k]; }
                                    else cand |= 1u << k;
                                }
                            }
                            rpk += rl[k];
                        }
                    }
                    uint32_t skip_before = wave_total_dpp(acc_skip);
                    if (__ballot(skipped != 0)) {                 // rare: clip past the contig end
                        skip_before += wave_incl_sum_dpp(lane_skip) - lane_skip;
                        acc_skip += lane_skip;
                    }
                    if (__ballot(cand != 0)) {
                        uint32_t rpk = rp, qpk = qp, skk = skip_before;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const bool e = (cand >> k) & 1u;
                            const uint64_t m = __ballot(e);
                            if (m) {
                                const uint32_t n_e = (uint32_t)__popcll(m);
                                const uint32_t rank = (uint32_t)__popcll(m & lanemask_lt());
                                csv_sig sg;
                                sg.start = rpk + 1u;
                                sg.end = sg.start + len[k] - 1u;
                                sg.read = (uint32_t)r;
                                const uint32_t kind = (op[k] == OP_I) ? CSV_KIND_INS : (op[k] == OP_D ? CSV_KIND_DEL : CSV_KIND_CLIP);
                                sg.qpos_kind = ((qpk - skk) << 2) | kind;
                                // reserve n_e slots of the workgroup buffer (LDS CAS), else go straight to HBM
                                uint32_t slot = 0xffffffffu;
                                if (lane == 0) {
                                    uint32_t old = buf_n;
                                    while (old + n_e <= SIG_BUF) {
                                        const uint32_t seen = atomicCAS(&buf_n, old, old + n_e);
                                        if (seen == old) { slot = old; break; }
                                        old = seen;
                                    }
                                }
                                slot = __shfl(slot, 0, 64);
                                if (slot != 0xffffffffu) {
                                    if (e) buf[slot + rank] = sg;
                                } else {
                                    unsigned long long g = 0;
                                    if (lane == 0) g = atomicAdd(&cnt->n_sig, (unsigned long long)n_e);
                                    g = __shfl(g, 0, 64);
                                    if (e && g + rank < sig_cap) sig_out[g + rank] = sg;
                                    if (lane == 0) atomicAdd(&blk_direct, 1u);
                                }
                                if (e) {
                                    my_n_del += (kind == CSV_KIND_DEL);
                                    my_overflow |= (sg.start >= start_limit);      // start does not fit the sort key (never for sane input)
                                }
                            }
                            if ((skipped >> k) & 1u) skk += len[k];
                            rpk += rl[k];
                            qpk += ql[k];
                        }
                    }
                }
            }
            acc_q += lane_q;
            ref_carry += (uint32_t)__builtin_amdgcn_readlane((int)incl_ref, 63);
            if (more) cur = nxt;
        }
        if (have_next && !next_first_issued) first = load_chunk(cigar, next_base, lane, n_cigar, vec_ok);

        const uint32_t q_total = wave_total_dpp(acc_q);
        if (lane == 0) {
            // htslib bam_endpos: pos + rlen, rlen == 0 (or unmapped) -> 1
            uint32_t rlen = (fl & F_UNMAP) ? 0u : ref_carry;
            if (rlen == 0) rlen = 1;
            ref_end[r] = (int32_t)(p0 + rlen);
            q_start[r] = qs < 0 ? 0 : qs;
            q_end[r] = (int32_t)q_total;
        }
    }
    }


    if (!emit) return;
    // workgroup epilogue: ONE global atomic per workgroup reserves the output range of the LDS buffer (the DEL count rides
    // in a second word only when non-zero). Every extra same-line atomic per workgroup costs tens of microseconds chip-wide.
    my_n_del = wave_sum(my_n_del);
    const bool wave_overflow = __ballot(my_overflow != 0) != 0;
    if (lane == 0) {
        if (my_n_del) atomicAdd(&blk_n_del, my_n_del);
        if (wave_overflow) atomicOr(&blk_overflow, 1u);
    }
    __syncthreads();
    const uint32_t nb = buf_n;
    if (threadIdx.x == 0) {
        blk_gbase = nb ? atomicAdd(&cnt->n_sig, (unsigned long long)nb) : 0ull;
        if (blk_n_del) atomicAdd(&cnt->n_del, (unsigned long long)blk_n_del);
        if (blk_overflow) cnt->max_start = 0xffffffffu;
    }
    __syncthreads();
    const unsigned long long g = blk_gbase;
    for (uint32_t i = threadIdx.x; i < nb; i += SCAN_THREADS)
        if (g + i < sig_cap) sig_out[g + i] = buf[i];
}

// Signature starts are < depth_len for coordinate-sorted input (start = pos + 1 <= contig length); the ordering pass sizes its
// radix keys from this bound, and the scan raises ScanCounters::max_start to 0xffffffff if a start ever exceeds it.
uint32_t scan_start_limit(uint32_t depth_len)
{
    int b = bits_of(depth_len);
    if (b < 8) b = 8;
    return b >= 32 ? 0xffffffffu : (1u << b);
}

void launch_cigar_scan(hipStream_t s, int n_cu, const csv_reads &d, uint32_t depth_len, uint32_t min_oplen,
                       uint32_t min_mapq, int emit, csv_sig *sig_out, uint64_t sig_cap,
                       int32_t *ref_end, int32_t *q_start, int32_t *q_end, uint32_t *ckpt, ScanCounters *cnt)
{
    if (d.n_reads == 0) return;
    // persistent-style grid: exactly as many workgroups as are resident at once (waves stride over the reads),
    // so there is no partially filled second round of workgroups
    static int blocks_per_cu = 0;
    if (blocks_per_cu == 0) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, cigar_scan_kernel, SCAN_THREADS, 0) != hipSuccess || occ <= 0) occ = 4;
        blocks_per_cu = occ;
    }
    uint64_t want = (d.n_reads + SCAN_WAVES - 1) / SCAN_WAVES;
    // two rounds of workgroups measured best (0.237 vs 0.247 ms at one round: the second round evens out the tail)
    uint64_t cap = (uint64_t)n_cu * blocks_per_cu * 2;
    unsigned grid = (unsigned)(want < cap ? want : cap);
    const int vec_ok = (((uintptr_t)d.cigar) & 15u) == 0;
    hipLaunchKernelGGL(cigar_scan_kernel, dim3(grid), dim3(SCAN_THREADS), 0, s, d.n_reads, d.n_cigar, d.pos, d.flag,
                       d.mapq, d.cigar_off, d.cigar, vec_ok, depth_len, scan_start_limit(depth_len), min_oplen, min_mapq, emit, sig_out, sig_cap,
                       ref_end, q_start, q_end, ckpt, cnt);
}

}  // namespace csv
