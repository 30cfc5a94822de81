// scan.hip — kernel #1: wavefront-per-read CIGAR scan (gfx950).
//
// Replaces SVCaller::findCIGARSVs / processCIGARRecord (sv_caller.cpp:506-661) and
// getAlignmentReadPositions + bam_endpos (sv_caller.cpp:663-690) for one shard of reads.
//
// One 64-lane wave owns one read at a time. Each lane takes 4 consecutive packed CIGAR words of a 1 KiB chunk (chunks are aligned
// to 256 words; the words of a boundary chunk that belong to the neighbouring reads are masked), a wave prefix sum turns op
// lengths into reference / query cursors, and ops with len >= min_oplen of kind I / S / D become 16-byte signatures.
//
// The CIGAR words reach the wave through a per-wave LDS ring filled by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave
// instruction, no VGPR destination). A wave's reads are contiguous in the word array, so its loads are one sequential stream of
// chunks; RING_D chunks are kept in flight per wave at no register cost (one chunk ahead in registers left the walk bound by
// load latency x occupancy: 6 waves/SIMD x 1 KiB in flight), and a chunk shared by two reads is fetched once and read twice
// from LDS (it used to be fetched per read: +22 % load instructions at 1.1 k words per read).
//
// Signatures are rare (~1e-3 of ops) so they are staged in a per-workgroup LDS buffer
// (each emitting lane reserves its slot with an LDS atomic) and flushed with ONE global atomic per workgroup;
// a single hot global counter would otherwise serialise the chip. Emission order is arbitrary —
// the ordering pass (sort.hip) reproduces the reference's addSVCall order afterwards.
//
// HBM traffic per read: 4*n_cigar + 23 B in (pos 4, flag 2, mapq 1, cigar_off 2x8 shared) and
// 12 B out (ref_end, q_start, q_end) + 16 B per signature.
#include <type_traits>

#include "../common.hpp"
#include "../devutil.hpp"

namespace csv {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_WAVES = SCAN_THREADS / WAVE;
constexpr uint32_t SIG_BUF = 512;            // signatures staged per workgroup (8 KiB LDS); more go straight to HBM
constexpr int RING_D = 4;                    // 1 KiB chunks in flight per wave (16 KiB LDS per workgroup)
constexpr int SCAN_OCC = 6;                  // workgroups per CU the LDS allows. 8 (RING_D 3, SIG_BUF 256) measured the same: the walk is
                                             // bound by instruction issue (VALU 60 %, scalar unit 57 % busy), not by latency or occupancy
constexpr int CHUNK_WORDS = 4 * WAVE;

struct Chunk {
    uint32_t w[4];
};

// LDS-DMA: lane l's 16 bytes at gbase + lane_byte_off land at lds_dst + 16 l (lds_dst wave-uniform, carried in M0). hipcc does not count an asm
// load in its own s_waitcnt bookkeeping, which is the point: the consumer waits with a counted vmcnt (ring_wait) instead of the
// vmcnt(0) hipcc puts in front of every LDS read while a __builtin_amdgcn_global_load_lds is outstanding.
__device__ __forceinline__ void glds16(const uint32_t *gbase, uint32_t lane_byte_off, uint32_t lds_dst)      // gbase, lds_dst wave-uniform
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_byte_off), "s"(gbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p; }

// Fill one ring slot with the chunk starting at word `base` (wave-uniform, multiple of 256). Only the last chunk of the whole
// array can be partial (or the base pointer unaligned, never for our own uploads): those go through registers, padded with P ops.
__device__ __forceinline__ void ring_issue(uint32_t *slot, const uint32_t *__restrict__ cigar, uint64_t base, int lane, uint64_t n_cigar, int vec_ok)
{
    const uint64_t idx = base + (uint64_t)lane * 4;
    if (vec_ok && base + CHUNK_WORDS <= n_cigar) {
        glds16(cigar + base, (uint32_t)lane * 16u, (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(slot)));
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) slot[lane * 4 + k] = (idx + k < n_cigar) ? cigar[idx + k] : (uint32_t)OP_P;
    }
}

// Loads return in issue order, so "at most RING_D - 1 vector-memory operations outstanding" means the DMA issued RING_D - 1
// fills ago has landed (stores in between only make the wait more conservative).
__device__ __forceinline__ void ring_wait_steady()
{
    static_assert(RING_D >= 2 && RING_D <= 8, "add the immediate for RING_D - 1");
    if (RING_D == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    if (RING_D == 7) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    if (RING_D == 6) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    if (RING_D == 5) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (RING_D == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    if (RING_D == 3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    if (RING_D == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
}
__device__ __forceinline__ void ring_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

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
// 54 VGPRs; the LDS (ring + signature buffer) sets the occupancy
__global__ __launch_bounds__(SCAN_THREADS, SCAN_OCC) void cigar_scan_kernel(
    uint64_t n_reads, uint64_t n_cigar, const int32_t *__restrict__ pos, const uint16_t *__restrict__ flag,
    const uint8_t *__restrict__ mapq, const uint64_t *__restrict__ cigar_off, const uint32_t *__restrict__ cigar,
    int vec_ok, uint32_t depth_len, uint32_t start_limit, uint32_t min_oplen, uint32_t min_mapq, int emit,
    csv_sig *__restrict__ sig_out, uint64_t sig_cap, int32_t *__restrict__ ref_end, int32_t *__restrict__ q_start,
    int32_t *__restrict__ q_end, uint32_t *__restrict__ ckpt, ScanCounters *__restrict__ cnt,
    unsigned long long *__restrict__ tile_range, uint32_t n_tiles, uint32_t *__restrict__ bucket_hist, int hist_type_pos, int hist_shift,
    const uint64_t *__restrict__ split)      // scan_split_kernel's table for this grid (resident shards: first read and its first word per wave), or null
{
    __shared__ csv_sig buf[SIG_BUF];
    __shared__ alignas(16) uint32_t ring[SCAN_WAVES][RING_D][CHUNK_WORDS];      // 16-byte aligned: LDS-DMA destination and ds_read_b128 source
    __shared__ uint32_t buf_n, blk_n_del, blk_overflow;
    __shared__ unsigned long long blk_gbase;

    const int lane = lane_id();
    const int wave = (int)uniform32(threadIdx.x >> 6);
    if (threadIdx.x == 0) { buf_n = 0; blk_n_del = 0; blk_overflow = 0; }
    // (the __syncthreads() after the split-point search below also publishes these)

    uint32_t my_n_del = 0, my_overflow = 0, my_bucket_max = 0;

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
    __shared__ uint64_t split_w[SCAN_WAVES + 1];
    if (split) {
        // the split points depend on the shard and the grid only: a resident shard has them in a table, with the first word of each (five
        // dependent round trips to HBM per wave in front of its first CIGAR word otherwise)
        if (threadIdx.x <= SCAN_WAVES) {
            const uint64_t g = (uint64_t)blockIdx.x * SCAN_WAVES + threadIdx.x;
            split_s[threadIdx.x] = split[2 * g]; split_w[threadIdx.x] = split[2 * g + 1];
        }
    } else {
        const uint64_t b = wave_lower_bound(cigar_off, n_reads, wave_gid * share, lane);
        if (lane == 0) split_s[wave] = b;
        if (wave == SCAN_WAVES - 1) {
            const uint64_t e = (wave_gid + 1 >= n_waves) ? n_reads : wave_lower_bound(cigar_off, n_reads, (wave_gid + 1) * share, lane);
            if (lane == 0) split_s[SCAN_WAVES] = e;
        }
    }
    __syncthreads();
    const uint64_t r_begin = uniform64(split_s[wave]);
    const uint64_t r_end = uniform64(split_s[wave + 1]);

    // This wave's load stream: the chunks [ring_base, ring_base + 256 ring_total) cover its reads' words.
    const uint64_t w_begin = split ? uniform64(split_w[wave]) : uniform64(r_begin < r_end ? cigar_off[r_begin] : 0);
    const uint64_t w_end = split ? uniform64(split_w[wave + 1]) : uniform64(r_begin < r_end ? cigar_off[r_end] : 0);
    const uint64_t ring_base = w_begin & ~(uint64_t)(CHUNK_WORDS - 1);
    const uint32_t ring_total = r_begin < r_end ? (uint32_t)((w_end - ring_base + CHUNK_WORDS - 1) / CHUNK_WORDS) : 0u;
    uint32_t ring_ready = 0;              // chunks [0, ring_ready) have landed; chunks [ring_ready, ring_ready + RING_D - 1) are in flight
    for (uint32_t a = 0; a < (uint32_t)(RING_D - 1) && a < ring_total; a++)
        ring_issue(ring[wave][a], cigar, ring_base + (uint64_t)a * CHUNK_WORDS, lane, n_cigar, vec_ok);

    // Per-read metadata is loaded for 64 reads at a time, lane-parallel and coalesced (lane l <-> read rb + l), and handed to
    // the wave with readlane; a wave's whole share is usually one batch. Inside the read loop only the ring traffic remains.
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
    const uint64_t batch_end_word = uniform64(cigar_off[rb + nb_reads]);
    if (__ballot(l_uns != 0) && lane == 0) cnt->unsorted = 1u;
    for (uint32_t ri = 0; ri < (uint32_t)nb_reads; ri++) {
        const uint64_t r = rb + ri;
        const bool have_next = ri + 1 < (uint32_t)nb_reads;
        const uint64_t c0 = bcast64(l_c0, ri);
        const uint64_t c1 = have_next ? bcast64(l_c0, ri + 1) : batch_end_word;     // cigar_off[r + 1]
        const uint32_t p0 = bcast32(l_p0, ri);
        const uint32_t flmq = bcast32(l_flmq, ri);
        const uint32_t fl = flmq & 0xffffu, mq = flmq >> 16;
        // sv_caller.cpp:526
        const bool emit_ok = emit && !(fl & (F_SECONDARY | F_UNMAP | F_DUP | F_QCFAIL | F_SUPP)) && mq >= min_mapq;

        // The common chunk does the minimum: decode, ONE DPP scan (reference
        // cursor: needed for the checkpoint and ref_end), lane-local query sums. Query cursors, skipped-clip
        // bookkeeping and signature assembly run only in chunks that contain an op >= min_oplen or that still
        // have to find query_start.
        uint32_t ref_carry = 0;      // reference bases consumed before this chunk (wave-uniform)
        uint32_t acc_q = 0;          // this lane's share of the query bases consumed before this chunk
        uint32_t acc_skip = 0;       // this lane's share of soft clips skipped by the `continue` at sv_caller.cpp:602-604
        bool had_skip = false;       // ... any so far (wave-uniform)
        int32_t  qs = -1;            // query_start (wave-uniform, kept in a scalar register)

        // Chunks are aligned to 256 words (1 KiB; a chunk boundary is a checkpoint slot). Inside a read everything is 32-bit and
        // relative to its first word (csv_reads validation bounds a read's word count): chunk t of the read holds the
        // read-relative words [256 t - head, 256 t - head + 256).
        const uint32_t n_words = (uint32_t)(c1 - c0);
        const uint32_t head = (uint32_t)c0 & (uint32_t)(CHUNK_WORDS - 1);          // words of the first chunk that belong to earlier reads
        const uint32_t n_chunks = n_words ? (head + n_words + CHUNK_WORDS - 1) / CHUNK_WORDS : 0u;
        const uint32_t j0 = (uint32_t)((c0 - head - ring_base) / CHUNK_WORDS);     // the first chunk's index in this wave's stream
        uint32_t *__restrict__ const ck_read = ckpt + ((c0 - head) >> CKPT_SHIFT);
        int32_t rel = lane * 4 - (int32_t)head;          // index of this lane's first word relative to the read's first word
        for (uint32_t t = 0; t < n_chunks; t++, rel += CHUNK_WORDS) {
            // chunk j of the stream: top the ring up (chunks before j are dead, so slots (j, j + RING_D) mod RING_D are free), wait for j
            const uint32_t j = j0 + t;
            if (j == ring_ready) {                // first visit (chunks are visited in stream order; a boundary chunk is visited again by the next read)
                const uint32_t ahead = j + (RING_D - 1);
                if (ahead < ring_total) {
                    ring_issue(ring[wave][ahead % RING_D], cigar, ring_base + (uint64_t)ahead * CHUNK_WORDS, lane, n_cigar, vec_ok);
                    ring_wait_steady();           // chunks j + 1 .. j + RING_D - 1 may still be in flight
                } else {
                    ring_wait_all();
                }
                ring_ready = j + 1;
            }
            Chunk cur;
            {
                const uint4 v = *reinterpret_cast<const uint4 *>(&ring[wave][j % RING_D][lane * 4]);
                cur.w[0] = v.x; cur.w[1] = v.y; cur.w[2] = v.z; cur.w[3] = v.w;
            }

            // only the first and the last chunk of a read can hold words of its neighbours: mask those to a no-op (P, length 0)
            const bool edge = (t == 0 && head != 0) || (t + 1) * CHUNK_WORDS - head > n_words;
            if (edge) {
#pragma unroll
                for (int k = 0; k < 4; k++) if (!((uint32_t)(rel + k) < n_words)) cur.w[k] = (uint32_t)OP_P;   // also catches rel + k < 0
            }
            uint32_t len[4], op[4], rl[4], ql[4];
            uint32_t lane_ref = 0, lane_q = 0, trig = 0, qst = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                op[k] = cur.w[k] & 15u;                  // (only the slow paths read op[])
                len[k] = cur.w[k] >> 4;
                // all-ones mask when the op consumes the reference / the query. v_bfe_i32 takes its bit offset from the low FIVE bits
                // of the word — the op and the length's lowest bit — so the op masks are repeated in the upper half-word.
                rl[k] = len[k] & (uint32_t)__builtin_amdgcn_sbfe((int)(REF_OPS | (REF_OPS << 16)), cur.w[k], 1u);
                ql[k] = len[k] & (uint32_t)__builtin_amdgcn_sbfe((int)(QRY_OPS | (QRY_OPS << 16)), cur.w[k], 1u);
                lane_ref += rl[k];
                lane_q += ql[k];
                trig |= rl[k] ^ ql[k];                   // the length of an op that consumes exactly one of the two: I, S, D (and N)
            }
            // a conservative trigger for the slow path, which re-tests every op exactly: an OR is at least its largest operand
            const bool big = trig >= min_oplen;
            if (qs < 0) {
#pragma unroll
                for (int k = 0; k < 4; k++) qst |= (QST_OPS >> op[k]) & 1u;
            }
            const uint32_t incl_ref = wave_incl_sum_dpp(lane_ref);
            {   // checkpoints: the reference offset of this read at every CKPT_WORDS-th word strictly inside it (depth.hip starts its
                // walks there); one store instruction per chunk, lanes 0, 16, 32, 48
                static_assert(CKPT_WORDS == 64, "one checkpoint per DPP row of 16 lanes x 4 words");
                uint32_t *__restrict__ ck = ck_read + t * (CHUNK_WORDS / CKPT_WORDS);
                const bool inside = (edge || t == 0) ? (rel > 0 && (uint32_t)rel < n_words) : true;       // wave-uniform choice of the test
                if ((lane & 15) == 0 && inside) ck[lane >> 4] = ref_carry + (incl_ref - lane_ref);
            }
            const bool any_big = emit_ok && __ballot(big) != 0;
            const bool need_q = (qs < 0) || any_big;
            if (need_q) {
                // ---------------- slow path: query cursors --------------------------------------------------
                const uint32_t q_carry = t ? wave_total_dpp(acc_q) : 0u;     // (nothing consumed before the first chunk)
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
                        qs = (int32_t)bcast32(q_at, (uint32_t)(__ffsll((long long)m) - 1));
                    }
                }

                if (any_big) {
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
                    uint32_t skip_before = had_skip ? wave_total_dpp(acc_skip) : 0u;
                    if (__ballot(skipped != 0)) {                 // rare: clip past the contig end
                        skip_before += wave_incl_sum_dpp(lane_skip) - lane_skip;
                        acc_skip += lane_skip;
                        had_skip = true;
                    }
                    if (__ballot(cand != 0)) {
                        uint32_t rpk = rp, qpk = qp, skk = skip_before;
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if ((cand >> k) & 1u) {               // few lanes, usually one: each reserves its own slot of the workgroup buffer
                                csv_sig sg;
                                sg.start = rpk + 1u;
                                sg.end = sg.start + len[k] - 1u;
                                sg.read = (uint32_t)r;
                                const uint32_t kind = (op[k] == OP_I) ? CSV_KIND_INS : (op[k] == OP_D ? CSV_KIND_DEL : CSV_KIND_CLIP);
                                sg.qpos_kind = ((qpk - skk) << 2) | kind;
                                const uint32_t slot = atomicAdd(&buf_n, 1u);
                                if (slot < SIG_BUF) buf[slot] = sg;
                                else {                            // buffer full: straight to HBM (buf_n keeps counting; the epilogue clamps it)
                                    const unsigned long long g = atomicAdd(&cnt->n_sig, 1ull);
                                    if (g < sig_cap) sig_out[g] = sg;
                                    if (bucket_hist) my_bucket_max = max(my_bucket_max, atomicAdd(&bucket_hist[bk_bucket(sg, hist_type_pos, hist_shift)], 1u) + 1u);
                                }
                                my_n_del += (kind == CSV_KIND_DEL);
                                my_overflow |= (sg.start >= start_limit);          // start does not fit the sort key (never for sane input)
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
        }

        // Depth tiles this read reaches (ScanExtras::tile_range): the depth pass starts from these ranges, so no search kernel
        // has to run between the two. Same filter as the depth pass (cnv_caller.cpp:491-495); positions as there, in uint32.
        if (tile_range && ref_carry != 0 && !(fl & (F_UNMAP | F_SECONDARY | F_QCFAIL | F_DUP))) {
            const uint32_t first = p0 + 1u, last = first + ref_carry - 1u;
            const uint32_t t0 = first >> DEPTH_TILE_SHIFT;
            if (t0 < n_tiles && last >= first) {
                const uint32_t t1 = min(last >> DEPTH_TILE_SHIFT, n_tiles - 1u);
                for (uint32_t t = t0 + (uint32_t)lane; t <= t1; t += WAVE) {
                    atomicMax(&tile_range[2 * (uint64_t)t], ~(unsigned long long)r);
                    atomicMax(&tile_range[2 * (uint64_t)t + 1], (unsigned long long)r + 1ull);
                }
            }
        }

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
    ring_wait_all();        // nothing may still be in flight towards this workgroup's LDS when it retires

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
    const uint32_t nb = min(buf_n, SIG_BUF);
    if (threadIdx.x == 0) {
        blk_gbase = nb ? atomicAdd(&cnt->n_sig, (unsigned long long)nb) : 0ull;
        if (blk_n_del) atomicAdd(&cnt->n_del, (unsigned long long)blk_n_del);
        if (blk_overflow) cnt->max_start = 0xffffffffu;
    }
    __syncthreads();
    const unsigned long long g = blk_gbase;
    for (uint32_t i = threadIdx.x; i < nb; i += SCAN_THREADS) {
        const csv_sig sg = buf[i];
        if (g + i < sig_cap) sig_out[g + i] = sg;
        // the ordering pass's bucket counts, taken here so that no histogram kernel runs between the scan and the depth pass
        if (bucket_hist) my_bucket_max = max(my_bucket_max, atomicAdd(&bucket_hist[bk_bucket(sg, hist_type_pos, hist_shift)], 1u) + 1u);
    }
    if (bucket_hist) {
        // The host only needs to know whether a bucket outgrew what one wave ranks (BK_LOCAL_MAX: then the radix path orders the
        // signatures), so nothing is written for ordinary input: twelve thousand waves queueing an atomicMax on the counters' cache
        // line held up the epilogues' slot reservations on the same line (a plain look at the value first was worse: 2x the kernel).
        my_bucket_max = wave_max(my_bucket_max);
        if (lane == 0 && my_bucket_max > BK_LOCAL_MAX) atomicMax(&cnt->max_len, my_bucket_max);
    }
}

// ------------------------------------------------------------------------------------------------------------- short-read form
// The same walk for shards of SHORT reads (PacBio HiFi: ~40 ops per read; BASELINE configs[4]): a wave per read spends a whole 256-word
// visit on 40 words and pays the per-read prologue once per 160 bytes. Here a GROUP of GL lanes (8 or 16) owns a read, 64 / GL reads
// are in flight per wave, a window is 64 words (64 / GL per lane) starting at the read's first word rounded down to 16 bytes — a
// 40-word read is one window — and every group runs through ITS reads at its own pace (the groups of a wave share nothing but the
// instruction stream): per-read state lives in vector registers, the cursors are group prefix sums (DPP row shifts; row_newbcast for
// the totals), per-read metadata reaches the groups through a 1 KiB LDS table per wave (one 16-byte broadcast read per read), and the
// words of a group's NEXT window — the next read's first, usually — are requested before the current one is decoded.
// Same outputs as cigar_scan_kernel, word for word (signatures as a set: the ordering pass fixes the order).
// The depth tiles' candidate ranges go through a per-workgroup LDS table first (a workgroup's reads reach a handful of tiles, and at
// 60x every tile is reached by a hundred reads: two same-address device atomics per (read, tile) otherwise).
constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / WAVE;
constexpr int rs_occ(int gl) { return gl == 8 ? 5 : 7; }    // workgroups per CU the registers allow (8 words per lane and their prefetch: 96 registers)
constexpr uint32_t TR_SLOTS = 64;            // depth tiles (from the workgroup's first read's tile on) whose ranges are combined in LDS
constexpr uint32_t RS_MIN_READS = 32;        // reads per wave below which the grid shrinks instead

template <int GL, bool PADDED>
__global__ __launch_bounds__(RS_THREADS, rs_occ(GL)) void cigar_scan_rows_kernel(
    uint64_t n_reads, uint64_t n_cigar, const int32_t *__restrict__ pos, const uint16_t *__restrict__ flag,
    const uint8_t *__restrict__ mapq, const uint64_t *__restrict__ cigar_off, const uint32_t *__restrict__ cigar,
    uint32_t depth_len, uint32_t start_limit, uint32_t min_oplen, uint32_t min_mapq, int emit,
    csv_sig *__restrict__ sig_out, uint64_t sig_cap, int32_t *__restrict__ ref_end, int32_t *__restrict__ q_start,
    int32_t *__restrict__ q_end, uint32_t *__restrict__ ckpt, ScanCounters *__restrict__ cnt,
    unsigned long long *__restrict__ tile_range, uint32_t n_tiles, uint32_t *__restrict__ bucket_hist, int hist_type_pos, int hist_shift,
    const uint64_t *__restrict__ split)
{
    constexpr int WPL = WAVE / GL;            // words per lane per window
    constexpr int NG = WAVE / GL;             // groups = reads in flight per wave
    __shared__ csv_sig buf[SIG_BUF];
    __shared__ alignas(16) uint4 meta[RS_WAVES][WAVE];               // {first word, words, pos, flag | mapq << 16} of the wave's current 64 reads
    __shared__ uint32_t tr_first[TR_SLOTS], tr_last[TR_SLOTS];
    __shared__ uint32_t buf_n, blk_n_del, blk_overflow, tr_base;
    __shared__ unsigned long long blk_gbase;
    __shared__ uint64_t split_s[RS_WAVES + 1];

    const int lane = lane_id();
    const int wave = (int)uniform32(threadIdx.x >> 6);
    const uint32_t sub = (uint32_t)lane & (uint32_t)(GL - 1);       // lane within its group
    const uint32_t grp = (uint32_t)lane / (uint32_t)GL;
    const uint32_t upper = (lane & 8) ? 0xffffffffu : 0u;
    if (threadIdx.x == 0) { buf_n = 0; blk_n_del = 0; blk_overflow = 0; }
    if (threadIdx.x < TR_SLOTS) { tr_first[threadIdx.x] = 0; tr_last[threadIdx.x] = 0; }

    const uint64_t n_waves = (uint64_t)gridDim.x * RS_WAVES;
    const uint64_t wave_gid = (uint64_t)blockIdx.x * RS_WAVES + wave;
    const uint64_t share = (n_cigar + n_waves - 1) / n_waves;
    if (split) {
        if (threadIdx.x <= RS_WAVES) split_s[threadIdx.x] = split[2 * ((uint64_t)blockIdx.x * RS_WAVES + threadIdx.x)];
    } else {
        const uint64_t b = wave_lower_bound(cigar_off, n_reads, wave_gid * share, lane);
        if (lane == 0) split_s[wave] = b;
        if (wave == RS_WAVES - 1) {
            const uint64_t e = (wave_gid + 1 >= n_waves) ? n_reads : wave_lower_bound(cigar_off, n_reads, (wave_gid + 1) * share, lane);
            if (lane == 0) split_s[RS_WAVES] = e;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint64_t r0 = split_s[0];
        tr_base = (tile_range && r0 < n_reads) ? (((uint32_t)pos[r0] + 1u) >> DEPTH_TILE_SHIFT) : 0u;
    }
    __syncthreads();
    const uint64_t r_begin = uniform64(split_s[wave]);
    const uint64_t r_end = uniform64(split_s[wave + 1]);
    const uint32_t tbase = tr_base;

    uint32_t my_n_del = 0, my_overflow = 0, my_bucket_max = 0;

    // one window's words of this lane: words [g, g + WPL) of the array (g a multiple of 4)
    auto loadw = [&](uint32_t g, uint32_t (&w)[WPL]) {
        if (PADDED) {
#pragma unroll
            for (int q = 0; q < WPL; q += 4) {
                const uint4 x = *reinterpret_cast<const uint4 *>(cigar + g + q);
                w[q] = x.x; w[q + 1] = x.y; w[q + 2] = x.z; w[q + 3] = x.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < WPL; k++) w[k] = ((uint64_t)g + k < n_cigar) ? cigar[g + k] : (uint32_t)OP_P;
        }
    };

    for (uint64_t rb = r_begin; rb < r_end; rb += WAVE) {
        const uint32_t nb = (uint32_t)min((uint64_t)WAVE, r_end - rb);
        uint32_t l_p0 = 0, l_fl = 0;      // lane l: pos and flag of read rb + l (for the batch's write-out)
        {   // the batch's metadata, lane-parallel and coalesced
            uint4 m = make_uint4(0u, 0u, 0u, 0u);
            uint32_t l_uns = 0;
            if ((uint32_t)lane < nb) {
                const uint64_t rr = rb + lane;
                const uint64_t c0 = cigar_off[rr], c1 = cigar_off[rr + 1];
                const int32_t p = pos[rr];
                m = make_uint4((uint32_t)c0, (uint32_t)(c1 - c0), (uint32_t)p, (uint32_t)flag[rr] | ((uint32_t)mapq[rr] << 16));
                l_p0 = m.z; l_fl = m.w & 0xffffu;
                l_uns = (rr > 0 && p < pos[rr - 1]) ? 1u : 0u;
            }
            if (__ballot(l_uns != 0) && lane == 0) cnt->unsorted = 1u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            meta[wave][lane] = m;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // group `grp` takes the batch's reads grp, grp + NG, ...; every loop step is one window of every group's current read
        uint32_t slot = grp;
        bool act = slot < nb;
        uint4 cm = meta[wave][act ? slot : 0u];
        uint32_t v = 0;                        // window of the current read
        uint32_t ref_carry = 0, q_carry = 0, skip_carry = 0;
        int32_t qs = -1;
        uint32_t w[WPL];
        loadw(act ? (cm.x & ~3u) + sub * WPL : 0u, w);
        while (__ballot(act)) {
            const uint32_t c0 = cm.x, nw = act ? cm.y : 0u, p0 = cm.z, fl = cm.w & 0xffffu, mq = cm.w >> 16;
            const uint32_t a = c0 & ~3u;
            const uint32_t n_win = max(1u, (c0 - a + nw + 63u) >> 6);
            const bool last = v + 1 >= n_win;
            // the group's next window: the next one of this read, or the first one of its next read
            const uint32_t nslot = last ? slot + NG : slot;
            const bool nact = act && nslot < nb;
            uint4 nm = cm;
            if (last) nm = meta[wave][nact ? nslot : 0u];
            const uint32_t nv = last ? 0u : v + 1u;
            uint32_t wn[WPL];
            loadw(nact ? (nm.x & ~3u) + nv * 64u + sub * WPL : 0u, wn);

            // ---- window v of the current read ------------------------------------------------------------------------
            const uint32_t g = a + v * 64u + sub * WPL;                  // this lane's first word
            const uint32_t t = g - c0;                                   // ... relative to the read's first word (wraps in front of it)
            // sv_caller.cpp:526
            const bool emit_ok = emit && act && !(fl & (F_SECONDARY | F_UNMAP | F_DUP | F_QCFAIL | F_SUPP)) && mq >= min_mapq;
            uint32_t lane_ref = 0, lane_q = 0, trig = 0;
#pragma unroll
            for (int k = 0; k < WPL; k++) {
                if (!((uint32_t)(t + k) < nw)) w[k] = (uint32_t)OP_P;    // words of the neighbouring reads (and everything of an idle group)
                const uint32_t len = w[k] >> 4;
                const uint32_t rl = len & (uint32_t)__builtin_amdgcn_sbfe((int)(REF_OPS | (REF_OPS << 16)), w[k], 1u);
                const uint32_t ql = len & (uint32_t)__builtin_amdgcn_sbfe((int)(QRY_OPS | (QRY_OPS << 16)), w[k], 1u);
                lane_ref += rl; lane_q += ql;
                trig |= rl ^ ql;
            }
            const uint32_t incl_ref = grp_incl_sum<GL>(lane_ref, upper);
            const uint32_t incl_q = grp_incl_sum<GL>(lane_q, upper);
            const uint32_t tot_ref = grp_last<GL>(incl_ref), tot_q = grp_last<GL>(incl_q);
            const uint32_t ro = ref_carry + (incl_ref - lane_ref);       // reference offset in front of this lane's first word
            const uint32_t qp = q_carry + (incl_q - lane_q);             // plain query cursor there
            // checkpoints: the read's reference offset at every CKPT_WORDS-th word strictly inside it (a lane's words start at a multiple of 4)
            {
                if ((g & (uint32_t)(CKPT_WORDS - 1)) == 0 && t - 1u < nw - 1u && nw) ckpt[g >> CKPT_SHIFT] = ro;
                if (WPL == 8 && ((g + 4u) & (uint32_t)(CKPT_WORDS - 1)) == 0 && t + 3u < nw - 1u && nw) {
                    uint32_t x = ro;
#pragma unroll
                    for (int k = 0; k < 4; k++) x += (w[k] >> 4) & (uint32_t)__builtin_amdgcn_sbfe((int)(REF_OPS | (REF_OPS << 16)), w[k], 1u);
                    ckpt[(g + 4u) >> CKPT_SHIFT] = x;
                }
            }
            // query_start = cursor at the first M/I/=/X op (sv_caller.cpp:674-676)
            if (__ballot(qs < 0 && act)) {
                // (cursors grow with the lane: the group's minimum over the lanes that hold such an op is the first one's)
                uint32_t q_at = 0xffffffffu, acc = qp;
                bool found = false;
#pragma unroll
                for (int k = 0; k < WPL; k++) {
                    const uint32_t op = w[k] & 15u;
                    if (!found && ((QST_OPS >> op) & 1u)) { found = true; q_at = acc; }
                    acc += (w[k] >> 4) & (uint32_t)__builtin_amdgcn_sbfe((int)(QRY_OPS | (QRY_OPS << 16)), w[k], 1u);
                }
                const uint32_t first = grp_min<GL>(q_at);
                if (qs < 0 && first != 0xffffffffu) qs = (int32_t)first;
            }
            uint32_t skip_tot = 0;
            if (__ballot(emit_ok && trig >= min_oplen)) {
                // candidate ops: len >= min_oplen and I / S / D (sv_caller.cpp:566-643). A soft clip whose position is at or beyond the contig's
                // end is skipped there together with its cursor update (the `continue` at :602-604); no op of this window can be one while
                // even the window's last cursor stays in front of the end — the ordinary case, which then needs no cursor per word.
                constexpr uint32_t ISD_OPS = (1u << OP_I) | (1u << OP_S) | (1u << OP_D);
                uint32_t cand = 0, skipped = 0;
                if (emit_ok) {
#pragma unroll
                    for (int k = 0; k < WPL; k++)
                        if ((w[k] >> 4) >= min_oplen && __builtin_amdgcn_sbfe((int)(ISD_OPS | (ISD_OPS << 16)), w[k], 1u)) cand |= 1u << k;
                }
                const bool may_skip = ((p0 | ref_carry | tot_ref) >= 0x20000000u) || (p0 + ref_carry + tot_ref + 1u >= depth_len);
                uint32_t skip_before = skip_carry;
                if (__ballot(cand != 0 && may_skip)) {         // rare: a read that reaches the contig's end (or absurd coordinates)
                    uint32_t lane_skip = 0, rpk = p0 + ro;
#pragma unroll
                    for (int k = 0; k < WPL; k++) {
                        const uint32_t len = w[k] >> 4, op = w[k] & 15u;
                        if (((cand >> k) & 1u) && op == OP_S && (uint32_t)(rpk + 1u) >= depth_len) { skipped |= 1u << k; lane_skip += len; }
                        rpk += len & (uint32_t)__builtin_amdgcn_sbfe((int)(REF_OPS | (REF_OPS << 16)), w[k], 1u);
                    }
                    cand &= ~skipped;
                    const uint32_t is = grp_incl_sum<GL>(lane_skip, upper);
                    skip_before += is - lane_skip;
                    skip_tot = grp_last<GL>(is);
                }
                if (cand) {
                    uint32_t rpk = p0 + ro, qpk = qp, skk = skip_before;
#pragma unroll
                    for (int k = 0; k < WPL; k++) {
                        const uint32_t len = w[k] >> 4, op = w[k] & 15u;
                        if ((cand >> k) & 1u) {
                            csv_sig sg;
                            sg.start = rpk + 1u;
                            sg.end = sg.start + len - 1u;
                            sg.read = (uint32_t)rb + slot;
                            const uint32_t kind = (op == OP_I) ? CSV_KIND_INS : (op == OP_D ? CSV_KIND_DEL : CSV_KIND_CLIP);
                            sg.qpos_kind = ((qpk - skk) << 2) | kind;
                            const uint32_t sl = atomicAdd(&buf_n, 1u);
                            if (sl < SIG_BUF) buf[sl] = sg;
                            else {
                                const unsigned long long gi = atomicAdd(&cnt->n_sig, 1ull);
                                if (gi < sig_cap) sig_out[gi] = sg;
                                if (bucket_hist) my_bucket_max = max(my_bucket_max, atomicAdd(&bucket_hist[bk_bucket(sg, hist_type_pos, hist_shift)], 1u) + 1u);
                            }
                            my_n_del += (kind == CSV_KIND_DEL);
                            my_overflow |= (sg.start >= start_limit);
                        }
                        if ((skipped >> k) & 1u) skk += len;
                        rpk += len & (uint32_t)__builtin_amdgcn_sbfe((int)(REF_OPS | (REF_OPS << 16)), w[k], 1u);
                        qpk += len & (uint32_t)__builtin_amdgcn_sbfe((int)(QRY_OPS | (QRY_OPS << 16)), w[k], 1u);
                    }
                }
            }
            const uint32_t ref_total = ref_carry + tot_ref, q_total = q_carry + tot_q;
            // the read is done: its totals take its place in the table (written out for the whole batch below)
            if (last && act && sub == 0) meta[wave][slot] = make_uint4(ref_total, (uint32_t)qs, q_total, 0u);
            // ---- advance
            if (last) { ref_carry = 0; q_carry = 0; skip_carry = 0; qs = -1; v = 0; slot = nslot; cm = nm; act = nact; }
            else { ref_carry = ref_total; q_carry = q_total; skip_carry += skip_tot; v = nv; }
#pragma unroll
            for (int k = 0; k < WPL; k++) w[k] = wn[k];
        }
        // ---- the batch is done: alignment intervals and the depth tiles each read reaches, lane l <-> read rb + l again (coalesced stores;
        // once per 64 reads instead of once per read)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        {
            const bool mine = (uint32_t)lane < nb;
            const uint4 res = meta[wave][lane];                       // {reference bases, query_start or -1, query bases}
            const uint64_t rr = rb + lane;
            if (mine) {
                uint32_t rlen = (l_fl & F_UNMAP) ? 0u : res.x;        // htslib bam_endpos: pos + rlen, rlen == 0 (or unmapped) -> 1
                if (rlen == 0) rlen = 1;
                ref_end[rr] = (int32_t)(l_p0 + rlen);
                q_start[rr] = (int32_t)res.y < 0 ? 0 : (int32_t)res.y;
                q_end[rr] = (int32_t)res.z;
            }
            // depth tiles (ScanExtras::tile_range): same filter as the depth pass (cnv_caller.cpp:491-495), positions in uint32 as there
            uint32_t t0 = 1, t1 = 0;
            if (mine && tile_range && res.x != 0 && !(l_fl & (F_UNMAP | F_SECONDARY | F_QCFAIL | F_DUP))) {
                const uint32_t first = l_p0 + 1u, lastp = first + res.x - 1u;
                if ((first >> DEPTH_TILE_SHIFT) < n_tiles && lastp >= first) { t0 = first >> DEPTH_TILE_SHIFT; t1 = min(lastp >> DEPTH_TILE_SHIFT, n_tiles - 1u); }
            }
            auto mark = [&](uint32_t tt, uint32_t rd) {
                const uint32_t sl = tt - tbase;
                if (sl < TR_SLOTS) {
                    atomicMax(&tr_first[sl], ~rd);
                    atomicMax(&tr_last[sl], rd + 1u);
                } else {
                    atomicMax(&tile_range[2 * (uint64_t)tt], ~(unsigned long long)rd);
                    atomicMax(&tile_range[2 * (uint64_t)tt + 1], (unsigned long long)rd + 1ull);
                }
            };
            const bool wide = t1 >= t0 && t1 - t0 >= 8u;              // a read across many tiles (long skips): the whole wave marks them
            if (t1 >= t0 && !wide) for (uint32_t tt = t0; tt <= t1; tt++) mark(tt, (uint32_t)rr);
            for (uint64_t mw = __ballot(wide); mw; mw &= mw - 1) {
                const uint32_t src = (uint32_t)__builtin_ctzll(mw);
                const uint32_t a0 = bcast32(t0, src), a1 = bcast32(t1, src), rd = (uint32_t)rb + src;
                for (uint32_t tt = a0 + (uint32_t)lane; tt <= a1; tt += WAVE) mark(tt, rd);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();          // the table is rewritten by the next batch
    }

    // workgroup epilogue (as cigar_scan_kernel's) + the tile table
    my_n_del = wave_sum(my_n_del);
    const bool wave_overflow = __ballot(my_overflow != 0) != 0;
    if (lane == 0) {
        if (my_n_del) atomicAdd(&blk_n_del, my_n_del);
        if (wave_overflow) atomicOr(&blk_overflow, 1u);
    }
    __syncthreads();
    if (tile_range && threadIdx.x < TR_SLOTS && tr_last[threadIdx.x]) {
        const uint64_t tt = (uint64_t)tbase + threadIdx.x;
        atomicMax(&tile_range[2 * tt], ~(unsigned long long)(uint32_t)~tr_first[threadIdx.x]);
        atomicMax(&tile_range[2 * tt + 1], (unsigned long long)tr_last[threadIdx.x]);
    }
    if (!emit) return;
    const uint32_t nbuf = min(buf_n, SIG_BUF);
    if (threadIdx.x == 0) {
        blk_gbase = nbuf ? atomicAdd(&cnt->n_sig, (unsigned long long)nbuf) : 0ull;
        if (blk_n_del) atomicAdd(&cnt->n_del, (unsigned long long)blk_n_del);
        if (blk_overflow) cnt->max_start = 0xffffffffu;
    }
    __syncthreads();
    const unsigned long long gb = blk_gbase;
    for (uint32_t i = threadIdx.x; i < nbuf; i += RS_THREADS) {
        const csv_sig sg = buf[i];
        if (gb + i < sig_cap) sig_out[gb + i] = sg;
        if (bucket_hist) my_bucket_max = max(my_bucket_max, atomicAdd(&bucket_hist[bk_bucket(sg, hist_type_pos, hist_shift)], 1u) + 1u);
    }
    if (bucket_hist) {
        my_bucket_max = wave_max(my_bucket_max);
        if (lane == 0 && my_bucket_max > BK_LOCAL_MAX) atomicMax(&cnt->max_len, my_bucket_max);
    }
}

// ------------------------------------------------------------------------------------------------------------- lane-per-read form
// The group form above still spends ~60 vector instructions per 40-op read on the machinery of sharing a read between lanes (masks for the
// neighbours' words, two group scans, broadcasts, a window's bookkeeping): for short reads the cheapest walk is the reference's own — one
// LANE walks one read op by op (sv_caller.cpp:563-656 is that loop), 64 reads per wave at a time. What makes it work on this machine: the
// 64 reads' words are one contiguous stretch of the array (~10 KiB), staged into the wave's LDS slice with coalesced 16-byte loads, and
// every lane then reads ITS words from LDS (the strided global loads a lane-per-read walk would issue thrash the L1). Per op: one LDS
// word, the two length masks, a compare against min_oplen — ~16 instructions for 64 reads at once; signatures, query_start, checkpoints
// are lane-local events. A batch takes as many consecutive reads as fit the stage; a read longer than the stage (never in a HiFi shard,
// possible in any input) is walked straight from global memory by its lane.
constexpr int LN_THREADS = 256;
constexpr int LN_WAVES = LN_THREADS / WAVE;
constexpr uint32_t LN_STAGE = 3072;          // words per wave (12 KiB): 64 HiFi reads are ~2 400 words

template <bool PADDED>
__global__ __launch_bounds__(LN_THREADS, 2) void cigar_scan_lanes_kernel(
    uint64_t n_reads, uint64_t n_cigar, const int32_t *__restrict__ pos, const uint16_t *__restrict__ flag,
    const uint8_t *__restrict__ mapq, const uint64_t *__restrict__ cigar_off, const uint32_t *__restrict__ cigar,
    uint32_t depth_len, uint32_t start_limit, uint32_t min_oplen, uint32_t min_mapq, int emit,
    csv_sig *__restrict__ sig_out, uint64_t sig_cap, int32_t *__restrict__ ref_end, int32_t *__restrict__ q_start,
    int32_t *__restrict__ q_end, uint32_t *__restrict__ ckpt, ScanCounters *__restrict__ cnt,
    unsigned long long *__restrict__ tile_range, uint32_t n_tiles, uint32_t *__restrict__ bucket_hist, int hist_type_pos, int hist_shift,
    const uint64_t *__restrict__ split)
{
    __shared__ csv_sig buf[SIG_BUF];
    __shared__ alignas(16) uint32_t stage[LN_WAVES][LN_STAGE + 8];      // (+ slack: the walk requests four words at a time)
    __shared__ uint32_t tr_first[TR_SLOTS], tr_last[TR_SLOTS];
    __shared__ uint32_t buf_n, blk_n_del, blk_overflow, tr_base;
    __shared__ unsigned long long blk_gbase;
    __shared__ uint64_t split_s[LN_WAVES + 1];

    const int lane = lane_id();
    const int wave = (int)uniform32(threadIdx.x >> 6);
    if (threadIdx.x == 0) { buf_n = 0; blk_n_del = 0; blk_overflow = 0; }
    if (threadIdx.x < TR_SLOTS) { tr_first[threadIdx.x] = 0; tr_last[threadIdx.x] = 0; }
    const uint64_t n_waves = (uint64_t)gridDim.x * LN_WAVES;
    const uint64_t wave_gid = (uint64_t)blockIdx.x * LN_WAVES + wave;
    const uint64_t share = (n_cigar + n_waves - 1) / n_waves;
    if (split) {
        if (threadIdx.x <= LN_WAVES) split_s[threadIdx.x] = split[2 * ((uint64_t)blockIdx.x * LN_WAVES + threadIdx.x)];
    } else {
        const uint64_t b = wave_lower_bound(cigar_off, n_reads, wave_gid * share, lane);
        if (lane == 0) split_s[wave] = b;
        if (wave == LN_WAVES - 1) {
            const uint64_t e = (wave_gid + 1 >= n_waves) ? n_reads : wave_lower_bound(cigar_off, n_reads, (wave_gid + 1) * share, lane);
            if (lane == 0) split_s[LN_WAVES] = e;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint64_t r0 = split_s[0];
        tr_base = (tile_range && r0 < n_reads) ? (((uint32_t)pos[r0] + 1u) >> DEPTH_TILE_SHIFT) : 0u;
    }
    __syncthreads();
    const uint64_t r_begin = uniform64(split_s[wave]);
    const uint64_t r_end = uniform64(split_s[wave + 1]);
    const uint32_t tbase = tr_base;
    uint32_t *__restrict__ const st = stage[wave];
    uint32_t my_n_del = 0, my_overflow = 0, my_bucket_max = 0;

    for (uint64_t rb = r_begin; rb < r_end;) {
        const uint32_t avail = (uint32_t)min((uint64_t)WAVE, r_end - rb);
        // lane l <-> read rb + l: offsets first, to see how many reads the stage takes
        uint32_t c0 = 0, nw = 0, p0 = 0, fl = 0, mq = 0, l_uns = 0;
        const uint64_t rr = rb + lane;
        if ((uint32_t)lane < avail) {
            const uint64_t a = cigar_off[rr], b = cigar_off[rr + 1];
            c0 = (uint32_t)a; nw = (uint32_t)(b - a);
            const int32_t p = pos[rr];
            p0 = (uint32_t)p; fl = flag[rr]; mq = mapq[rr];
            l_uns = (rr > 0 && p < pos[rr - 1]) ? 1u : 0u;
        }
        if (__ballot(l_uns != 0) && lane == 0) cnt->unsorted = 1u;
        const uint32_t base = bcast32(c0, 0) & ~3u;                  // the stage starts on a 16-byte boundary
        const bool fits = (uint32_t)lane < avail && (c0 + nw) - base <= LN_STAGE;          // (offsets grow with the lane: a prefix of the lanes)
        const uint64_t fit_mask = __ballot(fits);
        uint32_t nb = ~fit_mask ? (uint32_t)__builtin_ctzll(~fit_mask) : (uint32_t)WAVE;      // leading run of reads that fit together
        const bool direct = nb == 0;                                 // the first read alone is larger than the stage: its lane reads global memory
        if (direct) nb = 1;
        const bool mine = (uint32_t)lane < nb;
        if (!direct) {
            const uint32_t n_stage = bcast32(c0 + nw, nb - 1) - base;
            if (PADDED) {
                // LDS-DMA, 1 KiB per instruction, all of them in flight at once (a loop of load -> store pairs waited a round trip per KiB)
                const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_addr(st));
                for (uint32_t i = 0; i < n_stage; i += CHUNK_WORDS) glds16(cigar + base + i, (uint32_t)lane * 16u, lds0 + i * 4u);
                ring_wait_all();
            } else {
                for (uint32_t i = (uint32_t)lane * 4; i < n_stage; i += WAVE * 4) {
#pragma unroll
                    for (int k = 0; k < 4; k++) st[i + k] = ((uint64_t)base + i + k < n_cigar) ? cigar[base + i + k] : (uint32_t)OP_P;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // ---- every lane walks its read (sv_caller.cpp:563-656, :663-690)
        const bool emit_ok = emit && mine && !(fl & (F_SECONDARY | F_UNMAP | F_DUP | F_QCFAIL | F_SUPP)) && mq >= min_mapq;
        const uint32_t n_mine = mine ? nw : 0u;
        uint32_t ref = 0, q = 0, skipped_sum = 0;
        int32_t qs = -1;
        const uint32_t off = c0 - base;
        uint32_t next_ck = (CKPT_WORDS - (c0 & (CKPT_WORDS - 1))) & (CKPT_WORDS - 1);      // first k > 0 whose word sits on a checkpoint boundary
        if (next_ck == 0) next_ck = CKPT_WORDS;
        // (two instances of the loop: with the source chosen inside it the compiler merges the two loads into one FLAT load through a generic
        // pointer — a memory-latency operation per op even when the word sits in LDS: 2x the whole kernel)
        auto walk = [&](auto direct_c) {
        constexpr bool DIRECT = decltype(direct_c)::value;
        // four words per step: the four LDS reads are issued together and waited for once (a word per step paid the LDS latency per op —
        // with two waves per SIMD nothing hides it), the words of the NEXT step are requested before this step's are worked on
        uint32_t wn[4];
        auto fetch = [&](uint32_t k0) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (DIRECT) wn[j] = (k0 + j < n_mine) ? cigar[(uint64_t)c0 + k0 + j] : (uint32_t)OP_P;
                else wn[j] = st[off + k0 + j];                       // (up to three words past the read's end: inside the stage's slack, never used)
            }
        };
        fetch(0);
        for (uint32_t k = 0; __ballot(k < n_mine); k += 4) {
            uint32_t wc[4];
#pragma unroll
            for (int j = 0; j < 4; j++) wc[j] = wn[j];
            if (__ballot(k + 4 < n_mine)) fetch(k + 4);
#pragma unroll
            for (int j = 0; j < 4; j++) {
            if (k + j < n_mine) {
                const uint32_t w = wc[j];
                const uint32_t kk = k + j;
                const uint32_t len = w >> 4;
                const uint32_t rl = len & (uint32_t)__builtin_amdgcn_sbfe((int)(REF_OPS | (REF_OPS << 16)), w, 1u);
                const uint32_t ql = len & (uint32_t)__builtin_amdgcn_sbfe((int)(QRY_OPS | (QRY_OPS << 16)), w, 1u);
                if (kk == next_ck) { ckpt[(c0 + kk) >> CKPT_SHIFT] = ref; next_ck += CKPT_WORDS; }
                if (qs < 0 && __builtin_amdgcn_sbfe((int)(QST_OPS | (QST_OPS << 16)), w, 1u)) qs = (int32_t)q;
                if (len >= min_oplen && emit_ok) {
                    const uint32_t op = w & 15u;
                    if (op == OP_I || op == OP_D || op == OP_S) {
                        const uint32_t rp = p0 + ref;
                        if (op == OP_S && (uint32_t)(rp + 1u) >= depth_len) skipped_sum += len;      // the `continue` at :602-604: no call, no cursor update
                        else {
                            csv_sig sg;
                            sg.start = rp + 1u;
                            sg.end = sg.start + len - 1u;
                            sg.read = (uint32_t)rr;
                            const uint32_t kind = (op == OP_I) ? CSV_KIND_INS : (op == OP_D ? CSV_KIND_DEL : CSV_KIND_CLIP);
                            sg.qpos_kind = ((q - skipped_sum) << 2) | kind;
                            const uint32_t sl = atomicAdd(&buf_n, 1u);
                            if (sl < SIG_BUF) buf[sl] = sg;
                            else {
                                const unsigned long long gi = atomicAdd(&cnt->n_sig, 1ull);
                                if (gi < sig_cap) sig_out[gi] = sg;
                                if (bucket_hist) my_bucket_max = max(my_bucket_max, atomicAdd(&bucket_hist[bk_bucket(sg, hist_type_pos, hist_shift)], 1u) + 1u);
                            }
                            my_n_del += (kind == CSV_KIND_DEL);
                            my_overflow |= (sg.start >= start_limit);
                        }
                    }
                }
                ref += rl; q += ql;
            }
            }
        }
        };
#ifndef LN_KO_WALK
        if (direct) walk(std::true_type{}); else walk(std::false_type{});
#endif
        // ---- the batch's outputs, lane-parallel
        uint32_t t0 = 1, t1 = 0;
        if (mine) {
            uint32_t rlen = (fl & F_UNMAP) ? 0u : ref;               // htslib bam_endpos: pos + rlen, rlen == 0 (or unmapped) -> 1
            if (rlen == 0) rlen = 1;
            ref_end[rr] = (int32_t)(p0 + rlen);
            q_start[rr] = qs < 0 ? 0 : qs;
            q_end[rr] = (int32_t)q;
            if (tile_range && ref != 0 && !(fl & (F_UNMAP | F_SECONDARY | F_QCFAIL | F_DUP))) {
                const uint32_t first = p0 + 1u, lastp = first + ref - 1u;
                if ((first >> DEPTH_TILE_SHIFT) < n_tiles && lastp >= first) { t0 = first >> DEPTH_TILE_SHIFT; t1 = min(lastp >> DEPTH_TILE_SHIFT, n_tiles - 1u); }
            }
        }
        auto mark = [&](uint32_t tt, uint32_t rd) {
            const uint32_t sl = tt - tbase;
            if (sl < TR_SLOTS) {
                atomicMax(&tr_first[sl], ~rd);
                atomicMax(&tr_last[sl], rd + 1u);
            } else {
                atomicMax(&tile_range[2 * (uint64_t)tt], ~(unsigned long long)rd);
                atomicMax(&tile_range[2 * (uint64_t)tt + 1], (unsigned long long)rd + 1ull);
            }
        };
        const bool wide = t1 >= t0 && t1 - t0 >= 8u;
        if (t1 >= t0 && !wide) for (uint32_t tt = t0; tt <= t1; tt++) mark(tt, (uint32_t)rr);
        for (uint64_t mw = __ballot(wide); mw; mw &= mw - 1) {
            const uint32_t src = (uint32_t)__builtin_ctzll(mw);
            const uint32_t a0 = bcast32(t0, src), a1 = bcast32(t1, src), rd = (uint32_t)rb + src;
            for (uint32_t tt = a0 + (uint32_t)lane; tt <= a1; tt += WAVE) mark(tt, rd);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();          // the stage is refilled by the next batch
        rb += nb;
    }

    // workgroup epilogue (as cigar_scan_rows_kernel's)
    my_n_del = wave_sum(my_n_del);
    const bool wave_overflow = __ballot(my_overflow != 0) != 0;
    if (lane == 0) {
        if (my_n_del) atomicAdd(&blk_n_del, my_n_del);
        if (wave_overflow) atomicOr(&blk_overflow, 1u);
    }
    __syncthreads();
    if (tile_range && threadIdx.x < TR_SLOTS && tr_last[threadIdx.x]) {
        const uint64_t tt = (uint64_t)tbase + threadIdx.x;
        atomicMax(&tile_range[2 * tt], ~(unsigned long long)(uint32_t)~tr_first[threadIdx.x]);
        atomicMax(&tile_range[2 * tt + 1], (unsigned long long)tr_last[threadIdx.x]);
    }
    if (!emit) return;
    const uint32_t nbuf = min(buf_n, SIG_BUF);
    if (threadIdx.x == 0) {
        blk_gbase = nbuf ? atomicAdd(&cnt->n_sig, (unsigned long long)nbuf) : 0ull;
        if (blk_n_del) atomicAdd(&cnt->n_del, (unsigned long long)blk_n_del);
        if (blk_overflow) cnt->max_start = 0xffffffffu;
    }
    __syncthreads();
    const unsigned long long gb = blk_gbase;
    for (uint32_t i = threadIdx.x; i < nbuf; i += LN_THREADS) {
        const csv_sig sg = buf[i];
        if (gb + i < sig_cap) sig_out[gb + i] = sg;
        if (bucket_hist) my_bucket_max = max(my_bucket_max, atomicAdd(&bucket_hist[bk_bucket(sg, hist_type_pos, hist_shift)], 1u) + 1u);
    }
    if (bucket_hist) {
        my_bucket_max = wave_max(my_bucket_max);
        if (lane == 0 && my_bucket_max > BK_LOCAL_MAX) atomicMax(&cnt->max_len, my_bucket_max);
    }
}

// Signature starts are < depth_len for coordinate-sorted input (start = pos + 1 <= contig length); the ordering pass sizes its
// radix keys from this bound, and the scan raises ScanCounters::max_start to 0xffffffff if a start ever exceeds it.
uint32_t scan_start_limit(uint32_t depth_len)
{
    int b = bits_of(depth_len);
    if (b < 8) b = 8;
    return b >= 32 ? 0xffffffffu : (1u << b);
}

// persistent-style grid: exactly as many workgroups as are resident at once (waves stride over the reads), so there is no partially
// filled round of workgroups; two rounds measured best (0.177 ms; 0.191 at one round, 0.179 at three to four, 0.192 at six)
static unsigned scan_grid(int n_cu, uint64_t n_reads, int form)
{
    static int blocks_per_cu[4] = {0, 0, 0, 0};
    if (blocks_per_cu[form] == 0) {
        int occ = 0;
        hipError_t e;
        if (form == SCAN_FORM_WAVE) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, cigar_scan_kernel, SCAN_THREADS, 0);
        else if (form == SCAN_FORM_ROWS16) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, cigar_scan_rows_kernel<16, true>, RS_THREADS, 0);
        else if (form == SCAN_FORM_ROWS8) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, cigar_scan_rows_kernel<8, true>, RS_THREADS, 0);
        else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, cigar_scan_lanes_kernel<true>, LN_THREADS, 0);
        if (e != hipSuccess || occ <= 0) { (void)hipGetLastError(); occ = 4; }
        blocks_per_cu[form] = occ;
    }
    if (form == SCAN_FORM_WAVE) {
        const uint64_t want = (n_reads + SCAN_WAVES - 1) / SCAN_WAVES;
        const uint64_t cap = (uint64_t)n_cu * blocks_per_cu[form] * 2;
        return (unsigned)(want < cap ? want : cap);
    }
    // short reads: a wave wants a few dozen reads (its groups run through them side by side; the lane form takes 64 at a time and wants a few
    // batches per wave); one round of workgroups while they all fit, else two
    const uint64_t per_wave = form == SCAN_FORM_LANES ? 128 : RS_MIN_READS;
    const uint64_t want = (n_reads + (uint64_t)RS_WAVES * per_wave - 1) / ((uint64_t)RS_WAVES * per_wave);
    const uint64_t slots = (uint64_t)n_cu * blocks_per_cu[form];
    return (unsigned)(want <= slots ? want : 2 * slots);
}

// split[2 g] = first read of wave g of launch_cigar_scan's grid, split[2 g + 1] = that read's first word, g in [0, waves]: one wave per entry, the search the scan's waves would do
__global__ __launch_bounds__(256) void scan_split_kernel(const uint64_t *__restrict__ cigar_off, uint64_t n_reads, uint64_t n_cigar, uint64_t n_waves,
                                                        uint64_t *__restrict__ split)
{
    const uint64_t g = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g > n_waves) return;
    const uint64_t share = (n_cigar + n_waves - 1) / n_waves;
    const uint64_t b = g >= n_waves ? n_reads : wave_lower_bound(cigar_off, n_reads, g * share, lane_id());
    if (lane_id() == 0) { split[2 * g] = b; split[2 * g + 1] = cigar_off[b]; }
}

// Which form of the scan (and of the depth pass's walk) suits a shard: reads of a few dozen ops (HiFi) are walked by groups of lanes,
// several reads per wave; long reads (ONT: ~1 200 ops) by a wave each. CSV_SCAN_FORM = 0 / 1 / 2 / 3 overrides (experiments, tests).
int scan_form_for(uint64_t n_reads, uint64_t n_cigar)
{
    const char *e = getenv("CSV_SCAN_FORM");
    const int forced = e && *e ? atoi(e) : -1;
    if (n_cigar >= 0xffffffffull) return SCAN_FORM_WAVE;          // the short-read form indexes words in 32 bits
    if (forced >= 0 && forced <= 3) return forced;
    return (n_reads && n_cigar / n_reads < 192) ? SCAN_FORM_ROWS16 : SCAN_FORM_WAVE;
}

size_t scan_split_bytes(int n_cu, uint64_t n_reads, int form)
{
    const size_t w = (size_t)scan_grid(n_cu, n_reads, form) * SCAN_WAVES;
    return (w + 1) * 2 * sizeof(uint64_t);
}

void launch_scan_split(hipStream_t s, int n_cu, const csv_reads &d, uint64_t *split, int form)
{
    if (d.n_reads == 0) return;
    const uint64_t n_waves = (uint64_t)scan_grid(n_cu, d.n_reads, form) * SCAN_WAVES;
    hipLaunchKernelGGL(scan_split_kernel, dim3((unsigned)((n_waves + 1 + 3) / 4)), dim3(256), 0, s, d.cigar_off, d.n_reads, d.n_cigar, n_waves, split);
}

void launch_cigar_scan(hipStream_t s, int n_cu, const csv_reads &d, uint32_t depth_len, uint32_t min_oplen,
                       uint32_t min_mapq, int emit, csv_sig *sig_out, uint64_t sig_cap,
                       int32_t *ref_end, int32_t *q_start, int32_t *q_end, uint32_t *ckpt, ScanCounters *cnt, const ScanExtras &x, const uint64_t *split,
                       int form, uint32_t cigar_pad_words)
{
    if (d.n_reads == 0) return;
    static_assert(SCAN_WAVES == RS_WAVES, "one split table layout for both forms");
    if (form != SCAN_FORM_WAVE && d.n_cigar >= 0xffffffffull) { form = SCAN_FORM_WAVE; split = nullptr; }
    const unsigned grid = scan_grid(n_cu, d.n_reads, form);
    const int vec_ok = (((uintptr_t)d.cigar) & 15u) == 0;
    if (form == SCAN_FORM_WAVE) {
        hipLaunchKernelGGL(cigar_scan_kernel, dim3(grid), dim3(SCAN_THREADS), 0, s, d.n_reads, d.n_cigar, d.pos, d.flag,
                           d.mapq, d.cigar_off, d.cigar, vec_ok, depth_len, scan_start_limit(depth_len), min_oplen, min_mapq, emit, sig_out, sig_cap,
                           ref_end, q_start, q_end, ckpt, cnt, (unsigned long long *)x.tile_range, x.n_tiles, emit ? x.bucket_hist : nullptr, x.type_pos, x.bucket_shift,
                           split);
        return;
    }
    const bool padded = vec_ok && cigar_pad_words >= 128;
#define CSV_ROWS_LAUNCH(GL, PAD)                                                                                                                       \
    hipLaunchKernelGGL((cigar_scan_rows_kernel<GL, PAD>), dim3(grid), dim3(RS_THREADS), 0, s, d.n_reads, d.n_cigar, d.pos, d.flag, d.mapq, d.cigar_off,   \
                       d.cigar, depth_len, scan_start_limit(depth_len), min_oplen, min_mapq, emit, sig_out, sig_cap, ref_end, q_start, q_end, ckpt, cnt,  \
                       (unsigned long long *)x.tile_range, x.n_tiles, emit ? x.bucket_hist : nullptr, x.type_pos, x.bucket_shift, split)
    if (form == SCAN_FORM_ROWS16) { if (padded) CSV_ROWS_LAUNCH(16, true); else CSV_ROWS_LAUNCH(16, false); }
    else if (form == SCAN_FORM_ROWS8) { if (padded) CSV_ROWS_LAUNCH(8, true); else CSV_ROWS_LAUNCH(8, false); }
    else {
        static_assert(LN_WAVES == SCAN_WAVES, "one split table layout for every form");
#define CSV_LANES_LAUNCH(PAD)                                                                                                                        \
    hipLaunchKernelGGL((cigar_scan_lanes_kernel<PAD>), dim3(grid), dim3(LN_THREADS), 0, s, d.n_reads, d.n_cigar, d.pos, d.flag, d.mapq, d.cigar_off,   \
                       d.cigar, depth_len, scan_start_limit(depth_len), min_oplen, min_mapq, emit, sig_out, sig_cap, ref_end, q_start, q_end, ckpt, cnt,  \
                       (unsigned long long *)x.tile_range, x.n_tiles, emit ? x.bucket_hist : nullptr, x.type_pos, x.bucket_shift, split)
        if (padded) CSV_LANES_LAUNCH(true); else CSV_LANES_LAUNCH(false);
#undef CSV_LANES_LAUNCH
    }
#undef CSV_ROWS_LAUNCH
}

// cigar_off of arrays that already live in HBM (csvgpu_shard_wrap_dev) gets the test the host arrays get in check_reads
__global__ void validate_offsets_kernel(const uint64_t *__restrict__ cigar_off, uint64_t n_reads, uint64_t n_cigar, uint64_t max_words,
                                        uint32_t *__restrict__ bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_reads) return;
    const uint64_t a = cigar_off[i];
    bool wrong = a > n_cigar;
    if (i < n_reads) { const uint64_t b = cigar_off[i + 1]; wrong |= b < a || b - a >= max_words; }
    if (wrong) *bad = 1u;
}

void launch_validate_offsets(hipStream_t s, const uint64_t *cigar_off, uint64_t n_reads, uint64_t n_cigar, uint64_t max_words, uint32_t *bad)
{
    hipLaunchKernelGGL(validate_offsets_kernel, dim3((unsigned)((n_reads + 256) / 256)), dim3(256), 0, s, cigar_off, n_reads, n_cigar, max_words, bad);
}

}  // namespace csv
