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
constexpr uint32_t SIG_BUF = 1024;           // signatures staged per workgroup (16 KiB LDS)

struct Chunk {
    uint32_t w[4];
};

__device__ __forceinline__ Chunk load_chunk(const uint32_t *__restrict__ cigar, uint64_t idx, uint64_t n_cigar, int vec_ok)
{
    Chunk c;
    if (vec_ok && idx + 4 <= n_cigar) {
        uint4 v = *reinterpret_cast<const uint4 *>(cigar + idx);
        c.w[0] = v.x; c.w[1] = v.y; c.w[2] = v.z; c.w[3] = v.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) c.w[k] = (idx + k < n_cigar) ? cigar[idx + k] : (uint32_t)OP_P;
    }
    return c;
}

__global__ __launch_bounds__(SCAN_THREADS) void cigar_scan_kernel(
    uint64_t n_reads, uint64_t n_cigar, const int32_t *__restrict__ pos, const uint16_t *__restrict__ flag,
    const uint8_t *__restrict__ mapq, const uint64_t *__restrict__ cigar_off, const uint32_t *__restrict__ cigar,
    int vec_ok, uint32_t depth_len, uint32_t min_oplen, uint32_t min_mapq, int emit,
    csv_sig *__restrict__ sig_out, uint64_t sig_cap, int32_t *__restrict__ ref_end, int32_t *__restrict__ q_start,
    int32_t *__restrict__ q_end, uint32_t *__restrict__ ckpt, ScanCounters *__restrict__ cnt)
{
    __shared__ csv_sig buf[SIG_BUF];
    __shared__ uint32_t buf_n, blk_max_start, blk_max_len, blk_n_del;
    __shared__ unsigned long long blk_gbase;

    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) { buf_n = 0; blk_max_start = 0; blk_max_len = 0; blk_n_del = 0; }
    __syncthreads();

    uint32_t my_max_start = 0, my_max_len = 0, my_n_del = 0;

    const uint64_t wave_gid = (uint64_t)blockIdx.x * SCAN_WAVES + wave;
    const uint64_t wave_stride = (uint64_t)gridDim.x * SCAN_WAVES;

    for (uint64_t r = wave_gid; r < n_reads; r += wave_stride) {
        const uint64_t c0 = cigar_off[r], c1 = cigar_off[r + 1];
        const uint32_t p0 = (uint32_t)pos[r];
        const uint32_t fl = flag[r];
        const uint32_t mq = mapq[r];
        // sv_caller.cpp:526
        const bool emit_ok = emit && !(fl & (F_SECONDARY | F_UNMAP | F_DUP | F_QCFAIL | F_SUPP)) && mq >= min_mapq;
        if (lane == 0 && r > 0 && pos[r] < pos[r - 1]) cnt->unsorted = 1u;

        uint32_t ref_carry = 0;      // reference bases consumed so far (pos - aln_start)
        uint32_t q_carry = 0;        // query bases consumed so far (plain; getAlignmentReadPositions)
        uint32_t skip_carry = 0;     // lengths of soft clips skipped by the `continue` at sv_caller.cpp:602-604
        int32_t  qs = -1;            // query_start

        // chunks are aligned to 256 words (1 KiB) globally, so that a chunk boundary is a checkpoint slot
        const uint64_t base = c0 & ~255ull;
        Chunk cur = load_chunk(cigar, base + (uint64_t)lane * 4, n_cigar, vec_ok);
        for (uint64_t chunk = base; chunk < c1; chunk += 4 * WAVE) {
            const uint64_t idx = chunk + (uint64_t)lane * 4;
            if (lane == 0 && chunk > c0) ckpt[chunk >> 8] = ref_carry;     // reference offset of this read at word `chunk` (depth.hip)
            Chunk nxt;
            const bool more = chunk + 4 * WAVE < c1;
            if (more) nxt = load_chunk(cigar, idx + 4 * WAVE, n_cigar, vec_ok);   // prefetch next 1 KiB

            uint32_t len[4], op[4], rl[4], ql[4];
            uint32_t lane_ref = 0, lane_q = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const bool valid = (idx + k >= c0) && (idx + k < c1);
                len[k] = valid ? (cur.w[k] >> 4) : 0u;
                op[k] = valid ? (cur.w[k] & 15u) : (uint32_t)OP_P;
                rl[k] = ((REF_OPS >> op[k]) & 1u) ? len[k] : 0u;
                ql[k] = ((QRY_OPS >> op[k]) & 1u) ? len[k] : 0u;
                lane_ref += rl[k];
                lane_q += ql[k];
            }
            const uint32_t incl_ref = wave_incl_sum(lane_ref);
            const uint32_t incl_q = wave_incl_sum(lane_q);
            uint32_t rp = p0 + ref_carry + (incl_ref - lane_ref);   // reference `pos` before this lane's first op
            uint32_t qp = q_carry + (incl_q - lane_q);              // plain query cursor before this lane's first op

            // query_start = cursor at the first M/I/=/X op (sv_caller.cpp:674-676)
            if (qs < 0) {
                bool found = false;
                uint32_t q_at = 0, acc = qp;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (!found && ((QST_OPS >> op[k]) & 1u)) { found = true; q_at = acc; }
                    acc += ql[k];
                }
                const uint64_t m = __ballot(found);
                if (m) {
                    const int src = __ffsll((long long)m) - 1;
                    qs = (int32_t)__shfl(q_at, src, 64);
                }
            }

            if (emit_ok) {
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
                uint32_t skip_before = skip_carry;
                if (__ballot(skipped != 0)) {                 // rare: clip past the contig end
                    const uint32_t incl_s = wave_incl_sum(lane_skip);
                    skip_before += incl_s - lane_skip;
                    skip_carry += __shfl(incl_s, 63, 64);
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
                            csv_sig s;
                            s.start = rpk + 1u;
                            s.end = s.start + len[k] - 1u;
                            s.read = (uint32_t)r;
                            const uint32_t kind = (op[k] == OP_I) ? CSV_KIND_INS : (op[k] == OP_D ? CSV_KIND_DEL : CSV_KIND_CLIP);
                            s.qpos_kind = ((qpk - skk) << 2) | kind;
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
                                if (e) buf[slot + rank] = s;
                            } else {
                                unsigned long long g = 0;
                                if (lane == 0) g = atomicAdd(&cnt->n_sig, (unsigned long long)n_e);
                                g = __shfl(g, 0, 64);
                                if (e && g + rank < sig_cap) sig_out[g + rank] = s;
                            }
                            if (e) {
                                my_max_start = max(my_max_start, s.start);
                                my_max_len = max(my_max_len, s.end - s.start);
                                my_n_del += (kind == CSV_KIND_DEL);
                            }
                        }
                        if ((skipped >> k) & 1u) skk += len[k];
                        rpk += rl[k];
                        qpk += ql[k];
                    }
                }
            }

            ref_carry += __shfl(incl_ref, 63, 64);
            q_carry += __shfl(incl_q, 63, 64);
            if (more) cur = nxt;
        }

        if (lane == 0) {
            // htslib bam_endpos: pos + rlen, rlen == 0 (or unmapped) -> 1
            uint32_t rlen = (fl & F_UNMAP) ? 0u : ref_carry;
            if (rlen == 0) rlen = 1;
            ref_end[r] = (int32_t)(p0 + rlen);
            q_start[r] = qs < 0 ? 0 : qs;
            q_end[r] = (int32_t)q_carry;
        }
    }

    if (!emit) return;
    // workgroup epilogue: fold maxima, flush the LDS buffer with one global atomic
    my_max_start = wave_max(my_max_start);
    my_max_len = wave_max(my_max_len);
    my_n_del = wave_sum(my_n_del);
    if (lane == 0) {
        if (my_max_start) atomicMax(&blk_max_start, my_max_start);
        if (my_max_len) atomicMax(&blk_max_len, my_max_len);
        if (my_n_del) atomicAdd(&blk_n_del, my_n_del);
    }
    __syncthreads();
    const uint32_t nb = buf_n;
    if (threadIdx.x == 0) {
        blk_gbase = nb ? atomicAdd(&cnt->n_sig, (unsigned long long)nb) : 0ull;
        if (blk_max_start) atomicMax(&cnt->max_start, blk_max_start);
        if (blk_max_len) atomicMax(&cnt->max_len, blk_max_len);
        if (blk_n_del) atomicAdd(&cnt->n_del, (unsigned long long)blk_n_del);
    }
    __syncthreads();
    const unsigned long long g = blk_gbase;
    for (uint32_t i = threadIdx.x; i < nb; i += SCAN_THREADS)
        if (g + i < sig_cap) sig_out[g + i] = buf[i];
}

void launch_cigar_scan(hipStream_t s, int n_cu, const csv_reads &d, uint32_t depth_len, uint32_t min_oplen,
                       uint32_t min_mapq, int emit, csv_sig *sig_out, uint64_t sig_cap,
                       int32_t *ref_end, int32_t *q_start, int32_t *q_end, uint32_t *ckpt, ScanCounters *cnt)
{
    if (d.n_reads == 0) return;
    uint64_t want = (d.n_reads + SCAN_WAVES - 1) / SCAN_WAVES;
    uint64_t cap = (uint64_t)n_cu * 8;                    // 8 workgroups of 4 waves per CU = full occupancy
    unsigned grid = (unsigned)(want < cap ? want : cap);
    const int vec_ok = (((uintptr_t)d.cigar) & 15u) == 0;
    hipLaunchKernelGGL(cigar_scan_kernel, dim3(grid), dim3(SCAN_THREADS), 0, s, d.n_reads, d.n_cigar, d.pos, d.flag,
                       d.mapq, d.cigar_off, d.cigar, vec_ok, depth_len, min_oplen, min_mapq, emit, sig_out, sig_cap,
                       ref_end, q_start, q_end, ckpt, cnt);
}

}  // namespace csv
