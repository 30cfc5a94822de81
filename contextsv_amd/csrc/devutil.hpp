// devutil.hpp — wave64 device helpers (gfx950). Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace csv {

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v)
{
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (l >= d) v += t;
    }
    return v;
}

// DPP forms (gfx9/CDNA): row_shr 1,2,4,8 inside each 16-lane row, then row_bcast:15 / row_bcast:31 carry the row
// totals across rows. Six full-rate VALU ops and no LDS crossbar traffic, against six ds_bpermute round trips for the
// shuffle form — the CIGAR kernels are VALU-issue-bound, so this matters.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_incl_sum_dpp(uint32_t v)
{
    v += dpp_u32<0x111, 0xf>(0u, v);   // row_shr:1
    v += dpp_u32<0x112, 0xf>(0u, v);   // row_shr:2
    v += dpp_u32<0x114, 0xf>(0u, v);   // row_shr:4
    v += dpp_u32<0x118, 0xf>(0u, v);   // row_shr:8
    v += dpp_u32<0x142, 0xa>(0u, v);   // row_bcast:15 -> rows 1 and 3
    v += dpp_u32<0x143, 0xc>(0u, v);   // row_bcast:31 -> rows 2 and 3
    return v;
}
// total of the wave, as a scalar (lane 63 of the inclusive scan)
__device__ __forceinline__ uint32_t wave_total_dpp(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_sum_dpp(v), 63);
}

// ---- groups of GL = 8 or 16 lanes inside a DPP row (the short-read forms of the scan and of the depth walk: a group per read) ----
// inclusive prefix sum inside every group
template <int GL>
__device__ __forceinline__ uint32_t grp_incl_sum(uint32_t v, uint32_t upper)      // upper: all-ones in the upper 8 lanes of a row (GL == 8 only)
{
    v += dpp_u32<0x111, 0xf>(0u, v);
    v += dpp_u32<0x112, 0xf>(0u, v);
    v += dpp_u32<0x114, 0xf>(0u, v);
    v += dpp_u32<0x118, 0xf>(0u, v);
    if (GL == 8) v -= dpp_u32<0x157, 0xf>(0u, v) & upper;          // row_newbcast:7 — the lower group's total leaves the upper group's sums
    return v;
}
// the last lane's value of every group, in all its lanes
template <int GL>
__device__ __forceinline__ uint32_t grp_last(uint32_t v)
{
    if (GL == 16) return dpp_u32<0x15F, 0xf>(0u, v);                // row_newbcast:15
    const uint32_t lo = dpp_u32<0x157, 0xf>(0u, v);                 // row_newbcast:7
    return (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)v, 0x15F, 0xf, 0xc, false);   // banks 2, 3 (lanes 8..15) take lane 15's
}
template <int GL>
__device__ __forceinline__ uint32_t grp_min(uint32_t v)
{
    v = min(v, dpp_u32<0x111, 0xf>(0xffffffffu, v));
    v = min(v, dpp_u32<0x112, 0xf>(0xffffffffu, v));
    v = min(v, dpp_u32<0x114, 0xf>(0xffffffffu, v));                // GL == 8: lane 7 / 15 now hold their group's minimum (a window of 8 lanes)
    if (GL == 16) v = min(v, dpp_u32<0x118, 0xf>(0xffffffffu, v));
    return grp_last<GL>(v);
}

// (a << 2) + b in one instruction (hipcc reassociates the C expression into an add and a shift when b is itself such a sum)
__device__ __forceinline__ uint32_t lshl2_add(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// clamp x to [0, hi] (hi >= 0, wave-uniform) in one instruction; hipcc only forms v_med3 from min(max()) when it can prove 0 <= hi
__device__ __forceinline__ int32_t clamp0_i32(int32_t x, int32_t hi)
{
    int32_t r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
    return r;
}

__device__ __forceinline__ int32_t wave_incl_max(int32_t v)
{
    const int l = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int32_t t = __shfl_up(v, d, 64);
        if (l >= d) v = max(v, t);
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__device__ __forceinline__ uint64_t wave_sum64(uint64_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}

__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// The reference's interval metric (dbscan.cpp:69-81): 1.0 - std::min(ov/len1, ov/len2) <= eps in IEEE double, with
// std::min(a,b) = (b<a)?b:a (NaN-asymmetric when a length is 0). Build with -ffp-contract=off.
// For positive lengths min(ov/len1, ov/len2) IS ov/max(len1,len2) bit for bit (division is correctly rounded, hence monotonic
// in the divisor), so one division decides; and a single-precision estimate settles every pair that is not within 1e-4 of the
// threshold, which leaves the double division to the rare boundary case. Lengths <= 0 take the literal expression.
__device__ __forceinline__ bool iv_neighbor(uint32_t s1, uint32_t e1, uint32_t s2, uint32_t e2, double eps)
{
    const int a = min((int)e1, (int)e2);
    const int b = max((int)s1, (int)s2);
    const int overlap = max(0, a - b);
    const int length1 = (int)(e1 - s1);
    const int length2 = (int)(e2 - s2);
    if (length1 > 0 && length2 > 0) {
        const int lmax = max(length1, length2);
        const float q = __fdividef((float)overlap, (float)lmax);      // relative error << 1e-4
        const float t = (float)(1.0 - eps);
        if (q > t + 1e-4f) return true;
        if (q < t - 1e-4f) return false;
        return (1.0 - (double)overlap / (double)lmax) <= eps;
    }
    const double x = (double)overlap / (double)length1;
    const double y = (double)overlap / (double)length2;
    const double mn = (y < x) ? y : x;
    return (1.0 - mn) <= eps;
}

// bucket of a signature in the ordering pass's most-significant-digit split (sort.hip); counted by the scan's epilogue
__device__ __forceinline__ uint32_t bk_bucket(const csv_sig &sg, int type_pos, int shift)
{
    uint64_t k = sg.start;
    if (type_pos >= 0 && (sg.qpos_kind & 3u) != CSV_KIND_DEL) k |= 1ull << type_pos;
    const uint64_t b = k >> shift;
    return b < BK_N ? (uint32_t)b : BK_N - 1u;          // starts beyond the key width only occur with the overflow flag (fallback)
}

// A value every lane holds alike, moved to scalar registers: hipcc cannot prove that what comes out of LDS or of a load is
// wave-uniform, and without the hint a loop nest driven by such values (scan.hip's read / chunk loops) is compiled as divergent control flow (exec-mask loops,
// 64-bit VALU compares) — a third of that walk's VALU instructions.
__device__ __forceinline__ uint32_t uniform32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uniform64(uint64_t v) { return ((uint64_t)uniform32((uint32_t)(v >> 32)) << 32) | uniform32((uint32_t)v); }

}  // namespace csv
