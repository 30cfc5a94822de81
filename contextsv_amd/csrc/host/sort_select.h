// sort_select.h — "which element does libstdc++'s std::sort leave at position nth?" in O(n).
//
// mergeSVs picks a cluster's representative from a fixed slot of an UNSTABLE std::sort by length
// (reference src/sv_object.cpp:190-230), so on ties the answer is whatever libstdc++'s introsort does on
// that input order. Sorting 6e4 noise signatures per chromosome just to read one slot dominated the host
// side, so this header reproduces the std::sort outcome for ONE slot:
//   * std::sort = __introsort_loop (median-of-3 to *first, Hoare-style __unguarded_partition, recursion on
//     the right part, loop on the left, depth limit 2*floor(log2 n) -> heap sort) followed by one insertion
//     sort over the whole range (bits/stl_algo.h). Sub-ranges are independent once partitioned, so the final
//     content of slot nth only depends on the chain of partitions that contain nth.
//   * after the introsort loop every <=16-element segment is ordered relative to its neighbours
//     (left <= pivot <= right), so the closing insertion sort never moves an element across a segment
//     boundary: inside the segment it is a plain stable insertion sort.
// Hence: partition exactly like libstdc++, descend only into the side holding nth, finish the last segment
// with a stable insertion sort, read the slot. When the depth limit runs out the same heap sort the library
// would run (std::partial_sort(first, last, last) == make_heap + sort_heap) is applied to that range.
// tests/test_sort_select.py checks every slot against std::sort on tie-heavy and adversarial inputs.
#pragma once
#include <algorithm>
#include <cstddef>
#include <type_traits>
#include <utility>

namespace csvhost {

template <class It, class Cmp>
inline void lib_move_median_to_first(It result, It a, It b, It c, Cmp comp)
{
    if (comp(*a, *b)) {
        if (comp(*b, *c)) std::iter_swap(result, b);
        else if (comp(*a, *c)) std::iter_swap(result, c);
        else std::iter_swap(result, a);
    } else if (comp(*a, *c)) std::iter_swap(result, a);
    else if (comp(*b, *c)) std::iter_swap(result, c);
    else std::iter_swap(result, b);
}

template <class It, class Cmp>
inline It lib_unguarded_partition(It first, It last, It pivot, Cmp comp)
{
    while (true) {
        while (comp(*first, *pivot)) ++first;
        --last;
        while (comp(*pivot, *last)) --last;
        if (!(first < last)) return first;
        std::iter_swap(first, last);
        ++first;
    }
}

// The same partition — the same swaps in the same order, hence the same cut and the same arrangement — without a data-dependent branch
// per element. The library's loop swaps the k-th element from the left that does not belong left (!comp(x, pivot): a_1 < a_2 < ...) with
// the k-th from the right that does not belong right (!comp(pivot, y): b_1 > b_2 > ...) while a_k < b_k; the elements between a_k and b_k
// are untouched when pair k + 1 is looked for. So both lists can be collected block by block (offsets appended with `n += flag`, no
// branch), whole blocks from either end that cannot overlap are paired and swapped unconditionally, and the last stretch — less than two
// blocks plus whatever was collected and not yet paired — is left to the library's loop, resumed behind the last swapped pair. On the
// noise bucket of a large contig (3e5 lengths in random order: one mispredicted branch in two) this is the difference between 1.9 and
// ~0.7 ms on the one thread that replays it.
template <class T, class Cmp>
inline T *lib_unguarded_partition_blocks(T *first, T *last, T *pivot, Cmp comp)
{
    constexpr std::ptrdiff_t B = 64;
    unsigned char offL[B], offR[B];
    std::ptrdiff_t nL = 0, iL = 0, nR = 0, iR = 0;
    T *l_scan = first, *r_scan = last;             // unscanned: [l_scan, r_scan)
    T *lbase = first, *rbase = last;               // current blocks: lbase[0 .. B), rbase[-B .. 0)
    T *res_first = first, *res_last = last;        // the library loop's (first, last) behind the last swapped pair
    const T pv = *pivot;
    for (;;) {
        if (iL == nL) {
            if (r_scan - l_scan < B) break;
            nL = iL = 0;
            lbase = l_scan;
            for (std::ptrdiff_t i = 0; i < B; i++) { offL[nL] = (unsigned char)i; nL += !comp(lbase[i], pv); }
            l_scan += B;
        }
        if (iR == nR) {
            if (r_scan - l_scan < B) break;
            nR = iR = 0;
            rbase = r_scan;
            for (std::ptrdiff_t j = 0; j < B; j++) { offR[nR] = (unsigned char)j; nR += !comp(pv, rbase[-1 - j]); }
            r_scan -= B;
        }
        const std::ptrdiff_t k = std::min(nL - iL, nR - iR);
        for (std::ptrdiff_t t = 0; t < k; t++) std::swap(lbase[offL[iL + t]], rbase[-1 - (std::ptrdiff_t)offR[iR + t]]);
        if (k > 0) { res_first = lbase + offL[iL + k - 1] + 1; res_last = rbase - 1 - (std::ptrdiff_t)offR[iR + k - 1]; }
        iL += k; iR += k;
    }
    return lib_unguarded_partition(res_first, res_last, pivot, comp);
}

// Permutes [first,last) partially and returns an iterator to the element std::sort(first,last,comp) would
// leave at position first+nth. Requires 0 <= nth < last-first.
template <class It, class Cmp>
inline It std_sort_select(It first, It last, std::ptrdiff_t nth, Cmp comp)
{
    std::ptrdiff_t n = last - first;
    int depth_limit = 0;
    for (std::ptrdiff_t k = n; k > 1; k >>= 1) depth_limit++;     // std::__lg(n)
    depth_limit *= 2;
    It lo = first, hi = last;
    const It target = first + nth;
    while (hi - lo > 16) {
        if (depth_limit == 0) {
            std::partial_sort(lo, hi, hi, comp);                   // the library's heap-sort fallback for this range
            return target;
        }
        --depth_limit;
        It mid = lo + (hi - lo) / 2;
        lib_move_median_to_first(lo, lo + 1, mid, hi - 1, comp);
        It cut;
        if constexpr (std::is_pointer<It>::value) cut = (hi - lo > 1024) ? lib_unguarded_partition_blocks(lo + 1, hi, lo, comp) : lib_unguarded_partition(lo + 1, hi, lo, comp);
        else cut = lib_unguarded_partition(lo + 1, hi, lo, comp);
        if (target >= cut) lo = cut; else hi = cut;
    }
    for (It i = lo + 1; i < hi; ++i) {                             // stable insertion sort of the closing segment
        auto v = std::move(*i);
        It j = i;
        while (j > lo && comp(v, *(j - 1))) { *j = std::move(*(j - 1)); --j; }
        *j = std::move(v);
    }
    return target;
}

}  // namespace csvhost
