// sort_select.h against std::sort under AddressSanitizer + UBSan: random lengths (few values: ties everywhere; many values; pre-sorted), range
// sizes on both sides of the block-wise partition's threshold, every case the element std::sort leaves at a random slot.
// usage: fuzz_sort_select_asan <iterations> <seed>      (make -C contextsv_amd/csrc asan; tests/test_sanitizers.py runs a short one)
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../contextsv_amd/csrc/host/sort_select.h"

struct Ent { uint32_t len, idx; };

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 300;
    std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], nullptr, 10) : 1);
    for (int it = 0; it < iters; it++) {
        const size_t n = 1 + rng() % (it % 10 == 0 ? 300000 : 5000);
        const int vals = 1 + (int)(rng() % (it % 3 == 0 ? 4 : 100000));
        std::vector<Ent> a(n);
        for (size_t i = 0; i < n; i++) a[i] = Ent{(uint32_t)(rng() % vals), (uint32_t)i};
        if (it % 7 == 0) std::sort(a.begin(), a.end(), [](const Ent &x, const Ent &y) { return x.len < y.len; });
        if (it % 11 == 0) std::reverse(a.begin(), a.end());
        std::vector<Ent> b = a, c = a;
        auto cmp = [](const Ent &x, const Ent &y) { return x.len > y.len; };
        std::sort(b.begin(), b.end(), cmp);
        const size_t nth = rng() % n;
        const Ent *r = csvhost::std_sort_select(c.data(), c.data() + n, (std::ptrdiff_t)nth, cmp);
        if (r->idx != b[nth].idx) { printf("select: MISMATCH it=%d n=%zu nth=%zu\n", it, n, nth); return 1; }
    }
    printf("select: %d cases equal to std::sort\n", iters);
    return 0;
}
