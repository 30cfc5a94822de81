/* dbscan_window.cpp — TEST INFRASTRUCTURE (oracle): an independent, order-free, windowed restatement of the reference's interval DBSCAN
 * (src/dbscan.cpp:9-81) that finishes in O(n * window) instead of the O(n^2) of the literal walk in csv_oracle.c — so that sets of a few
 * hundred thousand signatures (chr1's INS set) can be checked on the CPU. Only tests/ load it; never the product.
 *
 * PINNED: bit for bit against the reference's own dbscan.cpp (oracle/_ref) on the 266 golden fits and on the 160 sweep inputs
 * (tests/test_oracle_windowed.py), and against the literal port orc_dbscan_iv on random inputs.
 *
 * What the reference's visit order amounts to (derived from :13-59, then checked against the reference itself):
 *   neighbours   N(i) = { j : distance(i, j) <= eps }, distance = 1 - min(ov / len_i, ov / len_j) in double (:69-81), std::min's
 *                argument order kept (it matters when a length is 0: 0/0 is NaN and such an interval neighbours nothing, itself included)
 *   core         |N(i)| >= minPts
 *   clusters     connected components of the cores; a component's first visited point is its smallest-index core ("start point"), so the
 *                clusters are numbered by their start points' indices
 *   border       a non-core with a core neighbour: the start step (:33-35) overwrites whatever label a neighbour of the start point has,
 *                expansion (:47-53) only claims unlabelled / noise points — so the LARGEST cluster among neighbouring start points wins,
 *                else the SMALLEST cluster among neighbouring cores
 *   noise        -2 (a non-core without core neighbour); -1 never survives
 * Neighbour pairs of positive-length intervals must share a position (ov > 0 whenever eps < 1), so on start-sorted intervals the
 * candidates of i are the j with start_j < end_i: a forward sweep; intervals of length <= 0 take the literal all-pairs row. */
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace {

inline double dist(uint32_t s1, uint32_t e1, uint32_t s2, uint32_t e2)
{
    const int overlap = std::max(0, std::min((int)e1, (int)e2) - std::max((int)s1, (int)s2));
    const int length1 = (int)(e1 - s1), length2 = (int)(e2 - s2);
    return 1.0 - std::min((double)overlap / (double)length1, (double)overlap / (double)length2);
}

struct Dsu {
    std::vector<uint32_t> p;
    explicit Dsu(size_t n) : p(n) { std::iota(p.begin(), p.end(), 0u); }
    uint32_t find(uint32_t x) { while (p[x] != x) { p[x] = p[p[x]]; x = p[x]; } return x; }
    void unite(uint32_t a, uint32_t b) { a = find(a); b = find(b); if (a != b) p[std::max(a, b)] = std::min(a, b); }     // root = smallest index
};

}  // namespace

extern "C" int orc_dbscan_iv_windowed(const uint32_t *s, const uint32_t *e, uint64_t n64, double eps, int32_t min_pts, int32_t *labels)
{
    const size_t n = (size_t)n64;
    if (!(eps >= 0.0) || !(eps < 1.0)) return -1;                        /* the window argument needs eps < 1 (as the device seam does) */
    std::vector<uint32_t> ord(n);
    std::iota(ord.begin(), ord.end(), 0u);
    std::stable_sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return s[a] < s[b]; });
    std::vector<uint32_t> degenerate;                                    /* length <= 0 as the reference's int arithmetic sees it */
    for (size_t i = 0; i < n; i++) if ((int)(e[i] - s[i]) <= 0) degenerate.push_back((uint32_t)i);
    auto positive = [&](uint32_t i) { return (int)(e[i] - s[i]) > 0; };

    // neighbour pairs (directed): enumerate(f) calls f(i, j) for every j in N(i)
    auto enumerate = [&](auto &&f) {
        for (size_t a = 0; a < n; a++) {
            const uint32_t i = ord[a];
            if (!positive(i)) continue;
            if (dist(s[i], e[i], s[i], e[i]) <= eps) f(i, i);
            for (size_t b = a + 1; b < n; b++) {
                const uint32_t j = ord[b];
                if ((int)s[j] >= (int)e[i] && (int)s[j] >= 0 && (int)e[i] >= 0) break;      // starts only grow: nothing further shares a position with i
                if (!positive(j)) continue;
                if (dist(s[i], e[i], s[j], e[j]) <= eps) f(i, j);
                if (dist(s[j], e[j], s[i], e[i]) <= eps) f(j, i);
            }
        }
        for (uint32_t i : degenerate)
            for (size_t j = 0; j < n; j++) {
                if (dist(s[i], e[i], s[j], e[j]) <= eps) f(i, (uint32_t)j);
                if (positive((uint32_t)j) && dist(s[j], e[j], s[i], e[i]) <= eps) f((uint32_t)j, i);
            }
    };

    std::vector<uint32_t> cnt(n, 0);
    enumerate([&](uint32_t i, uint32_t) { cnt[i]++; });
    std::vector<char> core(n);
    for (size_t i = 0; i < n; i++) core[i] = (int64_t)cnt[i] >= (int64_t)min_pts;
    Dsu dsu(n);
    enumerate([&](uint32_t i, uint32_t j) { if (core[i] && core[j]) dsu.unite(i, j); });
    // cluster ids: rank of the component's smallest-index core
    std::vector<int32_t> cid(n, -1);
    int32_t next = 0;
    for (size_t i = 0; i < n; i++) if (core[i] && dsu.find((uint32_t)i) == i) cid[i] = next++;
    for (size_t i = 0; i < n; i++) labels[i] = core[i] ? cid[dsu.find((uint32_t)i)] : -2;
    // borders: j in N(i) with i a core and j not
    std::vector<int32_t> by_start(n, -1), by_core(n, INT32_MAX);
    enumerate([&](uint32_t i, uint32_t j) {
        if (!core[i] || core[j]) return;
        const int32_t c = cid[dsu.find(i)];
        if (dsu.find(i) == i) by_start[j] = std::max(by_start[j], c);
        by_core[j] = std::min(by_core[j], c);
    });
    for (size_t j = 0; j < n; j++)
        if (!core[j]) labels[j] = by_start[j] >= 0 ? by_start[j] : (by_core[j] != INT32_MAX ? by_core[j] : -2);
    return 0;
}
