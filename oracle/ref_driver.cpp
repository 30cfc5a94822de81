// ref_driver.cpp — TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" shim around the REFERENCE's own translation units, compiled unmodified from
// where they lie under /root/reference (src/dbscan.cpp, src/dbscan1d.cpp, src/kc.cpp) into
// oracle/_ref/libcsvref.so by oracle/Makefile. Nothing from the reference is copied into this
// repository: this file only includes the reference headers at build time and forwards calls.
// It is used to (1) validate oracle/csv_oracle.c, (2) generate tests/golden/*.json
// (tests/golden/make_golden.py) and (3) as the "reference" CPU baseline of bench.py.
//
// Only these three TUs build here: every other TU on the path includes <htslib/...> (directly or
// through include/utils.h), which this image does not have, so they are treated as unbuildable.
#include <cstdint>
#include <vector>

#include "dbscan.h"     // /root/reference/include
#include "dbscan1d.h"
#include "kc.h"

extern "C" {

// DBSCAN::fit + getClusters (dbscan.cpp:9-24)
void ref_dbscan_iv(const uint32_t *start, const uint32_t *end, uint64_t n, double eps, int32_t min_pts,
                   int32_t *labels)
{
    std::vector<SVCall> calls(n);
    for (uint64_t i = 0; i < n; i++) { calls[i].start = start[i]; calls[i].end = end[i]; }
    DBSCAN dbscan(eps, min_pts);
    dbscan.fit(calls);
    const std::vector<int> &cl = dbscan.getClusters();
    for (uint64_t i = 0; i < n; i++) labels[i] = cl[i];
}

// DBSCAN1D::fit + getClusters (dbscan1d.cpp:8-24)
void ref_dbscan_1d(const int32_t *pts, uint64_t n, double eps, int32_t min_pts, int32_t *labels)
{
    std::vector<int> p(pts, pts + n);
    DBSCAN1D dbscan(eps, min_pts);
    dbscan.fit(p);
    const std::vector<int> &cl = dbscan.getClusters();
    for (uint64_t i = 0; i < n; i++) labels[i] = cl[i];
}

// DBSCAN1D::fit + getLargestCluster (dbscan1d.cpp:72-90); returns the member count
int64_t ref_dbscan_1d_largest(const int32_t *pts, uint64_t n, double eps, int32_t min_pts, int32_t *out)
{
    std::vector<int> p(pts, pts + n);
    DBSCAN1D dbscan(eps, min_pts);
    dbscan.fit(p);
    std::vector<int> big = dbscan.getLargestCluster(p);
    for (size_t i = 0; i < big.size(); i++) out[i] = big[i];
    return (int64_t)big.size();
}

double ref_pdf_normal(double x, double mu, double sigma) { return pdf_normal(x, mu, sigma); }   // kc.cpp:2658
double ref_cdf_normal(double x, double mu, double sigma) { return cdf_normal(x, mu, sigma); }   // kc.cpp:2565

}  // extern "C"
