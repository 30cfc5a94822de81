// dbscan.cpp — DBSCAN / DBSCAN1D classes forwarding to the C-ABI (no clustering arithmetic here).
#include "dbscan.h"

#include <iostream>
#include <map>
#include <mutex>
#include <stdexcept>

#include "log.h"

namespace {
csv_ctx *g_ctx = nullptr;
thread_local csv_ctx *t_ctx = nullptr;
std::mutex g_print;
bool g_quiet = false;
}

void printMessage(const std::string &m) { if (g_quiet) return; std::lock_guard<std::mutex> l(g_print); std::cout << m << std::endl; }
void printError(const std::string &m) { std::lock_guard<std::mutex> l(g_print); std::cerr << m << std::endl; }

namespace csvhost {
void set_quiet(bool q) { g_quiet = q; }
void set_context(csv_ctx *ctx) { g_ctx = ctx; }
void set_thread_context(csv_ctx *ctx) { t_ctx = ctx; }
csv_ctx *context()
{
    if (t_ctx) return t_ctx;
    if (!g_ctx) throw std::runtime_error("csvhost: no GPU context set (csvhost::set_context) — the clustering path has no CPU fallback");
    return g_ctx;
}
}  // namespace csvhost

static void check(int rc, const char *what)
{
    if (rc != CSV_OK) throw std::runtime_error(std::string(what) + ": " + csvgpu_last_error(csvhost::context()));
}

void DBSCAN::fit(const std::vector<SVCall> &sv_calls)
{
    const size_t n = sv_calls.size();
    std::vector<uint32_t> s(n), e(n);
    for (size_t i = 0; i < n; i++) { s[i] = sv_calls[i].start; e[i] = sv_calls[i].end; }
    clusters.assign(n, -1);
    if (n) check(csvgpu_dbscan_iv(csvhost::context(), s.data(), e.data(), n, epsilon, minPts, clusters.data()), "DBSCAN::fit");
}

void DBSCAN::fitBatch(const std::vector<const std::vector<SVCall> *> &sets, double epsilon, int minPts, std::vector<std::vector<int>> &labels)
{
    std::vector<uint64_t> off(sets.size() + 1, 0);
    for (size_t k = 0; k < sets.size(); k++) off[k + 1] = off[k] + sets[k]->size();
    std::vector<uint32_t> s(off.back()), e(off.back());
    std::vector<int> lab(off.back(), -1);
    for (size_t k = 0; k < sets.size(); k++)
        for (size_t i = 0; i < sets[k]->size(); i++) { s[off[k] + i] = (*sets[k])[i].start; e[off[k] + i] = (*sets[k])[i].end; }
    if (!s.empty()) check(csvgpu_dbscan_iv_batch(csvhost::context(), s.data(), e.data(), off.data(), sets.size(), epsilon, minPts, lab.data()), "DBSCAN::fitBatch");
    labels.resize(sets.size());
    for (size_t k = 0; k < sets.size(); k++) labels[k].assign(lab.begin() + (std::ptrdiff_t)off[k], lab.begin() + (std::ptrdiff_t)off[k + 1]);
}

void DBSCAN1D::fit(const std::vector<int> &points)
{
    clusters.assign(points.size(), -1);
    if (points.empty()) return;
    const uint64_t off[2] = {0, points.size()};
    check(csvgpu_dbscan_1d(csvhost::context(), points.data(), off, 1, epsilon, minPts, clusters.data()), "DBSCAN1D::fit");
}

void DBSCAN1D::fitBatch(const std::vector<std::vector<int>> &sets, double epsilon, int minPts, std::vector<std::vector<int>> &labels)
{
    std::vector<uint64_t> off(sets.size() + 1, 0);
    for (size_t k = 0; k < sets.size(); k++) off[k + 1] = off[k] + sets[k].size();
    std::vector<int> flat(off.back()), lab(off.back());
    for (size_t k = 0; k < sets.size(); k++) std::copy(sets[k].begin(), sets[k].end(), flat.begin() + off[k]);
    if (!flat.empty())
        check(csvgpu_dbscan_1d(csvhost::context(), flat.data(), off.data(), sets.size(), epsilon, minPts, lab.data()), "DBSCAN1D::fitBatch");
    labels.resize(sets.size());
    for (size_t k = 0; k < sets.size(); k++) labels[k].assign(lab.begin() + off[k], lab.begin() + off[k + 1]);
}

void DBSCAN1D::fitBatchFlat(const std::vector<int> &points, const std::vector<uint64_t> &off, double epsilon, int minPts, std::vector<int> &labels)
{
    labels.assign(points.size(), -1);
    if (points.empty() || off.size() < 2) return;
    check(csvgpu_dbscan_1d(csvhost::context(), points.data(), off.data(), off.size() - 1, epsilon, minPts, labels.data()), "DBSCAN1D::fitBatch");
}

// members of the most populated cluster in index order; the first strictly larger bucket in ascending
// id order wins, so ties go to the lowest id; no cluster -> empty (reference dbscan1d.cpp:72-90)
std::vector<int> DBSCAN1D::getLargestCluster(const std::vector<int> &points)
{
    std::map<int, size_t> sizes;
    for (int c : clusters) if (c >= 0) sizes[c]++;
    int best = -1; size_t best_n = 0;
    for (const auto &kv : sizes) if (kv.second > best_n) { best_n = kv.second; best = kv.first; }
    std::vector<int> out;
    if (best < 0) return out;
    for (size_t i = 0; i < clusters.size(); i++) if (clusters[i] == best) out.push_back(points[i]);
    return out;
}
