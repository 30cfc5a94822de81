// dbscan.h / dbscan1d — the reference's clustering classes (include/dbscan.h:11-33, include/dbscan1d.h:11-32)
// with fit() forwarded to the HIP kernels through the C-ABI. The context is process-global per thread
// (one GPU per process): set it once with csvhost::set_context().
#pragma once
#include <vector>

#include "../../../include/csvgpu.h"
#include "sv_object.h"

namespace csvhost {
void set_context(csv_ctx *ctx);      // borrowed, not owned
void set_thread_context(csv_ctx *ctx);   // this thread's own context (nullptr: the process-wide one): a context serves one host thread at a time
csv_ctx *context();                  // throws std::runtime_error when unset (there is no CPU fallback)
}

class DBSCAN {
public:
    DBSCAN(double epsilon, int minPts) : epsilon(epsilon), minPts(minPts) {}
    void fit(const std::vector<SVCall> &sv_calls);
    const std::vector<int> &getClusters() const { return clusters; }

    // batched form: one device call for many interval sets (set k = sets[k]'s (start, end) pairs, labels[k] per set in its order)
    static void fitBatch(const std::vector<const std::vector<SVCall> *> &sets, double epsilon, int minPts, std::vector<std::vector<int>> &labels);

private:
    double epsilon;
    int minPts;
    std::vector<int> clusters;
};

class DBSCAN1D {
public:
    DBSCAN1D(double epsilon, int minPts) : epsilon(epsilon), minPts(minPts) {}
    void fit(const std::vector<int> &points);
    const std::vector<int> &getClusters() const { return clusters; }
    std::vector<int> getLargestCluster(const std::vector<int> &points);

    // batched form: one device call for many point sets (the six fits per overlap group of the split-read path)
    static void fitBatch(const std::vector<std::vector<int>> &sets, double epsilon, int minPts, std::vector<std::vector<int>> &labels);
    // the same on flat arrays: set k = points[off[k] .. off[k+1]), labels in the same layout
    static void fitBatchFlat(const std::vector<int> &points, const std::vector<uint64_t> &off, double epsilon, int minPts, std::vector<int> &labels);

private:
    double epsilon;
    int minPts;
    std::vector<int> clusters;
};
