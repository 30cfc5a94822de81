// sv_object.h — SVCall and the clustering entry points, same names/signatures as the reference
// (include/sv_object.h:16-49) so callers switch by relinking. mergeSVs gets its DBSCAN labels from the
// HIP kernels through the C-ABI (host/dbscan.h); everything order-defining stays on the host with the
// same libstdc++ algorithms (std::sort is unstable: its result on a given input order is part of the
// observable output — SURVEY §7 hard part 2).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "sv_types.h"

using namespace sv_types;

struct SVCall {
    uint32_t start = 0;
    uint32_t end = 0;
    SVType sv_type = SVType::UNKNOWN;
    std::string alt_allele = ".";
    SVEvidenceFlags aln_type;
    Genotype genotype = Genotype::UNKNOWN;
    double hmm_likelihood = 0.0;
    int cn_state = 0;
    int aln_offset = 0;
    int cluster_size = 0;

    bool operator<(const SVCall &o) const { return start < o.start || (start == o.start && end < o.end); }

    SVCall() = default;
    SVCall(uint32_t start, uint32_t end, SVType sv_type, const std::string &alt_allele, SVEvidenceFlags aln_type,
           Genotype genotype, double hmm_likelihood, int cn_state, int aln_offset, int cluster_size)
        : start(start), end(end), sv_type(sv_type), alt_allele(alt_allele), aln_type(aln_type), genotype(genotype),
          hmm_likelihood(hmm_likelihood), cn_state(cn_state), aln_offset(aln_offset), cluster_size(cluster_size) {}
};

void addSVCall(std::vector<SVCall> &sv_calls, SVCall &sv_call);
void mergeDuplicateSVs(std::vector<SVCall> &sv_calls);
uint32_t getSVCount(const std::vector<SVCall> &sv_calls);
void concatenateSVCalls(std::vector<SVCall> &target, const std::vector<SVCall> &source);
void mergeSVs(std::vector<SVCall> &sv_calls, double epsilon, int min_pts, bool keep_noise, const std::string &json_filepath = "");

// mergeSVs on many call vectors at once (the final merges of a run, one vector per contig): every (vector, type) set of two or more
// calls goes through ONE batched device fit, the representative choices run on the host pool. Same result as mergeSVs on each.
void mergeSVsMany(const std::vector<std::vector<SVCall> *> &sets, double epsilon, int min_pts, bool keep_noise, int threads = 0);

// The part of mergeSVs after DBSCAN::fit: `type_calls` are the calls of one SV type in vector order and
// `labels` their cluster labels; appends the representatives to `merged` (sv_object.cpp:97-264 of the reference).
// Exposed so the per-chromosome pipeline can feed labels that are already on hand from the device pipeline.
void mergeTypeWithLabels(std::vector<SVCall> &type_calls, const int32_t *labels, bool keep_noise, std::vector<SVCall> &merged);
