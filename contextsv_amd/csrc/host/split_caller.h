// split_caller.h — host mirror of SVCaller::findSplitSVSignatures (src/sv_caller.cpp:68-504): split-alignment evidence ->
// SPLITDIST1 insertion / unknown candidates and SPLIT (or inversion) dummy calls for the copy-number pass.
//
// What moved to the GPU: the per-record CIGAR walk of the reference's third BAM pass (getAlignmentReadPositions +
// bam_endpos, :152, :162) — the scan kernel already produced ref_end / q_start / q_end for every record — and the six
// DBSCAN1D(100, 5) fits per overlap group (:270-372), which are collected for ALL groups of a chromosome and solved by
// one batched launch (DBSCAN1D::fitBatch -> csvgpu_dbscan_1d). What stays here, with the reference's own containers
// because their iteration order is observable (SURVEY §7 hard part 2): the qname-keyed unordered_maps, the
// (unbalanced) interval tree built in hash order, the overlap groups, the medians and votes.
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "sv_object.h"

// One alignment record as the split pass sees it (file order).
struct SplitRecord {
    int32_t  tid;
    int32_t  pos;        // 0-based
    uint16_t flag;
    uint8_t  mapq;
    int32_t  ref_end;    // bam_endpos           } from the scan kernel
    int32_t  q_start;    // query_start          } (csv_chr_result.ref_end / q_start / q_end
    int32_t  q_end;      // query_end            }  or csvgpu_aln_intervals)
};

struct SplitParams {
    int min_mapq = 20;        // sv_caller.h:72
    double eps = 100;         // DBSCAN1D(100, 5) at sv_caller.cpp:270
    int min_pts = 5;
    int min_length = 2000;    // :243
    int max_length = 1000000; // :244
};

// records[i] belongs to qnames[i]; target_names[tid] is the contig name. Fills sv_calls[contig] like the reference.
void findSplitSVSignatures(const std::vector<SplitRecord> &records, const std::vector<std::string> &qnames,
                           const std::vector<std::string> &target_names, const SplitParams &params,
                           std::unordered_map<std::string, std::vector<SVCall>> &sv_calls);
