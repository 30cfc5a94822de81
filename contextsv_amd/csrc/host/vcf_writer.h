// vcf_writer.h — the reference's VCF output (SVCaller::saveToVCF, src/sv_caller.cpp:1067-1330, and
// SVCaller::getReadDepth, :1332-1344) over depth maps that stay resident in HBM: the SUPPORT / DP values are one
// small device gather per contig (csvgpu_depth_lookup_resident) instead of a 4·(contig length) byte host vector.
// Record text is byte-identical to the reference's for the same calls, genome and depth values.
#pragma once
#include <cstdint>
#include <ostream>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../../include/csvgpu.h"
#include "fasta_query.h"
#include "sv_object.h"

// Where depth[pos] comes from. out[i] = depth at pos[i], or -1 when pos[i] is outside the contig's map
// (std::vector::at -> out_of_range in the reference: the writer then warns and uses 0).
// A contig without a map: std::out_of_range (chr_pos_depth_map.at(chr), :1306).
struct DepthSource {
    virtual ~DepthSource() = default;
    virtual void depthAt(const std::string &chr, const std::vector<uint32_t> &pos, std::vector<int32_t> &out) const = 0;
};

// the reference's own container (chr -> per-position depth vector)
struct HostDepthSource : DepthSource {
    explicit HostDepthSource(const std::unordered_map<std::string, std::vector<uint32_t>> &m) : map(m) {}
    void depthAt(const std::string &chr, const std::vector<uint32_t> &pos, std::vector<int32_t> &out) const override;
    const std::unordered_map<std::string, std::vector<uint32_t>> &map;
};

// depth maps resident in shards (what SVCaller::processChromosome(..., &shard) hands back)
struct ShardDepthSource : DepthSource {
    explicit ShardDepthSource(csv_ctx *ctx) : ctx(ctx) {}
    void add(const std::string &chr, csv_shard *shard) { shards[chr] = shard; }
    void depthAt(const std::string &chr, const std::vector<uint32_t> &pos, std::vector<int32_t> &out) const override;
    csv_ctx *ctx;
    std::unordered_map<std::string, csv_shard *> shards;
};

struct VCFOptions {
    std::string assembly_gaps;       // --assembly-gaps BED path, "" = none (input_data.getAssemblyGaps())
    std::string output_dir;          // the file is <output_dir>/output.vcf (:1103-1104)
    std::string file_date;           // "" = today, strftime("%Y%m%d") of local time (:1153-1160); settable so tests are reproducible
};

struct VCFCounts { int total = 0, unclassified = 0, assembly_gap_filtered = 0; };

// "ContextSV v<major>.<minor>.<patch>" (:1163, include/version.h)
std::string svMethodString();

// chr -> [start, end] 0-based BED rows in file order; malformed rows are reported and skipped (:1073-1099).
// false when the file cannot be opened.
bool loadAssemblyGaps(const std::string &path, std::unordered_map<std::string, std::vector<std::pair<uint32_t, uint32_t>>> &gaps);

// Header + records to any stream; contigs are written in the order given.
VCFCounts writeVCF(std::ostream &out, const std::vector<std::pair<std::string, const std::vector<SVCall> *>> &contigs,
                   const VCFOptions &opt, const std::unordered_map<std::string, std::vector<std::pair<uint32_t, uint32_t>>> &gaps,
                   const ReferenceGenome &ref_genome, const DepthSource &depth);

// The reference's entry point: iterates the map (so contig order is the unordered_map's, as there) and writes
// <output_dir>/output.vcf. Returns false where the reference returns early (gap file or output file not openable).
bool saveToVCF(const std::unordered_map<std::string, std::vector<SVCall>> &sv_calls, const VCFOptions &opt,
               const ReferenceGenome &ref_genome, const DepthSource &depth, VCFCounts *counts = nullptr);
