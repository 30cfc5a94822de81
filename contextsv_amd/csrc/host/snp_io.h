// snp_io.h — SNP / population-frequency VCF ingestion without htslib (reference: CNVCaller::readSNPAlleleFrequencies,
// src/cnv_caller.cpp:558-809; the PFB path table of InputData, src/input_data.cpp:211-310).
// The reference opens, indexes and region-queries both VCFs once per SV. Here each file is streamed once (BGZF blocks
// inflated by a pool, or plain text), the records that pass the reference's filters are kept per contig in sorted arrays,
// and query() answers a region from those arrays with the reference's result, including its quirks:
//   * positions come back in file order, duplicates included; the BAF map keeps the last duplicate;
//   * at most ONE population frequency per region: the first gnomAD SNP record inside [min SNP, max SNP] whose position is a
//     kept SNP, that has the AF key, and whose value is not <= 0.01 or >= 0.99 (a "." value is NaN and passes that test).
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "bgzf.h"
#include "cnv_caller.h"
#include "hugevec.h"

// Lines of a text file that is BGZF-compressed (.vcf.gz) or plain, streamed in batches.
class TextFile {
public:
    bool open(const std::string &path, std::string *err);
    // on_lines(chunk): chunk holds whole lines ('\n'-terminated except possibly the last of the file); return false to stop early.
    bool forEachChunk(int threads, const std::function<bool(std::string_view)> &on_lines, std::string *err);
    bool compressed() const { return is_bgzf; }
private:
    bgzf::MappedFile file;
    bool is_bgzf = false;
};

// --pfb table: "<chr without the chr prefix>=<path>" lines (input_data.cpp:211-292)
class AlleleFreqFiles {
public:
    // false + err when the table or one of the listed files cannot be opened (the reference exits there)
    bool load(const std::string &table_path, std::string *err);
    std::string get(std::string chr) const;              // "" when the contig has no entry (input_data.cpp:294-310)
    bool empty() const { return paths.empty(); }
private:
    std::unordered_map<std::string, std::string> paths;
};

// One contig's SNPs and population-frequency hits.
struct SNPFileTable : SNPSource {
    std::vector<uint32_t> pos;            // file order (coordinate-sorted files: ascending; duplicates kept)
    std::vector<double> baf;
    std::vector<uint32_t> af_pos;         // gnomAD records that can become a region's hit, file order
    std::vector<double> af;
    void query(uint32_t start_pos, uint32_t end_pos, std::vector<uint32_t> &snp_pos, std::unordered_map<uint32_t, double> &snp_baf,
               std::unordered_map<uint32_t, double> &snp_pfb) const override;
};

// The sample's SNP VCF parsed once for every contig; population frequencies attached per contig on request.
class SNPFile {
public:
    // Streams the file once and keeps, per contig, the records that pass the reference's filters (SNP alleles only, QUAL > 30,
    // FORMAT/DP > 10, FILTER PASS or ".", FORMAT/AD with two values; cnv_caller.cpp:679-727). Like the reference's synced reader
    // it wants an index next to a compressed file (.tbi or .csi): without one, false.
    bool load(const std::string &snp_vcf, int threads, std::string *err);
    // The contig's table with gnomAD hits from `pfb_vcf` ("" = none) under INFO key AF[_<ethnicity>]; built on first use.
    // chr naming against the gnomAD file follows the reference's path heuristic (cnv_caller.cpp:624-640).
    const SNPFileTable &table(const std::string &chr, const std::string &pfb_vcf, const std::string &ethnicity, int threads);
    uint64_t records_kept() const { return n_kept; }
private:
    std::unordered_map<std::string, SNPFileTable> tables;
    std::unordered_map<std::string, bool> af_done;
    SNPFileTable none;
    uint64_t n_kept = 0;
};

// the gnomAD contig name for `chr` given the file path (cnv_caller.cpp:624-640)
std::string gnomadContigName(const std::string &chr, const std::string &pfb_filepath);
