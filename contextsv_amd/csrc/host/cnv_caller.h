// cnv_caller.h — host mirror of the copy-number pass of the reference's CNVCaller / SVCaller
// (src/cnv_caller.cpp:41-387, src/sv_caller.cpp:983-1064), batched over the device kernels:
//   * the per-window depth sums + log2 ratios of querySNPRegion (cnv_caller.cpp:76-113) run in hmm.hip on the depth map
//     that stays resident in the shard (csvgpu_window_log2_resident), for all SVs of a chromosome in one launch;
//   * the observation sequences are assembled on the host exactly like the reference does — through a
//     std::unordered_map<std::string,double> keyed "ws-we", whose libstdc++ iteration order IS the observation
//     order (cnv_caller.cpp:77,124) — and all of them go through ONE batched Viterbi call (csvgpu_viterbi);
//   * the state votes and SVCall update rules are the reference's, line for line in behaviour.
// The reference re-opens and re-indexes the SNP VCF for every SV (readSNPAlleleFrequencies, cnv_caller.cpp:558-809);
// that reader is I/O outside this path, so SNPs come from a caller-supplied SNPSource.
#pragma once
#include <cstdint>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "../../../include/csvgpu.h"
#include "khmm.h"
#include "sv_object.h"

struct SNPData {
    std::vector<uint32_t> pos;
    std::vector<double> pfb;
    std::vector<double> baf;
    std::vector<double> log2_cov;
    std::vector<int> state_sequence;
    std::vector<bool> is_snp;
    double mean_chr_cov = 0;
};

// What readSNPAlleleFrequencies hands back for one region: positions (file order) and the two hash maps.
struct SNPSource {
    virtual ~SNPSource() = default;
    virtual void query(uint32_t start_pos, uint32_t end_pos, std::vector<uint32_t> &snp_pos,
                       std::unordered_map<uint32_t, double> &snp_baf, std::unordered_map<uint32_t, double> &snp_pfb) const = 0;
    // The same result without the maps, APPENDED to the three vectors: positions in the order query() gives them, and for each the
    // value snp_baf[pos] / snp_pfb[pos] would hold after query() (pfb: 0.0 where the map has no entry). Default: through query().
    virtual void queryFlat(uint32_t start_pos, uint32_t end_pos, std::vector<uint32_t> &snp_pos, std::vector<double> &baf_at,
                           std::vector<double> &pfb_at) const;
};

// SNPs of one chromosome held in sorted arrays (the one-time-per-chromosome load the reference lacks).
// has_pfb[i] == 0 reproduces the reference's default-constructed 0.0 for SNPs without a gnomAD hit (cnv_caller.cpp:138).
struct SNPTable : SNPSource {
    std::vector<uint32_t> pos;
    std::vector<double> baf, pfb;
    std::vector<uint8_t> has_pfb;
    void query(uint32_t start_pos, uint32_t end_pos, std::vector<uint32_t> &snp_pos, std::unordered_map<uint32_t, double> &snp_baf,
               std::unordered_map<uint32_t, double> &snp_pfb) const override;
    void queryFlat(uint32_t start_pos, uint32_t end_pos, std::vector<uint32_t> &snp_pos, std::vector<double> &baf_at,
                   std::vector<double> &pfb_at) const override;
};

class CNVCaller {
public:
    explicit CNVCaller(csv_ctx *ctx) : ctx(ctx) {}
    int sample_size = 20;            // --sample-size (input_data.cpp:18-37)
    uint32_t min_cnv_length = 2000;  // --min-cnv
    bool save_cnv_data = false;      // --save-cnv: CNVCalls.json records for the plotting scripts (cnv_caller.cpp:179-198, :243-284)
    std::string cnv_output_file;     //   <outdir>/CNVCalls.json (main.cpp:109-118)

    static Genotype getGenotypeFromCNState(int cn_state);   // cnv_caller.h:76-97 of the reference

    // querySNPRegion for many regions at once (one window launch); out[i] belongs to regions[i].
    void querySNPRegions(const std::vector<std::pair<uint32_t, uint32_t>> &regions, csv_shard *shard, double mean_chr_cov,
                         const SNPSource &snps, std::vector<SNPData> &out) const;

    // runViterbi for many SNPData at once (one Viterbi launch)
    void runViterbi(const CHMM &hmm, const std::vector<SNPData> &data, std::vector<std::pair<std::vector<int>, double>> &predictions) const;

    // cnv_caller.cpp:290-387 — updates sv_candidates in place; returns the number of candidates that went through the HMM
    size_t runCIGARCopyNumberPrediction(const std::string &chr, std::vector<SVCall> &sv_candidates, const CHMM &hmm, double mean_chr_cov,
                                      csv_shard *shard, const SNPSource &snps) const;

    // cnv_caller.cpp:166-287 for a batch of regions. With save_cnv_data the flanking half-length windows are queried too and
    // every region with a copy-number change of >= 30 kb appends a record to cnv_output_file; depth_len (the depth map's size)
    // bounds the right flank as pos_depth_map.size() does there.
    void runCopyNumberPredictions(const std::string &chr, const CHMM &hmm, const std::vector<std::pair<uint32_t, uint32_t>> &regions,
                                  double mean_chr_cov, csv_shard *shard, const SNPSource &snps,
                                  std::vector<std::tuple<double, SVType, Genotype, int>> &results, uint32_t depth_len = 0) const;

    // cnv_caller.cpp:811-974 (appends one record, opening the array when the file is empty) and utils.cpp:63-71 (closes it)
    void saveSVCopyNumberToJSON(SNPData &before_sv, SNPData &after_sv, SNPData &snp_data, const std::string &chr, uint32_t start, uint32_t end,
                                const std::string &sv_type, double likelihood, const std::string &filepath) const;
    static void closeJSON(const std::string &filepath);

    // One contig's share of a genome-wide copy-number pass.
    struct ContigJob {
        std::string chr;
        std::vector<SVCall> *calls = nullptr;
        double mean_chr_cov = 0.0;
        csv_shard *shard = nullptr;
        const SNPSource *snps = nullptr;
        uint32_t depth_len = 0;
    };
    // runCIGARCopyNumberPrediction / runSplitReadCopyNumberPredictions for every contig of a run at once: the window launches go
    // contig by contig (each on its own resident depth map), the observation vectors of all candidates are assembled on the host
    // pool, and ONE Viterbi launch covers the genome. Same results as the per-contig forms.
    size_t runCIGARCopyNumberPredictionAll(std::vector<ContigJob> &jobs, const CHMM &hmm) const;
    void runSplitReadCopyNumberPredictionsAll(std::vector<ContigJob> &jobs, const CHMM &hmm) const;
    int host_threads = 0;            // host pool threads for the per-region work (0: all); results do not depend on it

    // sv_caller.cpp:983-1064 — the five-way update / duplicate rule for split-read candidates
    void runSplitReadCopyNumberPredictions(const std::string &chr, std::vector<SVCall> &split_sv_calls, const CHMM &hmm,
                                           double mean_chr_cov, csv_shard *shard, const SNPSource &snps, uint32_t depth_len = 0) const;

private:
    csv_ctx *ctx;
    struct RegionBatch;
    struct SnpChunk;
    struct ObsChunk;
    struct GenomeObs;
    void prepareWindows(RegionBatch &B, const SNPSource &snps) const;
    void queryChunk(RegionBatch &B, const SNPSource &snps, size_t c) const;
    void finishWindows(RegionBatch &B) const;
    void launchWindows(RegionBatch &B, csv_shard *shard, double mean_chr_cov) const;
    void assembleRegion(const RegionBatch &B, size_t i, ObsChunk &out) const;
    void observeAndDecode(std::vector<RegionBatch> &batches, const std::vector<ContigJob> &jobs, const CHMM &hmm, GenomeObs &G) const;
    void applyCIGARPrediction(const std::string &chr, SVCall &sv_call, const uint32_t *obs_pos, const int *state_sequence, size_t T, double likelihood) const;
    static int splitVote(const int *seq, size_t T);
    static void applySplitPredictions(std::vector<SVCall> &split_sv_calls, const std::vector<std::tuple<double, SVType, Genotype, int>> &results);
};
