// synth.h — deterministic synthetic long-read shards: the benchmark's and the large tests'
// input. Not a reference component: the reference has no data generator and its sample data are not
// available offline, so inputs of the same shape are generated here.
#pragma once
#include <cstdint>
#include <vector>

#include "../../../include/csvgpu.h"

struct SynthParams {
    uint64_t seed = 0x5EED0000ull;
    uint32_t chr_len = 50818468;     // GRCh38 chr22
    double   depth = 30.0;
    int      tech = 0;               // 0 = ONT (lognormal lengths, 5 % indel events/base), 1 = HiFi (18 kb, 0.1 %)
    double   sv_per_bp = 1.0 / 120000.0;
    int      threads = 8;
    int      with_seq = 0;           // also generate 4-bit packed sequences (needed only for 50-bp INS ALT strings)
};

struct SynthShard {
    std::vector<int32_t>  pos;
    std::vector<uint16_t> flag;
    std::vector<uint8_t>  mapq;
    std::vector<int32_t>  tid;
    std::vector<uint64_t> cigar_off;
    std::vector<uint32_t> cigar;
    std::vector<uint64_t> seq_off;
    std::vector<uint8_t>  seq;
    std::vector<uint32_t> qname_id;     // [n_reads] read identity: a primary and its supplementary record share it (query name "r<tid>_<id>")
    uint32_t depth_len = 0;
    uint64_t n_truth_sv = 0;
    csv_reads view() const;
};

void synth_generate(const SynthParams &p, SynthShard &out);
void synth_snps(uint64_t seed, uint32_t chr_len, std::vector<uint32_t> &pos, std::vector<double> &baf);
