// khmm.h — CHMM + ReadCHMM + testVit_CHMM with the reference's signatures (include/khmm.h:14-49).
// The Viterbi itself runs in hmm.hip through csvgpu_viterbi.
#pragma once
#include <string>
#include <utility>
#include <vector>

struct CHMM {
    int N = 0;
    int M = 0;
    std::vector<std::vector<double>> A;
    std::vector<std::vector<double>> B;
    std::vector<double> pi;
    std::vector<double> B1_mean;
    std::vector<double> B1_sd;
    double B1_uf = 0.0;
    std::vector<double> B2_mean;
    std::vector<double> B2_sd;
    double B2_uf = 0.0;
    int NP_flag = 0;
    std::vector<double> B3_mean;
    std::vector<double> B3_sd;
    double B3_uf = 0.0;
    int dist = 0;
};

CHMM ReadCHMM(const std::string filename);
std::pair<std::vector<int>, double> testVit_CHMM(CHMM hmm, int T, std::vector<double> &O1, std::vector<double> &O2, std::vector<double> &pfb);

// batched form: one device call for many observation sequences (all SVs of a chromosome)
struct VitBatch {
    std::vector<double> o1, o2, pfb;
    std::vector<uint64_t> seq_off{0};
    void add(const std::vector<double> &a, const std::vector<double> &b, const std::vector<double> &c);
};
void testVit_CHMM_batch(const CHMM &hmm, const VitBatch &batch, std::vector<int> &states, std::vector<double> &loglik);
