// khmm.cpp — HMM parameter file parser (grammar and stopping point of the reference's ReadCHMM,
// src/khmm.cpp:395-553) and the testVit_CHMM seam forwarded to the device.
#include "khmm.h"

#include <fstream>
#include <limits>
#include <sstream>
#include <stdexcept>

#include "../../../include/csvgpu.h"
#include "dbscan.h"
#include "log.h"

namespace {

bool read_numbers(std::ifstream &f, size_t n, std::vector<double> &out)
{
    out.resize(n);
    for (size_t i = 0; i < n; i++) if (!(f >> out[i])) return false;
    f.ignore(std::numeric_limits<std::streamsize>::max(), '\n');
    return true;
}

bool expect_line(std::ifstream &f, const char *tag)
{
    std::string line;
    return std::getline(f, line) && line == tag;
}

bool read_scalar_line(std::ifstream &f, double &v)
{
    std::string line;
    if (!std::getline(f, line)) return false;
    try { v = std::stod(line); } catch (const std::exception &) { return false; }
    return true;
}

csv_hmm to_pod(const CHMM &h)
{
    if (h.N != 6 || h.A.size() != 6 || h.pi.size() != 6 || h.B1_mean.size() != 6 || h.B1_sd.size() != 6 || h.B2_mean.size() != 5 || h.B2_sd.size() != 5)
        throw std::runtime_error("testVit_CHMM: the device Viterbi is a fixed 6-state DP (N=6, 5 BAF components)");
    csv_hmm p;
    for (int i = 0; i < 6; i++) {
        if (h.A[i].size() != 6) throw std::runtime_error("testVit_CHMM: A must be 6x6");
        for (int j = 0; j < 6; j++) p.A[i * 6 + j] = h.A[i][j];
        p.pi[i] = h.pi[i]; p.B1_mean[i] = h.B1_mean[i]; p.B1_sd[i] = h.B1_sd[i];
    }
    for (int i = 0; i < 5; i++) { p.B2_mean[i] = h.B2_mean[i]; p.B2_sd[i] = h.B2_sd[i]; }
    p.B1_uf = h.B1_uf; p.B2_uf = h.B2_uf;
    return p;
}

}  // namespace

CHMM ReadCHMM(const std::string filename)
{
    std::ifstream f(filename);
    if (!f.is_open()) { printError("Error opening file"); return CHMM(); }
    CHMM h;
    std::string line;
    auto fail = [](const char *what) { printError(std::string("Error reading ") + what); return CHMM(); };
    if (!std::getline(f, line) || sscanf(line.c_str(), "M=%d", &h.M) != 1) return fail("M");
    if (!std::getline(f, line) || sscanf(line.c_str(), "N=%d", &h.N) != 1) return fail("N");
    if (h.N <= 0 || h.M <= 0) return fail("N");
    std::vector<double> flat;
    if (!expect_line(f, "A:") || !read_numbers(f, (size_t)h.N * h.N, flat)) return fail("A");
    h.A.assign(h.N, std::vector<double>(h.N));
    for (int i = 0; i < h.N; i++) for (int j = 0; j < h.N; j++) h.A[i][j] = flat[(size_t)i * h.N + j];
    if (!expect_line(f, "B:") || !read_numbers(f, (size_t)h.N * h.M, flat)) return fail("B");
    h.B.assign(h.N, std::vector<double>(h.M));
    for (int i = 0; i < h.N; i++) for (int j = 0; j < h.M; j++) h.B[i][j] = flat[(size_t)i * h.M + j];
    if (!expect_line(f, "pi:") || !read_numbers(f, h.N, h.pi)) return fail("pi");
    if (!expect_line(f, "B1_mean:") || !read_numbers(f, h.N, h.B1_mean)) return fail("B1_mean");
    if (!expect_line(f, "B1_sd:") || !read_numbers(f, h.N, h.B1_sd)) return fail("B1_sd");
    if (!expect_line(f, "B1_uf:") || !read_scalar_line(f, h.B1_uf)) return fail("B1_uf");
    if (!expect_line(f, "B2_mean:") || !read_numbers(f, 5, h.B2_mean)) return fail("B2_mean");
    if (!expect_line(f, "B2_sd:") || !read_numbers(f, 5, h.B2_sd)) return fail("B2_sd");
    if (!expect_line(f, "B2_uf:") || !read_scalar_line(f, h.B2_uf)) return fail("B2_uf");
    return h;   // B3_* lines, if present, are not read (the reference stops here too)
}

std::pair<std::vector<int>, double> testVit_CHMM(CHMM hmm, int T, std::vector<double> &O1, std::vector<double> &O2, std::vector<double> &pfb)
{
    const csv_hmm p = to_pod(hmm);
    const uint64_t off[2] = {0, (uint64_t)(T > 0 ? T : 0)};
    std::vector<int> states(off[1]);
    double ll = 0.0;
    const int rc = csvgpu_viterbi(csvhost::context(), &p, O1.data(), O2.data(), pfb.data(), off, 1, states.data(), &ll);
    if (rc != CSV_OK) throw std::runtime_error(std::string("testVit_CHMM: ") + csvgpu_last_error(csvhost::context()));
    return std::make_pair(states, ll);
}

void VitBatch::add(const std::vector<double> &a, const std::vector<double> &b, const std::vector<double> &c)
{
    o1.insert(o1.end(), a.begin(), a.end());
    o2.insert(o2.end(), b.begin(), b.end());
    pfb.insert(pfb.end(), c.begin(), c.end());
    seq_off.push_back(o1.size());
}

void testVit_CHMM_batch(const CHMM &hmm, const VitBatch &b, std::vector<int> &states, std::vector<double> &loglik)
{
    const csv_hmm p = to_pod(hmm);
    const uint64_t n_seq = b.seq_off.size() - 1;
    states.assign(b.o1.size(), 0);
    loglik.assign(n_seq, 0.0);
    if (!n_seq) return;
    const int rc = csvgpu_viterbi(csvhost::context(), &p, b.o1.data(), b.o2.data(), b.pfb.data(), b.seq_off.data(), n_seq, states.data(), loglik.data());
    if (rc != CSV_OK) throw std::runtime_error(std::string("testVit_CHMM_batch: ") + csvgpu_last_error(csvhost::context()));
}
