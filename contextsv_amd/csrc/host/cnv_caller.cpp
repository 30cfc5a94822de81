#include "cnv_caller.h"

#include <fstream>
#include <stdexcept>

#include <algorithm>
#include <stdexcept>

#include <charconv>

#include "log.h"
#include "par.h"
#include "umap_order.h"

namespace {
void check(csv_ctx *ctx, int rc, const char *what)
{
    if (rc != CSV_OK) throw std::runtime_error(std::string(what) + ": " + csvgpu_last_error(ctx));
}
}  // namespace

void SNPTable::query(uint32_t start_pos, uint32_t end_pos, std::vector<uint32_t> &snp_pos, std::unordered_map<uint32_t, double> &snp_baf,
                     std::unordered_map<uint32_t, double> &snp_pfb) const
{
    const size_t a = std::lower_bound(pos.begin(), pos.end(), start_pos) - pos.begin();
    for (size_t i = a; i < pos.size() && pos[i] <= end_pos; i++) {
        snp_pos.push_back(pos[i]);
        snp_baf[pos[i]] = baf[i];
        if (!has_pfb.empty() && has_pfb[i]) snp_pfb[pos[i]] = pfb[i];
    }
}

Genotype CNVCaller::getGenotypeFromCNState(int cn_state)
{
    switch (cn_state) {
        case 0: return Genotype::UNKNOWN;
        case 1: case 4: case 6: return Genotype::HOMOZYGOUS_ALT;
        case 2: case 5: return Genotype::HETEROZYGOUS;
        case 3: return Genotype::HOMOZYGOUS_REF;
    }
    printError("ERROR: Invalid CN state: " + std::to_string(cn_state));
    return Genotype::UNKNOWN;
}

// ---- querySNPRegion, batched ---------------------------------------------------------------------------------------------------
// One RegionBatch = the regions of one contig: (A) SNP look-ups per region, (B) ONE window launch on the contig's resident depth map,
// (C) the observation vectors per region. A and C are independent per region and run on the host pool in chunks of kChunk regions,
// every chunk appending to its own flat arrays (a genome has ~1e4 regions of ~25 observations: per-region containers were most
// of the pass's time).
namespace { constexpr size_t kChunk = 64; }

struct CNVCaller::SnpChunk {                   // SNPs of up to kChunk regions back to back
    std::vector<uint32_t> pos;
    std::vector<double> baf, pfb;              // the values the reference's two hash maps hold for pos[i] (pfb: 0.0 when it has none)
    std::vector<uint32_t> off{0};
};
struct CNVCaller::ObsChunk {                   // observation vectors of up to kChunk regions back to back
    std::vector<uint32_t> pos;
    std::vector<double> baf, pfb, l2;
    std::vector<uint8_t> is_snp;
    std::vector<uint64_t> off{0};
};
struct CNVCaller::RegionBatch {
    std::vector<std::pair<uint32_t, uint32_t>> regions;
    std::vector<SnpChunk> snp;                 // region i: chunk i / kChunk, entry i % kChunk
    std::vector<uint32_t> r_start, r_end;
    std::vector<int32_t> r_ss;
    std::vector<uint64_t> win_off{0};
    std::vector<size_t> slot;                  // region -> row of the device batch (invalid regions have none)
    std::vector<double> log2_cov;
    std::vector<uint32_t> ws, we;
};

void SNPSource::queryFlat(uint32_t start_pos, uint32_t end_pos, std::vector<uint32_t> &snp_pos, std::vector<double> &baf_at, std::vector<double> &pfb_at) const
{
    std::vector<uint32_t> pos;
    std::unordered_map<uint32_t, double> baf, pfb;
    query(start_pos, end_pos, pos, baf, pfb);
    for (uint32_t p : pos) {
        snp_pos.push_back(p);
        baf_at.push_back(baf[p]);
        const auto f = pfb.find(p);
        pfb_at.push_back(f != pfb.end() ? f->second : 0.0);      // operator[] default-constructs 0.0 (cnv_caller.cpp:138)
    }
}

// sorted arrays: duplicates of a position are neighbours, and the maps keep the LAST duplicate's value
void SNPTable::queryFlat(uint32_t start_pos, uint32_t end_pos, std::vector<uint32_t> &snp_pos, std::vector<double> &baf_at, std::vector<double> &pfb_at) const
{
    const size_t a = std::lower_bound(pos.begin(), pos.end(), start_pos) - pos.begin();
    size_t b = a;
    while (b < pos.size() && pos[b] <= end_pos) b++;
    for (size_t i = a; i < b;) {
        size_t j = i;
        while (j + 1 < b && pos[j + 1] == pos[i]) j++;
        double f = 0.0;
        if (!has_pfb.empty()) for (size_t k = i; k <= j; k++) if (has_pfb[k]) f = pfb[k];
        for (size_t k = i; k <= j; k++) { snp_pos.push_back(pos[i]); baf_at.push_back(baf[j]); pfb_at.push_back(f); }
        i = j + 1;
    }
}

// SNP look-ups of a batch's regions in chunks of kChunk (independent: one task each), then the window tables
void CNVCaller::queryChunk(RegionBatch &B, const SNPSource &snps, size_t c) const
{
    const size_t n = B.regions.size();
    SnpChunk &ch = B.snp[c];
    for (size_t i = c * kChunk; i < std::min(n, (c + 1) * kChunk); i++) {
        snps.queryFlat(B.regions[i].first, B.regions[i].second, ch.pos, ch.baf, ch.pfb);
        ch.off.push_back((uint32_t)ch.pos.size());
    }
}

void CNVCaller::finishWindows(RegionBatch &B) const
{
    const size_t n = B.regions.size();
    for (size_t i = 0; i < n; i++) {
        const uint32_t start_pos = B.regions[i].first, end_pos = B.regions[i].second;
        if (start_pos > end_pos) {                          // the reference logs and leaves snp_data empty (cnv_caller.cpp:69-73)
            printError("ERROR: Invalid SNP region for copy number prediction: " + std::to_string((int)start_pos) + "-" + std::to_string((int)end_pos));
            continue;
        }
        const SnpChunk &ch = B.snp[i / kChunk];
        const int n_snps = (int)(ch.off[i % kChunk + 1] - ch.off[i % kChunk]);
        const int ss = std::max(n_snps, sample_size);       // :65
        B.slot[i] = B.r_start.size();
        B.r_start.push_back(start_pos); B.r_end.push_back(end_pos); B.r_ss.push_back(ss);
        B.win_off.push_back(B.win_off.back() + (uint64_t)ss);
    }
}

void CNVCaller::prepareWindows(RegionBatch &B, const SNPSource &snps) const
{
    const size_t n = B.regions.size();
    B.snp.assign((n + kChunk - 1) / kChunk, SnpChunk());
    B.slot.assign(n, SIZE_MAX);
    csvhost::parallel_for(B.snp.size(), host_threads, [&](size_t c) { queryChunk(B, snps, c); });
    finishWindows(B);
}

void CNVCaller::launchWindows(RegionBatch &B, csv_shard *shard, double mean_chr_cov) const
{
    const uint64_t nw = B.win_off.back();
    B.log2_cov.resize(nw); B.ws.resize(nw); B.we.resize(nw);
    if (!B.r_start.empty())
        check(ctx, csvgpu_window_log2_resident(ctx, shard, B.r_start.data(), B.r_end.data(), B.r_ss.data(), B.win_off.data(), B.r_start.size(), mean_chr_cov,
                                               B.log2_cov.data(), B.ws.data(), B.we.data()), "querySNPRegion");
}

// The reference keys the windows by the string "ws-we" in an unordered_map<std::string,double>: equal keys collapse (the later window's
// value wins, the node stays where the first put it) and the iteration order of that libstdc++ container is the observation order
// (:77, :111-112, :124). The order is replayed by UMapOrder on the keys' hashes (umap_order.h) — no strings, no nodes.
// Appends region i's observations to `out` (and closes its entry in out.off).
void CNVCaller::assembleRegion(const RegionBatch &B, size_t i, ObsChunk &out) const
{
    if (B.slot[i] != SIZE_MAX) {
        const uint64_t w0 = B.win_off[B.slot[i]], w1 = B.win_off[B.slot[i] + 1];
        static thread_local csvhost::UMapOrder order;
        static thread_local std::vector<uint64_t> first_w, last_w;            // by node: the window that created the key / that wrote it last
        order.clear(); first_w.clear(); last_w.clear();
        for (uint64_t w = w0; w < w1; w++) {
            char key[24];
            char *e = std::to_chars(key, key + 11, B.ws[w]).ptr;
            *e++ = '-';
            e = std::to_chars(e, e + 11, B.we[w]).ptr;
            const uint64_t h = csvhost::std_string_hash(key, (size_t)(e - key));
            const int64_t node = order.find(h, [&](uint32_t nd) { return B.ws[first_w[nd]] == B.ws[w] && B.we[first_w[nd]] == B.we[w]; });
            if (node >= 0) { last_w[(size_t)node] = w; continue; }
            order.insert_new(h);
            first_w.push_back(w); last_w.push_back(w);
        }
        const SnpChunk &sc = B.snp[i / kChunk];
        const uint32_t s0 = sc.off[i % kChunk], s1 = sc.off[i % kChunk + 1];
        order.for_each([&](uint32_t node) {
            const uint32_t window_start = B.ws[first_w[node]], window_end = B.we[first_w[node]];
            const double l2 = B.log2_cov[last_w[node]];
            bool snp_found = false;
            for (uint32_t k = s0; k < s1; k++) {
                const uint32_t pos = sc.pos[k];
                if (pos >= window_start && pos <= window_end) {        // inclusive both ends: a SNP can land in two windows (:132)
                    out.pos.push_back(pos); out.baf.push_back(sc.baf[k]); out.pfb.push_back(sc.pfb[k]);
                    out.l2.push_back(l2); out.is_snp.push_back(1);
                    snp_found = true;
                }
            }
            if (!snp_found) {                                           // dummy observation at the window centre (:144-155)
                out.pos.push_back((window_start + window_end) / 2); out.baf.push_back(-1.0); out.pfb.push_back(0.5);
                out.l2.push_back(l2); out.is_snp.push_back(0);
            }
        });
    }
    out.off.push_back(out.pos.size());
}

void CNVCaller::querySNPRegions(const std::vector<std::pair<uint32_t, uint32_t>> &regions, csv_shard *shard, double mean_chr_cov,
                                const SNPSource &snps, std::vector<SNPData> &out) const
{
    RegionBatch B;
    B.regions = regions;
    prepareWindows(B, snps);
    launchWindows(B, shard, mean_chr_cov);
    out.assign(regions.size(), SNPData());
    csvhost::parallel_for(regions.size(), host_threads, [&](size_t i) {
        ObsChunk o;
        assembleRegion(B, i, o);
        SNPData &d = out[i];
        d.pos = std::move(o.pos); d.baf = std::move(o.baf); d.pfb = std::move(o.pfb); d.log2_cov = std::move(o.l2);
        d.is_snp.assign(o.is_snp.begin(), o.is_snp.end());
    });
}

void CNVCaller::runViterbi(const CHMM &hmm, const std::vector<SNPData> &data, std::vector<std::pair<std::vector<int>, double>> &predictions) const
{
    VitBatch b;
    for (const SNPData &d : data) b.add(d.log2_cov, d.baf, d.pfb);
    std::vector<int> states;
    std::vector<double> ll;
    testVit_CHMM_batch(hmm, b, states, ll);
    predictions.resize(data.size());
    for (size_t i = 0; i < data.size(); i++) {
        if (data[i].pos.empty()) printError("ERROR: No SNP data found for Viterbi algorithm.");      // runViterbi logs and still calls testVit_CHMM with T = 0 (:43-49)
        predictions[i] = std::make_pair(std::vector<int>(states.begin() + (std::ptrdiff_t)b.seq_off[i], states.begin() + (std::ptrdiff_t)b.seq_off[i + 1]), ll[i]);
    }
}

// Observation vectors of ALL candidates of a genome-wide pass (chunked over the pool) and ONE Viterbi launch over them.
struct CNVCaller::GenomeObs {
    struct Cand { size_t job, call, region; };
    std::vector<Cand> cands;
    std::vector<ObsChunk> chunks;              // candidate q: chunk q / kChunk, entry q % kChunk
    std::vector<uint64_t> seq_off;             // candidate q's observations in the flat Viterbi batch
    std::vector<int> states;
    std::vector<double> ll;
    const uint32_t *pos(size_t q) const { const ObsChunk &c = chunks[q / kChunk]; return c.pos.data() + c.off[q % kChunk]; }
    size_t len(size_t q) const { return (size_t)(seq_off[q + 1] - seq_off[q]); }
};

void CNVCaller::observeAndDecode(std::vector<RegionBatch> &batches, const std::vector<ContigJob> &jobs, const CHMM &hmm, GenomeObs &G) const
{
    {
        csvhost::TraceScope tr("cn: snp queries");
        // the chunks of ALL contigs as one parallel section (24 sections of a few chunks each left most of the pool idle)
        std::vector<std::pair<uint32_t, uint32_t>> tasks;
        for (size_t j = 0; j < jobs.size(); j++) {
            RegionBatch &B = batches[j];
            const size_t n = B.regions.size();
            if (!n) continue;
            B.snp.assign((n + kChunk - 1) / kChunk, SnpChunk());
            B.slot.assign(n, SIZE_MAX);
            for (size_t c = 0; c < B.snp.size(); c++) tasks.emplace_back((uint32_t)j, (uint32_t)c);
        }
        csvhost::parallel_for(tasks.size(), host_threads, [&](size_t t) { queryChunk(batches[tasks[t].first], *jobs[tasks[t].first].snps, tasks[t].second); });
        for (size_t j = 0; j < jobs.size(); j++) if (!batches[j].regions.empty()) finishWindows(batches[j]);
    }
    {
        csvhost::TraceScope tr("cn: window launches");
        std::vector<csv_shard *> sh;
        std::vector<const uint32_t *> rs, re;
        std::vector<const int32_t *> ss;
        std::vector<const uint64_t *> wo;
        std::vector<uint64_t> nr;
        std::vector<double> mean;
        std::vector<double *> l2;
        std::vector<uint32_t *> ws, we;
        for (size_t j = 0; j < jobs.size(); j++) {
            RegionBatch &B = batches[j];
            if (B.r_start.empty()) continue;
            const uint64_t nw = B.win_off.back();
            B.log2_cov.resize(nw); B.ws.resize(nw); B.we.resize(nw);
            sh.push_back(jobs[j].shard); rs.push_back(B.r_start.data()); re.push_back(B.r_end.data()); ss.push_back(B.r_ss.data()); wo.push_back(B.win_off.data());
            nr.push_back(B.r_start.size()); mean.push_back(jobs[j].mean_chr_cov); l2.push_back(B.log2_cov.data()); ws.push_back(B.ws.data()); we.push_back(B.we.data());
        }
        if (!sh.empty())
            check(ctx, csvgpu_window_log2_resident_many(ctx, (int)sh.size(), sh.data(), rs.data(), re.data(), ss.data(), wo.data(), nr.data(), mean.data(), l2.data(),
                                                        ws.data(), we.data()), "querySNPRegion");
    }
    const size_t n = G.cands.size();
    G.chunks.assign((n + kChunk - 1) / kChunk, ObsChunk());
    {
        csvhost::TraceScope tr("cn: assemble");
        csvhost::parallel_for(G.chunks.size(), host_threads, [&](size_t c) {
            for (size_t q = c * kChunk; q < std::min(n, (c + 1) * kChunk); q++) assembleRegion(batches[G.cands[q].job], G.cands[q].region, G.chunks[c]);
        });
    }
    csvhost::TraceScope tr("cn: viterbi");
    VitBatch b;
    b.seq_off.assign(n + 1, 0);
    std::vector<uint64_t> chunk_base(G.chunks.size() + 1, 0);
    for (size_t c = 0; c < G.chunks.size(); c++) chunk_base[c + 1] = chunk_base[c] + G.chunks[c].pos.size();
    b.o1.resize(chunk_base.back()); b.o2.resize(chunk_base.back()); b.pfb.resize(chunk_base.back());
    csvhost::parallel_for(G.chunks.size(), host_threads, [&](size_t c) {
        const ObsChunk &ch = G.chunks[c];
        std::copy(ch.l2.begin(), ch.l2.end(), b.o1.begin() + (std::ptrdiff_t)chunk_base[c]);
        std::copy(ch.baf.begin(), ch.baf.end(), b.o2.begin() + (std::ptrdiff_t)chunk_base[c]);
        std::copy(ch.pfb.begin(), ch.pfb.end(), b.pfb.begin() + (std::ptrdiff_t)chunk_base[c]);
        for (size_t k = 0; k + 1 < ch.off.size(); k++) b.seq_off[c * kChunk + k + 1] = chunk_base[c] + ch.off[k + 1];
    });
    testVit_CHMM_batch(hmm, b, G.states, G.ll);
    G.seq_off = b.seq_off;
    for (size_t q = 0; q < n; q++)
        if (G.len(q) == 0) printError("ERROR: No SNP data found for Viterbi algorithm.");              // runViterbi logs and still calls testVit_CHMM with T = 0 (:43-49)
}

// cnv_caller.cpp:337-384 for one candidate given its observation positions and state path
void CNVCaller::applyCIGARPrediction(const std::string &chr, SVCall &sv_call, const uint32_t *obs_pos, const int *state_sequence, size_t T, double likelihood) const
{
    if (T == 0) {
        printError("ERROR: No SNP data found for Viterbi algorithm for CIGAR SV at " + chr + ":" + std::to_string((int)sv_call.start) + "-" + std::to_string((int)sv_call.end));
        return;
    }
    int counts[7] = {0, 0, 0, 0, 0, 0, 0}, n_in = 0;                // states of observations inside [start,end] (:337-346)
    for (size_t i = 0; i < T; i++)
        if (obs_pos[i] >= sv_call.start && obs_pos[i] <= sv_call.end) { counts[state_sequence[i]]++; n_in++; }
    int max_state = 0, max_count = 0;                                // first maximum wins (:350-360)
    for (int s = 1; s <= 6; s++) if (counts[s] > max_count) { max_state = s; max_count = counts[s]; }
    if ((double)max_count / (double)n_in < 0.50) max_state = 0;      // :363-367 (0/0 -> NaN < 0.5 is false, as in the reference)
    const Genotype genotype = getGenotypeFromCNState(max_state);
    SVType updated = getSVTypeFromCNState(max_state);
    updated = (updated == SVType::LOH) ? sv_call.sv_type : updated;  // :375
    if (isValidCopyNumberUpdate(sv_call.sv_type, updated)) {
        sv_call.sv_type = updated;
        sv_call.aln_type.set((size_t)SVDataType::HMM);
        sv_call.hmm_likelihood = likelihood;
        sv_call.genotype = genotype;
        sv_call.cn_state = max_state;
    }
}

size_t CNVCaller::runCIGARCopyNumberPrediction(const std::string &chr, std::vector<SVCall> &sv_candidates, const CHMM &hmm, double mean_chr_cov,
                                               csv_shard *shard, const SNPSource &snps) const
{
    std::vector<ContigJob> one(1);
    one[0].chr = chr; one[0].calls = &sv_candidates; one[0].mean_chr_cov = mean_chr_cov; one[0].shard = shard; one[0].snps = &snps;
    return runCIGARCopyNumberPredictionAll(one, hmm);
}

// Every contig's candidates in one go: window launches contig by contig (each on its own resident depth map), the observation
// vectors of ALL candidates assembled on the host pool, ONE Viterbi launch, the votes on the pool.
size_t CNVCaller::runCIGARCopyNumberPredictionAll(std::vector<ContigJob> &jobs, const CHMM &hmm) const
{
    GenomeObs G;
    std::vector<RegionBatch> batches(jobs.size());
    for (size_t j = 0; j < jobs.size(); j++) {
        std::vector<SVCall> &v = *jobs[j].calls;
        for (size_t k = 0; k < v.size(); k++) {
            const SVCall &c = v[k];
            if (c.start > c.end) {
                printError("ERROR: Invalid SV region for copy number prediction: " + jobs[j].chr + ":" + std::to_string((int)c.start) + "-" + std::to_string((int)c.end));
                continue;
            }
            if ((c.end - c.start) < min_cnv_length) continue;               // :315
            G.cands.push_back(GenomeObs::Cand{j, k, batches[j].regions.size()});
            batches[j].regions.emplace_back(c.start, c.end);
        }
    }
    if (G.cands.empty()) return 0;
    observeAndDecode(batches, jobs, hmm, G);
    csvhost::TraceScope tr("cn: votes");
    csvhost::parallel_for(G.chunks.size(), host_threads, [&](size_t c) {
        for (size_t q = c * kChunk; q < std::min(G.cands.size(), (c + 1) * kChunk); q++)
            applyCIGARPrediction(jobs[G.cands[q].job].chr, (*jobs[G.cands[q].job].calls)[G.cands[q].call], G.pos(q), G.states.data() + G.seq_off[q], G.len(q), G.ll[q]);
    });
    return G.cands.size();
}

// cnv_caller.cpp:214-238: the state a split-read region is given from its path: the largest non-neutral share if above 0.3, else
// neutral if its share is above 0.3, else 0
int CNVCaller::splitVote(const int *seq, size_t T)
{
    double pct[7] = {0, 0, 0, 0, 0, 0, 0};
    const double state_count = (double)T;
    double largest_non_neutral_pct = 0.0; int non_neutral_state = 0;
    for (int i = 0; i < 6; i++) {                                    // :214-224
        pct[i + 1] = (double)std::count(seq, seq + T, i + 1) / state_count;
        if (i + 1 != 3 && pct[i + 1] > largest_non_neutral_pct) { largest_non_neutral_pct = pct[i + 1]; non_neutral_state = i + 1; }
    }
    int max_state = 0;
    if (largest_non_neutral_pct > 0.3) max_state = non_neutral_state;   // :227-238
    else if (pct[3] > 0.3) max_state = 3;
    return max_state;
}

void CNVCaller::runCopyNumberPredictions(const std::string &chr, const CHMM &hmm, const std::vector<std::pair<uint32_t, uint32_t>> &regions,
                                         double mean_chr_cov, csv_shard *shard, const SNPSource &snps,
                                         std::vector<std::tuple<double, SVType, Genotype, int>> &results, uint32_t depth_len) const
{
    results.assign(regions.size(), std::make_tuple(0.0, SVType::UNKNOWN, Genotype::UNKNOWN, 0));
    std::vector<size_t> idx;
    std::vector<std::pair<uint32_t, uint32_t>> valid;
    for (size_t k = 0; k < regions.size(); k++) {
        if (regions[k].first > regions[k].second) {                      // :169-173
            printError("ERROR: Invalid SV region for copy number prediction: " + chr + ":" + std::to_string((int)regions[k].first) + "-" + std::to_string((int)regions[k].second));
            continue;
        }
        idx.push_back(k); valid.push_back(regions[k]);
    }
    if (idx.empty()) return;
    // --save-cnv: the half-length windows before and after each region ride in the same device batch (:176-198)
    const size_t nv = valid.size();
    std::vector<std::pair<uint32_t, uint32_t>> batch = valid;
    std::vector<long> before_at(nv, -1), after_at(nv, -1);
    if (save_cnv_data) {
        for (size_t q = 0; q < nv; q++) {
            const int start_pos = (int)valid[q].first, end_pos = (int)valid[q].second;
            const int half = (end_pos - start_pos) / 2;
            const int b0 = std::max(1, start_pos - half), b1 = std::max(1, start_pos - 1);
            if (b0 < b1) { before_at[q] = (long)batch.size(); batch.emplace_back((uint32_t)b0, (uint32_t)b1); }
            const int last = (int)depth_len - 1;
            const int a0 = std::min(last, end_pos + 1), a1 = std::min(last, end_pos + half);
            if (a0 < a1) { after_at[q] = (long)batch.size(); batch.emplace_back((uint32_t)a0, (uint32_t)a1); }
        }
    }
    std::vector<SNPData> data;
    querySNPRegions(batch, shard, mean_chr_cov, snps, data);
    std::vector<SNPData> sv_data(data.begin(), data.begin() + (std::ptrdiff_t)nv);
    std::vector<std::pair<std::vector<int>, double>> pred;
    runViterbi(hmm, sv_data, pred);
    for (size_t q = 0; q < idx.size(); q++) {
        std::vector<int> &seq = pred[q].first;
        if (seq.empty()) continue;                                       // :206-209
        const int max_state = splitVote(seq.data(), seq.size());
        const SVType predicted = getSVTypeFromCNState(max_state);
        results[idx[q]] = std::make_tuple(pred[q].second, predicted, getGenotypeFromCNState(max_state), max_state);

        const uint32_t start_pos = valid[q].first, end_pos = valid[q].second;
        const bool change = predicted != SVType::UNKNOWN && predicted != SVType::NEUTRAL;
        if (save_cnv_data && change && (end_pos - start_pos) >= 30000u) {          // :243-284
            SNPData none_before, none_after;
            SNPData &snp_data = data[q];
            SNPData &before_sv = before_at[q] >= 0 ? data[(size_t)before_at[q]] : none_before;
            SNPData &after_sv = after_at[q] >= 0 ? data[(size_t)after_at[q]] : none_after;
            snp_data.state_sequence = std::move(seq);
            for (SNPData *d : {&snp_data, &before_sv, &after_sv})
                for (size_t i = 0; i < d->pos.size(); i++)
                    if (!d->is_snp[i]) { d->baf[i] = 0.0; d->pfb[i] = 0.0; }
            printMessage("Saving SV copy number predictions to " + cnv_output_file + "...");
            saveSVCopyNumberToJSON(before_sv, after_sv, snp_data, chr, start_pos, end_pos, getSVTypeString(predicted), pred[q].second, cnv_output_file);
        }
    }
}

namespace {
template <class V>
void json_array(std::ostream &out, const char *indent_key, const V &v, const char *tail)
{
    out << indent_key << "[";
    for (size_t i = 0; i < v.size(); ++i) {
        out << v[i];
        if (i + 1 < v.size()) out << ", ";
    }
    out << tail;
}
void json_block(std::ostream &out, const char *name, const SNPData &d, bool with_states, const char *close)
{
    out << "  \"" << name << "\": {\n";
    json_array(out, "    \"positions\": ", d.pos, "],\n");
    json_array(out, "    \"b_allele_freq\": ", d.baf, "],\n");
    json_array(out, "    \"population_freq\": ", d.pfb, "],\n");
    json_array(out, "    \"log2_ratio\": ", d.log2_cov, "],\n");
    if (with_states) json_array(out, "    \"states\": ", d.state_sequence, "],\n");
    json_array(out, "    \"is_snp\": ", d.is_snp, "]\n");
    out << close;
}
}  // namespace

void CNVCaller::saveSVCopyNumberToJSON(SNPData &before_sv, SNPData &after_sv, SNPData &snp_data, const std::string &chr, uint32_t start, uint32_t end,
                                       const std::string &sv_type, double likelihood, const std::string &filepath) const
{
    std::ofstream json_file(filepath, std::ios::app);
    if (!json_file.is_open()) throw std::runtime_error("ERROR: Could not open JSON file for writing: " + filepath);   // the reference exits (:815-819)
    json_file.seekp(0, std::ios::end);
    if (json_file.tellp() == std::streampos(0)) json_file << "[\n";      // first record opens the array (:823-829)
    else json_file << "},\n";                                            // later ones close their predecessor
    json_file << "{\n";
    json_file << "  \"chromosome\": \"" << chr << "\",\n";
    json_file << "  \"start\": " << start << ",\n";
    json_file << "  \"end\": " << end << ",\n";
    json_file << "  \"sv_type\": \"" << sv_type << "\",\n";
    json_file << "  \"likelihood\": " << likelihood << ",\n";
    json_file << "  \"size\": " << (end - start + 1) << ",\n";
    json_block(json_file, "before_sv", before_sv, false, "  },\n");
    json_block(json_file, "after_sv", after_sv, false, "  },\n");
    json_block(json_file, "sv", snp_data, true, "  }\n");
    json_file.close();
    printMessage("Saved copy number predictions for " + chr + ":" + std::to_string(start) + "-" + std::to_string(end) + " to " + filepath);
}

void CNVCaller::closeJSON(const std::string &filepath)
{
    std::ofstream json_file(filepath, std::ios::app);
    json_file << "}\n";
    json_file << "]";
}

void CNVCaller::runSplitReadCopyNumberPredictions(const std::string &chr, std::vector<SVCall> &split_sv_calls, const CHMM &hmm, double mean_chr_cov,
                                                  csv_shard *shard, const SNPSource &snps, uint32_t depth_len) const
{
    std::vector<std::pair<uint32_t, uint32_t>> regions;
    for (const SVCall &c : split_sv_calls) regions.emplace_back(c.start, c.end);
    std::vector<std::tuple<double, SVType, Genotype, int>> results;
    runCopyNumberPredictions(chr, hmm, regions, mean_chr_cov, shard, snps, results, depth_len);
    applySplitPredictions(split_sv_calls, results);
}

// Every contig's split-read candidates in one go (no --save-cnv records: those are written region by region in the reference's order
// by the per-contig form above): window launches contig by contig, all observation vectors on the host pool, ONE Viterbi launch.
void CNVCaller::runSplitReadCopyNumberPredictionsAll(std::vector<ContigJob> &jobs, const CHMM &hmm) const
{
    if (save_cnv_data) {
        for (ContigJob &j : jobs) runSplitReadCopyNumberPredictions(j.chr, *j.calls, hmm, j.mean_chr_cov, j.shard, *j.snps, j.depth_len);
        return;
    }
    GenomeObs G;
    std::vector<RegionBatch> batches(jobs.size());
    std::vector<std::vector<std::tuple<double, SVType, Genotype, int>>> results(jobs.size());
    for (size_t j = 0; j < jobs.size(); j++) {
        const std::vector<SVCall> &v = *jobs[j].calls;
        results[j].assign(v.size(), std::make_tuple(0.0, SVType::UNKNOWN, Genotype::UNKNOWN, 0));
        for (size_t k = 0; k < v.size(); k++) {
            if (v[k].start > v[k].end) {                                  // :169-173
                printError("ERROR: Invalid SV region for copy number prediction: " + jobs[j].chr + ":" + std::to_string((int)v[k].start) + "-" + std::to_string((int)v[k].end));
                continue;
            }
            G.cands.push_back(GenomeObs::Cand{j, k, batches[j].regions.size()});
            batches[j].regions.emplace_back(v[k].start, v[k].end);
        }
    }
    if (!G.cands.empty()) {
        observeAndDecode(batches, jobs, hmm, G);
        for (size_t q = 0; q < G.cands.size(); q++) {
            const size_t T = G.len(q);
            if (T == 0) continue;                                         // :206-209
            const int max_state = splitVote(G.states.data() + G.seq_off[q], T);
            results[G.cands[q].job][G.cands[q].call] = std::make_tuple(G.ll[q], getSVTypeFromCNState(max_state), getGenotypeFromCNState(max_state), max_state);
        }
    }
    for (size_t j = 0; j < jobs.size(); j++) applySplitPredictions(*jobs[j].calls, results[j]);
}

// sv_caller.cpp:995-1063: what a split-read candidate takes from its region's prediction
void CNVCaller::applySplitPredictions(std::vector<SVCall> &split_sv_calls, const std::vector<std::tuple<double, SVType, Genotype, int>> &results)
{
    auto take_prediction = [](SVCall &c, double lh, Genotype g, int cn) {
        c.aln_type.set((size_t)SVDataType::HMM); c.hmm_likelihood = lh; c.genotype = g; c.cn_state = cn;
    };
    std::vector<SVCall> additional_calls;
    for (size_t k = 0; k < split_sv_calls.size(); k++) {
        SVCall &c = split_sv_calls[k];
        const double supp_lh = std::get<0>(results[k]);
        const SVType supp_type = std::get<1>(results[k]);
        const Genotype genotype = std::get<2>(results[k]);
        const int cn_state = std::get<3>(results[k]);
        if (supp_type == SVType::UNKNOWN) continue;
        const bool supp_cnv = supp_type == SVType::DEL || supp_type == SVType::DUP;
        if (c.sv_type == SVType::UNKNOWN && supp_cnv) {                                          // sv_caller.cpp:999-1005
            c.sv_type = supp_type; c.alt_allele = getSVTypeSymbol(supp_type); take_prediction(c, supp_lh, genotype, cn_state);
        } else if (c.sv_type != SVType::UNKNOWN && (supp_type == c.sv_type || supp_type == SVType::LOH || supp_type == SVType::NEUTRAL)) {
            take_prediction(c, supp_lh, genotype, cn_state);                                     // :1009-1013
        } else if (c.sv_type != SVType::UNKNOWN && supp_type != c.sv_type && supp_cnv) {
            if (c.sv_type == SVType::INV) {                                                      // :1020-1024
                take_prediction(c, supp_lh, genotype, cn_state);
            } else if (c.sv_type == SVType::INS && supp_type == SVType::DUP) {                   // :1026-1032
                c.sv_type = supp_type; c.alt_allele = getSVTypeSymbol(supp_type); take_prediction(c, supp_lh, genotype, cn_state);
            } else {                                                                             // :1033-1043
                SVCall extra = c;
                extra.sv_type = supp_type; extra.alt_allele = getSVTypeSymbol(supp_type); take_prediction(extra, supp_lh, genotype, cn_state);
                additional_calls.push_back(extra);
            }
        }
    }
    for (SVCall &extra : additional_calls) {                                                     // :1050-1063
        bool found = false;
        for (SVCall &existing : split_sv_calls) {
            if (existing.start == extra.start && existing.end == extra.end && existing.sv_type == extra.sv_type) { existing = extra; found = true; break; }
        }
        if (!found) addSVCall(split_sv_calls, extra);
    }
}
