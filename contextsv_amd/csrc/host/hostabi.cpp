// hostabi.cpp — extern "C" hooks over the C++ host mirror so the parity tests and bench.py (ctypes)
// can drive it. POD only; every function returns 0 or -1 (message via csvhost_last_error()).
#include <chrono>
#include <cstring>
#include <stdexcept>
#include <string>

#include "cnv_caller.h"
#include "dbscan.h"
#include "khmm.h"
#include "log.h"
#include "sv_caller.h"
#include "sort_select.h"
#include "split_caller.h"
#include "synth.h"
#include "fasta_query.h"
#include "vcf_writer.h"
#include "bam_io.h"
#include "snp_io.h"
#include "fast_inflate.h"
#include "par.h"
#include "umap_order.h"
#include <atomic>
#include <mutex>
#include <thread>
#include <fstream>
#include <memory>
#include <algorithm>
#include <vector>

namespace { std::string g_err; }
#define GUARD(...) try { __VA_ARGS__; return 0; } catch (const std::exception &e) { g_err = e.what(); return -1; }

extern "C" {

// POD view of an SVCall for tests (alt allele / flags travel separately)
struct csvhost_call {
    uint32_t start, end;
    int32_t  sv_type;
    int32_t  cluster_size;
    double   hmm_likelihood;
    int64_t  id;            // caller's tag, carried through merges in aln_offset-independent form
    uint32_t aln_flags;     // SVEvidenceFlags bits
    int32_t  genotype, cn_state, aln_offset;
};

static SVCall from_pod(const csvhost_call &c)
{
    SVCall s(c.start, c.end, (SVType)c.sv_type, std::to_string(c.id), SVEvidenceFlags(c.aln_flags), (Genotype)c.genotype,
             c.hmm_likelihood, c.cn_state, c.aln_offset, c.cluster_size);
    return s;   // the id rides in alt_allele so it survives every copy the merge makes
}
static csvhost_call to_pod(const SVCall &s)
{
    csvhost_call c;
    c.start = s.start; c.end = s.end; c.sv_type = (int32_t)s.sv_type; c.cluster_size = s.cluster_size; c.hmm_likelihood = s.hmm_likelihood;
    c.id = -1;
    try { c.id = std::stoll(s.alt_allele); } catch (...) {}
    c.aln_flags = (uint32_t)s.aln_type.to_ulong(); c.genotype = (int32_t)s.genotype; c.cn_state = s.cn_state; c.aln_offset = s.aln_offset;
    return c;
}

const char *csvhost_last_error(void) { return g_err.c_str(); }
void csvhost_set_context(csv_ctx *ctx) { csvhost::set_context(ctx); }
void csvhost_set_quiet(int q) { csvhost::set_quiet(q != 0); }

// mergeSVs through the GPU (DBSCAN::fit -> csvgpu_dbscan_iv). out capacity >= n. *n_out = merged count.
int csvhost_merge_svs(const csvhost_call *calls, uint64_t n, double eps, int32_t min_pts, int keep_noise, csvhost_call *out, uint64_t *n_out)
{
    GUARD({
        std::vector<SVCall> v; v.reserve(n);
        for (uint64_t i = 0; i < n; i++) v.push_back(from_pod(calls[i]));
        mergeSVs(v, eps, min_pts, keep_noise != 0);
        for (size_t i = 0; i < v.size(); i++) out[i] = to_pod(v[i]);
        *n_out = v.size();
    })
}

// the representative choice alone, labels supplied (CPU-testable: no device involved)
int csvhost_merge_type_with_labels(const csvhost_call *calls, const int32_t *labels, uint64_t n, int keep_noise, csvhost_call *out, uint64_t *n_out)
{
    GUARD({
        std::vector<SVCall> v, merged; v.reserve(n);
        for (uint64_t i = 0; i < n; i++) v.push_back(from_pod(calls[i]));
        mergeTypeWithLabels(v, labels, keep_noise != 0, merged);
        for (size_t i = 0; i < merged.size(); i++) out[i] = to_pod(merged[i]);
        *n_out = merged.size();
    })
}

int csvhost_merge_duplicates(csvhost_call *calls, uint64_t n, uint64_t *n_out)
{
    GUARD({
        std::vector<SVCall> v; v.reserve(n);
        for (uint64_t i = 0; i < n; i++) v.push_back(from_pod(calls[i]));
        mergeDuplicateSVs(v);
        for (size_t i = 0; i < v.size(); i++) calls[i] = to_pod(v[i]);
        *n_out = v.size();
    })
}

// addSVCall order check: insert the calls one by one, return the resulting order of ids
int csvhost_add_sv_calls(const csvhost_call *calls, uint64_t n, int64_t *ids_out, uint64_t *n_out)
{
    GUARD({
        std::vector<SVCall> v;
        for (uint64_t i = 0; i < n; i++) { SVCall c = from_pod(calls[i]); addSVCall(v, c); }
        for (size_t i = 0; i < v.size(); i++) ids_out[i] = std::stoll(v[i].alt_allele);
        *n_out = v.size();
    })
}

// test hook for sort_select.h: ids that std::sort and std_sort_select leave at slot nth when sorting `keys`
// descending by key only (ties make the difference between stable and unstable visible through the ids)
int csvhost_sort_select_check(const uint32_t *keys, uint64_t n, uint64_t nth, int64_t *id_std, int64_t *id_sel)
{
    GUARD({
        std::vector<uint32_t> a(n), b(n);
        for (uint64_t i = 0; i < n; i++) a[i] = b[i] = (uint32_t)i;
        auto cmp = [&](uint32_t x, uint32_t y) { return keys[x] > keys[y]; };
        std::sort(a.begin(), a.end(), cmp);
        *id_std = a[nth];
        *id_sel = *csvhost::std_sort_select(b.data(), b.data() + n, (std::ptrdiff_t)nth, cmp);       // (pointers: the block-wise partition for large ranges)
    })
}

// test hook: one partition step of sort_select.h on ids sorted descending by key — the library's loop and the block-wise form must leave
// the same cut and the same arrangement. *differ = 0 when they agree (ranges of more than 16, as in std::sort: the unguarded loops need the
// median-of-three's sentinels).
int csvhost_partition_check(const uint32_t *keys, uint64_t n, int *differ)
{
    GUARD({
        std::vector<uint32_t> a(n), b(n);
        for (uint64_t i = 0; i < n; i++) a[i] = b[i] = (uint32_t)i;
        auto cmp = [&](uint32_t x, uint32_t y) { return keys[x] > keys[y]; };
        *differ = 0;
        if (n > 16) {
            csvhost::lib_move_median_to_first(a.data(), a.data() + 1, a.data() + n / 2, a.data() + n - 1, cmp);
            csvhost::lib_move_median_to_first(b.data(), b.data() + 1, b.data() + n / 2, b.data() + n - 1, cmp);
            uint32_t *ca = csvhost::lib_unguarded_partition(a.data() + 1, a.data() + n, a.data(), cmp);
            uint32_t *cb = csvhost::lib_unguarded_partition_blocks(b.data() + 1, b.data() + n, b.data(), cmp);
            *differ = (ca - a.data() != cb - b.data()) || a != b;
        }
    })
}

// ---- synthetic shards -------------------------------------------------------------------------
struct csvhost_synth { SynthShard sh; };

csvhost_synth *csvhost_synth_generate2(uint64_t seed, uint32_t chr_len, double depth, int tech, int threads, int with_seq, double sv_per_bp);
csvhost_synth *csvhost_synth_generate(uint64_t seed, uint32_t chr_len, double depth, int tech, int threads, int with_seq)
{
    return csvhost_synth_generate2(seed, chr_len, depth, tech, threads, with_seq, 0.0);
}
// sv_per_bp <= 0: the default density (one truth SV per 120 kb)
csvhost_synth *csvhost_synth_generate2(uint64_t seed, uint32_t chr_len, double depth, int tech, int threads, int with_seq, double sv_per_bp)
{
    try {
        SynthParams p; p.seed = seed; p.chr_len = chr_len; p.depth = depth; p.tech = tech; p.threads = threads; p.with_seq = with_seq;
        if (sv_per_bp > 0) p.sv_per_bp = sv_per_bp;
        csvhost_synth *h = new csvhost_synth();
        synth_generate(p, h->sh);
        return h;
    } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
void csvhost_synth_view(const csvhost_synth *h, csv_reads *out, uint32_t *depth_len, const uint64_t **seq_off, const uint8_t **seq)
{
    *out = h->sh.view(); *depth_len = h->sh.depth_len;
    if (seq_off) *seq_off = h->sh.seq_off.empty() ? nullptr : h->sh.seq_off.data();
    if (seq) *seq = h->sh.seq.empty() ? nullptr : h->sh.seq.data();
}
const uint32_t *csvhost_synth_qname_ids(const csvhost_synth *h) { return h->sh.qname_id.data(); }
void csvhost_synth_free(csvhost_synth *h) { delete h; }

// ---- per-chromosome CIGAR path (SVCaller::processChromosome mirror) ---------------------------
struct csvhost_chr_stats {
    uint64_t n_signatures, n_del, n_ins, depth_sum;
    uint32_t depth_nonzero; int32_t min_pts;
    double mean_cov, ms_device, ms_host_merge;
    uint64_t n_calls;
};

// Runs the whole path on a resident shard and writes the merged calls (POD) + their ALT strings' first
// bytes are not returned: alt_tag[i] = 0 "<DEL>", 1 "<INS>", 2 literal sequence.
int csvhost_process_resident_chromosome(csv_ctx *ctx, csv_shard *shard, const uint64_t *seq_off, const uint8_t *seq, double eps,
                                        double min_pts_pct, csvhost_call *out, uint8_t *alt_tag, uint64_t cap, csvhost_chr_stats *st)
{
    GUARD({
        SVCaller caller(ctx);
        SeqStore ss; ss.seq_off = seq_off; ss.seq = seq;
        std::vector<SVCall> calls;
        ChrStats cs;
        caller.processResidentChromosome("chr", shard, seq ? &ss : nullptr, eps, min_pts_pct, calls, cs);
        st->n_signatures = cs.n_signatures; st->n_del = cs.n_del; st->n_ins = cs.n_ins; st->depth_sum = cs.depth_sum;
        st->depth_nonzero = cs.depth_nonzero; st->min_pts = cs.dbscan_min_pts; st->mean_cov = cs.mean_chr_cov;
        st->ms_device = cs.ms_device; st->ms_host_merge = cs.ms_host_merge; st->n_calls = calls.size();
        for (size_t i = 0; i < calls.size() && i < cap; i++) {
            const SVCall &c = calls[i];
            csvhost_call p;
            p.start = c.start; p.end = c.end; p.sv_type = (int32_t)c.sv_type; p.cluster_size = c.cluster_size; p.hmm_likelihood = c.hmm_likelihood;
            p.id = -1; p.aln_flags = (uint32_t)c.aln_type.to_ulong(); p.genotype = (int32_t)c.genotype; p.cn_state = c.cn_state; p.aln_offset = c.aln_offset;
            out[i] = p;
            if (alt_tag) alt_tag[i] = c.alt_allele == "<DEL>" ? 0 : (c.alt_allele == "<INS>" ? 1 : 2);
        }
    })
}

// ALT strings only: SVCaller::toSVCall for n signatures ('\n'-joined into buf, at most cap bytes; returns the full length through *len).
// Host only — the 50-bp insertion ALT is cut from the 4-bit sequences here (sv_caller.cpp:572-590, :607-625 of the reference).
int csvhost_sig_alts(const csv_sig *sig, uint64_t n, const uint64_t *seq_off, const uint8_t *seq, char *buf, uint64_t cap, uint64_t *len)
{
    GUARD({
        SeqStore ss; ss.seq_off = seq_off; ss.seq = seq;
        std::string t;
        for (uint64_t i = 0; i < n; i++) { t += SVCaller::toSVCall(sig[i], seq ? &ss : nullptr).alt_allele; t += '\n'; }
        if (buf && cap) memcpy(buf, t.data(), (size_t)std::min<uint64_t>(cap, t.size()));
        *len = t.size();
    })
}

// processChromosome on a resident shard, returning "<start>\t<end>\t<ALT>\n" per merged call (the string fields the POD view drops)
int csvhost_process_resident_chromosome_alts(csv_ctx *ctx, csv_shard *shard, const uint64_t *seq_off, const uint8_t *seq, double eps, double min_pts_pct,
                                             char *buf, uint64_t cap, uint64_t *len)
{
    GUARD({
        SVCaller caller(ctx);
        SeqStore ss; ss.seq_off = seq_off; ss.seq = seq;
        std::vector<SVCall> calls;
        ChrStats cs;
        caller.processResidentChromosome("chr", shard, seq ? &ss : nullptr, eps, min_pts_pct, calls, cs);
        std::string t;
        for (const SVCall &c : calls) t += std::to_string(c.start) + "\t" + std::to_string(c.end) + "\t" + c.alt_allele + "\n";
        if (buf && cap) memcpy(buf, t.data(), (size_t)std::min<uint64_t>(cap, t.size()));
        *len = t.size();
    })
}

// The host half alone (no device): ordered signatures + labels -> merged calls; `reps` timed repetitions, *ms = mean time of one.
int csvhost_merge_ordered(const csv_sig *sig, const int32_t *labels, uint64_t n_del, uint64_t n_ins, int reps, csvhost_call *out, uint64_t cap,
                          uint64_t *n_out, double *ms)
{
    GUARD({
        std::vector<SVCall> calls;
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < std::max(reps, 1); r++) SVCaller::mergeOrdered(sig, labels, n_del, n_ins, nullptr, calls);
        if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / std::max(reps, 1);
        for (size_t i = 0; i < calls.size() && i < cap; i++) out[i] = to_pod(calls[i]);
        *n_out = calls.size();
    })
}

// n_steps passes over the same resident shard, software-pipelined (device chain of step i+1 overlaps the host merge of
// step i). Returns the merged calls of the LAST step and its stats; ms_total = wall time of all steps.
int csvhost_process_resident_pipelined(csv_ctx *ctx, csv_shard *shard, uint64_t n_steps, const uint64_t *seq_off, const uint8_t *seq,
                                       double eps, double min_pts_pct, csvhost_call *out, uint8_t *alt_tag, uint64_t cap,
                                       csvhost_chr_stats *st, double *ms_total, uint64_t *total_calls)
{
    GUARD({
        SVCaller caller(ctx);
        SeqStore ss; ss.seq_off = seq_off; ss.seq = seq;
        std::vector<csv_shard *> shards(n_steps, shard);
        std::vector<std::vector<SVCall>> calls;
        std::vector<ChrStats> stats;
        const auto t0 = std::chrono::steady_clock::now();
        caller.processResidentChromosomesPipelined(shards, seq ? &ss : nullptr, eps, min_pts_pct, calls, stats);
        *ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        uint64_t tot = 0;
        for (auto &c : calls) tot += c.size();
        *total_calls = tot;
        if (n_steps) {
            const ChrStats &cs = stats.back();
            const std::vector<SVCall> &last = calls.back();
            st->n_signatures = cs.n_signatures; st->n_del = cs.n_del; st->n_ins = cs.n_ins; st->depth_sum = cs.depth_sum;
            st->depth_nonzero = cs.depth_nonzero; st->min_pts = cs.dbscan_min_pts; st->mean_cov = cs.mean_chr_cov;
            double dev = 0, hm = 0;
            for (auto &x : stats) { dev += x.ms_device; hm += x.ms_host_merge; }
            st->ms_device = dev / n_steps; st->ms_host_merge = hm / n_steps; st->n_calls = last.size();
            for (size_t i = 0; i < last.size() && i < cap; i++) {
                const SVCall &c = last[i];
                csvhost_call p;
                p.start = c.start; p.end = c.end; p.sv_type = (int32_t)c.sv_type; p.cluster_size = c.cluster_size; p.hmm_likelihood = c.hmm_likelihood;
                p.id = -1; p.aln_flags = (uint32_t)c.aln_type.to_ulong(); p.genotype = (int32_t)c.genotype; p.cn_state = c.cn_state; p.aln_offset = c.aln_offset;
                out[i] = p;
                if (alt_tag) alt_tag[i] = c.alt_allele == "<DEL>" ? 0 : (c.alt_allele == "<INS>" ? 1 : 2);
            }
        }
    })
}

// n_lanes contexts of one GPU, lane l runs steps[l] passes over its own resident shard concurrently with the others.
// Returns lane 0's last merged calls + stats averaged over every step of every lane; ms_total = wall time of the whole job.
int csvhost_process_resident_lanes(int n_lanes, csv_ctx *const *ctxs, csv_shard *const *shards, const uint64_t *steps, double eps, double min_pts_pct,
                                   csvhost_call *out, uint64_t cap, csvhost_chr_stats *st, double *ms_total, uint64_t *total_calls)
{
    GUARD({
        std::vector<SVCaller::Lane> lanes((size_t)n_lanes);
        for (int l = 0; l < n_lanes; l++) { lanes[l].ctx = ctxs[l]; lanes[l].shards.assign(steps[l], shards[l]); }
        std::vector<std::vector<std::vector<SVCall>>> calls;
        std::vector<std::vector<ChrStats>> stats;
        const auto t0 = std::chrono::steady_clock::now();
        SVCaller::processResidentLanes(lanes, nullptr, eps, min_pts_pct, calls, stats);
        *ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        uint64_t tot = 0, n_steps = 0;
        double dev = 0, hm = 0;
        for (auto &lane : calls) for (auto &c : lane) tot += c.size();
        for (auto &lane : stats) for (auto &x : lane) { dev += x.ms_device; hm += x.ms_host_merge; n_steps++; }
        *total_calls = tot;
        if (n_steps && !stats[0].empty()) {
            const ChrStats &cs = stats[0].back();
            const std::vector<SVCall> &last = calls[0].back();
            st->n_signatures = cs.n_signatures; st->n_del = cs.n_del; st->n_ins = cs.n_ins; st->depth_sum = cs.depth_sum;
            st->depth_nonzero = cs.depth_nonzero; st->min_pts = cs.dbscan_min_pts; st->mean_cov = cs.mean_chr_cov;
            st->ms_device = dev / n_steps; st->ms_host_merge = hm / n_steps; st->n_calls = last.size();
            for (size_t i = 0; i < last.size() && i < cap; i++) out[i] = to_pod(last[i]);
        }
    })
}

// ---- split-read signatures (findSplitSVSignatures mirror) ---------------------------------------
struct csvhost_split_call { uint32_t start, end; int32_t sv_type, cluster_size, aln_offset; uint32_t aln_flags; int32_t tid; };

// records in file order; qname = "r<qname_id>". Output sorted by contig id, each contig's calls in the reference's order.
int csvhost_split_signatures(csv_ctx *ctx, uint64_t n, const int32_t *tid, const int32_t *pos, const uint16_t *flag, const uint8_t *mapq,
                             const int32_t *ref_end, const int32_t *q_start, const int32_t *q_end, const uint32_t *qname_id, int n_targets,
                             int min_mapq, csvhost_split_call *out, uint64_t cap, uint64_t *n_out)
{
    GUARD({
        csvhost::set_context(ctx);
        std::vector<SplitRecord> rec(n);
        std::vector<std::string> qn(n), targets((size_t)n_targets);
        for (int t = 0; t < n_targets; t++) targets[(size_t)t] = "contig" + std::to_string(t);
        for (uint64_t i = 0; i < n; i++) {
            rec[i] = SplitRecord{tid[i], pos[i], flag[i], mapq[i], ref_end[i], q_start[i], q_end[i]};
            qn[i] = "r" + std::to_string(qname_id[i]);
        }
        SplitParams p; p.min_mapq = min_mapq;
        std::unordered_map<std::string, std::vector<SVCall>> calls;
        findSplitSVSignatures(rec, qn, targets, p, calls);
        uint64_t k = 0;
        for (int t = 0; t < n_targets; t++) {
            auto it = calls.find(targets[(size_t)t]);
            if (it == calls.end()) continue;
            for (const SVCall &c : it->second) {
                if (k < cap) out[k] = csvhost_split_call{c.start, c.end, (int32_t)c.sv_type, c.cluster_size, c.aln_offset, (uint32_t)c.aln_type.to_ulong(), t};
                k++;
            }
        }
        *n_out = k;
    })
}

// std::hash<std::string> of n '\n'-joined names (what the staging code attaches to a shard as its query-name column)
int csvhost_string_hashes(const char *names, uint64_t n, uint64_t *out)
{
    GUARD({
        const char *p = names;
        for (uint64_t i = 0; i < n; i++) {
            const char *e = strchr(p, '\n');
            const size_t len = e ? (size_t)(e - p) : strlen(p);
            out[i] = csvhost::std_string_hash(p, len);
            p = e ? e + 1 : p + len;
        }
    })
}

// Test hook for par.h (CPU): parallel_for visits every index exactly once whatever the thread count, runs nested sections inline, hands
// the first exception to the caller and stays usable; WorkerThreads runs tasks side by side and reuses its threads. 0 = all held.
// SVCaller::assignShards: rank_of[i] for every shard weight
int csvhost_assign_shards(const double *weights, uint64_t n, int world, int32_t *rank_of)
{
    GUARD({
        const std::vector<int> r = SVCaller::assignShards(std::vector<double>(weights, weights + n), world);
        for (uint64_t i = 0; i < n; i++) rank_of[i] = r[i];
    })
}

int csvhost_par_selftest(int n_items, int threads)
{
    GUARD({
        std::vector<std::atomic<int>> hit((size_t)n_items);
        for (auto &h : hit) h = 0;
        csvhost::parallel_for((size_t)n_items, threads, [&](size_t i) {
            hit[i]++;
            csvhost::parallel_for(3, threads, [&](size_t) {});                     // nested: inline
        });
        for (auto &h : hit) if (h != 1) throw std::runtime_error("parallel_for: an index was not visited exactly once");
        bool thrown = false;
        try { csvhost::parallel_for((size_t)n_items, threads, [&](size_t i) { if (i == (size_t)n_items / 2) throw std::runtime_error("x"); }); }
        catch (const std::runtime_error &) { thrown = true; }
        if (!thrown && n_items > 0) throw std::runtime_error("parallel_for: exception lost");
        std::atomic<long> sum{0};
        csvhost::parallel_for((size_t)n_items, threads, [&](size_t i) { sum += (long)i; });
        if (sum != (long)n_items * (n_items - 1) / 2) throw std::runtime_error("parallel_for: wrong sum after an exception");
        csvhost::WorkerThreads &w = csvhost::WorkerThreads::instance();
        for (int round = 0; round < 3; round++) {
            std::atomic<int> inside{0}, peak{0};
            std::vector<csvhost::WorkerThreads::Ticket> t;
            for (int k = 0; k < 4; k++)
                t.push_back(w.start([&] {
                    const int now = ++inside;
                    int p = peak.load();
                    while (now > p && !peak.compare_exchange_weak(p, now)) {}
                    const auto until = std::chrono::steady_clock::now() + std::chrono::milliseconds(20);
                    while (std::chrono::steady_clock::now() < until && peak.load() < 4) std::this_thread::yield();
                    --inside;
                }));
            for (auto x : t) w.wait(x);
            if (peak != 4) throw std::runtime_error("WorkerThreads: tasks did not run side by side");
        }
        // several threads in parallel sections at once, two on each pool (the one that finds a pool taken runs its items itself and
        // leaves the owner's flag alone)
        std::atomic<long> bad{0};
        std::vector<csvhost::WorkerThreads::Ticket> t;
        for (int k = 0; k < 4; k++)
            t.push_back(w.start([&, k] {
                if (k & 1) csvhost::HostPool::second_pool_flag() = true;
                for (int rep = 0; rep < 50; rep++) {
                    const size_t n = (size_t)n_items / 4 + (size_t)rep + 1;
                    std::vector<std::atomic<int>> seen(n);
                    for (auto &h : seen) h = 0;
                    csvhost::parallel_for(n, threads, [&](size_t i) { seen[i]++; });
                    for (auto &h : seen) if (h != 1) bad++;
                }
                csvhost::HostPool::second_pool_flag() = false;
            }));
        for (auto x : t) w.wait(x);
        if (bad) throw std::runtime_error("parallel_for: concurrent sections lost or repeated an index");
    })
}

// Test hook for umap_order.h (CPU): keys = n names ('\n'-joined) inserted in order with operator[], then every key with
// erase_mask[i] != 0 is erased. order_real = for every surviving node of a real std::unordered_map<std::string,int>, the index of
// the key's first insertion, in iteration order; order_emu = the same from csvhost::UMapOrder. Returns the count through *n_out
// (both have the same length when the emulation is right; the caller compares).
int csvhost_umap_order_check(const char *names, uint64_t n, const uint8_t *erase_mask, int64_t *order_real, int64_t *order_emu, uint64_t *n_real, uint64_t *n_emu,
                             uint64_t *buckets_real, uint64_t *buckets_emu)
{
    GUARD({
        std::vector<std::string> keys;
        keys.reserve(n);
        const char *p = names;
        for (uint64_t i = 0; i < n; i++) {
            const char *e = strchr(p, '\n');
            keys.emplace_back(p, e ? (size_t)(e - p) : strlen(p));
            p = e ? e + 1 : p + strlen(p);
        }
        std::unordered_map<std::string, int> real;
        csvhost::UMapOrder emu;
        std::vector<int64_t> first;
        for (uint64_t i = 0; i < n; i++) {
            real.emplace(keys[i], (int)i);                 // keeps the first index, as the emulation's `first` does
            real[keys[i]];                                 // (operator[] on an existing key: no structural change)
            const uint64_t h = csvhost::std_string_hash(keys[i].data(), keys[i].size());
            if (emu.find(h, [&](uint32_t nd) { return keys[(size_t)first[nd]] == keys[i]; }) < 0) { emu.insert_new(h); first.push_back((int64_t)i); }
        }
        *buckets_real = real.bucket_count(); *buckets_emu = emu.bucket_count();
        std::vector<char> dead(n, 0);
        for (uint64_t i = 0; i < n; i++) if (erase_mask && erase_mask[i]) { real.erase(keys[i]); }
        uint64_t a = 0, b = 0;
        for (const auto &kv : real) order_real[a++] = kv.second;
        emu.for_each([&](uint32_t nd) { if (real.count(keys[(size_t)first[nd]])) order_emu[b++] = first[nd]; });
        *n_real = a; *n_emu = b;
    })
}

// ---- copy-number pass (CNVCaller mirror) -------------------------------------------------------
static CHMM chmm_from_pod(const csv_hmm *h)
{
    CHMM c; c.N = 6; c.M = 6;
    c.A.assign(6, std::vector<double>(6)); c.B.assign(6, std::vector<double>(6, 0.0));
    c.pi.resize(6); c.B1_mean.resize(6); c.B1_sd.resize(6); c.B2_mean.resize(5); c.B2_sd.resize(5);
    for (int i = 0; i < 6; i++) { for (int j = 0; j < 6; j++) c.A[i][j] = h->A[i * 6 + j]; c.pi[i] = h->pi[i]; c.B1_mean[i] = h->B1_mean[i]; c.B1_sd[i] = h->B1_sd[i]; }
    for (int i = 0; i < 5; i++) { c.B2_mean[i] = h->B2_mean[i]; c.B2_sd[i] = h->B2_sd[i]; }
    c.B1_uf = h->B1_uf; c.B2_uf = h->B2_uf;
    return c;
}
static SNPTable snp_table(const uint32_t *pos, const double *baf, const double *pfb, const uint8_t *has_pfb, uint64_t n)
{
    SNPTable t;
    t.pos.assign(pos, pos + n); t.baf.assign(baf, baf + n); t.pfb.assign(pfb, pfb + n); t.has_pfb.assign(has_pfb, has_pfb + n);
    return t;
}

// querySNPRegion for one region: observation arrays in the reference's order. cap = capacity of the outputs.
int csvhost_query_snp_region(csv_ctx *ctx, csv_shard *shard, uint32_t start, uint32_t end, double mean_cov, int sample_size,
                             const uint32_t *snp_pos, const double *snp_baf, const double *snp_pfb, const uint8_t *snp_has_pfb, uint64_t n_snp,
                             uint32_t *pos_out, double *baf_out, double *pfb_out, double *log2_out, uint8_t *is_snp_out, uint64_t cap, uint64_t *n_out)
{
    GUARD({
        CNVCaller cnv(ctx); cnv.sample_size = sample_size;
        SNPTable t = snp_table(snp_pos, snp_baf, snp_pfb, snp_has_pfb, n_snp);
        std::vector<SNPData> d;
        cnv.querySNPRegions({{start, end}}, shard, mean_cov, t, d);
        *n_out = d[0].pos.size();
        for (size_t i = 0; i < d[0].pos.size() && i < cap; i++) {
            pos_out[i] = d[0].pos[i]; baf_out[i] = d[0].baf[i]; pfb_out[i] = d[0].pfb[i]; log2_out[i] = d[0].log2_cov[i]; is_snp_out[i] = d[0].is_snp[i];
        }
    })
}

// runCIGARCopyNumberPrediction (split == 0, in place) or runSplitReadCopyNumberPredictions (split == 1, may grow the list)
int csvhost_cn_prediction(csv_ctx *ctx, csv_shard *shard, int split, csvhost_call *calls, uint64_t n, uint64_t cap, uint64_t *n_out,
                          const csv_hmm *hmm, double mean_cov, int sample_size, uint32_t min_cnv,
                          const uint32_t *snp_pos, const double *snp_baf, const double *snp_pfb, const uint8_t *snp_has_pfb, uint64_t n_snp)
{
    GUARD({
        csvhost::set_context(ctx);
        CNVCaller cnv(ctx); cnv.sample_size = sample_size; cnv.min_cnv_length = min_cnv;
        SNPTable t = snp_table(snp_pos, snp_baf, snp_pfb, snp_has_pfb, n_snp);
        CHMM h = chmm_from_pod(hmm);
        std::vector<SVCall> v; v.reserve(n);
        for (uint64_t i = 0; i < n; i++) v.push_back(from_pod(calls[i]));
        if (split) cnv.runSplitReadCopyNumberPredictions("chr", v, h, mean_cov, shard, t);
        else cnv.runCIGARCopyNumberPrediction("chr", v, h, mean_cov, shard, t);
        *n_out = v.size();
        for (size_t i = 0; i < v.size() && i < cap; i++) {
            // alt_allele carries the id for untouched calls; calls whose ALT was rewritten report id -1 (and a symbol ALT)
            calls[i] = to_pod(v[i]);
        }
    })
}

struct csvhost_fasta;
static const ReferenceGenome *fasta_genome(const csvhost_fasta *h);

// ---- whole run (SVCaller::run mirror) over concatenated contig arrays -----------------------------
// reads of contig t = [read_off[t], read_off[t+1]); cigar_off is global over the concatenated cigar array; SNPs of contig t =
// [snp_off[t], snp_off[t+1]); qname of read i = "r<qname_id[i]>". Output: merged calls with contig id, contigs ascending.
int csvhost_run(csv_ctx *ctx, int n_contigs, const uint64_t *read_off, const uint32_t *depth_len, const int32_t *pos, const uint16_t *flag,
                const uint8_t *mapq, const uint64_t *cigar_off, const uint32_t *cigar, const uint32_t *qname_id,
                const uint64_t *snp_off, const uint32_t *snp_pos, const double *snp_baf, const double *snp_pfb, const uint8_t *snp_has,
                const csv_hmm *hmm, double eps, double min_pts_pct, int sample_size, uint32_t min_cnv,
                csvhost_call *out, int32_t *out_tid, uint64_t cap, uint64_t *n_out,
                const csvhost_fasta *fasta, const char *vcf_dir, const char *gap_path, const char *file_date,
                char *alt_buf, uint64_t alt_cap, uint64_t *alt_off, int save_cnv)
{
    GUARD({
        std::vector<ChromosomeInput> contigs((size_t)n_contigs);
        std::vector<std::vector<uint64_t>> local_off((size_t)n_contigs);
        std::vector<std::vector<std::string>> qn((size_t)n_contigs);
        std::vector<SNPTable> tabs((size_t)n_contigs);
        for (int t = 0; t < n_contigs; t++) {
            const uint64_t r0 = read_off[t], r1 = read_off[t + 1], n = r1 - r0;
            local_off[t].resize(n + 1);
            for (uint64_t i = 0; i <= n; i++) local_off[t][i] = cigar_off[r0 + i] - cigar_off[r0];
            qn[t].resize(n);
            for (uint64_t i = 0; i < n; i++) qn[t][i] = "r" + std::to_string(qname_id[r0 + i]);
            tabs[t] = snp_table(snp_pos + snp_off[t], snp_baf + snp_off[t], snp_pfb + snp_off[t], snp_has + snp_off[t], snp_off[t + 1] - snp_off[t]);
            ChromosomeInput &c = contigs[t];
            c.name = "contig" + std::to_string(t);
            c.reads.n_reads = n; c.reads.n_cigar = local_off[t][n];
            c.reads.pos = pos + r0; c.reads.flag = flag + r0; c.reads.mapq = mapq + r0; c.reads.tid = nullptr;
            c.reads.cigar_off = local_off[t].data(); c.reads.cigar = cigar + cigar_off[r0];
            c.depth_len = depth_len[t]; c.qnames = &qn[t]; c.snps = &tabs[t];
        }
        RunParams P; P.dbscan_epsilon = eps; P.dbscan_min_pts_pct = min_pts_pct; P.sample_size = sample_size; P.min_cnv_length = min_cnv;
        if (fasta && vcf_dir) {
            P.ref_genome = fasta_genome(fasta);
            P.vcf.output_dir = vcf_dir; P.vcf.assembly_gaps = gap_path ? gap_path : ""; P.vcf.file_date = file_date ? file_date : "";
        }
        P.save_cnv = save_cnv != 0;
        SVCaller caller(ctx);
        std::unordered_map<std::string, std::vector<SVCall>> calls;
        caller.run(contigs, chmm_from_pod(hmm), P, calls);
        uint64_t k = 0, alt_used = 0;
        if (alt_off) alt_off[0] = 0;
        for (int t = 0; t < n_contigs; t++) {
            for (const SVCall &c : calls[contigs[t].name]) {
                if (k < cap) {
                    csvhost_call p;
                    p.start = c.start; p.end = c.end; p.sv_type = (int32_t)c.sv_type; p.cluster_size = c.cluster_size; p.hmm_likelihood = c.hmm_likelihood;
                    p.id = -1; p.aln_flags = (uint32_t)c.aln_type.to_ulong(); p.genotype = (int32_t)c.genotype; p.cn_state = c.cn_state; p.aln_offset = c.aln_offset;
                    out[k] = p; out_tid[k] = t;
                    if (alt_off) {                     // ALT strings, concatenated; truncated once alt_cap is reached
                        const uint64_t len = std::min<uint64_t>(c.alt_allele.size(), alt_cap - alt_used);
                        if (alt_buf && len) memcpy(alt_buf + alt_used, c.alt_allele.data(), len);
                        alt_used += len;
                        alt_off[k + 1] = alt_used;
                    }
                }
                k++;
            }
        }
        *n_out = k;
    })
}

// ---- a staged genome: contigs resident in HBM + the host arrays of the host-side passes (SVCaller::runResident) -----------
struct csvhost_genome {
    struct Contig {
        std::string name;
        int32_t global_tid = 0;
        csv_ctx *ctx = nullptr;                 // the context the shard was uploaded with (frees it)
        csv_shard *shard = nullptr;
        uint32_t depth_len = 0;
        uint64_t n_reads = 0, n_cigar = 0;
        std::vector<int32_t> pos; std::vector<uint16_t> flag; std::vector<uint8_t> mapq;
        std::vector<uint64_t> qhash, name_id;
        bool unique_names = false;
        SNPTable snps;
    };
    std::vector<std::unique_ptr<Contig>> contigs;
    ~csvhost_genome() { for (auto &c : contigs) if (c->shard) csvgpu_shard_free(c->ctx, c->shard); }
};

csvhost_genome *csvhost_genome_create(void) { return new csvhost_genome(); }
void csvhost_genome_free(csvhost_genome *g) { delete g; }

// Stage one contig: upload the records (the shard stays resident), keep pos / flag / mapq and the query-name hash + identity of
// every record on the host. Record i's query name is "r<global_tid>_<qname_id[i]>" (what the synthetic BAM writer calls it), so a
// contig hashes the same whichever rank or lane it lands on (name_style 0: "r<id>", the naming of csvhost_run). SNPs: snp_pos / snp_baf (n_snp, sorted; no population frequency).
int csvhost_genome_add(csvhost_genome *g, csv_ctx *ctx, const char *name, int32_t global_tid, const csv_reads *reads, uint32_t depth_len,
                       const uint32_t *qname_id, int name_style /* 0: "r<id>" (names shared across contigs), 1: "r<tid>_<id>" */,
                       const uint32_t *snp_pos, const double *snp_baf, const double *snp_pfb /* nullable: 0.0 */, const uint8_t *snp_has_pfb /* nullable: none */,
                       uint64_t n_snp)
{
    GUARD({
        std::unique_ptr<csvhost_genome::Contig> c(new csvhost_genome::Contig());
        c->name = name; c->global_tid = global_tid; c->ctx = ctx; c->depth_len = depth_len;
        c->n_reads = reads->n_reads; c->n_cigar = reads->n_cigar;
        c->shard = csvgpu_shard_upload(ctx, reads, depth_len);
        if (!c->shard) throw std::runtime_error(std::string("genome_add: ") + csvgpu_last_error(ctx));
        const uint64_t n = reads->n_reads;
        c->pos.assign(reads->pos, reads->pos + n); c->flag.assign(reads->flag, reads->flag + n); c->mapq.assign(reads->mapq, reads->mapq + n);
        if (qname_id) {
            c->qhash.resize(n); c->name_id.resize(n);
            char buf[48];
            const int pre = name_style ? snprintf(buf, sizeof buf, "r%d_", (int)global_tid) : snprintf(buf, sizeof buf, "r");
            for (uint64_t i = 0; i < n; i++) {
                const int len = pre + snprintf(buf + pre, sizeof buf - (size_t)pre, "%u", qname_id[i]);
                c->qhash[i] = csvhost::std_string_hash(buf, (size_t)len);
                c->name_id[i] = (name_style ? ((uint64_t)(uint32_t)global_tid << 32) : 0) | qname_id[i];
            }
            // the query-name column goes to HBM with the shard; and, once, is any name hash shared by two non-supplementary records?
            // (then the reference's map holds ONE node for them and the contig's order is replayed on the host instead of the device)
            if (csvgpu_shard_set_qname_hash(ctx, c->shard, c->qhash.data()) != CSV_OK) throw std::runtime_error(std::string("genome_add: ") + csvgpu_last_error(ctx));
            std::vector<uint64_t> h;
            h.reserve(n);
            for (uint64_t i = 0; i < n; i++) if (!(reads->flag[i] & 0x800)) h.push_back(c->qhash[i]);
            std::sort(h.begin(), h.end());
            c->unique_names = std::adjacent_find(h.begin(), h.end()) == h.end();
        }
        if (n_snp) {
            c->snps.pos.assign(snp_pos, snp_pos + n_snp); c->snps.baf.assign(snp_baf, snp_baf + n_snp);
            if (snp_pfb) c->snps.pfb.assign(snp_pfb, snp_pfb + n_snp); else c->snps.pfb.assign(n_snp, 0.0);
            if (snp_has_pfb) c->snps.has_pfb.assign(snp_has_pfb, snp_has_pfb + n_snp); else c->snps.has_pfb.assign(n_snp, 0);
        }
        g->contigs.push_back(std::move(c));
    })
}

// the same from a generated shard (+ generated SNPs when with_snps)
int csvhost_genome_add_synth(csvhost_genome *g, csv_ctx *ctx, const char *name, int32_t global_tid, const csvhost_synth *syn, uint64_t snp_seed, int with_snps)
{
    const SynthShard &sh = syn->sh;
    std::vector<uint32_t> sp; std::vector<double> sb;
    if (with_snps) synth_snps(snp_seed, sh.depth_len - 1, sp, sb);
    const csv_reads r = sh.view();
    return csvhost_genome_add(g, ctx, name, global_tid, &r, sh.depth_len, sh.qname_id.data(), 1, sp.data(), sb.data(), nullptr, nullptr, sp.size());
}

struct csvhost_stage_times {
    double ms_cigar, ms_cigar_cn, ms_split_fetch, ms_split, ms_split_cn, ms_merge_split, ms_merge_final, ms_vcf, ms_total, ms_split_prepare;
    uint64_t n_reads, n_signatures, n_cigar_calls, n_cigar_cn_regions, n_split_calls, n_final_calls;
};

uint64_t csvhost_genome_n_contigs(const csvhost_genome *g) { return g->contigs.size(); }
void csvhost_genome_contig_info(const csvhost_genome *g, uint64_t i, uint64_t *n_reads, uint64_t *n_cigar, uint32_t *depth_len, int32_t *global_tid, csv_shard **shard)
{
    const auto &c = *g->contigs[i];
    if (n_reads) *n_reads = c.n_reads;
    if (n_cigar) *n_cigar = c.n_cigar;
    if (depth_len) *depth_len = c.depth_len;
    if (global_tid) *global_tid = c.global_tid;
    if (shard) *shard = c.shard;
}

// One step: SVCaller::runResident over every staged contig. passes: bit 0 split-read pass, bit 1 CIGAR copy-number pass, bit 2 the two
// final merges, bit 3 keep the qname map's order on the host (umap_order.h) instead of csvgpu_split_order,
// bit 4 do not run the split pass's first half beside the CIGAR pass. Calls come back grouped by contig in staging order with the contig's GLOBAL tid in out_tid; stats[i] per contig.
int csvhost_genome_run(csvhost_genome *g, csv_ctx *ctx, int n_lanes, csv_ctx *const *lane_ctxs, const csv_hmm *hmm, double eps, double min_pts_pct,
                       int sample_size, uint32_t min_cnv, int passes, int host_threads, csvhost_call *out, int32_t *out_tid, uint64_t cap, uint64_t *n_out,
                       csvhost_stage_times *times, csvhost_chr_stats *stats)
{
    GUARD({
        std::vector<ResidentContig> rc(g->contigs.size());
        for (size_t i = 0; i < rc.size(); i++) {
            auto &c = *g->contigs[i];
            rc[i].name = c.name; rc[i].shard = c.shard; rc[i].depth_len = c.depth_len; rc[i].snps = &c.snps;
            rc[i].split.n = c.n_reads; rc[i].split.pos = c.pos.data(); rc[i].split.flag = c.flag.data(); rc[i].split.mapq = c.mapq.data();
            if (!c.qhash.empty()) { rc[i].split.qhash = c.qhash.data(); rc[i].split.name_id = c.name_id.data(); rc[i].split.unique_names = c.unique_names; }
        }
        RunParams P; P.dbscan_epsilon = eps; P.dbscan_min_pts_pct = min_pts_pct; P.sample_size = sample_size; P.min_cnv_length = min_cnv;
        P.split_svs = (passes & 1) != 0; P.cigar_cn = (passes & 2) != 0; P.merge_split_svs = P.merge_final_svs = (passes & 4) != 0;
        P.host_threads = host_threads;
        P.split_order_on_device = (passes & 8) == 0;
        P.overlap_split_prepare = (passes & 16) == 0;
        std::vector<csv_ctx *> lanes(lane_ctxs, lane_ctxs + (n_lanes > 0 ? n_lanes : 0));
        SVCaller caller(ctx);
        // the call map of a genome is 3e4 records with strings in them: it is torn down beside whatever the caller does next — the next run,
        // in a loop of runs: the previous run's teardown is waited for when this run hands over its own map, not before it starts
        static std::mutex teardown_mu;                                          // (runs of different genomes may come from different threads)
        static csvhost::WorkerThreads::Ticket teardown = nullptr;
        auto swap_teardown = [](csvhost::WorkerThreads::Ticket next) {
            csvhost::WorkerThreads::Ticket prev;
            { std::lock_guard<std::mutex> l(teardown_mu); prev = teardown; teardown = next; }
            if (prev) csvhost::WorkerThreads::instance().wait(prev);
        };
        auto calls_p = std::make_shared<std::unordered_map<std::string, std::vector<SVCall>>>();
        std::unordered_map<std::string, std::vector<SVCall>> &calls = *calls_p;
        struct Hand {                                                           // (also when runResident throws)
            std::shared_ptr<std::unordered_map<std::string, std::vector<SVCall>>> &p;
            decltype(swap_teardown) &swap;
            ~Hand() { auto dead = std::move(p); swap(csvhost::WorkerThreads::instance().start([dead]() mutable { dead.reset(); })); }
        } hand{calls_p, swap_teardown};
        std::vector<ChrStats> cs;
        RunStageTimes T;
        {
            csvhost::TraceScope tr_run("genome_run: runResident (with its locals' teardown)");
            caller.runResident(rc, lanes, chmm_from_pod(hmm), P, calls, &cs, &T);
        }
        csvhost::TraceScope tr_flat("genome_run: calls -> flat records");
        uint64_t k = 0;
        for (size_t i = 0; i < rc.size(); i++) {
            const std::vector<SVCall> &v = calls[rc[i].name];
            for (const SVCall &c : v) {
                if (k < cap) {
                    csvhost_call p;
                    p.start = c.start; p.end = c.end; p.sv_type = (int32_t)c.sv_type; p.cluster_size = c.cluster_size; p.hmm_likelihood = c.hmm_likelihood;
                    p.id = -1; p.aln_flags = (uint32_t)c.aln_type.to_ulong(); p.genotype = (int32_t)c.genotype; p.cn_state = c.cn_state; p.aln_offset = c.aln_offset;
                    out[k] = p; out_tid[k] = g->contigs[i]->global_tid;
                }
                k++;
            }
            if (stats) {
                csvhost_chr_stats &st = stats[i];
                st.n_signatures = cs[i].n_signatures; st.n_del = cs[i].n_del; st.n_ins = cs[i].n_ins; st.depth_sum = cs[i].depth_sum;
                st.depth_nonzero = cs[i].depth_nonzero; st.min_pts = cs[i].dbscan_min_pts; st.mean_cov = cs[i].mean_chr_cov;
                st.ms_device = cs[i].ms_device; st.ms_host_merge = cs[i].ms_host_merge; st.n_calls = v.size();
            }
        }
        *n_out = k;
        if (times) {
            times->ms_cigar = T.ms_cigar; times->ms_cigar_cn = T.ms_cigar_cn; times->ms_split_fetch = T.ms_split_fetch; times->ms_split = T.ms_split;
            times->ms_split_cn = T.ms_split_cn; times->ms_merge_split = T.ms_merge_split; times->ms_merge_final = T.ms_merge_final; times->ms_vcf = T.ms_vcf;
            times->ms_total = T.ms_total; times->ms_split_prepare = T.ms_split_prepare; times->n_reads = T.n_reads; times->n_signatures = T.n_signatures; times->n_cigar_calls = T.n_cigar_calls;
            times->n_cigar_cn_regions = T.n_cigar_cn_regions; times->n_split_calls = T.n_split_calls; times->n_final_calls = T.n_final_calls;
        }
    })
}

// ---- reference genome + VCF writer (fasta_query.cpp, saveToVCF) ---------------------------------------
struct csvhost_fasta { ReferenceGenome g; };
static const ReferenceGenome *fasta_genome(const csvhost_fasta *h) { return &h->g; }

csvhost_fasta *csvhost_fasta_open(const char *path)
{
    try {
        csvhost_fasta *h = new csvhost_fasta();
        if (h->g.setFilepath(path ? path : "") != 0) { delete h; g_err = "No FASTA filepath provided"; return nullptr; }
        return h;
    } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
void csvhost_fasta_free(csvhost_fasta *h) { delete h; }

// length of the view, -1 on error (unknown contig); at most cap bytes are copied to buf
int64_t csvhost_fasta_query(const csvhost_fasta *h, const char *chr, uint32_t pos_start, uint32_t pos_end, char *buf, uint64_t cap)
{
    try {
        std::string_view v = h->g.query(chr, pos_start, pos_end);
        if (buf && cap) memcpy(buf, v.data(), (size_t)std::min<uint64_t>(cap, v.size()));
        return (int64_t)v.size();
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}
int csvhost_fasta_compare(const csvhost_fasta *h, const char *chr, uint32_t pos_start, uint32_t pos_end, const char *seq, float threshold)
{
    try { return h->g.compare(chr, pos_start, pos_end, seq, threshold) ? 1 : 0; } catch (const std::exception &e) { g_err = e.what(); return -1; }
}
uint32_t csvhost_fasta_length(const csvhost_fasta *h, const char *chr) { return h->g.getChromosomeLength(chr); }
// contig header and the sorted contig list ('\n'-joined) as text; returns the full length
int64_t csvhost_fasta_contig_header(const csvhost_fasta *h, char *buf, uint64_t cap)
{
    const std::string s = h->g.getContigHeader();
    if (buf && cap) memcpy(buf, s.data(), (size_t)std::min<uint64_t>(cap, s.size()));
    return (int64_t)s.size();
}
int64_t csvhost_fasta_chromosomes(const csvhost_fasta *h, char *buf, uint64_t cap)
{
    std::string s;
    for (const std::string &c : h->g.getChromosomes()) { if (!s.empty()) s += '\n'; s += c; }
    if (buf && cap) memcpy(buf, s.data(), (size_t)std::min<uint64_t>(cap, s.size()));
    return (int64_t)s.size();
}

// Calls of contig t = calls[call_off[t] .. call_off[t+1]) with ALT strings alts[i]. Depth comes from resident shards
// (shards != nullptr, ShardDepthSource) or from host arrays depth[t][0..depth_len[t]) (a null depth[t] = contig without a map).
// map_order != 0 goes through saveToVCF's unordered_map iteration; otherwise contigs are written in the order given.
// counts = {total, unclassified, assembly-gap filtered}.
int csvhost_save_vcf(csv_ctx *ctx, const char *out_dir, const csvhost_fasta *fasta, const char *gap_path, const char *file_date, int n_contigs,
                     const char *const *chr_names, const uint64_t *call_off, const csvhost_call *calls, const char *const *alts,
                     csv_shard *const *shards, const uint32_t *const *depth, const uint64_t *depth_len, int map_order, int32_t *counts)
{
    GUARD({
        std::vector<std::pair<std::string, std::vector<SVCall>>> per((size_t)n_contigs);
        std::unordered_map<std::string, std::vector<uint32_t>> host_depth;
        ShardDepthSource shard_src(ctx);
        for (int t = 0; t < n_contigs; t++) {
            per[t].first = chr_names[t];
            for (uint64_t i = call_off[t]; i < call_off[t + 1]; i++) {
                SVCall c = from_pod(calls[i]);
                c.alt_allele = alts[i];
                per[t].second.push_back(c);
            }
            if (shards) { if (shards[t]) shard_src.add(chr_names[t], shards[t]); }
            else if (depth[t]) host_depth[chr_names[t]].assign(depth[t], depth[t] + depth_len[t]);
        }
        HostDepthSource host_src(host_depth);
        const DepthSource &src = shards ? (const DepthSource &)shard_src : (const DepthSource &)host_src;
        VCFOptions opt;
        opt.assembly_gaps = gap_path ? gap_path : "";
        opt.output_dir = out_dir;
        opt.file_date = file_date ? file_date : "";
        VCFCounts c;
        if (map_order) {
            std::unordered_map<std::string, std::vector<SVCall>> m;
            for (auto &e : per) m[e.first] = e.second;
            if (!saveToVCF(m, opt, fasta->g, src, &c)) throw std::runtime_error("saveToVCF returned early");
        } else {
            std::unordered_map<std::string, std::vector<std::pair<uint32_t, uint32_t>>> gaps;
            if (!opt.assembly_gaps.empty() && !loadAssemblyGaps(opt.assembly_gaps, gaps)) throw std::runtime_error("cannot open assembly gap file");
            std::ofstream f(opt.output_dir + "/output.vcf");
            if (!f.is_open()) throw std::runtime_error("cannot open output.vcf");
            std::vector<std::pair<std::string, const std::vector<SVCall> *>> order;
            for (auto &e : per) order.emplace_back(e.first, &e.second);
            c = writeVCF(f, order, opt, gaps, fasta->g, src);
        }
        if (counts) { counts[0] = c.total; counts[1] = c.unclassified; counts[2] = c.assembly_gap_filtered; }
    })
}

// ---- BAM / BAI reader and writer (bam_io) --------------------------------------------------------------
struct csvhost_bam { BamReader r; std::vector<BamShard> shards; std::string names; };

csvhost_bam *csvhost_bam_open(const char *path, int load_index)
{
    csvhost_bam *h = new csvhost_bam();
    if (!h->r.open(path) || (load_index && !h->r.loadIndex())) { g_err = h->r.error(); delete h; return nullptr; }
    for (const std::string &n : h->r.header().names) { if (!h->names.empty()) h->names += '\n'; h->names += n; }
    return h;
}
void csvhost_bam_close(csvhost_bam *h) { delete h; }
int csvhost_bam_n_ref(const csvhost_bam *h) { return (int)h->r.header().names.size(); }
const char *csvhost_bam_names(const csvhost_bam *h) { return h->names.c_str(); }
const char *csvhost_bam_text(const csvhost_bam *h) { return h->r.header().text.c_str(); }
uint32_t csvhost_bam_ref_len(const csvhost_bam *h, int tid) { return h->r.header().lens[(size_t)tid]; }

// chr != nullptr: readContig (needs the index) -> one shard; chr == nullptr: readAll -> one shard per contig with records.
// Returns the number of shards now held by the handle (replacing the previous ones), or -1.
int csvhost_bam_read(csvhost_bam *h, const char *chr, int want_seq, int want_qnames, int threads, uint32_t window_blocks, uint64_t *n_unplaced)
{
    try {
        BamReadOptions opt;
        opt.want_seq = want_seq != 0; opt.want_qnames = want_qnames != 0; opt.threads = threads;
        if (window_blocks) opt.window_blocks = window_blocks;
        h->shards.clear();
        bool ok;
        if (chr) { h->shards.emplace_back(); ok = h->r.readContig(chr, opt, h->shards.back()); }
        else ok = h->r.readAll(opt, [&](BamShard &&s) { h->shards.push_back(std::move(s)); }, n_unplaced);
        if (!ok) { g_err = h->r.error(); h->shards.clear(); return -1; }
        return (int)h->shards.size();
    } catch (const std::exception &e) { g_err = e.what(); return -1; }
}
// view of shard i: arrays stay owned by the handle until the next read / close
void csvhost_bam_shard(const csvhost_bam *h, int i, csv_reads *reads, int32_t *tid, uint32_t *target_len, const uint64_t **seq_off, const uint8_t **seq)
{
    const BamShard &s = h->shards[(size_t)i];
    *reads = s.view();
    *tid = s.tid; *target_len = s.target_len;
    *seq_off = s.seq_off.data(); *seq = s.seq.data();
}
// query names of shard i, '\n'-joined; returns the text length (copying at most cap bytes)
int64_t csvhost_bam_shard_qnames(const csvhost_bam *h, int i, char *buf, uint64_t cap)
{
    std::string t;
    for (const std::string &q : h->shards[(size_t)i].qnames) { t += q; t += '\n'; }
    if (buf && cap) memcpy(buf, t.data(), (size_t)std::min<uint64_t>(cap, t.size()));
    return (int64_t)t.size();
}

// Write a coordinate-sorted BAM + BAI from arrays. qnames: '\n'-separated (one per record). seq_off / seq may be null (l_seq = 0);
// l_seq[i] gives the base count (the packed length alone cannot tell odd from even).
int csvhost_bam_write(const char *path, const char *text, int n_ref, const char *ref_names, const uint32_t *ref_lens, uint64_t n, const int32_t *tid,
                      const int32_t *pos, const uint16_t *flag, const uint8_t *mapq, const uint64_t *cigar_off, const uint32_t *cigar,
                      const char *qnames, const uint64_t *seq_off, const uint8_t *seq, const int32_t *l_seq, int level, int threads)
{
    GUARD({
        BamHeader hd;
        hd.text = text ? text : "";
        const char *p = ref_names;
        for (int i = 0; i < n_ref; i++) {
            const char *e = strchr(p, '\n');
            hd.names.emplace_back(p, e ? (size_t)(e - p) : strlen(p));
            hd.lens.push_back(ref_lens[i]);
            p = e ? e + 1 : p + strlen(p);
        }
        BamWriter w;
        if (!w.open(path, hd, level, threads)) throw std::runtime_error(w.error());
        const char *q = qnames;
        for (uint64_t i = 0; i < n; i++) {
            const char *e = strchr(q, '\n');
            std::string name(q, e ? (size_t)(e - q) : strlen(q));
            q = e ? e + 1 : q + strlen(q);
            const int32_t ls = (seq && l_seq) ? l_seq[i] : 0;
            w.add(tid[i], pos[i], mapq[i], flag[i], name, cigar + cigar_off[i], (uint32_t)(cigar_off[i + 1] - cigar_off[i]),
                  ls ? seq + seq_off[i] : nullptr, ls, nullptr);
        }
        if (!w.close()) throw std::runtime_error(w.error());
    })
}

// The synthetic shard as a coordinate-sorted BAM + BAI on one contig ("stage inputs ... as a real BGZF BAM + BAI").
// Query names are r<read id> (a primary and its supplementary record share one); sequences are written when the shard has them, else l_seq = 0 ("*").
int csvhost_synth_write_bam(const csvhost_synth *h, const char *path, const char *chr_name, int level, int threads, uint64_t *bam_bytes)
{
    GUARD({
        const SynthShard &sh = h->sh;
        BamHeader hd;
        hd.text = std::string("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:") + chr_name + "\tLN:" + std::to_string(sh.depth_len - 1) + "\n";
        hd.names.push_back(chr_name);
        hd.lens.push_back(sh.depth_len - 1);
        BamWriter w;
        if (!w.open(path, hd, level, threads)) throw std::runtime_error(w.error());
        const bool have_seq = !sh.seq.empty() && sh.seq_off.size() == sh.pos.size() + 1;
        // the generator emits reads in position order per thread slice and merges them sorted; verify rather than assume
        for (size_t i = 0; i < sh.pos.size(); i++) {
            if (i && sh.pos[i] < sh.pos[i - 1]) throw std::runtime_error("synthetic shard is not coordinate-sorted");
            const uint32_t *cg = sh.cigar.data() + sh.cigar_off[i];
            const uint32_t nc = (uint32_t)(sh.cigar_off[i + 1] - sh.cigar_off[i]);
            int32_t l_seq = 0;
            if (have_seq) for (uint32_t k = 0; k < nc; k++) if ((0x3C1A7u >> ((cg[k] & 15) << 1)) & 1) l_seq += (int32_t)(cg[k] >> 4);   // query-consuming ops
            w.add(0, sh.pos[i], sh.mapq[i], sh.flag[i], "r" + std::to_string(sh.qname_id[i]), cg, nc, have_seq ? sh.seq.data() + sh.seq_off[i] : nullptr, l_seq, nullptr);
        }
        if (!w.close()) throw std::runtime_error(w.error());
        if (bam_bytes) { bgzf::MappedFile f; std::string e; *bam_bytes = f.open(path, &e) ? f.size() : 0; }
    })
}

// Several synthetic shards into ONE coordinate-sorted BAM + BAI, one contig each, appended in tid order.
struct csvhost_bam_writer { BamWriter w; };
csvhost_bam_writer *csvhost_bam_writer_open(const char *path, int n_ref, const char *ref_names, const uint32_t *ref_lens, int level, int threads)
{
    try {
        BamHeader hd;
        hd.text = "@HD\tVN:1.6\tSO:coordinate\n";
        const char *p = ref_names;
        for (int i = 0; i < n_ref; i++) {
            const char *e = strchr(p, '\n');
            hd.names.emplace_back(p, e ? (size_t)(e - p) : strlen(p));
            hd.lens.push_back(ref_lens[i]);
            hd.text += "@SQ\tSN:" + hd.names.back() + "\tLN:" + std::to_string(ref_lens[i]) + "\n";
            p = e ? e + 1 : p + strlen(p);
        }
        csvhost_bam_writer *h = new csvhost_bam_writer();
        if (!h->w.open(path, hd, level, threads)) { g_err = h->w.error(); delete h; return nullptr; }
        return h;
    } catch (const std::exception &e) { g_err = e.what(); return nullptr; }
}
int csvhost_bam_writer_append_synth(csvhost_bam_writer *h, const csvhost_synth *syn, int tid)
{
    GUARD({
        const SynthShard &sh = syn->sh;
        for (size_t i = 0; i < sh.pos.size(); i++) {
            if (i && sh.pos[i] < sh.pos[i - 1]) throw std::runtime_error("synthetic shard is not coordinate-sorted");
            h->w.add(tid, sh.pos[i], sh.mapq[i], sh.flag[i], "r" + std::to_string(tid) + "_" + std::to_string(sh.qname_id[i]), sh.cigar.data() + sh.cigar_off[i],
                     (uint32_t)(sh.cigar_off[i + 1] - sh.cigar_off[i]), nullptr, 0, nullptr);
        }
    })
}
int csvhost_bam_writer_close(csvhost_bam_writer *h)
{
    GUARD({
        const bool ok = h->w.close();
        const std::string e = h->w.error();
        delete h;
        if (!ok) throw std::runtime_error(e);
    })
}

struct csvhost_bam_stats { uint64_t n_contigs, n_reads, n_cigar, bam_bytes; double ms_decode, ms_total; };

// SVCaller::runBam: chrs = '\n'-separated contig names or null (all). SNPs: none (every window gets the dummy observation).
// Calls come back grouped by contig in header order, with their contig index in out_tid.
int csvhost_run_bam(csv_ctx *ctx, const char *bam_path, const char *chrs, int threads, const csv_hmm *hmm, double eps, double min_pts_pct,
                    int sample_size, uint32_t min_cnv, int passes /* bit 0: split-read pass, bit 1: CIGAR copy-number pass, bit 2: --save-cnv */, const csvhost_fasta *fasta, const char *vcf_dir, const char *gap_path,
                    const char *file_date, csvhost_call *out, int32_t *out_tid, uint64_t cap, uint64_t *n_out, csvhost_bam_stats *stats,
                    const char *snp_vcf, const char *pfb_table, const char *ethnicity)
{
    GUARD({
        std::vector<std::string> list;
        for (const char *p = chrs; p && *p;) {
            const char *e = strchr(p, '\n');
            list.emplace_back(p, e ? (size_t)(e - p) : strlen(p));
            p = e ? e + 1 : p + strlen(p);
        }
        RunParams P; P.dbscan_epsilon = eps; P.dbscan_min_pts_pct = min_pts_pct; P.sample_size = sample_size; P.min_cnv_length = min_cnv;
        P.split_svs = (passes & 1) != 0; P.cigar_cn = (passes & 2) != 0; P.save_cnv = (passes & 4) != 0;
        P.snp_vcf = snp_vcf ? snp_vcf : ""; P.pfb_table = pfb_table ? pfb_table : ""; P.ethnicity = ethnicity ? ethnicity : "";
        if (fasta && vcf_dir) {
            P.ref_genome = fasta_genome(fasta);
            P.vcf.output_dir = vcf_dir; P.vcf.assembly_gaps = gap_path ? gap_path : ""; P.vcf.file_date = file_date ? file_date : "";
        }
        SVCaller caller(ctx);
        std::unordered_map<std::string, std::vector<SVCall>> calls;
        BamRunStats bs;
        caller.runBam(bam_path, list, threads, chmm_from_pod(hmm), P, calls, &bs);
        BamReader hdr_reader;
        if (!hdr_reader.open(bam_path)) throw std::runtime_error(hdr_reader.error());
        uint64_t k = 0;
        for (size_t t = 0; t < hdr_reader.header().names.size(); t++) {
            auto it = calls.find(hdr_reader.header().names[t]);
            if (it == calls.end()) continue;
            for (const SVCall &c : it->second) {
                if (k < cap) {
                    csvhost_call p;
                    p.start = c.start; p.end = c.end; p.sv_type = (int32_t)c.sv_type; p.cluster_size = c.cluster_size; p.hmm_likelihood = c.hmm_likelihood;
                    p.id = -1; p.aln_flags = (uint32_t)c.aln_type.to_ulong(); p.genotype = (int32_t)c.genotype; p.cn_state = c.cn_state; p.aln_offset = c.aln_offset;
                    out[k] = p; out_tid[k] = (int32_t)t;
                }
                k++;
            }
        }
        *n_out = k;
        if (stats) { stats->n_contigs = bs.n_contigs; stats->n_reads = bs.n_reads; stats->n_cigar = bs.n_cigar; stats->bam_bytes = bs.bam_bytes;
                     stats->ms_decode = bs.ms_decode; stats->ms_total = bs.ms_total; }
    })
}

// ---- SNP / population-frequency VCF ingestion (snp_io) ------------------------------------------------------
struct csvhost_snp { SNPFile f; };

csvhost_snp *csvhost_snp_open(const char *snp_vcf, int threads)
{
    csvhost_snp *h = new csvhost_snp();
    std::string e;
    if (!h->f.load(snp_vcf ? snp_vcf : "", threads, &e)) { g_err = e; delete h; return nullptr; }
    return h;
}
void csvhost_snp_free(csvhost_snp *h) { delete h; }
uint64_t csvhost_snp_kept(const csvhost_snp *h) { return h->f.records_kept(); }

// readSNPAlleleFrequencies for one region from the loaded tables. Returns the number of positions (file order), -1 when cap is too
// small. baf_out[i] = map value at pos_out[i]; at most one population frequency (has_pfb, pfb_pos, pfb_val).
int64_t csvhost_snp_query(csvhost_snp *h, const char *chr, const char *pfb_vcf, const char *ethnicity, uint32_t start_pos, uint32_t end_pos,
                          uint32_t *pos_out, double *baf_out, uint64_t cap, int *has_pfb, uint32_t *pfb_pos, double *pfb_val, int threads)
{
    const SNPFileTable &t = h->f.table(chr, pfb_vcf ? pfb_vcf : "", ethnicity ? ethnicity : "", threads);
    std::vector<uint32_t> pos;
    std::unordered_map<uint32_t, double> baf, pfb;
    t.query(start_pos, end_pos, pos, baf, pfb);
    *has_pfb = 0;
    if (pos.size() > cap) return -1;
    for (size_t i = 0; i < pos.size(); i++) { pos_out[i] = pos[i]; baf_out[i] = baf[pos[i]]; }
    if (!pfb.empty()) { *has_pfb = (int)pfb.size(); *pfb_pos = pfb.begin()->first; *pfb_val = pfb.begin()->second; }
    return (int64_t)pos.size();
}

// the --pfb table: path for `chr` copied to buf (empty when absent); -1 when the table cannot be loaded
int64_t csvhost_pfb_path(const char *table_path, const char *chr, char *buf, uint64_t cap)
{
    AlleleFreqFiles a;
    std::string e;
    if (!a.load(table_path, &e)) { g_err = e; return -1; }
    const std::string p = a.get(chr);
    if (buf && cap) { memcpy(buf, p.data(), (size_t)std::min<uint64_t>(cap, p.size())); }
    return (int64_t)p.size();
}
int64_t csvhost_gnomad_contig(const char *chr, const char *pfb_path, char *buf, uint64_t cap)
{
    const std::string g = gnomadContigName(chr, pfb_path);
    if (buf && cap) memcpy(buf, g.data(), (size_t)std::min<uint64_t>(cap, g.size()));
    return (int64_t)g.size();
}

// ---- fast inflate / CRC (fast_inflate.h) on their own, for the tests -------------------------------------------
// 1 = decoded (out filled), 0 = declined (the caller would use zlib); never touches memory outside the two buffers
int csvhost_fast_inflate(const uint8_t *in, uint64_t in_len, uint8_t *out, uint64_t out_len) { return fastz::inflate(in, in_len, out, out_len) ? 1 : 0; }
uint32_t csvhost_fast_crc32(const uint8_t *buf, uint64_t len) { return fastz::crc32(buf, len); }

// ---- HMM file + Viterbi seam ------------------------------------------------------------------
int csvhost_read_chmm(const char *path, csv_hmm *out, int32_t *N)
{
    GUARD({
        CHMM h = ReadCHMM(path);
        *N = h.N;
        if (h.N == 6) {
            for (int i = 0; i < 6; i++) { for (int j = 0; j < 6; j++) out->A[i * 6 + j] = h.A[i][j]; out->pi[i] = h.pi[i]; out->B1_mean[i] = h.B1_mean[i]; out->B1_sd[i] = h.B1_sd[i]; }
            for (int i = 0; i < 5; i++) { out->B2_mean[i] = h.B2_mean[i]; out->B2_sd[i] = h.B2_sd[i]; }
            out->B1_uf = h.B1_uf; out->B2_uf = h.B2_uf;
        }
    })
}

}  // extern "C"
