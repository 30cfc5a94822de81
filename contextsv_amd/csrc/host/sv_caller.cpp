#include "sv_caller.h"

#include <deque>

#include <algorithm>
#include <cstdio>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <thread>
#include <stdexcept>
#include <atomic>

#include <memory>

#include "dbscan.h"
#include "log.h"
#include "par.h"
#include "sort_select.h"
#include "umap_order.h"

namespace {
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void check(csv_ctx *ctx, int rc, const char *what)
{
    if (rc != CSV_OK) throw std::runtime_error(std::string(what) + ": " + csvgpu_last_error(ctx));
}
}  // namespace

// Field values of the SVCall the reference builds per op (sv_caller.cpp:569-643): INS from I (CIGARINS) or
// from a soft clip (CIGARCLIP), DEL from D (CIGARDEL); ALT is the inserted sequence only when op_len <= 50
// (i.e. exactly 50, ambiguity codes RYKMSWBDHV in either case -> N), otherwise "<INS>"; DEL -> "<DEL>".
SVCall SVCaller::toSVCall(const csv_sig &s, const SeqStore *seq)
{
    const uint32_t kind = CSV_SIG_KIND(s);
    SVEvidenceFlags flags;
    if (kind == CSV_KIND_DEL) {
        flags.set((size_t)SVDataType::CIGARDEL);
        return SVCall(s.start, s.end, SVType::DEL, "<DEL>", flags, Genotype::UNKNOWN, 0.0, 0, 0, 0);
    }
    flags.set((size_t)(kind == CSV_KIND_INS ? SVDataType::CIGARINS : SVDataType::CIGARCLIP));
    const uint32_t op_len = s.end - s.start + 1;
    std::string alt = "<INS>";
    if (op_len <= 50) {
        alt.assign(op_len, 'N');
        const uint32_t q0 = CSV_SIG_QPOS(s);
        // a record stored without its sequence ("*", l_seq = 0) has nothing to copy from: the ALT stays N's
        if (seq && seq->seq && seq->seq_off && ((uint64_t)q0 + op_len + 1) / 2 <= seq->seq_off[s.read + 1] - seq->seq_off[s.read]) {
            static const char nt16[] = "=ACMGRSVTWYHKDBN";              // BAM 4-bit code table (SAM spec §4.2.3)
            const uint8_t *p = seq->seq + seq->seq_off[s.read];
            for (uint32_t j = 0; j < op_len; j++) {
                const uint32_t q = q0 + j;
                const char b = nt16[(p[q >> 1] >> ((~q & 1u) << 2)) & 0xf];
                switch (b) {
                    case 'R': case 'Y': case 'K': case 'M': case 'S': case 'W': case 'B': case 'D': case 'H': case 'V': alt[j] = 'N'; break;
                    default: alt[j] = b;
                }
            }
        }
    }
    return SVCall(s.start, s.end, SVType::INS, alt, flags, Genotype::UNKNOWN, 0.0, 0, 0, 0);
}

// mergeTypeWithLabels (sv_object.cpp) specialised to raw CIGAR signatures, keep_noise = false: same bucket order,
// same std::sort comparator on an index vector, same top-20 % / middle element rule, cluster_size = bucket size.
void SVCaller::mergeSignaturesWithLabels(const csv_sig *sig, const int32_t *labels, uint64_t n, const SeqStore *seq, std::vector<SVCall> &merged)
{
    // members of every label, in input order, as (length, index) pairs: the representative rule only compares lengths, and the
    // selection below streams over this compact array instead of chasing indices into the 16-byte records
    struct Ent { uint32_t len, idx; };
    static thread_local std::vector<uint32_t> head, cur;
    static thread_local std::vector<Ent> member;
    int32_t max_label = -2;
    for (uint64_t i = 0; i < n; i++) max_label = std::max(max_label, labels[i]);
    const size_t n_slots = (size_t)(max_label + 3);
    head.assign(n_slots + 1, 0);
    for (uint64_t i = 0; i < n; i++) head[(size_t)(labels[i] + 2) + 1]++;
    for (size_t s = 0; s < n_slots; s++) head[s + 1] += head[s];
    cur.assign(head.begin(), head.end() - 1);
    if (member.size() < n) member.resize(n);
    for (uint64_t i = 0; i < n; i++) member[cur[(size_t)(labels[i] + 2)]++] = Ent{sig[i].end - sig[i].start, (uint32_t)i};
    for (size_t s = 0; s < n_slots; s++) {
        const size_t sz = head[s + 1] - head[s];
        if (sz < 2) continue;
        Ent *m = member.data() + head[s];
        // the slot std::sort(by length desc) would fill at [top/2] — selected in O(sz), see sort_select.h
        const size_t top = (size_t)std::max(1, (int)(sz * 0.2));
        auto by_len_desc = [](const Ent &a, const Ent &b) { return a.len > b.len; };
        const uint32_t pick = csvhost::std_sort_select(m, m + sz, (std::ptrdiff_t)(top / 2), by_len_desc)->idx;
        SVCall rep = toSVCall(sig[pick], seq);
        rep.cluster_size = (int)sz;
        merged.push_back(rep);
    }
}

// How many signatures did the chromosomes of this process need room for so far? New result buffers start at that size: a buffer that
// turns out too small costs a CSV_ECAPACITY round trip (grow + synchronous fetch), and a driver that is called once per batch of
// chromosomes would pay it again for every buffer of every call (3 lanes x 3 buffers x ~1 ms per call of the benchmark).
namespace {
std::atomic<uint64_t> g_result_hint{0};
uint64_t result_capacity_hint(const csv_ctx *) { return std::max<uint64_t>(1 << 16, g_result_hint.load(std::memory_order_relaxed)); }
void note_result_size(const csv_ctx *, uint64_t n_sig)
{
    uint64_t h = g_result_hint.load(std::memory_order_relaxed);
    while (n_sig > h && !g_result_hint.compare_exchange_weak(h, n_sig, std::memory_order_relaxed)) {}
}
}  // namespace

void SVCaller::DeviceOut::reserve(csv_ctx *c, uint64_t n)
{
    if (n <= cap && c == ctx) return;
    release();
    ctx = c;
    const uint64_t want = n + n / 4 + 4096;
    sig = (csv_sig *)csvgpu_host_alloc(c, want * sizeof(csv_sig));
    lab = (int32_t *)csvgpu_host_alloc(c, want * sizeof(int32_t));
    if (!sig || !lab) { release(); throw std::runtime_error("cannot allocate page-locked result buffers"); }
    cap = want;
}

void SVCaller::runDeviceChain(const std::string &chr, csv_shard *shard, double eps, double pct, DeviceOut &out, ChrStats &st)
{
    const double t0 = now_ms();
    csv_chr_result res;
    if (!out.cap) out.reserve(ctx, result_capacity_hint(ctx));
    int rc = csvgpu_chr_pipeline_fetch(ctx, shard, (uint32_t)min_oplen, (uint8_t)min_mapq, eps, pct, &res, out.sig, out.lab, out.cap);
    if (rc == CSV_ECAPACITY) {                         // first contig of this size: grow the buffers, fetch what the device already holds
        out.reserve(ctx, res.n_sig);
        rc = csvgpu_chr_fetch(ctx, shard, &res, out.sig, out.lab);
    }
    check(ctx, rc, "processChromosome");
    note_result_size(ctx, res.n_sig);
    st.n_signatures = res.n_sig; st.n_del = res.n_del; st.n_ins = res.n_ins;
    st.depth_sum = res.depth_sum; st.depth_nonzero = res.depth_nonzero; st.mean_chr_cov = res.mean_cov; st.dbscan_min_pts = res.min_pts;
    if (pct > 0.0)
        printMessage(chr + ": Mean chr. cov.: " + std::to_string(res.mean_cov) + " (DBSCAN min. pts.= " + std::to_string(res.min_pts) +
                     ", min. pts. pct.= " + std::to_string(pct) + ")");
    out.n_del = res.n_del; out.n_ins = res.n_ins;
    st.ms_device = now_ms() - t0;
}

// mergeSVs(chr_sv_calls, eps, min_pts, keep_noise=false) with the labels already computed on device.
// Every CIGAR call has hmm_likelihood == 0, so only the length-ranked branch of the representative choice
// can run (sv_object.cpp:187-244 of the reference); it is evaluated on the 16-byte signatures and an SVCall
// (with its strings) is materialised for the chosen member only.
void SVCaller::mergeOrdered(const csv_sig *sig, const int32_t *labels, uint64_t n_del, uint64_t n_ins, const SeqStore *seq, std::vector<SVCall> &chr_sv_calls, bool share_pool)
{
    const uint64_t n_sig = n_del + n_ins;
    chr_sv_calls.clear();
    if (n_sig < 2) {                                   // mergeSVs returns early (:49-51)
        for (uint64_t i = 0; i < n_sig; i++) chr_sv_calls.push_back(toSVCall(sig[i], seq));
        return;
    }
    const uint64_t type_n[2] = {n_del, n_ins};
    auto one_type = [&](int t, std::vector<SVCall> &dst) {
        const uint64_t base = t ? n_del : 0;
        if (type_n[t] < 2) for (uint64_t i = 0; i < type_n[t]; i++) dst.push_back(toSVCall(sig[base + i], seq));
        else mergeSignaturesWithLabels(sig + base, labels + base, type_n[t], seq, dst);
    };
    // (share_pool = false: a pass over many contigs — the other lanes' merges hide this one, and the host pool belongs to the split-read and
    // copy-number sections running beside the pass: a merge that took it made those run inline, 1 ms sections became 6 ms ones)
    if (n_sig < 40000 || !share_pool) {
        one_type(0, chr_sv_calls);
        one_type(1, chr_sv_calls);
        return;
    }
    // (a type with few signatures — the deletions of a synthetic ONT contig — goes through the sequential form, the other through the sections)
    const bool big[2] = {n_del >= 20000, n_ins >= 20000};
    // A large contig (a rank that holds chr1 alone waited 2.85 ms here, its device chain takes 1.9): the same bucket order and the same
    // selections as mergeSignaturesWithLabels, in sections the host pool can share — per-chunk label counts, the members scattered chunk by
    // chunk (a chunk's members of a label go behind the earlier chunks': input order kept), then the buckets: each type's noise bucket
    // (93 % of the signatures; its selection is one sequential replay) as an item of its own beside ranges of clusters. When the pool is
    // taken (a whole-genome pass: the lanes' other merges hide this one) the sections run inline, one after the other.
    struct Ent { uint32_t len, idx; };
    struct TypeWork {
        uint64_t base = 0, n = 0;
        size_t n_slots = 0;
        std::vector<std::vector<uint32_t>> cnt;      // per chunk: members per slot, then the chunk's first position per slot
        std::vector<uint32_t> head;
        std::vector<Ent> member;
        std::vector<std::vector<SVCall>> part;       // per bucket item, in slot order
    } W[2];
    constexpr size_t kChunks = 8;
    std::unique_ptr<csvhost::TraceScope> tr(new csvhost::TraceScope("merge: label counts + scatter"));
    for (int t = 0; t < 2; t++) { W[t].base = t ? n_del : 0; W[t].n = big[t] ? type_n[t] : 0; W[t].cnt.assign(kChunks, {}); W[t].member.resize(W[t].n); }
    auto chunk = [&](const TypeWork &w, size_t c, uint64_t &a, uint64_t &b) { a = w.n * c / kChunks; b = w.n * (c + 1) / kChunks; };
    csvhost::parallel_for(2 * kChunks, 0, [&](size_t it) {
        TypeWork &w = W[it / kChunks];
        uint64_t a, b; chunk(w, it % kChunks, a, b);
        const int32_t *lab = labels + w.base;
        int32_t mx = -2;
        for (uint64_t i = a; i < b; i++) mx = std::max(mx, lab[i]);
        std::vector<uint32_t> &c = w.cnt[it % kChunks];
        c.assign((size_t)(mx + 3), 0);
        for (uint64_t i = a; i < b; i++) c[(size_t)(lab[i] + 2)]++;
    });
    for (int t = 0; t < 2; t++) {
        TypeWork &w = W[t];
        for (auto &c : w.cnt) w.n_slots = std::max(w.n_slots, c.size());
        w.head.assign(w.n_slots + 1, 0);
        for (auto &c : w.cnt) { c.resize(w.n_slots, 0); for (size_t s2 = 0; s2 < w.n_slots; s2++) w.head[s2 + 1] += c[s2]; }
        for (size_t s2 = 0; s2 < w.n_slots; s2++) w.head[s2 + 1] += w.head[s2];
        std::vector<uint32_t> run(w.head.begin(), w.head.end() - 1);
        for (auto &c : w.cnt) for (size_t s2 = 0; s2 < w.n_slots; s2++) { const uint32_t k = c[s2]; c[s2] = run[s2]; run[s2] += k; }
    }
    csvhost::parallel_for(2 * kChunks, 0, [&](size_t it) {
        TypeWork &w = W[it / kChunks];
        uint64_t a, b; chunk(w, it % kChunks, a, b);
        const int32_t *lab = labels + w.base;
        const csv_sig *sg = sig + w.base;
        std::vector<uint32_t> &cur = w.cnt[it % kChunks];
        for (uint64_t i = a; i < b; i++) w.member[cur[(size_t)(lab[i] + 2)]++] = Ent{sg[i].end - sg[i].start, (uint32_t)i};
    });
    // bucket items per type: [0, 1) = the noise bucket, then the clusters in kRanges ranges
    constexpr size_t kRanges = 7;
    tr.reset(new csvhost::TraceScope("merge: buckets (noise bucket + cluster ranges)"));
    for (int t = 0; t < 2; t++) W[t].part.assign(1 + kRanges, {});
    csvhost::parallel_for(2 * (1 + kRanges), 0, [&](size_t it) {
        TypeWork &w = W[it / (1 + kRanges)];
        const size_t item = it % (1 + kRanges);
        if (!big[it / (1 + kRanges)]) { if (item == 0) one_type((int)(it / (1 + kRanges)), w.part[0]); return; }
        size_t s0 = 0, s1 = std::min<size_t>(1, w.n_slots);
        if (item > 0) {
            const size_t rest = w.n_slots > 1 ? w.n_slots - 1 : 0;
            s0 = 1 + rest * (item - 1) / kRanges; s1 = 1 + rest * item / kRanges;
        }
        std::vector<SVCall> &dst = w.part[item];
        const csv_sig *sg = sig + w.base;
        for (size_t s2 = s0; s2 < s1; s2++) {
            const size_t sz = w.head[s2 + 1] - w.head[s2];
            if (sz < 2) continue;
            Ent *m = w.member.data() + w.head[s2];
            const size_t top = (size_t)std::max(1, (int)(sz * 0.2));
            auto by_len_desc = [](const Ent &x, const Ent &y) { return x.len > y.len; };
            const uint32_t pick = csvhost::std_sort_select(m, m + sz, (std::ptrdiff_t)(top / 2), by_len_desc)->idx;
            SVCall rep = toSVCall(sg[pick], seq);
            rep.cluster_size = (int)sz;
            dst.push_back(std::move(rep));
        }
    });
    tr.reset();
    for (int t = 0; t < 2; t++)
        for (auto &v : W[t].part) chr_sv_calls.insert(chr_sv_calls.end(), std::make_move_iterator(v.begin()), std::make_move_iterator(v.end()));
}

void SVCaller::hostMerge(const std::string &chr, const DeviceOut &in, const SeqStore *seq, std::vector<SVCall> &chr_sv_calls, ChrStats &st, bool share_pool)
{
    const double t1 = now_ms();
    csvhost::TraceScope tr("lane: host merge");
    printMessage(chr + ": Merging CIGAR...");
    mergeOrdered(in.sig, in.lab, in.n_del, in.n_ins, seq, chr_sv_calls, share_pool);
    st.ms_host_merge = now_ms() - t1;
    printMessage(chr + ": Found " + std::to_string(getSVCount(chr_sv_calls)) + " SV candidates in the CIGAR string");
}

void SVCaller::processResidentChromosome(const std::string &chr, csv_shard *shard, const SeqStore *seq, double eps, double pct,
                                         std::vector<SVCall> &chr_sv_calls, ChrStats &st)
{
    DeviceOut d;
    runDeviceChain(chr, shard, eps, pct, d, st);
    hostMerge(chr, d, seq, chr_sv_calls, st);
}

void SVCaller::processResidentChromosomesPipelined(const std::vector<csv_shard *> &shards, const SeqStore *seq, double eps, double pct,
                                                   std::vector<std::vector<SVCall>> &calls, std::vector<ChrStats> &stats)
{
    processResidentChromosomesPipelined(shards, seq ? std::vector<const SeqStore *>(shards.size(), seq) : std::vector<const SeqStore *>(), eps, pct, calls, stats);
}

void SVCaller::processResidentChromosomesPipelined(const std::vector<csv_shard *> &shards, const std::vector<const SeqStore *> &seqs, double eps, double pct,
                                                   std::vector<std::vector<SVCall>> &calls, std::vector<ChrStats> &stats)
{
    const size_t n = shards.size();
    if (!seqs.empty() && seqs.size() != n) throw std::invalid_argument("processResidentChromosomesPipelined: one SeqStore per shard, or none");
    calls.assign(n, {});
    stats.assign(n, ChrStats());
    // The representative choice of one chromosome (~0.5 ms for chr22: a sequential selection over the noise bucket) takes about as
    // long as its device chain, so two merge threads take alternate chromosomes; three result slots rotate between the device
    // thread (filling one) and the two merges in progress.
    constexpr size_t kMergers = 2, kSlots = kMergers + 1;
    DeviceOut slot[kSlots];
    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> ready(n, 0), merged(n, 0);          // guarded by mu
    bool failed = false;
    std::exception_ptr worker_err;
    csvhost::WorkerThreads &pool = csvhost::WorkerThreads::instance();
    std::vector<csvhost::WorkerThreads::Ticket> workers;
    for (size_t w = 0; w < kMergers; w++) {
        workers.push_back(pool.start([&, w] {
            try {
                for (size_t i = w; i < n; i += kMergers) {
                    { std::unique_lock<std::mutex> l(mu); cv.wait(l, [&] { return ready[i] || failed; }); if (failed) return; }
                    hostMerge("shard" + std::to_string(i), slot[i % kSlots], seqs.empty() ? nullptr : seqs[i], calls[i], stats[i], n <= 2);
                    { std::lock_guard<std::mutex> l(mu); merged[i] = 1; }
                    cv.notify_all();
                    if (on_merged) on_merged(i);
                }
            } catch (...) {
                std::lock_guard<std::mutex> l(mu);
                if (!worker_err) worker_err = std::current_exception();
                failed = true;
                cv.notify_all();
            }
        }));
    }
    auto stop_workers = [&] {
        { std::lock_guard<std::mutex> l(mu); failed = true; }
        cv.notify_all();
        for (auto &t : workers) pool.wait(t);
    };
    // The scan + depth pair of the next shard (CSV_JOBS_AHEAD of them: 1) is queued ahead of the shard whose results are being fetched. More
    // than one ahead was measured and is no better (25.7 / 26.7 ms per genome step with one, 26.4 / 26.9 with three): the gaps between the
    // big kernels were not a starved gate but a lane's stream sharing the gate's hardware queue — see csvgpu_gate_open.
    static const size_t kAhead = [] { const char *e = getenv("CSV_JOBS_AHEAD"); const int v = e && *e ? atoi(e) : 1; return (size_t)std::min(std::max(v, 1), 8); }();
    std::deque<csv_job *> ahead;                          // jobs whose scan + depth pass is already queued, oldest first
    size_t next_begin = 0;
    auto begin_one = [&] {
        csv_job *j = csvgpu_chr_job_begin(ctx, shards[next_begin], (uint32_t)min_oplen, (uint8_t)min_mapq, pct);
        if (!j) throw std::runtime_error(std::string("processChromosome: ") + csvgpu_last_error(ctx));
        ahead.push_back(j);
        next_begin++;
    };
    auto abort_all = [&] { for (csv_job *j : ahead) csvgpu_chr_job_abort(ctx, j); ahead.clear(); };
    // (a job works in its shard's own buffers: a shard that is listed again — the benchmark's repeated passes over one contig — is only
    // begun again once the job before it on that shard has queued its clustering)
    std::vector<csv_shard *> open_shards;                 // shards of the jobs in `ahead`, same order
    auto shard_free = [&](csv_shard *sh) { return std::find(open_shards.begin(), open_shards.end(), sh) == open_shards.end(); };
    auto top_up = [&] {
        while (next_begin < n && ahead.size() < kAhead && shard_free(shards[next_begin])) { open_shards.push_back(shards[next_begin]); begin_one(); }
    };
    try {
        top_up();
        for (size_t i = 0; i < n; i++) {
            // slot i % kSlots is free once shard i - kSlots has been merged
            if (i >= kSlots) {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [&] { return merged[i - kSlots] || failed; });
                if (failed) break;
            }
            const double t0 = now_ms();
            DeviceOut &out = slot[i % kSlots];
            ChrStats &st = stats[i];
            if (!out.cap) out.reserve(ctx, result_capacity_hint(ctx));
            if (ahead.empty()) top_up();                   // (a repeated shard: its next pass could not be queued ahead)
            csv_job *job = ahead.front();
            ahead.pop_front();
            int rc = csvgpu_chr_job_cluster(ctx, job, eps, out.sig, out.lab, out.cap);
            if (rc) { csvgpu_chr_job_abort(ctx, job); check(ctx, rc, "processChromosome"); }      // abort keeps the failure's own message
            open_shards.erase(open_shards.begin());       // this shard's clustering is queued: a later pass over it may follow on the stream
            // more scan + depth passes go into the queue now, behind this shard's clustering and copies
            try { top_up(); } catch (...) { csvgpu_chr_job_abort(ctx, job); throw; }
            csv_chr_result res;
            rc = csvgpu_chr_job_end(ctx, job, &res);
            if (rc == CSV_ECAPACITY) {                     // first contig of this size: grow the buffers, fetch what the device still holds
                out.reserve(ctx, res.n_sig);
                rc = csvgpu_chr_fetch(ctx, shards[i], &res, out.sig, out.lab);
            }
            check(ctx, rc, "processChromosome");
            note_result_size(ctx, res.n_sig);
            st.n_signatures = res.n_sig; st.n_del = res.n_del; st.n_ins = res.n_ins;
            st.depth_sum = res.depth_sum; st.depth_nonzero = res.depth_nonzero; st.mean_chr_cov = res.mean_cov; st.dbscan_min_pts = res.min_pts;
            out.n_del = res.n_del; out.n_ins = res.n_ins;
            st.ms_device = now_ms() - t0;
            if (on_device) on_device(i);
            { std::lock_guard<std::mutex> l(mu); ready[i] = 1; }
            cv.notify_all();
        }
        abort_all();                                      // (only after a merge-thread failure cut the loop short)
    } catch (...) {
        abort_all();
        stop_workers();
        throw;
    }
    for (auto &t : workers) pool.wait(t);
    if (worker_err) std::rethrow_exception(worker_err);
}

std::vector<int> SVCaller::assignShards(const std::vector<double> &weights, int world)
{
    if (world < 1) throw std::invalid_argument("assignShards: world < 1");
    std::vector<size_t> order(weights.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return weights[a] != weights[b] ? weights[a] > weights[b] : a < b; });
    std::vector<double> load((size_t)world, 0.0);
    std::vector<int> rank_of(weights.size(), 0);
    for (size_t i : order) {
        int r = 0;
        for (int k = 1; k < world; k++) if (load[(size_t)k] < load[(size_t)r]) r = k;      // (least loaded, lowest rank on ties)
        rank_of[i] = r;
        load[(size_t)r] += weights[i];
    }
    return rank_of;
}

void SVCaller::processResidentLanes(const std::vector<Lane> &lanes, const SeqStore *seq, double eps, double pct,
                                    std::vector<std::vector<std::vector<SVCall>>> &calls, std::vector<std::vector<ChrStats>> &stats,
                                    const std::function<void(size_t lane, size_t k)> &on_merged, int min_mapq, int min_oplen,
                                    const std::function<void(size_t lane, size_t k)> &on_device)
{
    calls.assign(lanes.size(), {});
    stats.assign(lanes.size(), {});
    std::vector<std::exception_ptr> errs(lanes.size());
    csvhost::WorkerThreads &pool = csvhost::WorkerThreads::instance();
    std::vector<csvhost::WorkerThreads::Ticket> threads;
    for (size_t l = 0; l < lanes.size(); l++) {
        threads.push_back(pool.start([&, l] {
            try {
                SVCaller caller(lanes[l].ctx);
                caller.min_mapq = min_mapq; caller.min_oplen = min_oplen;
                if (on_merged) caller.on_merged = [&on_merged, l](size_t k) { on_merged(l, k); };
                if (on_device) caller.on_device = [&on_device, l](size_t k) { on_device(l, k); };
                if (!lanes[l].seqs.empty()) caller.processResidentChromosomesPipelined(lanes[l].shards, lanes[l].seqs, eps, pct, calls[l], stats[l]);
                else caller.processResidentChromosomesPipelined(lanes[l].shards, seq, eps, pct, calls[l], stats[l]);
            } catch (...) { errs[l] = std::current_exception(); }
        }));
    }
    for (auto &t : threads) pool.wait(t);
    for (auto &e : errs) if (e) std::rethrow_exception(e);
}

void SVCaller::processChromosome(const std::string &chr, const csv_reads &reads, const SeqStore *seq, uint32_t depth_len, double eps,
                                 double pct, std::vector<SVCall> &chr_sv_calls, ChrStats &stats, csv_shard **keep_shard)
{
    printMessage(chr + ": CIGAR SVs...");
    csv_shard *sh = csvgpu_shard_upload(ctx, &reads, depth_len);
    if (!sh) throw std::runtime_error(std::string("processChromosome: ") + csvgpu_last_error(ctx));
    try {
        processResidentChromosome(chr, sh, seq, eps, pct, chr_sv_calls, stats);
    } catch (...) {
        csvgpu_shard_free(ctx, sh);
        throw;
    }
    if (keep_shard) *keep_shard = sh; else csvgpu_shard_free(ctx, sh);
}


namespace {
// alignment intervals of selected records of resident shards (csvgpu_aln_intervals_gather_batch): shard_of[c] is the resident shard of
// split-pass contig c
struct ShardIntervals : IntervalSource {
    ShardIntervals(csv_ctx *ctx, std::vector<csv_shard *> shard_of) : ctx(ctx), shard_of(std::move(shard_of)) {}
    void gather(const std::vector<size_t> &which, const std::vector<uint32_t> &rec, const std::vector<uint64_t> &rec_off, int32_t *ref_end, int32_t *q_start,
                int32_t *q_end) const override
    {
        std::vector<csv_shard *> sh(which.size());
        for (size_t k = 0; k < which.size(); k++) sh[k] = shard_of[which[k]];
        check(ctx, csvgpu_aln_intervals_gather_batch(ctx, (int)sh.size(), sh.data(), rec.data(), rec_off.data(), ref_end, q_start, q_end), "alignment intervals");
    }
    csv_ctx *ctx;
    std::vector<csv_shard *> shard_of;
};

// the iteration order of the reference's per-chromosome qname map from the device (csvgpu_split_order): shard_of[c] is the resident
// shard of split-pass contig c (its query-name hashes were attached when it was staged)
struct ShardOrderSource : SplitOrderSource {
    ShardOrderSource(csv_ctx *ctx, std::vector<csv_shard *> shard_of) : ctx(ctx), shard_of(std::move(shard_of)) {}
    void begin(const std::vector<size_t> &which, int min_mapq, bool complete) const override
    {
        pending.clear();
        if (which.empty() || which.size() > 32) return;                           // (more than one batch: the one-call form below)
        { const char *e = getenv("CSV_SPLIT_ONE_CALL"); if (e && *e && *e != '0') return; }      // (A/B: no head start)
        std::vector<csv_shard *> sh(which.size());
        for (size_t k = 0; k < which.size(); k++) sh[k] = shard_of[which[k]];
        // every contig of the run in this one call: the supplementary hashes are the shards' own and the whole order is queued now
        const char *e = getenv("CSV_SPLIT_NO_SELF");                                 // (A/B, tests: wait for the collected hashes)
        pending_self = complete && !(e && *e && *e != '0');
        check(ctx, pending_self ? csvgpu_split_order_begin_self(ctx, (int)sh.size(), sh.data(), (uint8_t)min_mapq)
                                : csvgpu_split_order_begin(ctx, (int)sh.size(), sh.data(), (uint8_t)min_mapq), "split-read order (begin)");
        pending = which; pending_mapq = min_mapq;
    }
    void survivors(const std::vector<size_t> &which, int min_mapq, const std::vector<uint64_t> &supp_hash, std::vector<std::vector<uint32_t>> &recs) const override
    {
        recs.assign(which.size(), {});
        if (!pending.empty() && pending == which && pending_mapq == min_mapq) {      // the head start was taken: only the last epochs are left
            pending.clear();
            const size_t nb = which.size();
            std::vector<uint64_t> off(nb + 1, 0);
            std::vector<uint32_t> out(std::max<size_t>(supp_hash.size() * 2, 1024));
            const uint64_t *sh_p = pending_self ? nullptr : supp_hash.data();
            const uint64_t sh_n = pending_self ? 0 : supp_hash.size();
            int rc = csvgpu_split_order_finish(ctx, sh_p, sh_n, out.data(), out.size(), off.data());
            if (rc == CSV_ECAPACITY) {
                out.resize(off[nb]);
                rc = csvgpu_split_order_finish(ctx, sh_p, sh_n, out.data(), out.size(), off.data());
            }
            check(ctx, rc, "split-read order");
            for (size_t k = 0; k < nb; k++) recs[k].assign(out.begin() + (std::ptrdiff_t)off[k], out.begin() + (std::ptrdiff_t)off[k + 1]);
            return;
        }
        pending.clear();
        for (size_t b0 = 0; b0 < which.size(); b0 += 32) {                       // the entry point takes up to 32 contigs per call
            const size_t nb = std::min<size_t>(32, which.size() - b0);
            std::vector<csv_shard *> sh(nb);
            for (size_t k = 0; k < nb; k++) sh[k] = shard_of[which[b0 + k]];
            std::vector<uint64_t> off(nb + 1, 0);
            std::vector<uint32_t> out(std::max<size_t>(supp_hash.size() * 2, 1024));
            int rc = csvgpu_split_order(ctx, (int)nb, sh.data(), (uint8_t)min_mapq, supp_hash.data(), supp_hash.size(), out.data(), out.size(), off.data());
            if (rc == CSV_ECAPACITY) {
                out.resize(off[nb]);
                rc = csvgpu_split_order(ctx, (int)nb, sh.data(), (uint8_t)min_mapq, supp_hash.data(), supp_hash.size(), out.data(), out.size(), off.data());
            }
            check(ctx, rc, "split-read order");
            for (size_t k = 0; k < nb; k++) recs[b0 + k].assign(out.begin() + (std::ptrdiff_t)off[k], out.begin() + (std::ptrdiff_t)off[k + 1]);
        }
    }
    csv_ctx *ctx;
    std::vector<csv_shard *> shard_of;
    mutable std::vector<size_t> pending;          // the contigs csvgpu_split_order_begin was called for, until _finish
    mutable int pending_mapq = 0;
    mutable bool pending_self = false;            // ... by csvgpu_split_order_begin_self: _finish takes no hashes
};

}  // namespace

// what the split-read pass of a run works on (built once per run; prepare() may already be running while the CIGAR pass is on the device)
struct SVCaller::SplitSetup {
    std::vector<SplitContig> blocks;
    std::vector<int> block_of;                          // per contig of the run: its index in `blocks`, or -1
    std::vector<std::string> names;
    std::unique_ptr<ShardOrderSource> dev_order;
    std::unique_ptr<ShardIntervals> intervals;
    SplitParams sp;
    std::unique_ptr<SplitPass> pass;
    double ms_prepare = 0.0;
    std::exception_ptr err;
};

namespace {
struct EmptySnps : SNPSource {
    void query(uint32_t, uint32_t, std::vector<uint32_t> &, std::unordered_map<uint32_t, double> &, std::unordered_map<uint32_t, double> &) const override {}
};
}  // namespace

namespace {
struct VectorSource : ContigSource {
    explicit VectorSource(const std::vector<ChromosomeInput> &v) : v(v) {}
    bool next(ChromosomeInput &out) override { if (i >= v.size()) return false; out = v[i++]; return true; }
    const std::vector<ChromosomeInput> &v;
    size_t i = 0;
};
}  // namespace

void SVCaller::run(const std::vector<ChromosomeInput> &contigs, const CHMM &hmm, const RunParams &P,
                   std::unordered_map<std::string, std::vector<SVCall>> &whole_genome_sv_calls)
{
    VectorSource src(contigs);
    run(src, hmm, P, whole_genome_sv_calls);
}

void SVCaller::run(ContigSource &source, const CHMM &hmm, const RunParams &P,
                   std::unordered_map<std::string, std::vector<SVCall>> &whole_genome_sv_calls)
{
    csvhost::set_context(ctx);
    // what outlives a contig's host arrays: its shard (reads, depth map, alignment intervals in HBM), its statistics, its SNP
    // source, and per record the 7 bytes + query name (hash + bytes) that the split-read pass groups by
    struct Kept {
        std::vector<int32_t> pos; std::vector<uint16_t> flag; std::vector<uint8_t> mapq;
        std::vector<uint64_t> qhash, name_off; std::string names;
    };
    std::vector<std::unique_ptr<Kept>> kept;
    std::vector<ResidentContig> contigs;
    std::vector<ChrStats> stats;
    auto free_all = [&] { for (ResidentContig &c : contigs) if (c.shard) csvgpu_shard_free(ctx, c.shard); };
    try {
        // depth pass + CIGAR pass + CIGAR merge (sv_caller.cpp:794-863); the reference pre-seeds the map with every contig
        ChromosomeInput c;
        while (source.next(c)) {
            const size_t i = contigs.size();
            contigs.emplace_back();
            stats.emplace_back();
            kept.emplace_back(new Kept());
            contigs[i].name = c.name; contigs[i].depth_len = c.depth_len; contigs[i].snps = c.snps;
            contigs[i].seq = nullptr;                                 // the sequences are only valid until the next contig: ALT strings are cut now
            contigs[i].split.tid = (int32_t)i;
            std::vector<SVCall> calls;
            if (P.cigar_svs) processChromosome(c.name, c.reads, c.seq, c.depth_len, P.dbscan_epsilon, P.dbscan_min_pts_pct, calls, stats[i], &contigs[i].shard);
            whole_genome_sv_calls[c.name] = std::move(calls);
            if (P.split_svs && c.qnames && contigs[i].shard) {                          // inputs of the split-read pass (:133-175)
                const uint64_t n = c.reads.n_reads;
                Kept &K = *kept[i];
                K.pos.assign(c.reads.pos, c.reads.pos + n); K.flag.assign(c.reads.flag, c.reads.flag + n); K.mapq.assign(c.reads.mapq, c.reads.mapq + n);
                K.qhash.resize(n); K.name_off.resize(n + 1);
                size_t bytes = 0;
                for (uint64_t r = 0; r < n; r++) bytes += (*c.qnames)[r].size();
                K.names.reserve(bytes);
                for (uint64_t r = 0; r < n; r++) {
                    const std::string &q = (*c.qnames)[r];
                    K.qhash[r] = csvhost::std_string_hash(q.data(), q.size());
                    K.name_off[r] = K.names.size();
                    K.names += q;
                }
                K.name_off[n] = K.names.size();
                SplitContig &sc = contigs[i].split;
                sc.n = n; sc.pos = K.pos.data(); sc.flag = K.flag.data(); sc.mapq = K.mapq.data();
                sc.qhash = K.qhash.data(); sc.name_bytes = K.names.data(); sc.name_off = K.name_off.data();
            }
        }
        RunStageTimes T;
        finishRun(contigs, stats, hmm, P, whole_genome_sv_calls, T);
    } catch (...) {
        free_all();
        throw;
    }
    free_all();
}

namespace {
bool env_on(const char *name) { const char *e = getenv(name); return e && *e && *e != '0'; }       // (set, not empty, not "0")
}  // namespace

void SVCaller::runResident(const std::vector<ResidentContig> &contigs_in, const std::vector<csv_ctx *> &lane_ctxs, const CHMM &hmm, const RunParams &P,
                           std::unordered_map<std::string, std::vector<SVCall>> &whole_genome_sv_calls, std::vector<ChrStats> *stats_out, RunStageTimes *times)
{
    csvhost::set_context(ctx);
    const double t_begin = now_ms();
    RunStageTimes T;
    // The caller's own context works on another thread while the lanes run the CIGAR pass (split-read prepare, csvgpu_split_order, the early
    // copy-number batches): a context is one arena + one stream + one set of timers and belongs to one thread at a time.
    for (csv_ctx *lc : lane_ctxs)
        if (lc == ctx) throw std::invalid_argument("runResident: the caller's context must not be one of the lanes (each lane needs a context of its own)");
    for (size_t a = 0; a < lane_ctxs.size(); a++)
        for (size_t b = 0; b < a; b++)
            if (lane_ctxs[a] == lane_ctxs[b]) throw std::invalid_argument("runResident: the same context given for two lanes");
    std::vector<ResidentContig> contigs = contigs_in;
    const size_t n = contigs.size();
    std::vector<ChrStats> stats(n);
    std::vector<std::vector<SVCall>> per(n);
    // contigs over the lanes: longest processing time first by read count, every lane works down its list
    const size_t L = std::max<size_t>(1, lane_ctxs.size());
    std::vector<Lane> lanes(L);
    std::vector<std::vector<size_t>> which(L);
    std::vector<std::vector<std::vector<SVCall>>> lane_calls;
    std::vector<std::vector<ChrStats>> lane_stats;
    if (P.cigar_svs && n) {
        for (size_t l = 0; l < L; l++) lanes[l].ctx = lane_ctxs.empty() ? ctx : lane_ctxs[l];
        std::vector<size_t> order(n);
        for (size_t i = 0; i < n; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return contigs[a].split.n != contigs[b].split.n ? contigs[a].split.n > contigs[b].split.n : contigs[a].depth_len != contigs[b].depth_len ? contigs[a].depth_len > contigs[b].depth_len : a < b; });
        std::vector<double> load(L, 0.0);
        for (size_t i : order) {
            size_t l = 0;
            for (size_t k = 1; k < L; k++) if (load[k] < load[l]) l = k;
            lanes[l].shards.push_back(contigs[i].shard);
            lanes[l].seqs.push_back(contigs[i].seq);
            which[l].push_back(i);
            load[l] += (double)std::max<uint64_t>(contigs[i].split.n, contigs[i].depth_len / 400);
        }
    }
    // The first half of the split-read pass needs nothing the CIGAR pass produces (flags, name hashes -> the qname map's iteration order):
    // with lanes, this caller's own context is idle during the CIGAR pass, so that half runs beside it on another thread. When it is done
    // the pass is a little over half way: the same thread then makes the CIGAR copy-number predictions of the contigs whose calls are
    // final by then (own context, the host pool nobody else uses during the pass), and only the rest waits for the end of the pass.
    struct EarlyCn {
        std::mutex mu;
        std::vector<std::pair<size_t, size_t>> merged;     // (lane, k) in the order the merge threads finished them
        std::vector<char> done, finished;                  // per contig: CIGAR copy-number predictions made; every stage of the run made
        size_t n_split_calls = 0;
        std::exception_ptr err;
        size_t regions = 0;
        bool pass_over = false;                            // (under mu) the CIGAR pass has returned: the task stops taking batches
        bool prepare_over = false;                         // (under mu) the task is past prepare()
        // A rank with few contigs (or short reads: a first half as long as the pass) takes no batch of the kind above. Its split chain needs the
        // scans' alignment intervals and, for its copy-number pass, the depth maps and mean coverages — all there when the contigs' device chains
        // are over, before their host merges: the task runs it for ALL contigs then, beside the rest of the pass and the CIGAR copy-number pass.
        std::vector<std::pair<size_t, size_t>> device_done; // (under mu) (lane, k) of the contigs whose device chain is over, in that order
        bool split_only = false;                           // (under mu) the task has decided to do that; the run meets it in front of the split chain
        bool main_joins_later = false;                     // (under mu, set with pass_over) the run has gone on without waiting for the task
        bool pre_ready = false;                            // pre_split holds every contig's split-read calls (read after the join)
        std::unordered_map<std::string, std::vector<SVCall>> pre_split;
    } early;
    early.done.assign(n, 0);
    early.finished.assign(n, 0);
    size_t n_lane_contigs = 0;
    for (size_t l = 0; l < L; l++) n_lane_contigs += which[l].size();
    const bool early_cn = P.cigar_svs && P.cigar_cn && P.split_svs && !P.save_cnv && n && lane_ctxs.size() > 1 && P.overlap_split_prepare && !env_on("CSV_NO_EARLY_CN");
    std::unique_ptr<SplitSetup> split;
    csvhost::WorkerThreads::Ticket split_task = nullptr;
    const bool forced_batches = env_on("CSV_EARLY_CN_WAIT_ALL") || env_on("CSV_EARLY_SMALL_BATCHES");       // (tests)
    if (P.split_svs) {
        split = makeSplitSetup(contigs, P);
        if (P.cigar_svs && n && lane_ctxs.size() > 1 && P.overlap_split_prepare) {
            SplitSetup *S = split.get();
            split_task = csvhost::WorkerThreads::instance().start([&, S, forced_batches] {
                const double t0 = now_ms();
                if (const char *e = getenv("CSV_TEST_PREPARE_DELAY_MS")) std::this_thread::sleep_for(std::chrono::milliseconds(atoi(e)));      // (tests: a first half that outlasts the pass)
                try { S->pass->prepare(); } catch (...) { S->err = std::current_exception(); }
                S->ms_prepare = now_ms() - t0;
                if (S->err) { std::lock_guard<std::mutex> l(early.mu); early.prepare_over = true; return; }
                // the split chain + its copy-number pass for every contig, on this thread and the caller's context (see EarlyCn::split_only)
                auto split_only_batch = [&] {
                    try {
                        static const EmptySnps no_snps;
                        // whatever is ready goes now, the rest as it comes (a rank of six contigs has two ready when the first half is over and
                        // the last one when the pass ends: what runs behind the pass is that one's chain, not all six)
                        size_t taken_b = 0;
                        for (;;) {
                            std::vector<std::pair<size_t, size_t>> snap;
                            bool over;
                            { std::lock_guard<std::mutex> l(early.mu); snap.assign(early.device_done.begin() + (std::ptrdiff_t)taken_b, early.device_done.end()); over = early.pass_over; }
                            if (snap.empty()) {
                                if (taken_b >= n_lane_contigs || over) break;          // (over with contigs missing: the pass failed)
                                std::this_thread::sleep_for(std::chrono::microseconds(50));
                                continue;
                            }
                            taken_b += snap.size();
                            csvhost::TraceScope tr("run: split chain beside the pass");
                            csvhost::set_thread_context(ctx);
                            std::vector<size_t> blocks;
                            std::unordered_map<std::string, std::pair<size_t, size_t>> lane_of;
                            for (const auto &lk : snap) {
                                if (early.finished[which[lk.first][lk.second]]) continue;      // (went through an early batch: its split calls are merged in)
                                const int b = S->block_of[which[lk.first][lk.second]];
                                if (b >= 0) blocks.push_back((size_t)b);
                                lane_of[contigs[which[lk.first][lk.second]].name] = lk;
                            }
                            if (blocks.empty()) { csvhost::set_thread_context(nullptr); continue; }
                            std::unordered_map<std::string, std::vector<SVCall>> part;
                            S->pass->finishFor(blocks, part);
                            std::vector<CNVCaller::ContigJob> sj;
                            for (auto &entry : part) {
                                if (entry.second.empty()) continue;
                                const auto lk = lane_of.at(entry.first);
                                const size_t i = which[lk.first][lk.second];
                                CNVCaller::ContigJob j;
                                j.chr = entry.first; j.calls = &entry.second; j.mean_chr_cov = lane_stats[lk.first][lk.second].mean_chr_cov; j.shard = contigs[i].shard;
                                j.snps = contigs[i].snps ? contigs[i].snps : (const SNPSource *)&no_snps; j.depth_len = contigs[i].depth_len;
                                sj.push_back(j);
                            }
                            CNVCaller cn(ctx);
                            cn.sample_size = P.sample_size; cn.min_cnv_length = P.min_cnv_length; cn.host_threads = P.host_threads;
                            cn.runSplitReadCopyNumberPredictionsAll(sj, hmm);
                            for (auto &entry : part) early.pre_split[entry.first] = std::move(entry.second);
                            csvhost::set_thread_context(nullptr);
                        }
                        early.pre_ready = taken_b >= n_lane_contigs;
                    } catch (...) { early.err = std::current_exception(); csvhost::set_thread_context(nullptr); }
                };
                const bool can_split_only = !P.save_cnv && !forced_batches && !env_on("CSV_NO_SPLIT_BESIDE_PASS");
                {   // prepare() outlasted the CIGAR pass (short reads): the run has gone on without waiting and no batch of the first kind may be taken any more
                    std::unique_lock<std::mutex> l(early.mu);
                    early.prepare_over = true;
                    if (early.pass_over && !forced_batches) {
                        const bool go = early.main_joins_later && can_split_only;
                        l.unlock();
                        if (go) split_only_batch();
                        return;
                    }
                }
                if (!early_cn) {
                    bool go = false;
                    { std::lock_guard<std::mutex> l(early.mu); if (can_split_only && !early.pass_over) { early.split_only = true; go = true; } }
                    if (go) split_only_batch();
                    return;
                }
                try {
                    if (env_on("CSV_EARLY_CN_WAIT_ALL")) {                      // tests: every contig through this path, whatever the timing
                        for (int spin = 0; spin < 200000; spin++) {
                            { std::lock_guard<std::mutex> l(early.mu); if (early.merged.size() >= n_lane_contigs) break; }
                            std::this_thread::sleep_for(std::chrono::microseconds(50));
                        }
                    }
                    // One batch now (a little over half the genome is merged), then another whenever a few more contigs are: what is left
                    // for the end of the pass is the last contigs' share. The pass's end (`pass_over`) ends the loop; whatever was not
                    // taken here is done behind the pass as before.
                    size_t taken = 0;
                    for (bool first = true;; first = false) {
                        std::vector<std::pair<size_t, size_t>> snap;
                        bool over;
                        size_t unmerged;
                        { std::lock_guard<std::mutex> l(early.mu); snap.assign(early.merged.begin() + (std::ptrdiff_t)taken, early.merged.end()); over = early.pass_over;
                          unmerged = n_lane_contigs - early.merged.size(); }
                        // (a batch takes a few milliseconds beside the pass: with fewer than eight contigs still to come the pass could be
                        // over first and the run would wait for the batch — those go with the rest, behind the pass; a rank with a handful of
                        // contigs never takes one)
                        // (CSV_EARLY_SMALL_BATCHES: tests — a batch whenever three more contigs are merged, down to the last one)
                        const bool small_batches = env_on("CSV_EARLY_SMALL_BATCHES");
                        if (unmerged < 8 && !env_on("CSV_EARLY_CN_WAIT_ALL") && !small_batches) {
                            if (can_split_only) {                                       // no (more) batches of this kind: the split chain alone, for every contig that went through none
                                bool go = false;
                                { std::lock_guard<std::mutex> l(early.mu); if (!early.pass_over) { early.split_only = true; go = true; } }
                                if (go) split_only_batch();
                            }
                            break;
                        }
                        if (small_batches && first && snap.size() < 3 && !over) { std::this_thread::sleep_for(std::chrono::microseconds(100)); continue; }
                        if (!first && (over || snap.size() < 3)) {
                            if (over) break;
                            std::this_thread::sleep_for(std::chrono::microseconds(100));
                            continue;
                        }
                        taken += snap.size();
                        static const EmptySnps no_snps;
                        std::vector<CNVCaller::ContigJob> jobs;
                        for (const auto &lk : snap) {
                            const size_t i = which[lk.first][lk.second];
                            std::vector<SVCall> &v = lane_calls[lk.first][lk.second];
                            if (v.empty()) continue;
                            CNVCaller::ContigJob j;
                            j.chr = contigs[i].name; j.calls = &v; j.mean_chr_cov = lane_stats[lk.first][lk.second].mean_chr_cov; j.shard = contigs[i].shard;
                            j.snps = contigs[i].snps ? contigs[i].snps : (const SNPSource *)&no_snps; j.depth_len = contigs[i].depth_len;
                            jobs.push_back(j);
                        }
                        csvhost::TraceScope tr(jobs.empty() ? "cn: early batch (nothing merged yet)" : "cn: early batch");
                        if (!jobs.empty()) {
                            csvhost::set_thread_context(ctx);
                            CNVCaller cn(ctx);
                            cn.sample_size = P.sample_size; cn.min_cnv_length = P.min_cnv_length; cn.host_threads = P.host_threads;
                            early.regions += cn.runCIGARCopyNumberPredictionAll(jobs, hmm);
                            csvhost::set_thread_context(nullptr);
                        }
                        for (const auto &lk : snap) early.done[which[lk.first][lk.second]] = 1;
                        // ... and everything else the run does with these contigs (:885-927): nothing in it reaches across contigs, so the
                        // same functions run on this batch's contigs now and on the rest behind the pass
                        if (!env_on("CSV_NO_EARLY_SPLIT")) {
                            csvhost::TraceScope tr2("run: early split chain + merges");
                            csvhost::set_thread_context(ctx);
                            std::vector<size_t> blocks;                                // (merged => scanned: their alignment intervals exist)
                            for (const auto &lk : snap) { const int b = S->block_of[which[lk.first][lk.second]]; if (b >= 0) blocks.push_back((size_t)b); }
                            std::unordered_map<std::string, std::vector<SVCall>> split_calls;
                            S->pass->finishFor(blocks, split_calls);
                            std::unordered_map<std::string, std::pair<size_t, size_t>> lane_of;     // contig name -> (lane, k)
                            for (const auto &lk : snap) lane_of[contigs[which[lk.first][lk.second]].name] = lk;
                            {
                                std::vector<CNVCaller::ContigJob> sj;
                                for (auto &entry : split_calls) {
                                    if (entry.second.empty()) continue;
                                    const auto lk = lane_of.at(entry.first);
                                    const size_t i = which[lk.first][lk.second];
                                    CNVCaller::ContigJob j;
                                    j.chr = entry.first; j.calls = &entry.second; j.mean_chr_cov = lane_stats[lk.first][lk.second].mean_chr_cov; j.shard = contigs[i].shard;
                                    j.snps = contigs[i].snps ? contigs[i].snps : (const SNPSource *)&no_snps; j.depth_len = contigs[i].depth_len;
                                    sj.push_back(j);
                                }
                                CNVCaller cn(ctx);
                                cn.sample_size = P.sample_size; cn.min_cnv_length = P.min_cnv_length; cn.host_threads = P.host_threads;
                                cn.runSplitReadCopyNumberPredictionsAll(sj, hmm);
                            }
                            if (P.merge_split_svs) {
                                std::vector<std::vector<SVCall> *> sets;
                                for (auto &entry : split_calls) sets.push_back(&entry.second);
                                mergeSVsMany(sets, 0.1, 2, true, P.host_threads);
                            }
                            for (auto &entry : split_calls) {
                                const auto lk = lane_of.at(entry.first);
                                early.n_split_calls += entry.second.size();
                                std::vector<SVCall> &dst = lane_calls[lk.first][lk.second];
                                dst.insert(dst.end(), entry.second.begin(), entry.second.end());
                            }
                            if (P.merge_final_svs) {
                                std::vector<std::vector<SVCall> *> sets;
                                for (const auto &lk : snap) sets.push_back(&lane_calls[lk.first][lk.second]);
                                mergeSVsMany(sets, 0.1, 2, true, P.host_threads);
                            }
                            for (const auto &lk : snap) early.finished[which[lk.first][lk.second]] = 1;
                            csvhost::set_thread_context(nullptr);
                        }
                        if (env_on("CSV_EARLY_CN_WAIT_ALL") || env_on("CSV_EARLY_ONE_BATCH")) break;
                    }
                } catch (...) { early.err = std::current_exception(); csvhost::set_thread_context(nullptr); }
            });
        }
    }
    struct JoinSplit {                                                  // the task refers to this frame: never leave it with the task running
        csvhost::WorkerThreads::Ticket &t;
        ~JoinSplit() { if (t) { csvhost::WorkerThreads::instance().wait(t); t = nullptr; } }
    } join_split{split_task};
    struct EndPass {                                                    // (declared after join_split: runs before it, also when the pass throws)
        EarlyCn &e;
        ~EndPass() { std::lock_guard<std::mutex> l(e.mu); e.pass_over = true; }
    } end_pass{early};
    bool join_later = false;
    if (P.cigar_svs && n) {
        csvhost::TraceScope tr_pass("run: CIGAR pass");
        if (L == 1) {
            lane_calls.resize(1); lane_stats.resize(1);
            SVCaller one(lanes[0].ctx);
            one.min_mapq = min_mapq; one.min_oplen = min_oplen;
            one.processResidentChromosomesPipelined(lanes[0].shards, lanes[0].seqs, P.dbscan_epsilon, P.dbscan_min_pts_pct, lane_calls[0], lane_stats[0]);
        } else {
            std::function<void(size_t, size_t)> note;
            if (early_cn) note = [&early](size_t l, size_t k) { std::lock_guard<std::mutex> g(early.mu); early.merged.emplace_back(l, k); };
            std::function<void(size_t, size_t)> dev_note;
            if (split_task) dev_note = [&early](size_t l, size_t k) { std::lock_guard<std::mutex> g(early.mu); early.device_done.emplace_back(l, k); };
            processResidentLanes(lanes, nullptr, P.dbscan_epsilon, P.dbscan_min_pts_pct, lane_calls, lane_stats, note, min_mapq, min_oplen, dev_note);
        }
        T.ms_cigar = now_ms() - t_begin;
        // (the early copy-number batch works on lane_calls in place: it must be over before they move. A task still inside prepare()
        // takes no batch after the pass: the run goes on — the CIGAR copy-number pass needs nothing of prepare() — and meets the task in
        // front of the split chain.)
        { std::lock_guard<std::mutex> l(early.mu); early.pass_over = true;
          join_later = split_task && (!early.prepare_over || early.split_only) && !forced_batches && !env_on("CSV_NO_LATE_JOIN");
          early.main_joins_later = join_later; }
        if (split_task && !join_later) { csvhost::WorkerThreads::instance().wait(split_task); split_task = nullptr; }
        for (size_t l = 0; l < L; l++)
            for (size_t k = 0; k < which[l].size(); k++) { per[which[l][k]] = std::move(lane_calls[l][k]); stats[which[l][k]] = lane_stats[l][k]; }
    } else T.ms_cigar = now_ms() - t_begin;
    for (size_t i = 0; i < n; i++) {                                   // the map is filled in contig order, as run() does
        T.n_signatures += stats[i].n_signatures; T.n_cigar_calls += per[i].size(); T.n_reads += contigs[i].split.n;
        whole_genome_sv_calls[contigs[i].name] = std::move(per[i]);
    }
    const std::function<void()> join = [&] {
        if (split_task) { csvhost::WorkerThreads::instance().wait(split_task); split_task = nullptr; }
        if (split && split->err) std::rethrow_exception(split->err);
        if (early.err) std::rethrow_exception(early.err);
    };
    if (!join_later) join();
    T.n_cigar_cn_regions += early.regions;                                  // (a task joined later has taken no batch)
    T.n_split_calls += early.n_split_calls;
    finishRun(contigs, stats, hmm, P, whole_genome_sv_calls, T, split.get(), lane_ctxs.size() > 1 ? lane_ctxs[0] : nullptr, early_cn ? &early.done : nullptr,
              early_cn ? &early.finished : nullptr, join_later ? &join : nullptr, &early.pre_split, &early.pre_ready);
    T.ms_total = now_ms() - t_begin;
    if (stats_out) *stats_out = stats;
    if (times) *times = T;
}

std::unique_ptr<SVCaller::SplitSetup> SVCaller::makeSplitSetup(std::vector<ResidentContig> &contigs, const RunParams &P)
{
    std::unique_ptr<SplitSetup> S(new SplitSetup());
    std::vector<csv_shard *> shard_of;
    for (size_t i = 0; i < contigs.size(); i++) {
        ResidentContig &c = contigs[i];
        S->names.push_back(c.name);
        S->block_of.push_back(-1);
        if (!c.split.qhash || !c.shard || !c.split.n) continue;
        S->block_of.back() = (int)S->blocks.size();
        c.split.tid = (int32_t)i;
        c.split.ref_end = c.split.q_start = c.split.q_end = nullptr;     // the scan kernel's per-read intervals (the reference's third BAM pass, :137-172) stay in the shards
        S->blocks.push_back(c.split);
        shard_of.push_back(c.shard);
    }
    S->sp.min_mapq = min_mapq; S->sp.threads = P.host_threads;
    S->dev_order.reset(new ShardOrderSource(ctx, shard_of));
    S->intervals.reset(new ShardIntervals(ctx, shard_of));
    S->sp.intervals = S->intervals.get();
    if (P.split_order_on_device) S->sp.device_order = S->dev_order.get();          // only contigs staged with unique_names take it
    S->pass.reset(new SplitPass(S->blocks, S->names, S->sp));
    return S;
}

void SVCaller::finishRun(std::vector<ResidentContig> &contigs, const std::vector<ChrStats> &stats, const CHMM &hmm, const RunParams &P,
                         std::unordered_map<std::string, std::vector<SVCall>> &whole_genome_sv_calls, RunStageTimes &T, SplitSetup *split, csv_ctx *side_ctx,
                         const std::vector<char> *cigar_cn_done, const std::vector<char> *finished, const std::function<void()> *before_split,
                         std::unordered_map<std::string, std::vector<SVCall>> *pre_split, const bool *pre_split_ready)
{
    const EmptySnps no_snps;
    csvhost::WorkerThreads::Ticket teardown = nullptr;
    struct JoinTeardown {
        csvhost::WorkerThreads::Ticket &t;
        ~JoinTeardown() { if (t) csvhost::WorkerThreads::instance().wait(t); }
    } join_teardown{teardown};
    std::unordered_map<std::string, size_t> index_of;
    std::vector<std::string> names;
    for (size_t i = 0; i < contigs.size(); i++) { index_of[contigs[i].name] = i; names.push_back(contigs[i].name); }
    CNVCaller cnv(ctx);
    cnv.sample_size = P.sample_size; cnv.min_cnv_length = P.min_cnv_length; cnv.host_threads = P.host_threads;
    if (P.save_cnv && !P.vcf.output_dir.empty()) {                                 // main.cpp:109-118
        cnv.save_cnv_data = true;
        cnv.cnv_output_file = P.vcf.output_dir + "/CNVCalls.json";
        std::remove(cnv.cnv_output_file.c_str());
        printMessage("Saving CNV data to: " + cnv.cnv_output_file);
    }
    double t0 = now_ms();
    auto cn_jobs = [&](std::unordered_map<std::string, std::vector<SVCall>> &m, const std::vector<char> *skip = nullptr) {
        std::vector<CNVCaller::ContigJob> jobs;
        for (auto &entry : m) {                                                      // the map's own order, as the reference walks it
            if (entry.second.empty()) continue;
            const size_t i = index_of.at(entry.first);
            if (skip && (*skip)[i]) continue;                                            // (predicted while the CIGAR pass was still running)
            CNVCaller::ContigJob j;
            j.chr = entry.first; j.calls = &entry.second; j.mean_chr_cov = stats[i].mean_chr_cov; j.shard = contigs[i].shard;
            j.snps = contigs[i].snps ? contigs[i].snps : (const SNPSource *)&no_snps; j.depth_len = contigs[i].depth_len;
            jobs.push_back(j);
        }
        return jobs;
    };
    // The CIGAR copy-number pass (:865-881) and the split-read chain (:885-917) do not depend on each other: with a second context
    // at hand (`side_ctx`: a lane's, idle by now) the former runs on another thread — its own context, its own host pool — while
    // this thread goes on with the latter; they meet in front of the final merge.
    csvhost::WorkerThreads::Ticket cn_task = nullptr;
    std::exception_ptr cn_err;
    struct JoinCn {
        csvhost::WorkerThreads::Ticket &t;
        ~JoinCn() { if (t) csvhost::WorkerThreads::instance().wait(t); }
    } join_cn{cn_task};
    if (P.cigar_svs && P.cigar_cn) {
        printMessage("Running copy number predictions on CIGAR SVs...");
        if (side_ctx && side_ctx != ctx && P.split_svs && !cnv.save_cnv_data) {
            cn_task = csvhost::WorkerThreads::instance().start([&, t0] {
                try {
                    csvhost::HostPool::second_pool_flag() = true;
                    csvhost::set_thread_context(side_ctx);
                    CNVCaller side(side_ctx);
                    side.sample_size = cnv.sample_size; side.min_cnv_length = cnv.min_cnv_length; side.host_threads = cnv.host_threads;
                    std::vector<CNVCaller::ContigJob> jobs = cn_jobs(whole_genome_sv_calls, cigar_cn_done);
                    T.n_cigar_cn_regions += side.runCIGARCopyNumberPredictionAll(jobs, hmm);
                } catch (...) { cn_err = std::current_exception(); }
                csvhost::set_thread_context(nullptr);
                csvhost::HostPool::second_pool_flag() = false;
                T.ms_cigar_cn = now_ms() - t0;
            });
        } else {
            std::vector<CNVCaller::ContigJob> jobs = cn_jobs(whole_genome_sv_calls, cigar_cn_done);
            T.n_cigar_cn_regions += cnv.runCIGARCopyNumberPredictionAll(jobs, hmm);
            T.ms_cigar_cn = now_ms() - t0;
        }
    }
    if (P.split_svs) {                                                             // :885-917
        t0 = now_ms();
        std::unique_ptr<SplitSetup> own;
        if (!split) { own = makeSplitSetup(contigs, P); split = own.get(); }
        T.ms_split_fetch = 0.0;
        std::unordered_map<std::string, std::vector<SVCall>> split_calls;
        if (before_split) (*before_split)();                                       // (prepare() still running on its thread: meet it here)
        const bool pre = pre_split && pre_split_ready && *pre_split_ready;         // (the split chain and its copy-number pass ran beside the CIGAR pass)
        if (pre) split_calls = std::move(*pre_split);
        else split->pass->finish(split_calls);                                     // (runs prepare() first when nobody has)
        T.ms_split_prepare = split->ms_prepare;
        {   // the pass's working set (1e5 small vectors for a genome) is torn down beside the next stages, not between them
            std::shared_ptr<SplitPass> dead(split->pass.release());
            teardown = csvhost::WorkerThreads::instance().start([dead]() mutable { dead.reset(); });
        }
        T.ms_split = now_ms() - t0;
        t0 = now_ms();
        if (!pre) {
            std::vector<CNVCaller::ContigJob> jobs = cn_jobs(split_calls);
            cnv.runSplitReadCopyNumberPredictionsAll(jobs, hmm);
        }
        T.ms_split_cn = now_ms() - t0;
        t0 = now_ms();
        if (P.merge_split_svs) {
            std::vector<std::vector<SVCall> *> sets;
            for (auto &entry : split_calls) sets.push_back(&entry.second);
            mergeSVsMany(sets, 0.1, 2, true, P.host_threads);
        }
        if (cn_task) { csvhost::WorkerThreads::instance().wait(cn_task); cn_task = nullptr; }          // the CIGAR calls are final from here on
        if (cn_err) std::rethrow_exception(cn_err);
        for (auto &entry : split_calls) {
            T.n_split_calls += entry.second.size();
            std::vector<SVCall> &dst = whole_genome_sv_calls[entry.first];
            dst.insert(dst.end(), entry.second.begin(), entry.second.end());
        }
        T.ms_merge_split = now_ms() - t0;
    }
    if (cn_task) { csvhost::WorkerThreads::instance().wait(cn_task); cn_task = nullptr; }
    if (cn_err) std::rethrow_exception(cn_err);
    t0 = now_ms();
    if (P.merge_final_svs) {                                                                                 // :919-927
        std::vector<std::vector<SVCall> *> sets;
        for (auto &entry : whole_genome_sv_calls) {
            if (finished && (*finished)[index_of.at(entry.first)]) continue;                                 // (merged while the CIGAR pass was still running)
            sets.push_back(&entry.second);
        }
        mergeSVsMany(sets, 0.1, 2, true, P.host_threads);
    }
    T.ms_merge_final = now_ms() - t0;
    if (cnv.save_cnv_data) CNVCaller::closeJSON(cnv.cnv_output_file);                                       // :929-931
    uint32_t total = 0;
    for (const auto &entry : whole_genome_sv_calls) {
        total += getSVCount(entry.second);
        printMessage("Total SVs detected for " + entry.first + ": " + std::to_string(getSVCount(entry.second)));
    }
    T.n_final_calls = total;
    printMessage("Total SVs detected: " + std::to_string(total));
    if (P.ref_genome && !P.vcf.output_dir.empty()) {                               // :943-945
        t0 = now_ms();
        printMessage("Saving SVs to VCF...");
        ShardDepthSource depth(ctx);
        for (const ResidentContig &c : contigs) if (c.shard) depth.add(c.name, c.shard);
        saveToVCF(whole_genome_sv_calls, P.vcf, *P.ref_genome, depth);
        T.ms_vcf = now_ms() - t0;
    }
}

// ---- the run fed from a BAM file ----------------------------------------------------------------------------------
#include <future>

#include "bam_io.h"
#include "snp_io.h"

namespace {
// Contigs decoded one ahead: while run() has contig i on the device, the inflate pool works on contig i + 1.
struct BamSource : ContigSource {
    BamReader reader;
    std::vector<std::string> chrs;
    BamReadOptions opt;
    BamShard cur, ahead;
    std::future<bool> pending;
    SeqStore seq;
    size_t i = 0;
    BamRunStats st;
    SNPFile *snp_file = nullptr;                                  // loaded SNP VCF, or nullptr
    const AlleleFreqFiles *af_files = nullptr;
    std::string ethnicity;

    void launch(size_t k) { pending = std::async(std::launch::async, [this, k] { return reader.readContig(chrs[k], opt, ahead); }); }

    bool next(ChromosomeInput &out) override
    {
        if (i >= chrs.size()) return false;
        const double t0 = now_ms();
        if (!pending.valid()) launch(i);
        const bool ok = pending.get();
        st.ms_decode += now_ms() - t0;
        if (!ok) throw std::runtime_error(reader.error());
        std::swap(cur, ahead);
        if (i + 1 < chrs.size()) launch(i + 1);
        seq.seq_off = cur.seq_off.data();
        seq.seq = cur.seq.data();
        out = ChromosomeInput();
        out.name = cur.name;
        out.reads = cur.view();
        out.seq = opt.want_seq ? &seq : nullptr;
        out.depth_len = cur.target_len + 1;                       // cnv_caller.cpp:482
        out.qnames = opt.want_qnames ? &cur.qnames : nullptr;
        if (snp_file) out.snps = &snp_file->table(cur.name, af_files ? af_files->get(cur.name) : std::string(), ethnicity, opt.threads);
        st.n_contigs++; st.n_reads += cur.n_reads(); st.n_cigar += cur.cigar.size();
        i++;
        return true;
    }
    ~BamSource() override { if (pending.valid()) pending.wait(); }
};
}  // namespace

void SVCaller::runBam(const std::string &bam_path, const std::vector<std::string> &chromosomes, int threads, const CHMM &hmm, const RunParams &P,
                      std::unordered_map<std::string, std::vector<SVCall>> &whole_genome_sv_calls, BamRunStats *bam_stats)
{
    const double t0 = now_ms();
    BamSource src;
    if (!src.reader.open(bam_path)) throw std::runtime_error("ERROR failed to open BAM file " + bam_path + ": " + src.reader.error());
    if (!src.reader.loadIndex()) throw std::runtime_error("ERROR failed to load index for " + bam_path);
    src.chrs = chromosomes.empty() ? src.reader.header().names : chromosomes;      // sv_caller.cpp:766-774
    for (const std::string &chr : src.chrs) {
        if (src.reader.header().tid(chr) < 0) throw std::runtime_error("ERROR: Could not find chromosome " + chr + " in BAM file.");
        if (P.ref_genome && P.ref_genome->getChromosomeLength(chr) == 0) {        // :795-800: the reference gives up on the whole run
            printError("Chromosome " + chr + " not found in reference genome");
            return;
        }
    }
    src.opt.threads = std::max(1, threads);
    SNPFile snp_file;
    AlleleFreqFiles af_files;
    if (!P.snp_vcf.empty()) {
        std::string e;
        if (!af_files.load(P.pfb_table, &e)) throw std::runtime_error(e);          // the reference exits (input_data.cpp:221-225, :278-283)
        if (snp_file.load(P.snp_vcf, src.opt.threads, &e)) {
            src.snp_file = &snp_file; src.af_files = &af_files; src.ethnicity = P.ethnicity;
        } else {
            printError(e);                                                         // as there: every region then has no SNPs (cnv_caller.cpp:586-591)
        }
    }
    src.opt.want_seq = true;                                      // 50-bp insertion ALT strings (sv_caller.cpp:589-600)
    src.opt.want_qnames = P.split_svs;
    run(src, hmm, P, whole_genome_sv_calls);
    if (bam_stats) {
        *bam_stats = src.st;
        bam_stats->bam_bytes = src.reader.bytes();
        bam_stats->ms_total = now_ms() - t0;
    }
}
