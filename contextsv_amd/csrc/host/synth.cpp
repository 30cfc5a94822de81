#include "synth.h"

#include <algorithm>
#include <cmath>
#include <numeric>
#include <random>
#include <thread>

namespace {

enum : uint32_t { M = 0, I = 1, D = 2, S = 4 };
inline uint32_t cg(uint32_t op, uint32_t len) { return (len << 4) | op; }

struct TruthSV {
    uint32_t pos, len;
    uint8_t type;      // 0 DEL, 1 INS, 2 split-only (emitted as primary + supplementary)
    uint8_t hom;       // 1: on both haplotypes
    uint8_t hap;       // haplotype carrying a het SV
};

inline uint64_t mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

struct ReadRec {
    int32_t pos; uint16_t flag; uint8_t mapq;
    uint64_t c0, c1;     // range in the thread-local cigar vector
    uint32_t thread;
    uint32_t qid;        // the read this record belongs to (a primary and its supplementary share it)
};

struct ThreadOut {
    std::vector<ReadRec> recs;
    std::vector<uint32_t> cigar;
};

// geometric(p) on {1,2,...} by inversion
inline uint32_t geo(std::mt19937_64 &g, double log1mp)
{
    const double u = (double)((g() >> 11) + 1) * (1.0 / 9007199254740993.0);
    const double v = std::floor(std::log(u) / log1mp) + 1.0;
    return v > 1e9 ? 1000000000u : (uint32_t)v;
}
inline double unif(std::mt19937_64 &g) { return (double)(g() >> 11) * (1.0 / 9007199254740992.0); }

}  // namespace

csv_reads SynthShard::view() const
{
    csv_reads r;
    r.n_reads = pos.size(); r.n_cigar = cigar.size();
    r.pos = pos.data(); r.flag = flag.data(); r.mapq = mapq.data(); r.tid = tid.data();
    r.cigar_off = cigar_off.data(); r.cigar = cigar.data();
    return r;
}

void synth_generate(const SynthParams &p, SynthShard &out)
{
    std::mt19937_64 g0(p.seed);
    // truth SVs
    std::vector<TruthSV> svs((size_t)std::max(1.0, p.chr_len * p.sv_per_bp));
    for (TruthSV &v : svs) {
        v.pos = 1000 + (uint32_t)(unif(g0) * (p.chr_len > 4000 ? p.chr_len - 4000 : 1));
        const double u = unif(g0);
        v.type = u < 0.45 ? 0 : (u < 0.90 ? 1 : 2);
        const double hi = unif(g0) < 0.01 ? 1e6 : 1e4;
        v.len = (uint32_t)std::exp(std::log(50.0) + unif(g0) * (std::log(hi) - std::log(50.0)));
        if (v.type == 2 && v.len < 2500) v.len += 2500;
        v.hom = unif(g0) < 1.0 / 3.0;
        v.hap = (uint8_t)(g0() & 1);
    }
    std::sort(svs.begin(), svs.end(), [](const TruthSV &a, const TruthSV &b) { return a.pos < b.pos; });
    out.n_truth_sv = svs.size();

    const bool ont = p.tech == 0;
    const double mean_len = ont ? std::exp(9.2 + 0.18) : 18000.0;
    const uint64_t n_reads = (uint64_t)std::max(1.0, p.depth * (double)p.chr_len / mean_len);
    std::vector<uint32_t> starts(n_reads);
    for (uint64_t r = 0; r < n_reads; r++) starts[r] = (uint32_t)(unif(g0) * (double)(p.chr_len > 2000 ? p.chr_len - 1000 : 1));
    std::sort(starts.begin(), starts.end());

    const int T = std::max(1, p.threads);
    std::vector<ThreadOut> outs(T);
    const double ev_rate = ont ? 0.05 : 0.001;
    const double log1m_ev = std::log(1.0 - ev_rate);
    const double log1m_len = std::log(1.0 - 1.0 / 1.6);
    auto worker = [&](int t) {
        ThreadOut &o = outs[t];
        const uint64_t r0 = n_reads * t / T, r1 = n_reads * (t + 1) / T;
        o.recs.reserve((r1 - r0) + (r1 - r0) / 20);
        std::normal_distribution<double> nd(0.0, 1.0);
        for (uint64_t r = r0; r < r1; r++) {
            std::mt19937_64 g(mix(p.seed ^ (r * 0x9E3779B97F4A7C15ull)));
            uint32_t len;
            if (ont) len = (uint32_t)std::min(200000.0, std::max(1000.0, std::exp(9.2 + 0.6 * nd(g))));
            else len = (uint32_t)std::min(30000.0, std::max(5000.0, 18000.0 + 3000.0 * nd(g)));
            const uint32_t start = starts[r];
            if ((uint64_t)start + len > p.chr_len) len = p.chr_len - start;
            if (len < 100) len = 100;
            const uint8_t hap = (uint8_t)(g() & 1);
            uint16_t flag = (g() & 1) ? 0x10 : 0;
            const double uf = unif(g);
            if (uf < 0.01) flag |= 0x100; else if (uf < 0.015) flag |= 0x400;
            const uint8_t mq = unif(g) < 0.05 ? (uint8_t)(g() % 20) : 60;
            const double clip_p = ont ? 0.3 : 0.1;

            ReadRec rec; rec.pos = (int32_t)start; rec.flag = flag; rec.mapq = mq; rec.thread = (uint32_t)t; rec.qid = (uint32_t)r; rec.c0 = o.cigar.size();
            if (unif(g) < clip_p) o.cigar.push_back(cg(S, 10 + (uint32_t)(g() % 291)));
            uint32_t ref = start, end = start + len;
            size_t k = std::lower_bound(svs.begin(), svs.end(), ref, [](const TruthSV &a, uint32_t x) { return a.pos < x; }) - svs.begin();
            bool split_done = false;
            uint32_t pending_m = 0;
            while (ref < end) {
                // next small event
                uint32_t gap = geo(g, log1m_ev);
                uint32_t next_ev = ref + gap;
                // next truth SV carried by this read
                while (k < svs.size() && (svs[k].pos < ref || !(svs[k].hom || svs[k].hap == hap))) k++;
                const uint32_t sv_at = k < svs.size() ? (uint32_t)((int64_t)svs[k].pos + (int64_t)(g() % 11) - 5) : 0xffffffffu;
                if (sv_at <= next_ev && sv_at < end && sv_at > ref) {
                    pending_m += sv_at - ref; ref = sv_at;
                    if (pending_m) { o.cigar.push_back(cg(M, pending_m)); pending_m = 0; }
                    const TruthSV &v = svs[k++];
                    const uint32_t vlen = std::max<uint32_t>(1, (uint32_t)((double)v.len * (0.98 + 0.04 * unif(g))));
                    if (v.type == 2 || vlen >= 10000) {
                        // split alignment: primary stops here with the rest soft-clipped, a supplementary record carries the rest
                        const uint32_t rest = end - ref;
                        if (rest >= 200) {
                            o.cigar.push_back(cg(S, rest));
                            rec.c1 = o.cigar.size(); o.recs.push_back(rec);
                            ReadRec sup; sup.thread = (uint32_t)t; sup.qid = (uint32_t)r; sup.flag = (uint16_t)(flag | 0x800); sup.mapq = mq;
                            const uint32_t jump = v.type == 0 ? vlen : (v.type == 2 ? vlen : 0);
                            sup.pos = (int32_t)std::min<uint64_t>((uint64_t)ref + jump, p.chr_len > 300 ? p.chr_len - 300 : 0);
                            sup.c0 = o.cigar.size();
                            o.cigar.push_back(cg(5 /*H*/, (uint32_t)(ref - start) + 1));
                            o.cigar.push_back(cg(M, std::min<uint32_t>(rest, p.chr_len - (uint32_t)sup.pos)));
                            sup.c1 = o.cigar.size(); o.recs.push_back(sup);
                            split_done = true;
                            break;
                        }
                    } else if (v.type == 0) {
                        o.cigar.push_back(cg(D, vlen)); ref += vlen;
                    } else {
                        o.cigar.push_back(cg(I, vlen));
                    }
                    continue;
                }
                if (next_ev >= end) { pending_m += end - ref; ref = end; break; }
                pending_m += next_ev - ref; ref = next_ev;
                o.cigar.push_back(cg(M, pending_m)); pending_m = 0;
                const uint32_t elen = geo(g, log1m_len);
                if (unif(g) < 0.4) o.cigar.push_back(cg(I, elen));
                else { o.cigar.push_back(cg(D, elen)); ref += elen; }
            }
            if (split_done) continue;
            if (pending_m) o.cigar.push_back(cg(M, pending_m));
            if (unif(g) < clip_p) o.cigar.push_back(cg(S, 10 + (uint32_t)(g() % 291)));
            rec.c1 = o.cigar.size();
            o.recs.push_back(rec);
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) th.emplace_back(worker, t);
    for (auto &x : th) x.join();

    // merge in coordinate order (stable: generation order breaks ties, like a coordinate-sorted BAM)
    std::vector<ReadRec> all;
    size_t tot = 0, tot_c = 0;
    for (auto &o : outs) { tot += o.recs.size(); tot_c += o.cigar.size(); }
    all.reserve(tot);
    for (auto &o : outs) all.insert(all.end(), o.recs.begin(), o.recs.end());
    std::stable_sort(all.begin(), all.end(), [](const ReadRec &a, const ReadRec &b) { return a.pos < b.pos; });
    out.pos.resize(tot); out.flag.resize(tot); out.mapq.resize(tot); out.tid.assign(tot, 0);
    out.cigar_off.resize(tot + 1); out.cigar.resize(tot_c); out.qname_id.resize(tot);
    uint64_t w = 0;
    for (size_t i = 0; i < tot; i++) {
        const ReadRec &r = all[i];
        out.pos[i] = r.pos; out.flag[i] = r.flag; out.mapq[i] = r.mapq; out.cigar_off[i] = w; out.qname_id[i] = r.qid;
        const uint32_t *src = outs[r.thread].cigar.data();
        std::copy(src + r.c0, src + r.c1, out.cigar.begin() + w);
        w += r.c1 - r.c0;
    }
    out.cigar_off[tot] = w;
    out.depth_len = p.chr_len + 1;

    if (p.with_seq) {
        out.seq_off.resize(tot + 1);
        uint64_t b = 0;
        std::vector<uint32_t> qlen(tot);
        for (size_t i = 0; i < tot; i++) {
            uint64_t q = 0;
            for (uint64_t c = out.cigar_off[i]; c < out.cigar_off[i + 1]; c++) {
                const uint32_t op = out.cigar[c] & 15u;
                if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) q += out.cigar[c] >> 4;
            }
            qlen[i] = (uint32_t)q; out.seq_off[i] = b; b += (q + 1) / 2;
        }
        out.seq_off[tot] = b;
        out.seq.resize(b);
        static const uint8_t codes[8] = {1, 2, 4, 8, 1, 2, 4, 15};   // A C G T (+ a few N)
        for (size_t i = 0; i < tot; i++) {
            std::mt19937_64 g(mix(p.seed ^ 0xABCDEFull ^ (i * 0x9E3779B97F4A7C15ull)));
            uint8_t *d = out.seq.data() + out.seq_off[i];
            for (uint32_t q = 0; q < (qlen[i] + 1) / 2; q++) { const uint64_t x = g(); d[q] = (uint8_t)((codes[x & 7] << 4) | codes[(x >> 3) & 7]); }
        }
    }
}

// SNPs of a synthetic sample: about one per kilobase, heterozygous (BAF 0.5 +- 0.05) two times out of three, else
// homozygous (BAF 1.0); no population frequency (a run without --pfb: the reference's map then default-constructs 0.0).
void synth_snps(uint64_t seed, uint32_t chr_len, std::vector<uint32_t> &pos, std::vector<double> &baf)
{
    std::mt19937_64 g(mix(seed ^ 0x534E5053ull));
    pos.clear(); baf.clear();
    for (uint64_t p = 500 + g() % 1000; p < chr_len; p += 500 + g() % 1000) {
        pos.push_back((uint32_t)p);
        baf.push_back(unif(g) < 2.0 / 3.0 ? 0.45 + 0.1 * unif(g) : 1.0);
    }
}
