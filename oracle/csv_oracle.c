/*
 * csv_oracle.c — TEST INFRASTRUCTURE ONLY. NOT PART OF THE PRODUCT PATH.
 *
 * Plain-C, single-threaded restatement of the ContextSV hot path, written to follow the
 * reference line by line (citations are file:line under the reference root). Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; nothing under
 * contextsv_amd/ or include/ links, imports or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   orc_dbscan_iv / orc_dbscan_1d / orc_largest_cluster / orc_pdf_normal / orc_cdf_normal
 *       PINNED: checked bit-for-bit against the reference's own dbscan.cpp, dbscan1d.cpp and
 *       kc.cpp compiled unmodified into oracle/_ref (tests/test_oracle_vs_ref.py) and against
 *       the golden vectors generated from them (tests/golden/).
 *   orc_cigar_scan / orc_aln_intervals / orc_depth / orc_window_log2 / orc_viterbi
 *       PARITY UNPINNED by reference fixtures: sv_caller.cpp, cnv_caller.cpp and khmm.cpp need
 *       htslib headers that this image lacks, so they cannot be built here, and the reference's
 *       only known-answer test needs data that is not in the repository. They are line-by-line
 *       restatements plus hand-built known-answer cases, one per quirk (tests/test_oracle_kat.py).
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC (see oracle/Makefile). -ffp-contract=off keeps
 * every double expression un-fused, as in the reference's x86-64 build (Makefile:14, no -march).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* BAM constants (SAM spec; the reference gets them from htslib/sam.h) */
enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };
enum { F_REVERSE = 0x10, F_UNMAP = 0x4, F_SECONDARY = 0x100, F_QCFAIL = 0x200, F_DUP = 0x400, F_SUPP = 0x800 };

typedef struct { uint32_t start, end, read, qpos_kind; } orc_sig;   /* same layout as csv_sig */

/* ------------------------------------------------------------------------------------------ */
/* SVCall::operator< (sv_object.cpp:17-20) on (start,end) */
static int sig_less(const orc_sig *a, const orc_sig *b)
{
    return a->start < b->start || (a->start == b->start && a->end < b->end);
}

/* addSVCall (sv_object.cpp:22-33): reject start > end, std::lower_bound + insert */
static void add_sv_call(orc_sig *v, uint64_t *n, const orc_sig *c)
{
    if (c->start > c->end) return;                       /* :25-28 */
    uint64_t lo = 0, hi = *n;                            /* std::lower_bound: first elem !(elem < c) */
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if (sig_less(&v[mid], c)) lo = mid + 1; else hi = mid;
    }
    memmove(&v[lo + 1], &v[lo], (size_t)(*n - lo) * sizeof(orc_sig));
    v[lo] = *c;
    (*n)++;
}

/* findCIGARSVs + processCIGARRecord + addSVCall (sv_caller.cpp:506-537, 539-661).
 * Returns the number of signatures; writes at most `cap` of them (call with cap = upper bound:
 * the number of CIGAR words). */
int64_t orc_cigar_scan(uint64_t n_reads, const int32_t *pos_a, const uint16_t *flag_a,
                       const uint8_t *mapq_a, const uint64_t *cigar_off, const uint32_t *cigar,
                       uint32_t depth_len, uint32_t min_oplen, uint8_t min_mapq,
                       orc_sig *out, uint64_t cap)
{
    uint64_t n = 0;
    orc_sig *per_read = NULL;
    uint64_t per_read_cap = 0;
    for (uint64_t r = 0; r < n_reads; r++) {
        uint16_t flag = flag_a[r];
        /* sv_caller.cpp:526 */
        if ((flag & F_SECONDARY) || (flag & F_UNMAP) || (flag & F_DUP) || (flag & F_QCFAIL) ||
            mapq_a[r] < min_mapq || (flag & F_SUPP))
            continue;
        /* processCIGARRecord :539 */
        uint32_t pos = (uint32_t)pos_a[r];               /* :542-543 */
        uint32_t query_pos = 0;                          /* :546 */
        uint64_t c0 = cigar_off[r], c1 = cigar_off[r + 1];
        uint64_t m = 0;
        if (c1 - c0 > per_read_cap) {
            per_read_cap = c1 - c0;
            per_read = (orc_sig *)realloc(per_read, (size_t)per_read_cap * sizeof(orc_sig));
        }
        for (uint64_t i = c0; i < c1; i++) {             /* :563 */
            int op_len = (int)(cigar[i] >> 4);           /* :564 */
            int op = (int)(cigar[i] & 0xf);              /* :565 */
            if (op_len >= (int)min_oplen) {              /* :566 */
                if (op == OP_I) {                        /* :569 */
                    orc_sig s;
                    s.start = pos + 1;                   /* :584 */
                    s.end = s.start + (uint32_t)op_len - 1;   /* :585 */
                    s.read = (uint32_t)r;
                    s.qpos_kind = (query_pos << 2) | 0u; /* CIGARINS */
                    per_read[m++] = s;
                } else if (op == OP_S) {                 /* :599 */
                    if ((size_t)(uint32_t)(pos + 1u) >= (size_t)depth_len)   /* uint32 `pos + 1` vs size_t size() */
                        continue;                        /* :602-604 — also skips :648-655 */
                    orc_sig s;
                    s.start = pos + 1;                   /* :619 */
                    s.end = s.start + (uint32_t)op_len - 1;   /* :620 */
                    s.read = (uint32_t)r;
                    s.qpos_kind = (query_pos << 2) | 2u; /* CIGARCLIP */
                    per_read[m++] = s;
                } else if (op == OP_D) {                 /* :634 */
                    orc_sig s;
                    s.start = pos + 1;                   /* :636 */
                    s.end = s.start + (uint32_t)op_len - 1;   /* :637 */
                    s.read = (uint32_t)r;
                    s.qpos_kind = (query_pos << 2) | 1u; /* CIGARDEL */
                    per_read[m++] = s;
                }
            }
            if (op == OP_M || op == OP_D || op == OP_N || op == OP_EQ || op == OP_X)
                pos += (uint32_t)op_len;                 /* :648-650 */
            if (op == OP_M || op == OP_I || op == OP_S || op == OP_EQ || op == OP_X)
                query_pos += (uint32_t)op_len;           /* :653-655 */
        }
        for (uint64_t k = 0; k < m; k++) {               /* :658-660 */
            if (n < cap) add_sv_call(out, &n, &per_read[k]);
            else if (per_read[k].start <= per_read[k].end) n++;
        }
    }
    free(per_read);
    return (int64_t)n;
}

/* getAlignmentReadPositions (sv_caller.cpp:663-690) + htslib bam_endpos (pos + reference length of
 * the CIGAR over M,D,N,=,X; length 0 or unmapped -> pos + 1), as used at sv_caller.cpp:152,162. */
void orc_aln_intervals(uint64_t n_reads, const int32_t *pos_a, const uint16_t *flag_a,
                       const uint64_t *cigar_off, const uint32_t *cigar,
                       int32_t *ref_end, int32_t *q_start, int32_t *q_end)
{
    for (uint64_t r = 0; r < n_reads; r++) {
        int query_start = -1, query_end = 0;             /* :665-666 */
        int64_t rlen = 0;
        for (uint64_t i = cigar_off[r]; i < cigar_off[r + 1]; i++) {
            int op_len = (int)(cigar[i] >> 4), op = (int)(cigar[i] & 0xf);
            if (query_start == -1 && (op == OP_M || op == OP_I || op == OP_EQ || op == OP_X))
                query_start = query_end;                 /* :674-676 */
            if (op == OP_M || op == OP_I || op == OP_S || op == OP_EQ || op == OP_X)
                query_end += op_len;                     /* :680-682 */
            if (op == OP_M || op == OP_D || op == OP_N || op == OP_EQ || op == OP_X)
                rlen += op_len;
        }
        if (query_start == -1) query_start = 0;          /* :685-687 */
        if (flag_a[r] & F_UNMAP) rlen = 0;
        if (rlen == 0) rlen = 1;
        ref_end[r] = (int32_t)(pos_a[r] + rlen);
        q_start[r] = query_start;
        q_end[r] = query_end;
    }
}

/* per-chromosome body of calculateMeanChromosomeCoverage (cnv_caller.cpp:488-543) */
void orc_depth(uint64_t n_reads, const int32_t *pos_a, const uint16_t *flag_a,
               const uint64_t *cigar_off, const uint32_t *cigar, uint32_t depth_len,
               uint32_t *depth, uint64_t *sum_out, uint32_t *nonzero_out)
{
    memset(depth, 0, (size_t)depth_len * sizeof(uint32_t));
    for (uint64_t r = 0; r < n_reads; r++) {
        uint16_t flag = flag_a[r];
        if (flag & (F_UNMAP | F_SECONDARY | F_QCFAIL | F_DUP)) continue;   /* :491-495 */
        uint32_t ref_pos = (uint32_t)pos_a[r] + 1;                          /* :498-499 */
        for (uint64_t i = cigar_off[r]; i < cigar_off[r + 1]; i++) {
            uint32_t op = cigar[i] & 0xf, op_len = cigar[i] >> 4;
            if (op == OP_M || op == OP_EQ || op == OP_X) {                  /* :506 */
                for (uint32_t j = 0; j < op_len; j++) {
                    if ((uint64_t)(uint32_t)(ref_pos + j) >= (uint64_t)depth_len) continue;  /* :511-515 */
                    depth[(uint32_t)(ref_pos + j)]++;
                }
            }
            if (op == OP_M || op == OP_D || op == OP_N || op == OP_EQ || op == OP_X)
                ref_pos += op_len;                                          /* :522-523 */
        }
    }
    uint64_t cum = 0; uint32_t cnt = 0;                                     /* :531-532 */
    for (uint32_t p = 0; p < depth_len; p++) { cum += depth[p]; cnt += depth[p] > 0; }
    *sum_out = cum; *nonzero_out = cnt;
}

/* ------------------------------------------------------------------------------------------ */
/* DBSCAN (dbscan.cpp:9-81): sequential, visit-order dependent, O(n^2) regionQuery              */

static double iv_distance(uint32_t s1, uint32_t e1, uint32_t s2, uint32_t e2)
{   /* dbscan.cpp:69-81 */
    int a = (int)e1 < (int)e2 ? (int)e1 : (int)e2;          /* std::min(end1,end2) */
    int b = (int)s1 > (int)s2 ? (int)s1 : (int)s2;          /* std::max(start1,start2) */
    int overlap = (a - b) > 0 ? (a - b) : 0;                 /* std::max(0, ...) */
    int length1 = (int)(e1 - s1);
    int length2 = (int)(e2 - s2);
    double x = (double)overlap / (double)length1;
    double y = (double)overlap / (double)length2;
    double mn = (y < x) ? y : x;                             /* std::min(x,y) = (y<x)?y:x — NaN-asymmetric */
    return 1.0 - mn;
}

typedef struct { size_t *v; size_t n, cap; } idxvec;
static void iv_push(idxvec *q, size_t x)
{
    if (q->n == q->cap) { q->cap = q->cap ? q->cap * 2 : 64; q->v = (size_t *)realloc(q->v, q->cap * sizeof(size_t)); }
    q->v[q->n++] = x;
}

static void region_query_iv(const uint32_t *s, const uint32_t *e, size_t n, size_t p, double eps, idxvec *out)
{   /* dbscan.cpp:59-67 */
    out->n = 0;
    for (size_t i = 0; i < n; i++)
        if (iv_distance(s[p], e[p], s[i], e[i]) <= eps) iv_push(out, i);
}

void orc_dbscan_iv(const uint32_t *s, const uint32_t *e, uint64_t n64, double eps, int32_t min_pts, int32_t *cl)
{
    size_t n = (size_t)n64;
    int cluster_id = 0;
    idxvec seeds = {0, 0, 0}, res = {0, 0, 0};
    for (size_t i = 0; i < n; i++) cl[i] = -1;               /* :11 */
    for (size_t p = 0; p < n; p++) {                         /* :13 */
        if (cl[p] != -1) continue;
        /* expandCluster :26 */
        region_query_iv(s, e, n, p, eps, &seeds);
        if ((int)seeds.n < min_pts) { cl[p] = -2; continue; }         /* :28-31 */
        for (size_t k = 0; k < seeds.n; k++) cl[seeds.v[k]] = cluster_id;   /* :33-35 */
        {   /* :37 erase-remove pointIdx */
            size_t w = 0;
            for (size_t k = 0; k < seeds.n; k++) if (seeds.v[k] != p) seeds.v[w++] = seeds.v[k];
            seeds.n = w;
        }
        while (seeds.n) {                                    /* :39 */
            size_t cur = seeds.v[--seeds.n];                 /* back + pop_back */
            region_query_iv(s, e, n, cur, eps, &res);
            if ((int)res.n >= min_pts) {
                for (size_t k = 0; k < res.n; k++) {
                    size_t q = res.v[k];
                    if (cl[q] == -1 || cl[q] == -2) {
                        if (cl[q] == -1) iv_push(&seeds, q);
                        cl[q] = cluster_id;
                    }
                }
            }
        }
        cluster_id++;                                        /* :16-18 */
    }
    free(seeds.v); free(res.v);
}

/* DBSCAN1D (dbscan1d.cpp:8-70) */
static void region_query_1d(const int32_t *pts, size_t n, size_t p, double eps, idxvec *out)
{
    out->n = 0;
    for (size_t i = 0; i < n; i++) {
        double d = (double)abs(pts[p] - pts[i]);             /* :68-70 std::abs(int) -> double */
        if (d <= eps) iv_push(out, i);
    }
}

void orc_dbscan_1d(const int32_t *pts, uint64_t n64, double eps, int32_t min_pts, int32_t *cl)
{
    size_t n = (size_t)n64;
    int cluster_id = 0;
    idxvec seeds = {0, 0, 0}, res = {0, 0, 0};
    for (size_t i = 0; i < n; i++) cl[i] = -1;
    for (size_t p = 0; p < n; p++) {
        if (cl[p] != -1) continue;
        region_query_1d(pts, n, p, eps, &seeds);
        if ((int)seeds.n < min_pts) { cl[p] = -2; continue; }
        for (size_t k = 0; k < seeds.n; k++) cl[seeds.v[k]] = cluster_id;
        {
            size_t w = 0;
            for (size_t k = 0; k < seeds.n; k++) if (seeds.v[k] != p) seeds.v[w++] = seeds.v[k];
            seeds.n = w;
        }
        while (seeds.n) {
            size_t cur = seeds.v[--seeds.n];
            region_query_1d(pts, n, cur, eps, &res);
            if ((int)res.n >= min_pts) {
                for (size_t k = 0; k < res.n; k++) {
                    size_t q = res.v[k];
                    if (cl[q] == -1 || cl[q] == -2) {
                        if (cl[q] == -1) iv_push(&seeds, q);
                        cl[q] = cluster_id;
                    }
                }
            }
        }
        cluster_id++;
    }
    free(seeds.v); free(res.v);
}

/* DBSCAN1D::getLargestCluster (dbscan1d.cpp:72-90): members (in index order) of the cluster with
 * the most points; ties -> lowest id (ascending std::map walk with strict >); none -> empty. */
int64_t orc_largest_cluster(const int32_t *pts, const int32_t *cl, uint64_t n, int32_t *out)
{
    int32_t max_id = -1;
    for (uint64_t i = 0; i < n; i++) if (cl[i] > max_id) max_id = cl[i];
    int64_t best = -1; uint64_t best_size = 0;
    for (int32_t c = 0; c <= max_id; c++) {
        uint64_t sz = 0;
        for (uint64_t i = 0; i < n; i++) sz += (cl[i] == c);
        if (sz > best_size) { best_size = sz; best = c; }
    }
    if (best < 0) return 0;                                  /* cluster_map[-1] — empty (:89) */
    int64_t m = 0;
    for (uint64_t i = 0; i < n; i++) if (cl[i] == best) out[m++] = pts[i];
    return m;
}

/* ------------------------------------------------------------------------------------------ */
/* window log2 coverage (cnv_caller.cpp:76-113)                                                 */
void orc_window_log2(const uint32_t *depth, uint32_t depth_len, uint32_t start_pos, uint32_t end_pos,
                     int32_t sample_size, double mean_chr_cov,
                     double *log2_out, uint32_t *win_start, uint32_t *win_end)
{
    double pos_step = (double)(end_pos - start_pos + 1) / (double)sample_size;   /* :76 */
    for (int i = 0; i < sample_size; i++) {                                       /* :78 */
        uint32_t window_start = (uint32_t)(start_pos + i * pos_step);             /* :80 */
        uint32_t window_end = (uint32_t)(start_pos + (i + 1) * pos_step);         /* :81 */
        double cov_sum = 0.0; int pos_count = 0;
        for (int j = 0; j < pos_step; j++) {                                      /* :86 */
            uint32_t pos = (uint32_t)(start_pos + i * pos_step + j);              /* :88 */
            if (pos > end_pos) break;                                             /* :89-92 */
            if (pos < depth_len) { cov_sum += depth[pos]; pos_count++; }          /* :93-96 */
        }
        double log2_cov = 0.0;
        if (pos_count > 0) {
            if (cov_sum == 0) cov_sum = 1e-9;                                     /* :102-106 */
            log2_cov = log2((cov_sum / (double)pos_count) / mean_chr_cov);        /* :107 */
        }
        log2_out[i] = log2_cov; win_start[i] = window_start; win_end[i] = window_end;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* kc.cpp numerics reachable from the path                                                      */
#define KC_ITMAX 100
#define KC_EPS 3.0e-7
#define KC_FPMIN 1.0e-30
#define KC_PI 3.141592653579893            /* kc.cpp:150 — not pi's digits; reproduced on purpose */

static double kc_gammln(double x)           /* kc.cpp:3470-3490 */
{
    double tmp, ser;
    tmp = x + 4.5 - (x - 0.5) * log(x + 4.5);
    ser = 1.000000000190015 + (76.18009172947146 / x) - (86.50532032941677 / (x + 1.0)) +
          (24.01409824083091 / (x + 2.0)) - (1.231739572450155 / (x + 3.0)) +
          (0.1208650973866179e-2 / (x + 4.0)) - (0.5395239384953e-5 / (x + 5.0));
    return (log(2.5066282746310005 * ser) - tmp);
}

static void kc_gser(double *gamser, double a, double x, double *gln)   /* kc.cpp:3548-3577 */
{
    int n; double sum, del, ap;
    *gln = kc_gammln(a);
    if (x <= 0.0) { *gamser = 0.0; return; }
    ap = a; del = sum = 1.0 / a;
    for (n = 1; n <= KC_ITMAX; n++) {
        ++ap; del *= x / ap; sum += del;
        if (fabs(del) < fabs(sum) * KC_EPS) { *gamser = sum * exp(-x + a * log(x) - (*gln)); return; }
    }
}

static void kc_gcf(double *gammcf, double a, double x, double *gln)    /* kc.cpp:3579-3605 */
{
    int i; double an, b, c, d, del, h;
    *gln = kc_gammln(a);
    b = x + 1.0 - a; c = 1.0 / KC_FPMIN; d = 1.0 / b; h = d;
    for (i = 1; i <= KC_ITMAX; i++) {
        an = -i * (i - a); b += 2.0; d = an * d + b;
        if (fabs(d) < KC_FPMIN) d = KC_FPMIN;
        c = b + an / c;
        if (fabs(c) < KC_FPMIN) c = KC_FPMIN;
        d = 1.0 / d; del = d * c; h *= del;
        if (fabs(del - 1.0) < KC_EPS) break;
    }
    *gammcf = exp(-x + a * log(x) - (*gln)) * h;
}

static double kc_gammp(double a, double x)  /* kc.cpp:3512-3540 */
{
    double gamser = 0.0, gammcf, gln;
    if (x < (a + 1.0)) { kc_gser(&gamser, a, x, &gln); return gamser; }
    kc_gcf(&gammcf, a, x, &gln); return 1.0 - gammcf;
}

static double kc_errorf(double x)           /* kc.cpp:3703-3716 */
{
    return (x < 0.0) ? (-kc_gammp(0.5, x * x)) : kc_gammp(0.5, x * x);
}

double orc_cdf_normal(double x, double mu, double sigma)   /* kc.cpp:2565-2576 */
{
    return (1 + kc_errorf((x - mu) / (sigma * sqrt(2)))) / 2;
}

double orc_pdf_normal(double x, double mu, double sigma)   /* kc.cpp:2658-2662 */
{
    return exp(-(x - mu) * (x - mu) / (2 * sigma * sigma)) / (sigma * sqrt(2 * KC_PI));
}

/* ------------------------------------------------------------------------------------------ */
/* khmm.cpp                                                                                     */
typedef struct {
    double A[36], pi[6], B1_mean[6], B1_sd[6], B1_uf, B2_mean[5], B2_sd[5], B2_uf;
} orc_hmm;                                  /* same layout as csv_hmm */

#define VITHUGE 100000000000.0
#define FLOAT_MINIMUM 1.175494351e-38
#define PROB_MAX 0.9999999999999999

double orc_b1iot(int state, const double *mean, const double *sd, double uf, double o)
{   /* khmm.cpp:58-78 */
    if (o < mean[0]) o = mean[0];
    else if (o > mean[5]) o = mean[5];
    double p = uf + ((1 - uf) * orc_pdf_normal(o, mean[state - 1], sd[state - 1]));
    return log(p);
}

double orc_b2iot(int state, const double *mean, const double *sd, double uf, double pfb, double b)
{   /* khmm.cpp:80-206 */
    double p = 0;
    double mean0 = mean[0], mean25 = mean[1], mean33 = mean[2], mean50 = mean[3], mean50_state1 = mean[4];
    double sd0 = sd[0], sd25 = sd[1], sd33 = sd[2], sd50 = sd[3], sd50_state1 = sd[4];
    p = uf;
    if (state == 1) {
        if (b == 0) p += (1 - uf) * orc_cdf_normal(0, mean50_state1, sd50_state1);
        else if (b == 1) p += (1 - uf) * orc_cdf_normal(0, mean50_state1, sd50_state1);
        else p += (1 - uf) * orc_pdf_normal(b, mean50_state1, sd50_state1);
    } else if (state == 2) {
        if (b == 0) p += (1 - uf) * (1 - pfb) / 2;
        else if (b == 1) p += (1 - uf) * pfb / 2;
        else {
            p += (1 - uf) * (1 - pfb) * orc_pdf_normal(b, mean0, sd0);
            p += (1 - uf) * pfb * orc_pdf_normal(b, 1 - mean0, sd0);
        }
    } else if (state == 3) {
        if (b == 0) p += (1 - uf) * (1 - pfb) * (1 - pfb) / 2;
        else if (b == 1) p += (1 - uf) * pfb * pfb / 2;
        else {
            p += (1 - uf) * (1 - pfb) * (1 - pfb) * orc_pdf_normal(b, mean0, sd0);
            p += (1 - uf) * 2 * pfb * (1 - pfb) * orc_pdf_normal(b, mean50, sd50);
            p += (1 - uf) * pfb * pfb * orc_pdf_normal(b, 1 - mean0, sd0);
        }
    } else if (state == 4) {
        if (b == 0) p += (1 - uf) * (1 - pfb) / 2;
        else if (b == 1) p += (1 - uf) * pfb / 2;
        else {
            p += (1 - uf) * (1 - pfb) * orc_pdf_normal(b, mean0, sd0);
            p += (1 - uf) * pfb * orc_pdf_normal(b, 1 - mean0, sd0);
        }
    } else if (state == 5) {
        if (b == 0) p += (1 - uf) * (1 - pfb) * (1 - pfb) * (1 - pfb) / 2;
        else if (b == 1) p += (1 - uf) * pfb * pfb * pfb / 2;
        else {
            p += (1 - uf) * (1 - pfb) * (1 - pfb) * (1 - pfb) * orc_pdf_normal(b, mean0, sd0);
            p += (1 - uf) * 3 * (1 - pfb) * (1 - pfb) * pfb * orc_pdf_normal(b, mean33, sd33);
            p += (1 - uf) * 3 * (1 - pfb) * pfb * pfb * orc_pdf_normal(b, 1 - mean33, sd33);
            p += (1 - uf) * pfb * pfb * pfb * orc_pdf_normal(b, 1 - mean0, sd0);
        }
    } else if (state == 6) {
        if (b == 0) p += (1 - uf) * (1 - pfb) * (1 - pfb) * (1 - pfb) * (1 - pfb) / 2;
        else if (b == 1) p += (1 - uf) * pfb * pfb * pfb * pfb / 2;
        else {
            p += (1 - uf) * (1 - pfb) * (1 - pfb) * (1 - pfb) * (1 - pfb) * orc_pdf_normal(b, mean0, sd0);
            p += (1 - uf) * 4 * (1 - pfb) * (1 - pfb) * (1 - pfb) * pfb * orc_pdf_normal(b, mean25, sd25);
            p += (1 - uf) * 6 * (1 - pfb) * (1 - pfb) * pfb * pfb * orc_pdf_normal(b, mean50, sd50);
            p += (1 - uf) * 4 * (1 - pfb) * pfb * pfb * pfb * orc_pdf_normal(b, 1 - mean25, sd25);
            p += (1 - uf) * pfb * pfb * pfb * pfb * orc_pdf_normal(b, 1 - mean0, sd0);
        }
    }
    /* :203 std::max(FLOAT_MINIMUM, std::min(PROB_MAX, p)) */
    double q = (p < PROB_MAX) ? p : PROB_MAX;                /* std::min(PROB_MAX,p) = (p<PROB_MAX)?p:PROB_MAX */
    q = (FLOAT_MINIMUM < q) ? q : FLOAT_MINIMUM;             /* std::max(FLOAT_MINIMUM,q) = (FM<q)?q:FM */
    return log(q);
}

/* testVit_CHMM + ViterbiLogNP_CHMM (khmm.cpp:28-56, 225-393). states[0..T-1] in 1..6. */
void orc_viterbi(const orc_hmm *hmm, int32_t T, const double *O1, const double *O2, const double *pfb,
                 int32_t *states, double *loglik)
{
    const int N = 6;
    if (T <= 0) { *loglik = -VITHUGE; return; }              /* q[T]=1, final_lh=-VITHUGE, nothing returned */
    double pi[6];
    for (int i = 0; i < N; i++) {                            /* :276-283 */
        double v = hmm->pi[i];
        if (v == 0) v = 1e-9;
        pi[i] = log(v);
    }
    double *biot = (double *)malloc(sizeof(double) * (size_t)T * N);
    double *delta = (double *)malloc(sizeof(double) * (size_t)T * N);
    int *psi = (int *)malloc(sizeof(int) * (size_t)T * N);
    for (int i = 1; i <= N; i++)                             /* :287-320 */
        for (int t = 1; t <= T; t++) {
            double o1 = O1[t - 1];
            if (O2[t - 1] == -1) {
                biot[(t - 1) * N + (i - 1)] = orc_b1iot(i, hmm->B1_mean, hmm->B1_sd, hmm->B1_uf, o1);
            } else {
                double a = orc_b1iot(i, hmm->B1_mean, hmm->B1_sd, hmm->B1_uf, o1);
                double b = orc_b2iot(i, hmm->B2_mean, hmm->B2_sd, hmm->B2_uf, pfb[t - 1], O2[t - 1]);
                biot[(t - 1) * N + (i - 1)] = a + b;
            }
        }
    for (int i = 0; i < N; i++) { delta[i] = pi[i] + biot[i]; psi[i] = 0; }    /* :323-328 */
    for (int t = 1; t < T; t++)                              /* :334-356 */
        for (int j = 0; j < N; j++) {
            double maxval = -VITHUGE; int maxvalind = 1;
            for (int i = 0; i < N; i++) {
                double val = delta[(t - 1) * N + i] + log(hmm->A[i * 6 + j]);
                if (val > maxval) { maxval = val; maxvalind = i + 1; }
            }
            delta[t * N + j] = maxval + biot[t * N + j];
            psi[t * N + j] = maxvalind;
        }
    int q = 1; double final_lh = -VITHUGE;                   /* :362-371 */
    for (int i = 0; i < N; i++)
        if (delta[(T - 1) * N + i] > final_lh) { final_lh = delta[(T - 1) * N + i]; q = i + 1; }
    states[T - 1] = q;
    for (int t = T - 2; t >= 0; t--) {                       /* :378-381 */
        q = psi[(t + 1) * N + (q - 1)];
        states[t] = q;
    }
    *loglik = final_lh;
    free(biot); free(delta); free(psi);
}

/* batch wrapper with the C-ABI's segment layout */
void orc_viterbi_batch(const orc_hmm *hmm, const double *o1, const double *o2, const double *pfb,
                       const uint64_t *seq_off, uint64_t n_seq, int32_t *states, double *loglik)
{
    for (uint64_t s = 0; s < n_seq; s++) {
        uint64_t a = seq_off[s], b = seq_off[s + 1];
        orc_viterbi(hmm, (int32_t)(b - a), o1 + a, o2 + a, pfb + a, states + a, &loglik[s]);
    }
}
