// hmm.hip — kernel #4 (secondary): window log2 coverage and batched 6-state Viterbi (gfx950).
//
// window_log2: replaces the window loop of CNVCaller::querySNPRegion (cnv_caller.cpp:76-113).
//   One wave per window; every lane evaluates the reference's own double expression
//   (uint32)(start + i*step + j) for its j's, so even the rounding corner cases of the position
//   formula are reproduced; depth values are summed as integers (exact, order-free) and converted once.
// viterbi: replaces testVit_CHMM / ViterbiLogNP_CHMM (khmm.cpp:28-56, 225-393) with its emissions
//   b1iot / b2iot (khmm.cpp:58-206) and the kc.cpp numerics they reach (pdf_normal :2658,
//   cdf_normal :2565 -> errorf :3703 -> gammp/gser/gcf/gammln :3470-3605, PI as defined at :150).
//   Three kernels: parameters (log A, log pi, the one cdf_normal constant) -> emissions, one thread
//   per (observation, state), fully parallel -> DP, wave-synchronous: a wave carries 10 sequences,
//   6 lanes each (lane = state); the max-plus step reads the six predecessor deltas with shuffles,
//   strict '>' from -1e11 keeps the reference's lowest-index tie-break; psi goes to a byte array,
//   the group's first lane backtracks. fp64 throughout, compiled with -ffp-contract=off.
#include "../common.hpp"
#include "../devutil.hpp"

namespace csv {

// ------------------------------------------------------------------------------- window log2
__global__ __launch_bounds__(256) void window_log2_kernel(const uint32_t *__restrict__ depth, uint32_t depth_len,
                                                         const uint32_t *__restrict__ rs, const uint32_t *__restrict__ re,
                                                         const int32_t *__restrict__ ss, const uint64_t *__restrict__ win_off,
                                                         uint64_t n_regions, uint64_t n_windows, double mean_cov,
                                                         double *__restrict__ log2_cov, uint32_t *__restrict__ ws_out,
                                                         uint32_t *__restrict__ we_out)
{
    const int lane = lane_id();
    const uint64_t w = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= n_windows) return;
    // region of this window: last r with win_off[r] <= w
    uint64_t lo = 0, hi = n_regions;
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (win_off[mid] <= w) lo = mid; else hi = mid; }
    const uint64_t r = lo;
    const int i = (int)(w - win_off[r]);
    const uint32_t start_pos = rs[r], end_pos = re[r];
    const int sample_size = ss[r];
    const double pos_step = (double)(uint32_t)(end_pos - start_pos + 1u) / (double)sample_size;   // :76
    const uint32_t window_start = (uint32_t)((double)start_pos + (double)i * pos_step);           // :80
    const uint32_t window_end = (uint32_t)((double)start_pos + (double)(i + 1) * pos_step);       // :81
    unsigned long long sum = 0; uint32_t cnt = 0;
    const double base = (double)start_pos + (double)i * pos_step;
    for (int j = lane; (double)j < pos_step; j += WAVE) {                                          // :86
        const uint32_t pos = (uint32_t)(base + (double)j);                                         // :88
        if (pos > end_pos) break;                       // positions are monotone in j (:89-92)
        if (pos < depth_len) { sum += depth[pos]; cnt++; }                                         // :93-96
    }
    sum = wave_sum64(sum);
    cnt = wave_sum(cnt);
    if (lane == 0) {
        double log2_c = 0.0;
        if (cnt > 0) {
            double cov_sum = (double)sum;
            if (cov_sum == 0) cov_sum = 1e-9;                                                      // :102-106
            log2_c = log2((cov_sum / (double)(int)cnt) / mean_cov);                                // :107
        }
        log2_cov[w] = log2_c; ws_out[w] = window_start; we_out[w] = window_end;
    }
}

void launch_window_log2(hipStream_t s, const uint32_t *depth, uint32_t depth_len, const uint32_t *rs, const uint32_t *re,
                        const int32_t *ss, const uint64_t *win_off, uint64_t n_regions, uint64_t n_windows,
                        double mean_cov, double *log2_cov, uint32_t *ws, uint32_t *we)
{
    if (!n_windows) return;
    hipLaunchKernelGGL(window_log2_kernel, dim3((unsigned)((n_windows + 3) / 4)), dim3(256), 0, s, depth, depth_len, rs, re, ss,
                       win_off, n_regions, n_windows, mean_cov, log2_cov, ws, we);
}

// ------------------------------------------------------------------------------- kc.cpp numerics
#define KC_ITMAX 100
#define KC_EPS 3.0e-7
#define KC_FPMIN 1.0e-30
#define KC_PI 3.141592653579893      /* kc.cpp:150 */
#define VITHUGE 100000000000.0
#define FLOAT_MINIMUM 1.175494351e-38
#define PROB_MAX 0.9999999999999999

__device__ double kc_gammln(double x)
{
    double tmp = x + 4.5 - (x - 0.5) * log(x + 4.5);
    double ser = 1.000000000190015 + (76.18009172947146 / x) - (86.50532032941677 / (x + 1.0)) +
                 (24.01409824083091 / (x + 2.0)) - (1.231739572450155 / (x + 3.0)) +
                 (0.1208650973866179e-2 / (x + 4.0)) - (0.5395239384953e-5 / (x + 5.0));
    return (log(2.5066282746310005 * ser) - tmp);
}
__device__ double kc_gammp(double a, double x)
{
    const double gln = kc_gammln(a);
    if (x < (a + 1.0)) {                                   // gser
        if (x <= 0.0) return 0.0;
        double ap = a, del = 1.0 / a, sum = del, gamser = 0.0;
        for (int n = 1; n <= KC_ITMAX; n++) {
            ++ap; del *= x / ap; sum += del;
            if (fabs(del) < fabs(sum) * KC_EPS) { gamser = sum * exp(-x + a * log(x) - gln); break; }
        }
        return gamser;
    }
    double b = x + 1.0 - a, c = 1.0 / KC_FPMIN, d = 1.0 / b, h = d;   // gcf
    for (int i = 1; i <= KC_ITMAX; i++) {
        const double an = -i * (i - a);
        b += 2.0; d = an * d + b;
        if (fabs(d) < KC_FPMIN) d = KC_FPMIN;
        c = b + an / c;
        if (fabs(c) < KC_FPMIN) c = KC_FPMIN;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < KC_EPS) break;
    }
    return 1.0 - exp(-x + a * log(x) - gln) * h;
}
__device__ double kc_errorf(double x) { return (x < 0.0) ? (-kc_gammp(0.5, x * x)) : kc_gammp(0.5, x * x); }
__device__ double kc_cdf_normal(double x, double mu, double sigma) { return (1 + kc_errorf((x - mu) / (sigma * sqrt(2.0)))) / 2; }
__device__ __forceinline__ double kc_pdf_normal(double x, double mu, double sigma)
{
    return exp(-(x - mu) * (x - mu) / (2 * sigma * sigma)) / (sigma * sqrt(2 * KC_PI));
}

struct HmmDev {            // parameters + derived constants, device resident
    csv_hmm h;
    double logA[36];
    double logpi[6];
    double cdf1;           // cdf_normal(0, B2_mean[4], B2_sd[4]) — state 1, BAF exactly 0 or 1 (khmm.cpp:102-109)
};

__global__ void hmm_prep_kernel(HmmDev *d)
{
    const int t = threadIdx.x;
    if (t < 36) d->logA[t] = log(d->h.A[t]);                                   // khmm.cpp:344
    if (t < 6) { double v = d->h.pi[t]; if (v == 0) v = 1e-9; d->logpi[t] = log(v); }   // :276-283
    if (t == 63) d->cdf1 = kc_cdf_normal(0, d->h.B2_mean[4], d->h.B2_sd[4]);
}

__device__ double b1iot(const csv_hmm &h, int state, double o)
{   // khmm.cpp:58-78
    if (o < h.B1_mean[0]) o = h.B1_mean[0];
    else if (o > h.B1_mean[5]) o = h.B1_mean[5];
    const double uf = h.B1_uf;
    const double p = uf + ((1 - uf) * kc_pdf_normal(o, h.B1_mean[state - 1], h.B1_sd[state - 1]));
    return log(p);
}

__device__ double b2iot(const HmmDev &d, int state, double pfb, double b)
{   // khmm.cpp:80-206
    const csv_hmm &h = d.h;
    const double uf = h.B2_uf;
    const double mean0 = h.B2_mean[0], mean25 = h.B2_mean[1], mean33 = h.B2_mean[2], mean50 = h.B2_mean[3], mean50_s1 = h.B2_mean[4];
    const double sd0 = h.B2_sd[0], sd25 = h.B2_sd[1], sd33 = h.B2_sd[2], sd50 = h.B2_sd[3], sd50_s1 = h.B2_sd[4];
    double p = uf;
    if (state == 1) {
        if (b == 0) p += (1 - uf) * d.cdf1;
        else if (b == 1) p += (1 - uf) * d.cdf1;
        else p += (1 - uf) * kc_pdf_normal(b, mean50_s1, sd50_s1);
    } else if (state == 2 || state == 4) {
        if (b == 0) p += (1 - uf) * (1 - pfb) / 2;
        else if (b == 1) p += (1 - uf) * pfb / 2;
        else {
            p += (1 - uf) * (1 - pfb) * kc_pdf_normal(b, mean0, sd0);
            p += (1 - uf) * pfb * kc_pdf_normal(b, 1 - mean0, sd0);
        }
    } else if (state == 3) {
        if (b == 0) p += (1 - uf) * (1 - pfb) * (1 - pfb) / 2;
        else if (b == 1) p += (1 - uf) * pfb * pfb / 2;
        else {
            p += (1 - uf) * (1 - pfb) * (1 - pfb) * kc_pdf_normal(b, mean0, sd0);
            p += (1 - uf) * 2 * pfb * (1 - pfb) * kc_pdf_normal(b, mean50, sd50);
            p += (1 - uf) * pfb * pfb * kc_pdf_normal(b, 1 - mean0, sd0);
        }
    } else if (state == 5) {
        if (b == 0) p += (1 - uf) * (1 - pfb) * (1 - pfb) * (1 - pfb) / 2;
        else if (b == 1) p += (1 - uf) * pfb * pfb * pfb / 2;
        else {
            p += (1 - uf) * (1 - pfb) * (1 - pfb) * (1 - pfb) * kc_pdf_normal(b, mean0, sd0);
            p += (1 - uf) * 3 * (1 - pfb) * (1 - pfb) * pfb * kc_pdf_normal(b, mean33, sd33);
            p += (1 - uf) * 3 * (1 - pfb) * pfb * pfb * kc_pdf_normal(b, 1 - mean33, sd33);
            p += (1 - uf) * pfb * pfb * pfb * kc_pdf_normal(b, 1 - mean0, sd0);
        }
    } else {
        if (b == 0) p += (1 - uf) * (1 - pfb) * (1 - pfb) * (1 - pfb) * (1 - pfb) / 2;
        else if (b == 1) p += (1 - uf) * pfb * pfb * pfb * pfb / 2;
        else {
            p += (1 - uf) * (1 - pfb) * (1 - pfb) * (1 - pfb) * (1 - pfb) * kc_pdf_normal(b, mean0, sd0);
            p += (1 - uf) * 4 * (1 - pfb) * (1 - pfb) * (1 - pfb) * pfb * kc_pdf_normal(b, mean25, sd25);
            p += (1 - uf) * 6 * (1 - pfb) * (1 - pfb) * pfb * pfb * kc_pdf_normal(b, mean50, sd50);
            p += (1 - uf) * 4 * (1 - pfb) * pfb * pfb * pfb * kc_pdf_normal(b, 1 - mean25, sd25);
            p += (1 - uf) * pfb * pfb * pfb * pfb * kc_pdf_normal(b, 1 - mean0, sd0);
        }
    }
    double q = (p < PROB_MAX) ? p : PROB_MAX;               // std::min(PROB_MAX, p)
    q = (FLOAT_MINIMUM < q) ? q : FLOAT_MINIMUM;            // std::max(FLOAT_MINIMUM, q)   (:203)
    return log(q);
}

__global__ void hmm_emit_kernel(const HmmDev *__restrict__ d, const double *__restrict__ o1, const double *__restrict__ o2,
                                const double *__restrict__ pfb, uint64_t n_obs, double *__restrict__ biot)
{
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_obs * 6) return;
    const uint64_t t = g / 6;
    const int state = (int)(g % 6) + 1;
    const double a = b1iot(d->h, state, o1[t]);                              // khmm.cpp:296-317
    const double b = o2[t];
    biot[g] = (b == -1) ? a : a + b2iot(*d, state, pfb[t], b);
}

constexpr int VIT_GROUPS = 10;    // sequences per wave (6 lanes each, 4 lanes idle)
constexpr int VIT_LDS_T = 512;    // back-pointers of sequences up to this length stay in LDS (30 KiB per wave): the backtrack is T dependent
                                  // look-ups by one lane, ~1 us apiece from global memory, tens of ns from LDS; a copy-number pass has T ~ 25

__global__ __launch_bounds__(64) void hmm_viterbi_kernel(const HmmDev *__restrict__ d, const double *__restrict__ biot,
                                                        const uint64_t *__restrict__ seq_off, uint64_t n_seq,
                                                        uint8_t *__restrict__ psi, int32_t *__restrict__ states,
                                                        double *__restrict__ loglik)
{
    __shared__ uint8_t lpsi[VIT_GROUPS][VIT_LDS_T][6];
    const int lane = lane_id();
    const int g = lane / 6, j = lane % 6;
    const uint64_t s = (uint64_t)blockIdx.x * VIT_GROUPS + g;
    const bool live = g < VIT_GROUPS && s < n_seq;
    uint64_t a = 0; int64_t T = 0;
    if (live) { a = seq_off[s]; T = (int64_t)(seq_off[s + 1] - a); }
    int64_t Tmax = T;
#pragma unroll
    for (int dd = 32; dd > 0; dd >>= 1) Tmax = max(Tmax, (int64_t)__shfl_xor((long long)Tmax, dd, 64));
    const bool in_lds = Tmax <= (int64_t)VIT_LDS_T;                          // wave-uniform
    double la[6];
#pragma unroll
    for (int i = 0; i < 6; i++) la[i] = d->logA[i * 6 + j];
    double delta = 0.0;
    if (live && T > 0) delta = d->logpi[j] + biot[a * 6 + j];                // khmm.cpp:323-328
    const int gbase = g < VIT_GROUPS ? g * 6 : 0;
    for (int64_t t = 1; t < Tmax; t++) {                                     // :334-356
        double maxval = -VITHUGE; int ind = 1;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const double di = __shfl(delta, gbase + i, 64);
            const double val = di + la[i];
            if (val > maxval) { maxval = val; ind = i + 1; }
        }
        if (live && t < T) {
            delta = maxval + biot[(a + t) * 6 + j];
            if (in_lds) lpsi[g][t][j] = (uint8_t)ind; else psi[(a + t) * 6 + j] = (uint8_t)ind;
        }
    }
    // termination (:362-371) and backtrack (:378-381) by the group's first lane
    double dl[6];
#pragma unroll
    for (int i = 0; i < 6; i++) dl[i] = __shfl(delta, gbase + i, 64);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (live && j == 0) {
        if (T <= 0) { loglik[s] = -VITHUGE; return; }
        int q = 1; double final_lh = -VITHUGE;
#pragma unroll
        for (int i = 0; i < 6; i++) if (dl[i] > final_lh) { final_lh = dl[i]; q = i + 1; }
        loglik[s] = final_lh;
        states[a + T - 1] = q;
        for (int64_t t = T - 2; t >= 0; t--) {
            q = in_lds ? lpsi[g][t + 1][q - 1] : psi[(a + t + 1) * 6 + (q - 1)];
            states[a + t] = q;
        }
    }
}

size_t viterbi_tmp_bytes(uint64_t n_obs, uint64_t n_seq)
{
    (void)n_seq;
    return align_up(sizeof(HmmDev), 256) + align_up(n_obs * 6 * sizeof(double), 256) + align_up(n_obs * 6, 256);
}

void launch_viterbi(hipStream_t s, const csv_hmm &hmm, const double *o1, const double *o2, const double *pfb,
                    const uint64_t *seq_off, uint64_t n_seq, uint64_t n_obs, int32_t *states, double *loglik, void *tmp)
{
    if (n_seq == 0) return;
    char *p = (char *)tmp;
    HmmDev *d = (HmmDev *)p;        p += align_up(sizeof(HmmDev), 256);
    double *biot = (double *)p;     p += align_up(n_obs * 6 * sizeof(double), 256);
    uint8_t *psi = (uint8_t *)p;
    (void)hipMemcpyAsync(&d->h, &hmm, sizeof(csv_hmm), hipMemcpyHostToDevice, s);
    hipLaunchKernelGGL(hmm_prep_kernel, dim3(1), dim3(64), 0, s, d);
    if (n_obs) hipLaunchKernelGGL(hmm_emit_kernel, dim3((unsigned)((n_obs * 6 + 255) / 256)), dim3(256), 0, s, d, o1, o2, pfb, n_obs, biot);
    hipLaunchKernelGGL(hmm_viterbi_kernel, dim3((unsigned)((n_seq + VIT_GROUPS - 1) / VIT_GROUPS)), dim3(64), 0, s, d, biot, seq_off,
                       n_seq, psi, states, loglik);
}

}  // namespace csv
