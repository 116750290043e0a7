/*
 * scan_oracle.c -- plain-C restatement of the XFMamba selective scan, forward and backward.
 *
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Linked only by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg (through oracle/c_scan.py).  The product library
 * (xfmamba_amd/csrc) never links or calls it.
 *
 * Follows the reference's CPU path `selective_scan_torch` (models/csms6s.py:25-68) for the
 * forward, and for the backward the closed form that autograd through that function yields,
 * which is also what the reference's CUDA kernel evaluates
 * (models/selective_scan/csrc/selective_scan/selective_scan_bwd_kernel.cuh:141-273;
 * SURVEY.md appendix B).  Inputs are fp32 arrays (16-bit model inputs are up-cast exactly by
 * the caller, csms6s.py:52); all internal arithmetic is double so this file can adjudicate
 * between two fp32 implementations.  Pinned against golden vectors produced by the real
 * reference: tests/test_oracle_golden.py.
 *
 * Layouts (contiguous):  u, delta, out, dout, du, ddelta : (B, KD, L)
 *                        A, dA : (KD, N)      Bm, Cm, dB, dC : (B, K, N, L)
 *                        D, delta_bias, dD, ddelta_bias : (KD)   (NULL = absent)
 * Rows d of group k = d / (KD/K) share Bm[b,k], Cm[b,k]           (csms6s.py:53-54).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

static double softplus20(double x) { return x <= 20.0 ? log1p(exp(x)) : x; } /* torch softplus, threshold 20 */

int xfm_oracle_scan_fwd(const float *u, const float *delta, const float *A, const float *Bm, const float *Cm,
                        const float *D, const float *delta_bias, int delta_softplus, float *out, int B, int KD,
                        int K, int N, int L) {
    if (KD % K) return -1;
    const int Dg = KD / K;
    double *h = (double *)malloc(sizeof(double) * (size_t)N);
    if (!h) return -2;
    for (int b = 0; b < B; ++b)
        for (int r = 0; r < KD; ++r) {
            const int k = r / Dg;
            const float *ur = u + ((size_t)b * KD + r) * L, *dr = delta + ((size_t)b * KD + r) * L;
            const float *Bg = Bm + ((size_t)b * K + k) * N * L, *Cg = Cm + ((size_t)b * K + k) * N * L;
            float *yr = out + ((size_t)b * KD + r) * L;
            for (int n = 0; n < N; ++n) h[n] = 0.0;
            for (int t = 0; t < L; ++t) {
                double dl = (double)dr[t] + (delta_bias ? (double)delta_bias[r] : 0.0); /* csms6s.py:47-48 */
                if (delta_softplus) dl = softplus20(dl);                                /* :49-50 */
                const double du = dl * (double)ur[t];
                double y = 0.0;
                for (int n = 0; n < N; ++n) {
                    const double a = exp(dl * (double)A[(size_t)r * N + n]);            /* :55 */
                    h[n] = a * h[n] + du * (double)Bg[(size_t)n * L + t];               /* :56,62 */
                    y += h[n] * (double)Cg[(size_t)n * L + t];                          /* :63 */
                }
                if (D) y += (double)D[r] * (double)ur[t];                               /* :67 */
                yr[t] = (float)y;
            }
        }
    free(h);
    return 0;
}

/* Backward.  dB/dC/dA/dD/ddelta_bias are accumulated in double and written once. */
int xfm_oracle_scan_bwd(const float *u, const float *delta, const float *A, const float *Bm, const float *Cm,
                        const float *D, const float *delta_bias, const float *dout, int delta_softplus, float *du,
                        float *ddelta, float *dA, float *dB, float *dC, float *dD, float *ddelta_bias, int B, int KD,
                        int K, int N, int L) {
    if (KD % K) return -1;
    const int Dg = KD / K;
    const size_t nBC = (size_t)B * K * N * L;
    double *h = (double *)malloc(sizeof(double) * (size_t)N * L);     /* h_t[n] history of one row */
    double *a = (double *)malloc(sizeof(double) * (size_t)N * L);
    double *dl = (double *)malloc(sizeof(double) * (size_t)L);
    double *raw = (double *)malloc(sizeof(double) * (size_t)L);
    double *accA = (double *)calloc((size_t)KD * N, sizeof(double));
    double *accB = (double *)calloc(nBC, sizeof(double));
    double *accC = (double *)calloc(nBC, sizeof(double));
    double *accD = (double *)calloc((size_t)KD, sizeof(double));
    double *accb = (double *)calloc((size_t)KD, sizeof(double));
    double *dh = (double *)malloc(sizeof(double) * (size_t)N);
    if (!h || !a || !dl || !raw || !accA || !accB || !accC || !accD || !accb || !dh) return -2;
    for (int b = 0; b < B; ++b)
        for (int r = 0; r < KD; ++r) {
            const int k = r / Dg;
            const size_t ro = ((size_t)b * KD + r) * L, go = ((size_t)b * K + k) * N * L;
            const float *ur = u + ro, *dr = delta + ro, *gr = dout + ro;
            const float *Bg = Bm + go, *Cg = Cm + go;
            for (int t = 0; t < L; ++t) {
                raw[t] = (double)dr[t] + (delta_bias ? (double)delta_bias[r] : 0.0);
                dl[t] = delta_softplus ? softplus20(raw[t]) : raw[t];
            }
            for (int n = 0; n < N; ++n) { /* forward states */
                double hp = 0.0;
                for (int t = 0; t < L; ++t) {
                    const double at = exp(dl[t] * (double)A[(size_t)r * N + n]);
                    hp = at * hp + dl[t] * (double)ur[t] * (double)Bg[(size_t)n * L + t];
                    a[(size_t)n * L + t] = at;
                    h[(size_t)n * L + t] = hp;
                }
                dh[n] = 0.0; /* holds a_{t+1} * dh_{t+1} while walking back */
            }
            for (int t = L - 1; t >= 0; --t) {
                double s1 = 0.0, s2 = 0.0;
                for (int n = 0; n < N; ++n) {
                    const size_t i = (size_t)n * L + t;
                    const double cur = (double)Cg[i] * (double)gr[t] + dh[n];       /* dh_t */
                    const double bt = dl[t] * (double)ur[t] * (double)Bg[i];
                    const double ah = h[i] - bt;                                    /* a_t * h_{t-1} */
                    s1 += cur * (double)Bg[i];
                    s2 += cur * (double)A[(size_t)r * N + n] * ah;
                    accA[(size_t)r * N + n] += cur * dl[t] * ah;
                    accB[go + i] += cur * dl[t] * (double)ur[t];
                    accC[go + i] += (double)gr[t] * h[i];
                    dh[n] = a[i] * cur;
                }
                double dut = dl[t] * s1;
                if (D) {
                    dut += (double)D[r] * (double)gr[t];
                    accD[r] += (double)gr[t] * (double)ur[t];
                }
                double ddl = (double)ur[t] * s1 + s2;
                if (delta_softplus && raw[t] <= 20.0) ddl *= 1.0 / (1.0 + exp(-raw[t]));
                du[ro + t] = (float)dut;
                ddelta[ro + t] = (float)ddl;
                accb[r] += ddl;
            }
        }
    for (size_t i = 0; i < (size_t)KD * N; ++i) dA[i] = (float)accA[i];
    for (size_t i = 0; i < nBC; ++i) { dB[i] = (float)accB[i]; dC[i] = (float)accC[i]; }
    for (int r = 0; r < KD; ++r) {
        if (dD) dD[r] = (float)accD[r];
        if (ddelta_bias) ddelta_bias[r] = (float)accb[r];
    }
    free(h); free(a); free(dl); free(raw); free(accA); free(accB); free(accC); free(accD); free(accb); free(dh);
    return 0;
}
