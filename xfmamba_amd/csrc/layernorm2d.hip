// layernorm2d.hip -- LayerNorm over the channel axis of an NCHW tensor, in place of the reference's
// permute -> F.layer_norm -> permute (LayerNorm2d, models/fusion_vmamba.py:52-57).
//
// The reference pays two layout copies around a library LayerNorm; here lanes run along the contiguous
// H*W axis (coalesced 256-byte rows per wavefront), the four waves of a workgroup split the channels, and
// the per-position statistics are combined through LDS.  Forward: mean pass, centred-variance pass,
// normalise pass (the 2nd/3rd read hit L2).  Backward: one kernel for dx (same tiling), one for the
// per-channel dweight/dbias reductions (planes are contiguous per channel, so those are plain streaming
// sums + one atomic per workgroup).  HBM-bound: algorithmic bytes = 1 read + 1 write (fwd),
// 2 reads + 1 write (+ 2 re-reads for dw/db) (bwd).
#include "xfm_common.hpp"
#include <type_traits>
#include <cstdlib>

namespace xfm {

// NW waves per workgroup split the channels (4 for narrow rows; 16 for wide ones, where a sample's 64-position tile is
// hundreds of KB and four waves with one load in flight each left the kernel latency-bound: 1024- to 2048-channel rows
// of XFMamba-S / -B).  The channel loops run UNR loads ahead.
template <typename Tx, typename Ty, int NW>
__global__ void __launch_bounds__(64 * NW) ln2d_fwd_kernel(const Tx *__restrict__ x, const float *__restrict__ w,
                                                          const float *__restrict__ bias, Ty *__restrict__ y,
                                                          float *__restrict__ mean, float *__restrict__ rstd, int C,
                                                          int L, int tiles_pb, float eps) {
    constexpr int UNR = 4;
    __shared__ float red[NW][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / tiles_pb, p = (blockIdx.x - b * tiles_pb) * 64 + lane;
    const bool ok = p < L;
    const Tx *xb = x + (int64_t)b * C * L + p;
    float s = 0.f;
    if (ok) {
        int c = wave;
        for (; c + (UNR - 1) * NW < C; c += UNR * NW) {
            float t[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) t[q] = ldf<Tx>(xb + (int64_t)(c + q * NW) * L);
#pragma unroll
            for (int q = 0; q < UNR; ++q) s += t[q];
        }
        for (; c < C; c += NW) s += ldf<Tx>(xb + (int64_t)c * L);
    }
    red[wave][lane] = s;
    __syncthreads();
    float mu = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) mu += red[q][lane];
    mu /= (float)C;
    __syncthreads();
    float v = 0.f;
    if (ok) {
        int c = wave;
        for (; c + (UNR - 1) * NW < C; c += UNR * NW) {
            float t[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) t[q] = ldf<Tx>(xb + (int64_t)(c + q * NW) * L) - mu;
#pragma unroll
            for (int q = 0; q < UNR; ++q) v = fmaf(t[q], t[q], v);
        }
        for (; c < C; c += NW) {
            const float d = ldf<Tx>(xb + (int64_t)c * L) - mu;
            v = fmaf(d, d, v);
        }
    }
    red[wave][lane] = v;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) var += red[q][lane];
    const float rs = rsqrtf(var / (float)C + eps);
    if (ok) {
        if (wave == 0) {
            mean[(int64_t)b * L + p] = mu;
            rstd[(int64_t)b * L + p] = rs;
        }
        Ty *yb = y + (int64_t)b * C * L + p;
        int c = wave;
        for (; c + (UNR - 1) * NW < C; c += UNR * NW) {
            float t[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) t[q] = ldf<Tx>(xb + (int64_t)(c + q * NW) * L);
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                const int cc = c + q * NW;
                stf<Ty>(yb + (int64_t)cc * L, fmaf((t[q] - mu) * rs, w[cc], bias ? bias[cc] : 0.f));
            }
        }
        for (; c < C; c += NW) {
            const float xh = (ldf<Tx>(xb + (int64_t)c * L) - mu) * rs;
            stf<Ty>(yb + (int64_t)c * L, fmaf(xh, w[c], bias ? bias[c] : 0.f));
        }
    }
}

// dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),  g = dy * w
template <typename Tx, typename Ty, int NW>
__global__ void __launch_bounds__(64 * NW) ln2d_bwd_dx_kernel(const Tx *__restrict__ x, const float *__restrict__ w,
                                                             const Ty *__restrict__ dy, const float *__restrict__ mean,
                                                             const float *__restrict__ rstd, Tx *__restrict__ dx, int C,
                                                             int L, int tiles_pb) {
    constexpr int UNR = 4;
    __shared__ float red[2][NW][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x / tiles_pb, p = (blockIdx.x - b * tiles_pb) * 64 + lane;
    const bool ok = p < L;
    const int64_t o = (int64_t)b * C * L + p;
    const float mu = ok ? mean[(int64_t)b * L + p] : 0.f, rs = ok ? rstd[(int64_t)b * L + p] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    if (ok) {
        int c = wave;
        for (; c + (UNR - 1) * NW < C; c += UNR * NW) {
            float g[UNR], xv[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                g[q] = ldf<Ty>(dy + o + (int64_t)(c + q * NW) * L);
                xv[q] = ldf<Tx>(x + o + (int64_t)(c + q * NW) * L);
            }
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                const float gg = g[q] * w[c + q * NW];
                s1 += gg;
                s2 = fmaf(gg, (xv[q] - mu) * rs, s2);
            }
        }
        for (; c < C; c += NW) {
            const float g = ldf<Ty>(dy + o + (int64_t)c * L) * w[c];
            const float xh = (ldf<Tx>(x + o + (int64_t)c * L) - mu) * rs;
            s1 += g;
            s2 = fmaf(g, xh, s2);
        }
    }
    red[0][wave][lane] = s1;
    red[1][wave][lane] = s2;
    __syncthreads();
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) {
        m1 += red[0][q][lane];
        m2 += red[1][q][lane];
    }
    m1 /= (float)C;
    m2 /= (float)C;
    if (ok) {
        int c = wave;
        for (; c + (UNR - 1) * NW < C; c += UNR * NW) {
            float g[UNR], xv[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                g[q] = ldf<Ty>(dy + o + (int64_t)(c + q * NW) * L);
                xv[q] = ldf<Tx>(x + o + (int64_t)(c + q * NW) * L);
            }
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                const int cc = c + q * NW;
                const float xh = (xv[q] - mu) * rs;
                stf<Tx>(dx + o + (int64_t)cc * L, rs * (g[q] * w[cc] - m1 - xh * m2));
            }
        }
        for (; c < C; c += NW) {
            const float g = ldf<Ty>(dy + o + (int64_t)c * L) * w[c];
            const float xh = (ldf<Tx>(x + o + (int64_t)c * L) - mu) * rs;
            stf<Tx>(dx + o + (int64_t)c * L, rs * (g - m1 - xh * m2));
        }
    }
}

// ---- short maps with wide rows (the 7 x 7 stage: 768 ... 1536 channels, a few thousand positions in all).  Lanes along the
// positions leave the chip mostly idle there (64 x 49 positions = 49 workgroups) and the register-cached form needs 2 x 48
// values per thread under the 128-register cap of a 16-wave workgroup (58 spills, 65 us for 24 MB).  Here a workgroup owns
// 16 positions and splits the channels 16 ways across its threads; two passes over L2-resident data, no caching.
template <typename Tx, typename Ty>
__global__ void __launch_bounds__(256) ln2d_bwd_dx_short_kernel(const Tx *__restrict__ x, const float *__restrict__ w,
                                                                const Ty *__restrict__ dy, const float *__restrict__ mean,
                                                                const float *__restrict__ rstd, Tx *__restrict__ dx, int C,
                                                                int L, int NP) {
    constexpr int UNR = 4;
    __shared__ float red[2][16][17];
    const int pi = threadIdx.x & 15, cs = threadIdx.x >> 4;
    const int P = blockIdx.x * 16 + pi;
    const bool ok = P < NP;
    const int b = ok ? P / L : 0, p = ok ? P - b * L : 0;
    const int64_t o = (int64_t)b * C * L + p;
    const float mu = ok ? mean[(int64_t)b * L + p] : 0.f, rs = ok ? rstd[(int64_t)b * L + p] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    if (ok) {
        int c = cs;
        for (; c + (UNR - 1) * 16 < C; c += UNR * 16) {
            float g[UNR], xv[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                g[q] = ldf<Ty>(dy + o + (int64_t)(c + q * 16) * L);
                xv[q] = ldf<Tx>(x + o + (int64_t)(c + q * 16) * L);
            }
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                const float gg = g[q] * w[c + q * 16];
                s1 += gg;
                s2 = fmaf(gg, (xv[q] - mu) * rs, s2);
            }
        }
        for (; c < C; c += 16) {
            const float g = ldf<Ty>(dy + o + (int64_t)c * L) * w[c];
            s1 += g;
            s2 = fmaf(g, (ldf<Tx>(x + o + (int64_t)c * L) - mu) * rs, s2);
        }
    }
    red[0][cs][pi] = s1;
    red[1][cs][pi] = s2;
    __syncthreads();
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        m1 += red[0][q][pi];
        m2 += red[1][q][pi];
    }
    m1 /= (float)C;
    m2 /= (float)C;
    if (ok) {
        int c = cs;
        for (; c + (UNR - 1) * 16 < C; c += UNR * 16) {
            float g[UNR], xv[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                g[q] = ldf<Ty>(dy + o + (int64_t)(c + q * 16) * L);
                xv[q] = ldf<Tx>(x + o + (int64_t)(c + q * 16) * L);
            }
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                const int cc = c + q * 16;
                stf<Tx>(dx + o + (int64_t)cc * L, rs * (g[q] * w[cc] - m1 - (xv[q] - mu) * rs * m2));
            }
        }
        for (; c < C; c += 16) {
            const float g = ldf<Ty>(dy + o + (int64_t)c * L) * w[c];
            stf<Tx>(dx + o + (int64_t)c * L, rs * (g - m1 - (ldf<Tx>(x + o + (int64_t)c * L) - mu) * rs * m2));
        }
    }
}

// ---- the same maps, register-cached (round 4).  A workgroup owns EIGHT positions and splits the channels 32 ways: thread
// (position pi, slice cs) keeps channels cs + 32 j, j < CPT = C / 32, in registers, so x (and dy) are read once and 1568
// positions make 196 workgroups (the lanes-along-positions forms launch 25 for a 32 x 768 x 7 x 7 map: 27 us for 4.8 MB).
// The backward kernel can leave the weight / bias gradient's partial rows ([sum dy * xhat | sum dy] over its 8 positions).
// (Tried instead: ONE 1024-thread workgroup per sample reading its (C, L) block as a flat, fully coalesced run with the values
//  in registers -- 46 / 77 us against 33 / 56 us here at 96 x 1536 x 7 x 7: 96 workgroups leave 160 CUs idle and a workgroup's
//  77 loads per thread go out in register-limited batches.)
template <typename Tx, typename Ty, int CPT>
__global__ void __launch_bounds__(256) ln2d_fwd_short_kernel(const Tx *__restrict__ x, const float *__restrict__ w,
                                                             const float *__restrict__ bias, Ty *__restrict__ y,
                                                             float *__restrict__ mean, float *__restrict__ rstd, int C, int L,
                                                             int NP, float eps) {
    __shared__ float red[32][9];
    const int pi = threadIdx.x & 7, cs = threadIdx.x >> 3;
    const int P = blockIdx.x * 8 + pi;
    const bool ok = P < NP;
    const int b = ok ? P / L : 0, p = ok ? P - b * L : 0;
    const int64_t o = (int64_t)b * C * L + p;
    float v[CPT];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const float ld = ldf<Tx>(x + o + (int64_t)(cs + 32 * j) * L);      // (always issued: a load under a lane condition sits in
        v[j] = ok ? ld : 0.f;                                             //  its own block and the join waits for it -- 48 serial loads)
        s += v[j];
    }
    red[cs][pi] = s;
    __syncthreads();
    float mu = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) mu += red[q][pi];
    mu /= (float)C;
    __syncthreads();
    float q2 = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        v[j] -= mu;
        q2 = fmaf(v[j], v[j], q2);
    }
    red[cs][pi] = q2;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) var += red[q][pi];
    const float rs = rsqrtf(var / (float)C + eps);
    if (!ok) return;
    if (cs == 0) {
        mean[(int64_t)b * L + p] = mu;
        rstd[(int64_t)b * L + p] = rs;
    }
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = cs + 32 * j;
        stf<Ty>(y + o + (int64_t)c * L, fmaf(v[j] * rs, w[c], bias ? bias[c] : 0.f));
    }
}

template <typename Tx, typename Ty, int CPT>
__global__ void __launch_bounds__(256) ln2d_bwd_dx_short8_kernel(const Tx *__restrict__ x, const float *__restrict__ w,
                                                                 const Ty *__restrict__ dy, const float *__restrict__ mean,
                                                                 const float *__restrict__ rstd, Tx *__restrict__ dx, int C,
                                                                 int L, int NP, float *__restrict__ parts) {
    __shared__ float red[2][32][9];
    const int pi = threadIdx.x & 7, cs = threadIdx.x >> 3;
    const int P = blockIdx.x * 8 + pi;
    const bool ok = P < NP;
    const int b = ok ? P / L : 0, p = ok ? P - b * L : 0;
    const int64_t o = (int64_t)b * C * L + p;
    const float mu = ok ? mean[(int64_t)b * L + p] : 0.f, rs = ok ? rstd[(int64_t)b * L + p] : 0.f;
    float g[CPT], xh[CPT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int64_t a = o + (int64_t)(cs + 32 * j) * L;
        const float gl = ldf<Ty>(dy + a), xl = ldf<Tx>(x + a);           // (always issued, see the forward kernel)
        g[j] = ok ? gl : 0.f;
        xh[j] = ok ? (xl - mu) * rs : 0.f;
    }
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const float gw = g[j] * w[cs + 32 * j];
        s1 += gw;
        s2 = fmaf(gw, xh[j], s2);
    }
    red[0][cs][pi] = s1;
    red[1][cs][pi] = s2;
    __syncthreads();
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        m1 += red[0][q][pi];
        m2 += red[1][q][pi];
    }
    m1 /= (float)C;
    m2 /= (float)C;
    if (ok) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = cs + 32 * j;
            stf<Tx>(dx + o + (int64_t)c * L, rs * (g[j] * w[c] - m1 - xh[j] * m2));
        }
    }
    if (parts) {
        // the eight positions of a channel are eight adjacent lanes: three exchange steps, lane pi == 0 writes the pair
        float *pr = parts + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            float a1 = g[j] * xh[j], a2 = g[j];
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) {
                a1 += __shfl_xor(a1, off, 64);
                a2 += __shfl_xor(a2, off, 64);
            }
            if (pi == 0) {
                pr[cs + 32 * j] = a1;
                pr[C + cs + 32 * j] = a2;
            }
        }
    }
}

// ---- 7 x 7 maps on the whole chip, COALESCED: the slab form (round 4).  A workgroup owns a (sample, 64-channel slab): its
// 64 L elements are one flat run; thread t = q L + p (q < 5 channel groups, p < L <= 51 positions) reads element t + 5 L j of
// the run in step j (channel q + 5 j of the slab, ITS position p): contiguous across the workgroup, register accumulators per
// position, a 5-way fold through LDS.  The statistics over all C channels need the other slabs: kernel 1 leaves one partial
// pair per (sample, slab, position) in a workspace, kernel 2 folds the C / 64 pairs of its sample (L2 reads) and applies.
// Forward pairs: [sum (x - s) | sum (x - s)^2] with the shift s = x[b, 0, p] (no cancellation when |mean| >> std);
// backward pairs: [sum g w | sum g w xhat].  The backward's second kernel also leaves the weight / bias gradient's partial
// rows, one row pair per SAMPLE.  B * C / 64 workgroups (2304 at 96 x 1536) instead of 96 or position slices of 32 bytes.
constexpr int kSlabC = 64, kSlabG = 5, kSlabJ = (kSlabC + kSlabG - 1) / kSlabG;      // 13 steps

template <typename Tx>
__global__ void __launch_bounds__(256) ln2d_slab_stats_kernel(const Tx *__restrict__ x, float *__restrict__ ws, int C, int L) {
    __shared__ float red[2][kSlabG][52];
    const int t = threadIdx.x, slab = blockIdx.x, b = blockIdx.y, nslab = C / kSlabC;
    const bool act = t < kSlabG * L;
    const int q = act ? t / L : 0, p = act ? t - q * L : 0, GL = kSlabG * L;
    const Tx *xs = x + ((int64_t)b * C + (int64_t)slab * kSlabC) * L;
    const float sh = ldf<Tx>(x + (int64_t)b * C * L + p);
    float s = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < kSlabJ; ++j) {
        const bool ok = act && q + kSlabG * j < kSlabC;
        const float v = ldf<Tx>(xs + (ok ? t + GL * j : 0)) - sh;
        s += ok ? v : 0.f;
        s2 += ok ? v * v : 0.f;
    }
    if (act) {
        red[0][q][p] = s;
        red[1][q][p] = s2;
    }
    __syncthreads();
    if (t < 2 * L) {
        const int which = t >= L, pp = which ? t - L : t;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < kSlabG; ++k) a += red[which][k][pp];
        ws[(((int64_t)b * nslab + slab) * 2 + which) * L + pp] = a;
    }
}

// the nslab partial pairs of a sample, folded by all five thread groups (group q takes slabs q, q + 5, ...: the loads of a
// group's few steps are independent; one thread per position walking all 24 slabs was 17 us of serial L2 latency)
__device__ __forceinline__ void ln2d_slab_fold(const float *__restrict__ pw, int nslab, int L, int t,
                                               float (&fold)[2][kSlabG][52]) {
    if (t < kSlabG * L) {
        const int q = t / L, p = t - q * L;
        float a = 0.f, a2 = 0.f;
#pragma unroll 8
        for (int k = q; k < nslab; k += kSlabG) {
            a += pw[(int64_t)(2 * k) * L + p];
            a2 += pw[(int64_t)(2 * k + 1) * L + p];
        }
        fold[0][q][p] = a;
        fold[1][q][p] = a2;
    }
    __syncthreads();
}

template <typename Tx, typename Ty>
__global__ void __launch_bounds__(256) ln2d_slab_apply_kernel(const Tx *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ bias, Ty *__restrict__ y,
                                                              float *__restrict__ mean, float *__restrict__ rstd,
                                                              const float *__restrict__ ws, int C, int L, float eps) {
    __shared__ float st[2][52];
    __shared__ float fold[2][kSlabG][52];
    const int t = threadIdx.x, slab = blockIdx.x, b = blockIdx.y, nslab = C / kSlabC;
    ln2d_slab_fold(ws + (int64_t)b * nslab * 2 * L, nslab, L, t, fold);
    if (t < L) {
        float a = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k < kSlabG; ++k) {
            a += fold[0][k][t];
            a2 += fold[1][k][t];
        }
        const float sh = ldf<Tx>(x + (int64_t)b * C * L + t);
        const float d = a / (float)C;                       // mean - shift
        const float var = fmaxf(a2 / (float)C - d * d, 0.f);
        const float mu = sh + d, rs = rsqrtf(var + eps);
        st[0][t] = mu;
        st[1][t] = rs;
        if (slab == 0) {
            mean[(int64_t)b * L + t] = mu;
            rstd[(int64_t)b * L + t] = rs;
        }
    }
    __syncthreads();
    if (t >= kSlabG * L) return;
    const int q = t / L, p = t - q * L, GL = kSlabG * L;
    const float mu = st[0][p], rs = st[1][p];
    const int64_t o = ((int64_t)b * C + (int64_t)slab * kSlabC) * L + t;
#pragma unroll
    for (int j = 0; j < kSlabJ; ++j) {
        const int cl = q + kSlabG * j;
        if (cl < kSlabC) {
            const int c = slab * kSlabC + cl;
            stf<Ty>(y + o + GL * j, fmaf((ldf<Tx>(x + o + GL * j) - mu) * rs, w[c], bias ? bias[c] : 0.f));
        }
    }
}

template <typename Tx, typename Ty>
__global__ void __launch_bounds__(256) ln2d_slab_bwd_stats_kernel(const Tx *__restrict__ x, const float *__restrict__ w,
                                                                  const Ty *__restrict__ dy, const float *__restrict__ mean,
                                                                  const float *__restrict__ rstd, float *__restrict__ ws, int C,
                                                                  int L) {
    __shared__ float red[2][kSlabG][52];
    const int t = threadIdx.x, slab = blockIdx.x, b = blockIdx.y, nslab = C / kSlabC;
    const bool act = t < kSlabG * L;
    const int q = act ? t / L : 0, p = act ? t - q * L : 0, GL = kSlabG * L;
    const int64_t o = ((int64_t)b * C + (int64_t)slab * kSlabC) * L;
    const float mu = mean[(int64_t)b * L + p], rs = rstd[(int64_t)b * L + p];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < kSlabJ; ++j) {
        const int cl = q + kSlabG * j;
        const bool ok = act && cl < kSlabC;
        const int64_t a = o + (ok ? t + GL * j : 0);
        const float g = ldf<Ty>(dy + a) * w[slab * kSlabC + (ok ? cl : 0)];
        const float xh = (ldf<Tx>(x + a) - mu) * rs;
        s1 += ok ? g : 0.f;
        s2 += ok ? g * xh : 0.f;
    }
    if (act) {
        red[0][q][p] = s1;
        red[1][q][p] = s2;
    }
    __syncthreads();
    if (t < 2 * L) {
        const int which = t >= L, pp = which ? t - L : t;
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < kSlabG; ++k) a += red[which][k][pp];
        ws[(((int64_t)b * nslab + slab) * 2 + which) * L + pp] = a;
    }
}

template <typename Tx, typename Ty>
__global__ void __launch_bounds__(256) ln2d_slab_bwd_apply_kernel(const Tx *__restrict__ x, const float *__restrict__ w,
                                                                  const Ty *__restrict__ dy, const float *__restrict__ mean,
                                                                  const float *__restrict__ rstd, Tx *__restrict__ dx,
                                                                  const float *__restrict__ ws, float *__restrict__ parts, int C,
                                                                  int L) {
    __shared__ float st[2][52];
    __shared__ float fold[2][kSlabG][52];
    __shared__ float col[2][kSlabC * 51];                   // [dy xhat | dy][channel of the slab][position]: odd pitch L
    const int t = threadIdx.x, slab = blockIdx.x, b = blockIdx.y, nslab = C / kSlabC;
    ln2d_slab_fold(ws + (int64_t)b * nslab * 2 * L, nslab, L, t, fold);
    if (t < L) {
        float a = 0.f, a2 = 0.f;
#pragma unroll
        for (int k = 0; k < kSlabG; ++k) {
            a += fold[0][k][t];
            a2 += fold[1][k][t];
        }
        st[0][t] = a / (float)C;
        st[1][t] = a2 / (float)C;
    }
    __syncthreads();
    const bool act = t < kSlabG * L;
    const int q = act ? t / L : 0, p = act ? t - q * L : 0, GL = kSlabG * L;
    const float mu = mean[(int64_t)b * L + p], rs = rstd[(int64_t)b * L + p];
    const float m1 = st[0][p], m2 = st[1][p];
    const int64_t o = ((int64_t)b * C + (int64_t)slab * kSlabC) * L + t;
    if (act) {
#pragma unroll
        for (int j = 0; j < kSlabJ; ++j) {
            const int cl = q + kSlabG * j;
            if (cl < kSlabC) {
                const float gr = ldf<Ty>(dy + o + GL * j);
                const float xh = (ldf<Tx>(x + o + GL * j) - mu) * rs;
                stf<Tx>(dx + o + GL * j, rs * (gr * w[slab * kSlabC + cl] - m1 - xh * m2));
                if (parts) {
                    col[0][cl * L + p] = gr * xh;
                    col[1][cl * L + p] = gr;
                }
            }
        }
    }
    if (!parts) return;
    __syncthreads();
    if (t < 2 * kSlabC) {
        const int which = t >= kSlabC, cl = which ? t - kSlabC : t;
        const float *cp = col[which] + cl * L;
        float a = 0.f;
        for (int k = 0; k < L; ++k) a += cp[k];
        parts[((int64_t)b * 2 + which) * C + slab * kSlabC + cl] = a;
    }
}

static bool ln2d_slab_ok(int B, int C, int L) {
    return L >= 16 && L <= 51 && C >= 512 && C % kSlabC == 0 && (int64_t)B * L <= 16 * 1024 && !getenv("XFM_LN2D_NO_SLAB");
}

static bool ln2d_short8_ok(int B, int C, int L) {
    return C >= 512 && (int64_t)B * L <= 16 * 1024 && C % 32 == 0 && (C == 512 || C == 768 || C == 1024 || C == 1536) &&
           !getenv("XFM_LN2D_NO_SHORT8");
}

// ---------------------------------------------------------------------------------------------
// register-cached variants: C == CPT * NW.  The workgroup has NW waves (<= 16); lane = position, each thread
// keeps its CPT channel values in registers, so x (and dy) are read from HBM exactly once.
// ---------------------------------------------------------------------------------------------
// HALF: a workgroup covers 32 positions instead of 64 -- the lower lanes take the even, the upper lanes the odd channels of
// their wave -- so that maps with few positions (14 x 14: 12544 = 196 tiles of 64 for 256 CUs) launch twice the workgroups.
template <typename Tx, typename Ty, int CPT, bool HALF = false>
__global__ void __launch_bounds__(1024) ln2d_fwd_cached_kernel(const Tx *__restrict__ x, const float *__restrict__ w,
                                                               const float *__restrict__ bias, Ty *__restrict__ y,
                                                               float *__restrict__ mean, float *__restrict__ rstd,
                                                               int C, int L, int tiles_pb, float eps, int NW) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl = HALF ? (lane & 31) : lane, hh = HALF ? (lane >> 5) : 0;
    constexpr int PPW = HALF ? 32 : 64;
    auto chan = [&](int j) { return HALF ? (wave + j * NW) * 2 + hh : wave + j * NW; };
    // positions of all images form one index space (tiles_pb = B*L here): no ragged last tile per image
    const int64_t P = (int64_t)blockIdx.x * PPW + pl;
    const bool ok = P < tiles_pb;
    const int b = ok ? (int)(P / L) : 0, p = ok ? (int)(P - (int64_t)b * L) : 0;
    const int64_t o = (int64_t)b * C * L + p;
    float v[CPT];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        v[j] = ok ? ldf<Tx>(x + o + (int64_t)chan(j) * L) : 0.f;
        s += v[j];
    }
    red[wave][lane] = s;
    __syncthreads();
    float mu = 0.f;
    for (int q = 0; q < NW; ++q) mu += red[q][pl] + (HALF ? red[q][pl + 32] : 0.f);
    mu /= (float)C;
    __syncthreads();
    float q2 = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        v[j] -= mu;
        q2 = fmaf(v[j], v[j], q2);
    }
    red[wave][lane] = q2;
    __syncthreads();
    float var = 0.f;
    for (int q = 0; q < NW; ++q) var += red[q][pl] + (HALF ? red[q][pl + 32] : 0.f);
    const float rs = rsqrtf(var / (float)C + eps);
    if (ok) {
        if (wave == 0 && hh == 0) {
            mean[(int64_t)b * L + p] = mu;
            rstd[(int64_t)b * L + p] = rs;
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = chan(j);
            stf<Ty>(y + o + (int64_t)c * L, fmaf(v[j] * rs, w[c], bias ? bias[c] : 0.f));
        }
    }
}

// sums over the lanes of a wave on DPP adds: row-wise inclusive prefixes (row_shr 1, 2, 4, 8), row totals carried into the next
// row (row_bcast:15) and, for the whole wave, into rows 2 / 3 (row_bcast:31).  HALVES: lane 31 ends with the total of lanes
// 0..31, lane 63 with that of lanes 32..63; otherwise lane 63 with the total of all 64.
template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ float ln2d_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <bool HALVES> __device__ __forceinline__ float ln2d_lane_sum(float v) {
    v += ln2d_dpp<0x111>(v);
    v += ln2d_dpp<0x112>(v);
    v += ln2d_dpp<0x114>(v);
    v += ln2d_dpp<0x118>(v);
    v += ln2d_dpp<0x142, 0xa>(v);
    if constexpr (!HALVES) v += ln2d_dpp<0x143, 0xc>(v);
    return v;
}

// `parts` != null: the weight / bias gradient of the workgroup's positions as a partial row pair -- parts[(blk * 2 + k) * C + c],
// k = 0: sum dy * xhat, k = 1: sum dy -- instead of a second kernel reading x and dy again (ln2d_bwd_wb_kernel); the rows are
// folded by xfm_partial_sums_multi (deferred.py) or a sum over blocks.
template <typename Tx, typename Ty, int CPT, bool HALF = false>
__global__ void __launch_bounds__(1024) ln2d_bwd_dx_cached_kernel(const Tx *__restrict__ x, const float *__restrict__ w,
                                                                  const Ty *__restrict__ dy, const float *__restrict__ mean,
                                                                  const float *__restrict__ rstd, Tx *__restrict__ dx,
                                                                  int C, int L, int tiles_pb, int NW, float *__restrict__ parts) {
    __shared__ float red[2][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl = HALF ? (lane & 31) : lane, hh = HALF ? (lane >> 5) : 0;
    constexpr int PPW = HALF ? 32 : 64;
    auto chan = [&](int j) { return HALF ? (wave + j * NW) * 2 + hh : wave + j * NW; };
    // positions of all images form one index space (tiles_pb = B*L here): no ragged last tile per image
    const int64_t P = (int64_t)blockIdx.x * PPW + pl;
    const bool ok = P < tiles_pb;
    const int b = ok ? (int)(P / L) : 0, p = ok ? (int)(P - (int64_t)b * L) : 0;
    const int64_t o = (int64_t)b * C * L + p;
    const float mu = ok ? mean[(int64_t)b * L + p] : 0.f, rs = ok ? rstd[(int64_t)b * L + p] : 0.f;
    float gr[CPT], xh[CPT];                                        // RAW dy (the weight is applied at each use) and xhat
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = chan(j);
        gr[j] = ok ? ldf<Ty>(dy + o + (int64_t)c * L) : 0.f;
        xh[j] = ok ? (ldf<Tx>(x + o + (int64_t)c * L) - mu) * rs : 0.f;
        const float g = gr[j] * w[c];
        s1 += g;
        s2 = fmaf(g, xh[j], s2);
    }
    red[0][wave][lane] = s1;
    red[1][wave][lane] = s2;
    __syncthreads();
    float m1 = 0.f, m2 = 0.f;
    for (int q = 0; q < NW; ++q) {
        m1 += red[0][q][pl] + (HALF ? red[0][q][pl + 32] : 0.f);
        m2 += red[1][q][pl] + (HALF ? red[1][q][pl + 32] : 0.f);
    }
    m1 /= (float)C;
    m2 /= (float)C;
    if (ok) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = chan(j);
            stf<Tx>(dx + o + (int64_t)c * L, rs * (gr[j] * w[c] - m1 - xh[j] * m2));
        }
    }
    if (parts) {                                                    // (uniform)
        // a wave's channels over its positions: lane sums on DPP, gathered per workgroup in LDS, one coalesced row pair out
        __shared__ float prow[2 * 1536];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const float a1 = ln2d_lane_sum<HALF>(gr[j] * xh[j]);
            const float a2 = ln2d_lane_sum<HALF>(gr[j]);
            if (lane == 63 || (HALF && lane == 31)) {
                const int c = chan(j);
                prow[c] = a1;
                prow[C + c] = a2;
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 2 * C; e += blockDim.x) parts[(int64_t)blockIdx.x * 2 * C + e] = prow[e];
    }
}

// ---- vectorised forms of the two cached kernels for fp32 maps / bf16 outputs (the SS2D out_norm under autocast):
// a lane owns VP consecutive positions (16- / 8-byte accesses instead of 4- / 2-byte ones), 16 waves split the channels.
// ---- backward on 14 x 14 maps with 384 channels, split form (round 4).  The register-cached kernel above needs 16 waves per
// workgroup to hold a position's 2 x 384 values: one workgroup per CU, every barrier idles the CU -- 40.8 us for 48 MB
// (1.2 TB/s).  Nothing is cached here: kernel 1 (lane = position, four waves x CPW channels of a 4 CPW-channel slab) leaves
// [sum g w | sum g w xhat] per (slab, position) in a workspace, kernel 2 folds the C / (4 CPW) slab pairs of its position, reads
// x and dy again (just read by kernel 1: the step's 29 MB sit in the 256 MB last-level cache), writes dx and the weight /
// bias partial rows of its 64 positions (wave sums on DPP adds).  256-thread workgroups, 4 x (B L / 64) of them.
template <typename Tx, typename Ty, int CPW>
__global__ void __launch_bounds__(256) ln2d_bwd_split_stats_kernel(const Tx *__restrict__ x, const float *__restrict__ w,
                                                                   const Ty *__restrict__ dy, const float *__restrict__ mean,
                                                                   const float *__restrict__ rstd, float *__restrict__ ws, int C,
                                                                   int L, int NP) {
    __shared__ float red[2][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slab = blockIdx.y;
    const int P = blockIdx.x * 64 + lane;
    const bool ok = P < NP;
    const int b = ok ? P / L : 0, p = ok ? P - b * L : 0;
    const int c0 = (slab * 4 + wave) * CPW;
    const int64_t o = ((int64_t)b * C + c0) * L + p;
    const float mu = mean[(int64_t)b * L + p], rs = rstd[(int64_t)b * L + p];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        const float g = ldf<Ty>(dy + o + (int64_t)j * L) * w[c0 + j];
        const float xh = (ldf<Tx>(x + o + (int64_t)j * L) - mu) * rs;
        s1 += g;
        s2 = fmaf(g, xh, s2);
    }
    red[0][wave][lane] = s1;
    red[1][wave][lane] = s2;
    __syncthreads();
    if (wave < 2 && ok)
        ws[((int64_t)slab * 2 + wave) * NP + P] = (red[wave][0][lane] + red[wave][1][lane]) + (red[wave][2][lane] + red[wave][3][lane]);
}

template <typename Tx, typename Ty, int CPW>
__global__ void __launch_bounds__(256) ln2d_bwd_split_apply_kernel(const Tx *__restrict__ x, const float *__restrict__ w,
                                                                   const Ty *__restrict__ dy, const float *__restrict__ mean,
                                                                   const float *__restrict__ rstd, Tx *__restrict__ dx,
                                                                   const float *__restrict__ ws, float *__restrict__ parts, int C,
                                                                   int L, int NP) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slab = blockIdx.y, nslab = gridDim.y;
    const int P = blockIdx.x * 64 + lane;
    const bool ok = P < NP;
    const int Pc = ok ? P : 0;
    const int b = Pc / L, p = Pc - b * L;
    const int c0 = (slab * 4 + wave) * CPW;
    const int64_t o = ((int64_t)b * C + c0) * L + p;
    const float mu = mean[(int64_t)b * L + p], rs = rstd[(int64_t)b * L + p];
    float m1 = 0.f, m2 = 0.f;
    for (int k = 0; k < nslab; ++k) {
        m1 += ws[((int64_t)k * 2) * NP + Pc];
        m2 += ws[((int64_t)k * 2 + 1) * NP + Pc];
    }
    m1 /= (float)C;
    m2 /= (float)C;
    float *pr = parts ? parts + (int64_t)blockIdx.x * 2 * C : nullptr;
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        const float gr = ldf<Ty>(dy + o + (int64_t)j * L);
        const float xh = (ldf<Tx>(x + o + (int64_t)j * L) - mu) * rs;
        if (ok) stf<Tx>(dx + o + (int64_t)j * L, rs * (gr * w[c0 + j] - m1 - xh * m2));
        if (parts) {
            const float a1 = ln2d_lane_sum<false>(ok ? gr * xh : 0.f);
            const float a2 = ln2d_lane_sum<false>(ok ? gr : 0.f);
            if (lane == 63) {
                pr[c0 + j] = a1;
                pr[C + c0 + j] = a2;
            }
        }
    }
}

// channels per wave of the split form for this shape (24: slabs of 96 channels), or 0
static int ln2d_split_cpw(int B, int C, int L) {
    if (getenv("XFM_LN2D_NO_SPLIT")) return 0;
    return (C == 384 && (int64_t)B * L >= 4096 && L <= 256) ? 24 : 0;
}

template <int VP> __device__ __forceinline__ void ldv_f32(const float *p, float (&v)[VP]) {
    if constexpr (VP == 4) { const float4 t = *reinterpret_cast<const float4 *>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    else { const float2 t = *reinterpret_cast<const float2 *>(p); v[0] = t.x; v[1] = t.y; }
}
template <int VP> __device__ __forceinline__ void stv_f32(float *p, const float (&v)[VP]) {
    if constexpr (VP == 4) *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    else *reinterpret_cast<float2 *>(p) = make_float2(v[0], v[1]);
}
template <int VP> __device__ __forceinline__ void ldv_bf16(const bf16_t *p, float (&v)[VP]) {
    uint32_t w[VP / 2];
    if constexpr (VP == 4) { const uint2 t = *reinterpret_cast<const uint2 *>(p); w[0] = t.x; w[1] = t.y; }
    else w[0] = *reinterpret_cast<const uint32_t *>(p);
#pragma unroll
    for (int i = 0; i < VP / 2; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
template <int VP> __device__ __forceinline__ void stv_bf16(bf16_t *p, const float (&v)[VP]) {
    if constexpr (VP == 4) *reinterpret_cast<uint2 *>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    else *reinterpret_cast<uint32_t *>(p) = pack_bf16x2(v[0], v[1]);
}

template <int CPT, int VP>
__global__ void __launch_bounds__(1024) ln2d_fwd_vec_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                            const float *__restrict__ bias, bf16_t *__restrict__ y,
                                                            float *__restrict__ mean, float *__restrict__ rstd, int C, int L,
                                                            int64_t NP, float eps) {
    constexpr int NW = 16;
    __shared__ float red[NW][64 * VP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t Pv = (int64_t)blockIdx.x * 64 + lane;
    const bool ok = Pv < NP;
    const int64_t P = ok ? Pv * VP : 0;
    const int b = (int)(P / L), p = (int)(P - (int64_t)b * L);
    const int64_t o = (int64_t)b * C * L + p;
    float v[CPT][VP], s[VP];
#pragma unroll
    for (int i = 0; i < VP; ++i) s[i] = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        ldv_f32<VP>(x + o + (int64_t)(wave + j * NW) * L, v[j]);
#pragma unroll
        for (int i = 0; i < VP; ++i) s[i] += v[j][i];
    }
    stv_f32<VP>(&red[wave][lane * VP], s);
    __syncthreads();
    float mu[VP];
#pragma unroll
    for (int i = 0; i < VP; ++i) mu[i] = 0.f;
    for (int q = 0; q < NW; ++q) {
        float t[VP];
        ldv_f32<VP>(&red[q][lane * VP], t);
#pragma unroll
        for (int i = 0; i < VP; ++i) mu[i] += t[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VP; ++i) {
        mu[i] /= (float)C;
        s[i] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < CPT; ++j)
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            v[j][i] -= mu[i];
            s[i] = fmaf(v[j][i], v[j][i], s[i]);
        }
    stv_f32<VP>(&red[wave][lane * VP], s);
    __syncthreads();
    float rs[VP];
#pragma unroll
    for (int i = 0; i < VP; ++i) rs[i] = 0.f;
    for (int q = 0; q < NW; ++q) {
        float t[VP];
        ldv_f32<VP>(&red[q][lane * VP], t);
#pragma unroll
        for (int i = 0; i < VP; ++i) rs[i] += t[i];
    }
#pragma unroll
    for (int i = 0; i < VP; ++i) rs[i] = rsqrtf(rs[i] / (float)C + eps);
    if (ok) {
        if (wave == 0) {
            stv_f32<VP>(mean + (int64_t)b * L + p, mu);
            stv_f32<VP>(rstd + (int64_t)b * L + p, rs);
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int c = wave + j * NW;
            const float wc = w[c], bc = bias ? bias[c] : 0.f;
            float t[VP];
#pragma unroll
            for (int i = 0; i < VP; ++i) t[i] = fmaf(v[j][i] * rs[i], wc, bc);
            stv_bf16<VP>(y + o + (int64_t)c * L, t);
        }
    }
}

template <int CPT, int VP>
__global__ void __launch_bounds__(1024) ln2d_bwd_dx_vec_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                               const bf16_t *__restrict__ dy, const float *__restrict__ mean,
                                                               const float *__restrict__ rstd, float *__restrict__ dx, int C,
                                                               int L, int64_t NP, float *__restrict__ parts) {
    constexpr int NW = 16;
    __shared__ float red[2][NW][64 * VP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t Pv = (int64_t)blockIdx.x * 64 + lane;
    const bool ok = Pv < NP;
    const int64_t P = ok ? Pv * VP : 0;
    const int b = (int)(P / L), p = (int)(P - (int64_t)b * L);
    const int64_t o = (int64_t)b * C * L + p;
    float mu[VP], rs[VP];
    ldv_f32<VP>(mean + (int64_t)b * L + p, mu);
    ldv_f32<VP>(rstd + (int64_t)b * L + p, rs);
    float g[CPT][VP], xh[CPT][VP], s1[VP], s2[VP];            // g: RAW dy (the weight is applied at each use)
#pragma unroll
    for (int i = 0; i < VP; ++i) s1[i] = s2[i] = 0.f;
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int c = wave + j * NW;
        const float wc = w[c];
        ldv_bf16<VP>(dy + o + (int64_t)c * L, g[j]);
        ldv_f32<VP>(x + o + (int64_t)c * L, xh[j]);
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            if (!ok) g[j][i] = 0.f;
            xh[j][i] = ok ? (xh[j][i] - mu[i]) * rs[i] : 0.f;
            s1[i] = fmaf(g[j][i], wc, s1[i]);
            s2[i] = fmaf(g[j][i] * wc, xh[j][i], s2[i]);
        }
    }
    stv_f32<VP>(&red[0][wave][lane * VP], s1);
    stv_f32<VP>(&red[1][wave][lane * VP], s2);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VP; ++i) s1[i] = s2[i] = 0.f;
    for (int q = 0; q < NW; ++q) {
        float t1[VP], t2[VP];
        ldv_f32<VP>(&red[0][q][lane * VP], t1);
        ldv_f32<VP>(&red[1][q][lane * VP], t2);
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            s1[i] += t1[i];
            s2[i] += t2[i];
        }
    }
    if (ok) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const float wc = w[wave + j * NW];
            float t[VP];
#pragma unroll
            for (int i = 0; i < VP; ++i) t[i] = rs[i] * (g[j][i] * wc - s1[i] / (float)C - xh[j][i] * (s2[i] / (float)C));
            stv_f32<VP>(dx + o + (int64_t)(wave + j * NW) * L, t);
        }
    }
    if (parts) {                                                    // (uniform) see ln2d_bwd_dx_cached_kernel
        __shared__ float prow[2 * 16 * CPT];
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            float p1 = 0.f, p2 = 0.f;
#pragma unroll
            for (int i = 0; i < VP; ++i) {
                p1 = fmaf(g[j][i], xh[j][i], p1);
                p2 += g[j][i];
            }
            p1 = ln2d_lane_sum<false>(p1);
            p2 = ln2d_lane_sum<false>(p2);
            if (lane == 63) {
                prow[wave + j * NW] = p1;
                prow[C + wave + j * NW] = p2;
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 2 * C; e += 1024) parts[(int64_t)blockIdx.x * 2 * C + e] = prow[e];
    }
}

// (CPT, VP) of the vectorised kernels for C channels split over 16 waves, or false
// (16 waves per workgroup cap a thread at 128 registers: 4 positions per lane up to 96 channels, 2 up to 192 / -- forward
//  only -- 384; the 384-channel backward keeps the scalar kernel)
static bool ln2d_vec_plan(int C, int L, bool bwd, int &cpt, int &vp) {
    if (C % 16 != 0) return false;
    cpt = C / 16;
    if (cpt != 6 && cpt != 12 && cpt != 24) return false;
    if (cpt == 24 && bwd) return false;
    if (cpt != 24 && !bwd) return false;         // measured: the scalar forward is as fast or faster at 96 / 192 channels
    vp = cpt == 6 ? 4 : 2;
    return L % vp == 0;
}

// dw[c] = sum_{b,p} dy * xhat,  db[c] = sum_{b,p} dy;  one workgroup per (channel, batch slice)
template <typename Tx, typename Ty>
__global__ void __launch_bounds__(256) ln2d_bwd_wb_kernel(const Tx *__restrict__ x, const Ty *__restrict__ dy,
                                                          const float *__restrict__ mean, const float *__restrict__ rstd,
                                                          float *__restrict__ dw, float *__restrict__ db, int B, int C,
                                                          int L, int bsplit) {
    __shared__ float red[2][4];
    const int c = blockIdx.x / bsplit, sl = blockIdx.x - c * bsplit;
    const int b0 = (int)((int64_t)B * sl / bsplit), b1 = (int)((int64_t)B * (sl + 1) / bsplit);
    float a1 = 0.f, a2 = 0.f;
    if (L < 256) {
        for (int i = threadIdx.x; i < (b1 - b0) * L; i += 256) {    // short maps (7x7): (image, position) pairs
            const int b = b0 + i / L, p = i - (i / L) * L;           // flattened, so the workgroup stays full
            const int64_t o = ((int64_t)b * C + c) * L + p;
            const float g = ldf<Ty>(dy + o);
            a1 = fmaf(g, (ldf<Tx>(x + o) - mean[(int64_t)b * L + p]) * rstd[(int64_t)b * L + p], a1);
            a2 += g;
        }
    } else {
        for (int b = b0; b < b1; ++b) {
            const int64_t o = ((int64_t)b * C + c) * L;
            const float *mb = mean + (int64_t)b * L, *rb = rstd + (int64_t)b * L;
            for (int p = threadIdx.x; p < L; p += 256) {
                const float g = ldf<Ty>(dy + o + p);
                a1 = fmaf(g, (ldf<Tx>(x + o + p) - mb[p]) * rb[p], a1);
                a2 += g;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        a1 += __shfl_xor(a1, off, 64);
        a2 += __shfl_xor(a2, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = a1;
        red[1][threadIdx.x >> 6] = a2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(dw + c, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        if (db) atomicAdd(db + c, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

template <typename Tx, typename Ty>
static int ln_fwd(const void *x, const float *w, const float *b, void *y, float *mean, float *rstd, int B, int C, int L,
                  float eps, hipStream_t s) {
    const int tiles = (L + 63) / 64;
    if constexpr (std::is_same<Tx, float>::value && std::is_same<Ty, bf16_t>::value) {
        int cpt, vp;
        if (!getenv("XFM_LN2D_SCALAR") && ln2d_vec_plan(C, L, false, cpt, vp)) {
            const int64_t NP = (int64_t)B * L / vp;
            const dim3 grid((unsigned)((NP + 63) / 64)), block(1024);
            const float *xf = (const float *)x;
            bf16_t *yb = (bf16_t *)y;
            if (cpt == 6) hipLaunchKernelGGL((ln2d_fwd_vec_kernel<6, 4>), grid, block, 0, s, xf, w, b, yb, mean, rstd, C, L, NP, eps);
            else if (cpt == 12) hipLaunchKernelGGL((ln2d_fwd_vec_kernel<12, 2>), grid, block, 0, s, xf, w, b, yb, mean, rstd, C, L, NP, eps);
            else hipLaunchKernelGGL((ln2d_fwd_vec_kernel<24, 2>), grid, block, 0, s, xf, w, b, yb, mean, rstd, C, L, NP, eps);
            return check_launch();
        }
    }
    if (ln2d_short8_ok(B, C, L)) {                          // 7 x 7 maps with 512 ... 1536 channels
        const dim3 grid((B * L + 7) / 8), block(256);
#define XFM_LN2D_S8(CPT)                                                                                                    \
    hipLaunchKernelGGL((ln2d_fwd_short_kernel<Tx, Ty, CPT>), grid, block, 0, s, (const Tx *)x, w, b, (Ty *)y, mean, rstd, C, L,  \
                       B * L, eps)
        if (C == 512) XFM_LN2D_S8(16);
        else if (C == 768) XFM_LN2D_S8(24);
        else if (C == 1024) XFM_LN2D_S8(32);
        else XFM_LN2D_S8(48);
#undef XFM_LN2D_S8
        return check_launch();
    }
    // few positions (a 14 x 14 stage: 196 tiles of 64 for 256 CUs): 32-position workgroups, the lane halves split the channels
    // (the fp32 -> bf16 forward above measured the same either way: 14.1 vs 14.6 us; the backward 22.9 -> 17.8 us)
    if ((B * L + 63) / 64 < 256 && !getenv("XFM_LN2D_NO_HALF") && C % 24 == 0 && C / 24 <= 16) {
        const int NW = C / 24;
        hipLaunchKernelGGL((ln2d_fwd_cached_kernel<Tx, Ty, 12, true>), dim3((B * L + 31) / 32), dim3(64 * NW), 0, s, (const Tx *)x, w, b,
                           (Ty *)y, mean, rstd, C, L, B * L, eps, NW);
    } else if (C % 24 == 0 && C / 24 <= 16) {
        const int NW = C / 24;
        hipLaunchKernelGGL((ln2d_fwd_cached_kernel<Tx, Ty, 24>), dim3((B * L + 63) / 64), dim3(64 * NW), 0, s, (const Tx *)x, w, b,
                           (Ty *)y, mean, rstd, C, L, B * L, eps, NW);
    } else if (C % 48 == 0 && C / 48 <= 16) {
        const int NW = C / 48;
        hipLaunchKernelGGL((ln2d_fwd_cached_kernel<Tx, Ty, 48>), dim3((B * L + 63) / 64), dim3(64 * NW), 0, s, (const Tx *)x, w, b,
                           (Ty *)y, mean, rstd, C, L, B * L, eps, NW);
    } else if (C % 16 == 0 && C / 16 <= 16) {            // power-of-two widths (XFMamba-B: 256 / 512 / 1024 channels)
        const int NW = C / 16;
        hipLaunchKernelGGL((ln2d_fwd_cached_kernel<Tx, Ty, 16>), dim3((B * L + 63) / 64), dim3(64 * NW), 0, s, (const Tx *)x, w, b,
                           (Ty *)y, mean, rstd, C, L, B * L, eps, NW);
    } else if (C % 32 == 0 && C / 32 <= 16) {
        const int NW = C / 32;
        hipLaunchKernelGGL((ln2d_fwd_cached_kernel<Tx, Ty, 32>), dim3((B * L + 63) / 64), dim3(64 * NW), 0, s, (const Tx *)x, w, b,
                           (Ty *)y, mean, rstd, C, L, B * L, eps, NW);
    } else if (C % 64 == 0 && C / 64 <= 16) {
        const int NW = C / 64;
        hipLaunchKernelGGL((ln2d_fwd_cached_kernel<Tx, Ty, 64>), dim3((B * L + 63) / 64), dim3(64 * NW), 0, s, (const Tx *)x, w, b,
                           (Ty *)y, mean, rstd, C, L, B * L, eps, NW);
    } else {
        if (C >= 512)
            hipLaunchKernelGGL((ln2d_fwd_kernel<Tx, Ty, 16>), dim3(B * tiles), dim3(1024), 0, s, (const Tx *)x, w, b, (Ty *)y, mean,
                               rstd, C, L, tiles, eps);
        else
            hipLaunchKernelGGL((ln2d_fwd_kernel<Tx, Ty, 4>), dim3(B * tiles), dim3(256), 0, s, (const Tx *)x, w, b, (Ty *)y, mean,
                               rstd, C, L, tiles, eps);
    }
    return check_launch();
}

// workgroups (= partial row pairs) of the register-cached / vectorised dx kernels when they can emit the weight / bias gradient's
// partial rows for this shape, else 0 (the short-map / two-pass kernels do not: ln2d_bwd_wb_kernel follows them)
static int ln2d_parts_blocks(int B, int C, int L, bool vec_path) {
    if (vec_path) {                                                 // fp32 maps / bf16 gradients: the vectorised kernels
        int cpt, vp;
        if (!ln2d_vec_plan(C, L, true, cpt, vp)) return 0;
        return (int)(((int64_t)B * L / vp + 63) / 64);
    }
    if (C > 1536) return 0;
    if (ln2d_short8_ok(B, C, L)) return (B * L + 7) / 8;                           // the register-cached short-map kernel
    if (C >= 512 && (int64_t)B * L <= 16 * 1024) return 0;                         // the two-pass short-map kernel
    if ((B * L + 63) / 64 < 256 && !getenv("XFM_LN2D_NO_HALF") && C % 24 == 0 && C / 24 <= 16) return (B * L + 31) / 32;
    if ((C % 24 == 0 && C / 24 <= 16) || (C % 48 == 0 && C / 48 <= 16) || (C % 16 == 0 && C / 16 <= 16) ||
        (C % 32 == 0 && C / 32 <= 16))
        return (B * L + 63) / 64;
    return 0;
}

template <typename Tx, typename Ty>
static int ln_bwd(const void *x, const float *w, const void *dy, const float *mean, const float *rstd, void *dx,
                  float *dw, float *db, int B, int C, int L, hipStream_t s, float *parts = nullptr) {
    const int tiles = (L + 63) / 64;
    bool done = false;
    if constexpr (std::is_same<Tx, float>::value && std::is_same<Ty, bf16_t>::value) {
        int cpt, vp;
        if (!getenv("XFM_LN2D_SCALAR") && ln2d_vec_plan(C, L, true, cpt, vp)) {
            const int64_t NP = (int64_t)B * L / vp;
            const dim3 grid((unsigned)((NP + 63) / 64)), block(1024);
            const float *xf = (const float *)x;
            const bf16_t *dyb = (const bf16_t *)dy;
            float *dxf = (float *)dx;
            if (cpt == 6) hipLaunchKernelGGL((ln2d_bwd_dx_vec_kernel<6, 4>), grid, block, 0, s, xf, w, dyb, mean, rstd, dxf, C, L, NP, parts);
            else hipLaunchKernelGGL((ln2d_bwd_dx_vec_kernel<12, 2>), grid, block, 0, s, xf, w, dyb, mean, rstd, dxf, C, L, NP, parts);
            done = true;
        }
    }
    if (done) {
    } else if (ln2d_short8_ok(B, C, L)) {
        const dim3 grid((B * L + 7) / 8), block(256);
#define XFM_LN2D_S8(CPT)                                                                                                    \
    hipLaunchKernelGGL((ln2d_bwd_dx_short8_kernel<Tx, Ty, CPT>), grid, block, 0, s, (const Tx *)x, w, (const Ty *)dy, mean,  \
                       rstd, (Tx *)dx, C, L, B * L, parts)
        if (C == 512) XFM_LN2D_S8(16);
        else if (C == 768) XFM_LN2D_S8(24);
        else if (C == 1024) XFM_LN2D_S8(32);
        else XFM_LN2D_S8(48);
#undef XFM_LN2D_S8
    } else if (C >= 512 && (int64_t)B * L <= 16 * 1024) {
        // short maps, wide rows (7 x 7 at 768 / 1536 channels): positions x channel slices instead of lanes along positions
        hipLaunchKernelGGL((ln2d_bwd_dx_short_kernel<Tx, Ty>), dim3((B * L + 15) / 16), dim3(256), 0, s, (const Tx *)x, w,
                           (const Ty *)dy, mean, rstd, (Tx *)dx, C, L, B * L);
    } else if ((B * L + 63) / 64 < 256 && !getenv("XFM_LN2D_NO_HALF") && C % 24 == 0 && C / 24 <= 16) {
        const int NW = C / 24;
        hipLaunchKernelGGL((ln2d_bwd_dx_cached_kernel<Tx, Ty, 12, true>), dim3((B * L + 31) / 32), dim3(64 * NW), 0, s, (const Tx *)x, w,
                           (const Ty *)dy, mean, rstd, (Tx *)dx, C, L, B * L, NW, parts);
    } else if (C % 24 == 0 && C / 24 <= 16) {
        const int NW = C / 24;
        hipLaunchKernelGGL((ln2d_bwd_dx_cached_kernel<Tx, Ty, 24>), dim3((B * L + 63) / 64), dim3(64 * NW), 0, s, (const Tx *)x, w,
                           (const Ty *)dy, mean, rstd, (Tx *)dx, C, L, B * L, NW, parts);
    } else if (C % 48 == 0 && C / 48 <= 16) {
        const int NW = C / 48;
        hipLaunchKernelGGL((ln2d_bwd_dx_cached_kernel<Tx, Ty, 48>), dim3((B * L + 63) / 64), dim3(64 * NW), 0, s, (const Tx *)x, w,
                           (const Ty *)dy, mean, rstd, (Tx *)dx, C, L, B * L, NW, parts);
    } else if (C % 16 == 0 && C / 16 <= 16) {
        const int NW = C / 16;
        hipLaunchKernelGGL((ln2d_bwd_dx_cached_kernel<Tx, Ty, 16>), dim3((B * L + 63) / 64), dim3(64 * NW), 0, s, (const Tx *)x, w,
                           (const Ty *)dy, mean, rstd, (Tx *)dx, C, L, B * L, NW, parts);
    } else if (C % 32 == 0 && C / 32 <= 16) {
        const int NW = C / 32;
        hipLaunchKernelGGL((ln2d_bwd_dx_cached_kernel<Tx, Ty, 32>), dim3((B * L + 63) / 64), dim3(64 * NW), 0, s, (const Tx *)x, w,
                           (const Ty *)dy, mean, rstd, (Tx *)dx, C, L, B * L, NW, parts);
    } else {
        // (1024-channel rows, XFMamba-B stage 2: 2 x 64 values per thread do not fit the 128-register cap of a 16-wave
        //  workgroup -- measured 519 us with the spills against 183 us for the two-pass kernel below)
        if (C >= 512)
            hipLaunchKernelGGL((ln2d_bwd_dx_kernel<Tx, Ty, 16>), dim3(B * tiles), dim3(1024), 0, s, (const Tx *)x, w, (const Ty *)dy,
                               mean, rstd, (Tx *)dx, C, L, tiles);
        else
            hipLaunchKernelGGL((ln2d_bwd_dx_kernel<Tx, Ty, 4>), dim3(B * tiles), dim3(256), 0, s, (const Tx *)x, w, (const Ty *)dy,
                               mean, rstd, (Tx *)dx, C, L, tiles);
    }
    int rc = check_launch();
    if (rc) return rc;
    if (parts) return XFM_OK;                   // (the dx kernel left the partial rows: no second pass)
    int bsplit = 2048 / C;                      // aim at >= ~2048 workgroups
    if (bsplit < 1) bsplit = 1;
    if (bsplit > B) bsplit = B;
    hipLaunchKernelGGL((ln2d_bwd_wb_kernel<Tx, Ty>), dim3(C * bsplit), dim3(256), 0, s, (const Tx *)x, (const Ty *)dy,
                       mean, rstd, dw, db, B, C, L, bsplit);
    return check_launch();
}

}  // namespace xfm

extern "C" {

int xfm_layernorm2d_fwd(const void *x, const float *weight, const float *bias, void *y, float *mean, float *rstd, int B,
                        int C, int L, float eps, int x_dtype, int y_dtype, void *stream) {
    using namespace xfm;
    if (!x || !weight || !y || !mean || !rstd || B <= 0 || C <= 0 || L <= 0) return XFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == XFM_F32 && y_dtype == XFM_F32) return ln_fwd<float, float>(x, weight, bias, y, mean, rstd, B, C, L, eps, s);
    if (x_dtype == XFM_F32 && y_dtype == XFM_BF16) return ln_fwd<float, bf16_t>(x, weight, bias, y, mean, rstd, B, C, L, eps, s);
    if (x_dtype == XFM_F32 && y_dtype == XFM_F16) return ln_fwd<float, f16_t>(x, weight, bias, y, mean, rstd, B, C, L, eps, s);
    if (x_dtype == XFM_BF16 && y_dtype == XFM_BF16) return ln_fwd<bf16_t, bf16_t>(x, weight, bias, y, mean, rstd, B, C, L, eps, s);
    if (x_dtype == XFM_BF16 && y_dtype == XFM_F32) return ln_fwd<bf16_t, float>(x, weight, bias, y, mean, rstd, B, C, L, eps, s);
    if (x_dtype == XFM_F16 && y_dtype == XFM_F16) return ln_fwd<f16_t, f16_t>(x, weight, bias, y, mean, rstd, B, C, L, eps, s);
    if (x_dtype == XFM_F16 && y_dtype == XFM_F32) return ln_fwd<f16_t, float>(x, weight, bias, y, mean, rstd, B, C, L, eps, s);
    return XFM_EDTYPE;
}

/* Backward with the weight / bias gradient left as partial rows (no second pass over x and dy): parts
 * (xfm_layernorm2d_bwd_parts_blocks(...), 2, C) fp32, row pair j = [sum dy * xhat | sum dy] over the positions of workgroup j,
 * every element written.  0 blocks: the kernel chosen for this shape cannot (use xfm_layernorm2d_bwd). */
static bool ln2d_takes_vec(int C, int L, int x_dtype, int y_dtype) {
    int cpt, vp;
    return x_dtype == XFM_F32 && y_dtype == XFM_BF16 && !getenv("XFM_LN2D_SCALAR") && xfm::ln2d_vec_plan(C, L, true, cpt, vp);
}

int xfm_layernorm2d_bwd_parts_blocks(int B, int C, int L, int x_dtype, int y_dtype) {
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    return xfm::ln2d_parts_blocks(B, C, L, ln2d_takes_vec(C, L, x_dtype, y_dtype));
}

int xfm_layernorm2d_bwd_parts(const void *x, const float *weight, const void *dy, const float *mean, const float *rstd,
                              void *dx, float *parts, int B, int C, int L, int x_dtype, int y_dtype, void *stream) {
    using namespace xfm;
    if (!x || !weight || !dy || !mean || !rstd || !dx || !parts || B <= 0 || C <= 0 || L <= 0) return XFM_EINVAL;
    if (!xfm_layernorm2d_bwd_parts_blocks(B, C, L, x_dtype, y_dtype)) return XFM_ELIMIT;
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == XFM_F32 && y_dtype == XFM_F32) return ln_bwd<float, float>(x, weight, dy, mean, rstd, dx, nullptr, nullptr, B, C, L, s, parts);
    if (x_dtype == XFM_F32 && y_dtype == XFM_BF16) return ln_bwd<float, bf16_t>(x, weight, dy, mean, rstd, dx, nullptr, nullptr, B, C, L, s, parts);
    if (x_dtype == XFM_BF16 && y_dtype == XFM_BF16) return ln_bwd<bf16_t, bf16_t>(x, weight, dy, mean, rstd, dx, nullptr, nullptr, B, C, L, s, parts);
    if (x_dtype == XFM_BF16 && y_dtype == XFM_F32) return ln_bwd<bf16_t, float>(x, weight, dy, mean, rstd, dx, nullptr, nullptr, B, C, L, s, parts);
    return XFM_EDTYPE;
}

/* The slab form for 7 x 7 maps (two kernels, a workspace of xfm_layernorm2d_ws_floats(...) fp32 values between them; 0: the
 * form does not cover the shape, use the entries above).  Same results as xfm_layernorm2d_fwd / _bwd_parts; the backward's
 * partial rows are (B, 2, C): one row pair per sample. */
int xfm_layernorm2d_ws_floats(int B, int C, int L) {
    if (B <= 0 || C <= 0 || L <= 0 || !xfm::ln2d_slab_ok(B, C, L)) return 0;
    return B * (C / xfm::kSlabC) * 2 * L;
}

int xfm_layernorm2d_fwd_ws(const void *x, const float *weight, const float *bias, void *y, float *mean, float *rstd,
                           float *workspace, int B, int C, int L, float eps, int x_dtype, int y_dtype, void *stream) {
    using namespace xfm;
    if (!x || !weight || !y || !mean || !rstd || !workspace) return XFM_EINVAL;
    if (!xfm_layernorm2d_ws_floats(B, C, L)) return XFM_ELIMIT;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)(C / kSlabC), (unsigned)B), block(256);
#define XFM_LN2D_SLAB_F(TX, TY)                                                                                             \
    do {                                                                                                                    \
        hipLaunchKernelGGL((ln2d_slab_stats_kernel<TX>), grid, block, 0, s, (const TX *)x, workspace, C, L);                 \
        hipLaunchKernelGGL((ln2d_slab_apply_kernel<TX, TY>), grid, block, 0, s, (const TX *)x, weight, bias, (TY *)y, mean,  \
                           rstd, workspace, C, L, eps);                                                                     \
    } while (0)
    if (x_dtype == XFM_F32 && y_dtype == XFM_F32) XFM_LN2D_SLAB_F(float, float);
    else if (x_dtype == XFM_F32 && y_dtype == XFM_BF16) XFM_LN2D_SLAB_F(float, bf16_t);
    else if (x_dtype == XFM_BF16 && y_dtype == XFM_BF16) XFM_LN2D_SLAB_F(bf16_t, bf16_t);
    else if (x_dtype == XFM_BF16 && y_dtype == XFM_F32) XFM_LN2D_SLAB_F(bf16_t, float);
    else return XFM_EDTYPE;
#undef XFM_LN2D_SLAB_F
    return check_launch();
}

/* Workspace (fp32 values) and partial-row count of xfm_layernorm2d_bwd_parts_ws for this shape: the slab form (7 x 7 maps:
 * one row pair per sample) or the split form (14 x 14 maps with 384 channels: one row pair per 64 positions); 0: not covered. */
int xfm_layernorm2d_bwd_ws_floats(int B, int C, int L) {
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    if (const int n = xfm_layernorm2d_ws_floats(B, C, L)) return n;
    if (const int cpw = xfm::ln2d_split_cpw(B, C, L)) return (C / (4 * cpw)) * 2 * B * L;
    return 0;
}

int xfm_layernorm2d_bwd_ws_blocks(int B, int C, int L) {
    if (B <= 0 || C <= 0 || L <= 0) return 0;
    if (xfm_layernorm2d_ws_floats(B, C, L)) return B;
    if (xfm::ln2d_split_cpw(B, C, L)) return (B * L + 63) / 64;
    return 0;
}

int xfm_layernorm2d_bwd_parts_ws(const void *x, const float *weight, const void *dy, const float *mean, const float *rstd,
                                 void *dx, float *parts, float *workspace, int B, int C, int L, int x_dtype, int y_dtype,
                                 void *stream) {
    using namespace xfm;
    if (!x || !weight || !dy || !mean || !rstd || !dx || !parts || !workspace) return XFM_EINVAL;
    if (!xfm_layernorm2d_bwd_ws_floats(B, C, L)) return XFM_ELIMIT;
    hipStream_t s = (hipStream_t)stream;
    if (!xfm_layernorm2d_ws_floats(B, C, L)) {              // the split form
        const int cpw = ln2d_split_cpw(B, C, L), NP = B * L;
        const dim3 grid((unsigned)((NP + 63) / 64), (unsigned)(C / (4 * cpw))), block(256);
#define XFM_LN2D_SPLIT_B(TX, TY)                                                                                            \
    do {                                                                                                                    \
        hipLaunchKernelGGL((ln2d_bwd_split_stats_kernel<TX, TY, 24>), grid, block, 0, s, (const TX *)x, weight,              \
                           (const TY *)dy, mean, rstd, workspace, C, L, NP);                                                \
        hipLaunchKernelGGL((ln2d_bwd_split_apply_kernel<TX, TY, 24>), grid, block, 0, s, (const TX *)x, weight,              \
                           (const TY *)dy, mean, rstd, (TX *)dx, workspace, parts, C, L, NP);                               \
    } while (0)
        if (x_dtype == XFM_F32 && y_dtype == XFM_F32) XFM_LN2D_SPLIT_B(float, float);
        else if (x_dtype == XFM_F32 && y_dtype == XFM_BF16) XFM_LN2D_SPLIT_B(float, bf16_t);
        else if (x_dtype == XFM_BF16 && y_dtype == XFM_BF16) XFM_LN2D_SPLIT_B(bf16_t, bf16_t);
        else if (x_dtype == XFM_BF16 && y_dtype == XFM_F32) XFM_LN2D_SPLIT_B(bf16_t, float);
        else return XFM_EDTYPE;
#undef XFM_LN2D_SPLIT_B
        return check_launch();
    }
    const dim3 grid((unsigned)(C / kSlabC), (unsigned)B), block(256);
#define XFM_LN2D_SLAB_B(TX, TY)                                                                                             \
    do {                                                                                                                    \
        hipLaunchKernelGGL((ln2d_slab_bwd_stats_kernel<TX, TY>), grid, block, 0, s, (const TX *)x, weight, (const TY *)dy,   \
                           mean, rstd, workspace, C, L);                                                                    \
        hipLaunchKernelGGL((ln2d_slab_bwd_apply_kernel<TX, TY>), grid, block, 0, s, (const TX *)x, weight, (const TY *)dy,   \
                           mean, rstd, (TX *)dx, workspace, parts, C, L);                                                   \
    } while (0)
    if (x_dtype == XFM_F32 && y_dtype == XFM_F32) XFM_LN2D_SLAB_B(float, float);
    else if (x_dtype == XFM_F32 && y_dtype == XFM_BF16) XFM_LN2D_SLAB_B(float, bf16_t);
    else if (x_dtype == XFM_BF16 && y_dtype == XFM_BF16) XFM_LN2D_SLAB_B(bf16_t, bf16_t);
    else if (x_dtype == XFM_BF16 && y_dtype == XFM_F32) XFM_LN2D_SLAB_B(bf16_t, float);
    else return XFM_EDTYPE;
#undef XFM_LN2D_SLAB_B
    return check_launch();
}

int xfm_layernorm2d_bwd(const void *x, const float *weight, const void *dy, const float *mean, const float *rstd,
                        void *dx, float *dweight, float *dbias, int B, int C, int L, int x_dtype, int y_dtype,
                        void *stream) {
    using namespace xfm;
    if (!x || !weight || !dy || !mean || !rstd || !dx || !dweight || B <= 0 || C <= 0 || L <= 0) return XFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == XFM_F32 && y_dtype == XFM_F32) return ln_bwd<float, float>(x, weight, dy, mean, rstd, dx, dweight, dbias, B, C, L, s);
    if (x_dtype == XFM_F32 && y_dtype == XFM_BF16) return ln_bwd<float, bf16_t>(x, weight, dy, mean, rstd, dx, dweight, dbias, B, C, L, s);
    if (x_dtype == XFM_F32 && y_dtype == XFM_F16) return ln_bwd<float, f16_t>(x, weight, dy, mean, rstd, dx, dweight, dbias, B, C, L, s);
    if (x_dtype == XFM_BF16 && y_dtype == XFM_BF16) return ln_bwd<bf16_t, bf16_t>(x, weight, dy, mean, rstd, dx, dweight, dbias, B, C, L, s);
    if (x_dtype == XFM_BF16 && y_dtype == XFM_F32) return ln_bwd<bf16_t, float>(x, weight, dy, mean, rstd, dx, dweight, dbias, B, C, L, s);
    if (x_dtype == XFM_F16 && y_dtype == XFM_F16) return ln_bwd<f16_t, f16_t>(x, weight, dy, mean, rstd, dx, dweight, dbias, B, C, L, s);
    if (x_dtype == XFM_F16 && y_dtype == XFM_F32) return ln_bwd<f16_t, float>(x, weight, dy, mean, rstd, dx, dweight, dbias, B, C, L, s);
    return XFM_EDTYPE;
}
}
