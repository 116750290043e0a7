// bn_tokens.hip -- training-mode BatchNorm2d of the shallow fusion block on the token-major stream, both views in one go.
//
// Reference: ShallowFusionBlock_v4 applies ONE nn.BatchNorm2d to view 1, then to view 2 (models/fusion_vmamba.py:906-907):
// batch statistics per view, the running statistics updated twice.  BatchNorm2d over an NCHW map is the per-column
// normalisation of its (B H W, C) token matrix, so on the trunk's token stream x is (V, N, C) fp32 with V = 2 views, N = B H W
// rows.  Through the framework this was 2 x (step counter, column statistics, running-stat update, transform) + a cat + the
// cast for the following in_proj GEMM in the forward pass (62 us for 2 x 4.8 MB) and 78 us in the backward pass.  Here:
//   forward : column partial sums per (view, row slice) -> finalize (mean / rstd per view, running statistics view after view)
//             -> apply, writing the GEMM's dtype directly;
//   backward: partial sums of dy and dy * xhat -> finalize (per-view sums, the shared weight / bias gradient = their total)
//             -> dx.
// Variance from shifted sums (shift = the view's first row): no cancellation when |mean| >> std.  HBM-bound streams.
#include "xfm_common.hpp"

namespace xfm {

constexpr int kBnS = 8;       // row slices per view

// partial column sums.  MODE 0: [sum (x - s) | sum (x - s)^2], s = x[v, 0, c];  MODE 1: [sum dy | sum dy * xhat]
template <typename Tdy, int MODE>
__global__ void __launch_bounds__(256) bn_tokens_partial_kernel(const float *__restrict__ x, const Tdy *__restrict__ dy,
                                                                const float *__restrict__ mean, const float *__restrict__ rstd,
                                                                float *__restrict__ ws, int N, int C) {
    __shared__ float red[2][16][68];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4, v = blockIdx.y, s = blockIdx.z, V = gridDim.y;
    const int r0 = (int)((int64_t)N * s / kBnS), r1 = (int)((int64_t)N * (s + 1) / kBnS);
    const float *xv = x + (int64_t)v * N * C + c;
    float a[4], b[4], m[4], rs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = b[i] = 0.f;
        m[i] = MODE == 0 ? xv[i] : mean[v * C + c + i];                  // MODE 0: the shift
        rs[i] = MODE == 0 ? 1.f : rstd[v * C + c + i];
    }
#pragma unroll 4
    for (int r = r0 + rl; r < r1; r += 16) {
        float xr[4];
        Pack<float>::ld(xv + (int64_t)r * C, xr);
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d = xr[i] - m[i];
                a[i] += d;
                b[i] = fmaf(d, d, b[i]);
            }
        } else {
            float g[4];
            if constexpr (sizeof(Tdy) == 4) {
                Pack<float>::ld(reinterpret_cast<const float *>(dy) + ((int64_t)v * N + r) * C + c, g);
            } else {
                const uint2 w = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(dy) + ((int64_t)v * N + r) * C + c);
                g[0] = __uint_as_float(w.x << 16); g[1] = __uint_as_float(w.x & 0xffff0000u);
                g[2] = __uint_as_float(w.y << 16); g[3] = __uint_as_float(w.y & 0xffff0000u);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] += g[i];
                b[i] = fmaf(g[i], (xr[i] - m[i]) * rs[i], b[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        red[0][rl][cq * 4 + i] = a[i];
        red[1][rl][cq * 4 + i] = b[i];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int which = threadIdx.x >> 6, cc = threadIdx.x & 63;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[which][k][cc];
        ws[(((int64_t)s * V + v) * 2 + which) * C + blockIdx.x * 64 + cc] = t;
    }
}

// forward finalize: per (view, channel) mean / rstd; running statistics updated view after view (momentum, unbiased variance)
__global__ void __launch_bounds__(256) bn_tokens_fwd_finalize_kernel(const float *__restrict__ x, const float *__restrict__ ws,
                                                                     float *__restrict__ mean, float *__restrict__ rstd,
                                                                     float *__restrict__ running_mean,
                                                                     float *__restrict__ running_var, float momentum, float eps,
                                                                     int V, int N, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
    for (int v = 0; v < V; ++v) {
        float a = 0.f, a2 = 0.f;
#pragma unroll
        for (int s = 0; s < kBnS; ++s) {
            a += ws[(((int64_t)s * V + v) * 2) * C + c];
            a2 += ws[(((int64_t)s * V + v) * 2 + 1) * C + c];
        }
        const float d = a / (float)N;
        const float mu = x[(int64_t)v * N * C + c] + d;
        const float var = fmaxf(a2 / (float)N - d * d, 0.f);
        mean[v * C + c] = mu;
        rstd[v * C + c] = rsqrtf(var + eps);
        rm = fmaf(momentum, mu - rm, rm);
        rv = fmaf(momentum, var * ((float)N / (float)(N > 1 ? N - 1 : 1)) - rv, rv);
    }
    if (running_mean) running_mean[c] = rm;
    if (running_var) running_var[c] = rv;
}

template <typename Ty>
__global__ void __launch_bounds__(256) bn_tokens_apply_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, const float *__restrict__ mean,
                                                              const float *__restrict__ rstd, Ty *__restrict__ y, int N, int C,
                                                              int64_t nvec) {
    const int C4 = C / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const int v = (int)(i / ((int64_t)N * C4));
        float xr[4], o[4];
        Pack<float>::ld(x + i * 4, xr);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = fmaf((xr[k] - mean[v * C + c + k]) * rstd[v * C + c + k], gamma ? gamma[c + k] : 1.f, beta ? beta[c + k] : 0.f);
        if constexpr (sizeof(Ty) == 4) {
            Pack<float>::st(reinterpret_cast<float *>(y) + i * 4, o);
        } else {
            *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(y) + i * 4) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
        }
    }
}

// backward finalize: per-view sums st[v][0|1][c] = [sum dy | sum dy xhat]; the shared parameters' gradients are their totals
__global__ void __launch_bounds__(256) bn_tokens_bwd_finalize_kernel(const float *__restrict__ ws, float *__restrict__ st,
                                                                     float *__restrict__ dgamma, float *__restrict__ dbeta, int V,
                                                                     int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float tg = 0.f, tb = 0.f;
    for (int v = 0; v < V; ++v) {
        float a = 0.f, a2 = 0.f;
#pragma unroll
        for (int s = 0; s < kBnS; ++s) {
            a += ws[(((int64_t)s * V + v) * 2) * C + c];
            a2 += ws[(((int64_t)s * V + v) * 2 + 1) * C + c];
        }
        st[(v * 2) * C + c] = a;
        st[(v * 2 + 1) * C + c] = a2;
        tb += a;
        tg += a2;
    }
    if (dgamma) dgamma[c] = tg;
    if (dbeta) dbeta[c] = tb;
}

template <typename Tdy>
__global__ void __launch_bounds__(256) bn_tokens_dx_kernel(const float *__restrict__ x, const Tdy *__restrict__ dy,
                                                           const float *__restrict__ gamma, const float *__restrict__ mean,
                                                           const float *__restrict__ rstd, const float *__restrict__ st,
                                                           float *__restrict__ dx, int N, int C, int64_t nvec) {
    const int C4 = C / 4;
    const float invN = 1.f / (float)N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const int v = (int)(i / ((int64_t)N * C4));
        float xr[4], g[4], o[4];
        Pack<float>::ld(x + i * 4, xr);
        if constexpr (sizeof(Tdy) == 4) {
            Pack<float>::ld(reinterpret_cast<const float *>(dy) + i * 4, g);
        } else {
            const uint2 w = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(dy) + i * 4);
            g[0] = __uint_as_float(w.x << 16); g[1] = __uint_as_float(w.x & 0xffff0000u);
            g[2] = __uint_as_float(w.y << 16); g[3] = __uint_as_float(w.y & 0xffff0000u);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float rs = rstd[v * C + c + k];
            const float xh = (xr[k] - mean[v * C + c + k]) * rs;
            const float sb = st[(v * 2) * C + c + k], sg = st[(v * 2 + 1) * C + c + k];
            o[k] = (gamma ? gamma[c + k] : 1.f) * rs * (g[k] - (sb + xh * sg) * invN);
        }
        Pack<float>::st(dx + i * 4, o);
    }
}

}  // namespace xfm

extern "C" {

int xfm_bn_tokens_supported(int V, int N, int C) { return (V >= 1 && V <= 8 && N >= 2 && C > 0 && C % 64 == 0) ? 1 : 0; }

/* fp32 values of the workspace of xfm_bn_tokens_fwd / _bwd: partial sums (row slices x V x 2 x C) + the backward's per-view sums */
int xfm_bn_tokens_ws_floats(int V, int N, int C) {
    return xfm_bn_tokens_supported(V, N, C) ? (xfm::kBnS * V * 2 * C + V * 2 * C) : 0;
}

int xfm_bn_tokens_fwd(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                      float momentum, float eps, void *y, float *mean, float *rstd, float *workspace, int V, int N, int C,
                      int y_dtype, void *stream) {
    using namespace xfm;
    if (!x || !y || !mean || !rstd || !workspace) return XFM_EINVAL;
    if (!xfm_bn_tokens_supported(V, N, C)) return XFM_ELIMIT;
    if (y_dtype != XFM_F32 && y_dtype != XFM_BF16) return XFM_EDTYPE;
    if (((uintptr_t)x | (uintptr_t)y) & 15) return XFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((bn_tokens_partial_kernel<float, 0>), dim3(C / 64, V, kBnS), dim3(256), 0, s, x, (const float *)nullptr,
                       (const float *)nullptr, (const float *)nullptr, workspace, N, C);
    hipLaunchKernelGGL(bn_tokens_fwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, x, workspace, mean, rstd,
                       running_mean, running_var, momentum, eps, V, N, C);
    const int64_t nvec = (int64_t)V * N * (C / 4);
    const unsigned grid = (unsigned)std::min<int64_t>((nvec + 255) / 256, 256 * 8);
    if (y_dtype == XFM_F32)
        hipLaunchKernelGGL((bn_tokens_apply_kernel<float>), dim3(grid), dim3(256), 0, s, x, gamma, beta, mean, rstd, (float *)y, N, C, nvec);
    else
        hipLaunchKernelGGL((bn_tokens_apply_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, x, gamma, beta, mean, rstd, (bf16_t *)y, N, C, nvec);
    return check_launch();
}

int xfm_bn_tokens_bwd(const float *x, const void *dy, const float *gamma, const float *mean, const float *rstd, float *dx,
                      float *dgamma, float *dbeta, float *workspace, int V, int N, int C, int dy_dtype, void *stream) {
    using namespace xfm;
    if (!x || !dy || !mean || !rstd || !dx || !workspace) return XFM_EINVAL;
    if (!xfm_bn_tokens_supported(V, N, C)) return XFM_ELIMIT;
    if (dy_dtype != XFM_F32 && dy_dtype != XFM_BF16) return XFM_EDTYPE;
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) return XFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float *st = workspace + (int64_t)kBnS * V * 2 * C;
    const int64_t nvec = (int64_t)V * N * (C / 4);
    const unsigned grid = (unsigned)std::min<int64_t>((nvec + 255) / 256, 256 * 8);
    if (dy_dtype == XFM_F32) {
        hipLaunchKernelGGL((bn_tokens_partial_kernel<float, 1>), dim3(C / 64, V, kBnS), dim3(256), 0, s, x, (const float *)dy, mean,
                           rstd, workspace, N, C);
        hipLaunchKernelGGL(bn_tokens_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, workspace, st, dgamma, dbeta, V, C);
        hipLaunchKernelGGL((bn_tokens_dx_kernel<float>), dim3(grid), dim3(256), 0, s, x, (const float *)dy, gamma, mean, rstd, st, dx,
                           N, C, nvec);
    } else {
        hipLaunchKernelGGL((bn_tokens_partial_kernel<bf16_t, 1>), dim3(C / 64, V, kBnS), dim3(256), 0, s, x, (const bf16_t *)dy, mean,
                           rstd, workspace, N, C);
        hipLaunchKernelGGL(bn_tokens_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, workspace, st, dgamma, dbeta, V, C);
        hipLaunchKernelGGL((bn_tokens_dx_kernel<bf16_t>), dim3(grid), dim3(256), 0, s, x, (const bf16_t *)dy, gamma, mean, rstd, st, dx,
                           N, C, nvec);
    }
    return check_launch();
}

}  // extern "C"
