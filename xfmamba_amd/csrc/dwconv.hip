// dwconv.hip -- depthwise 3x3 convolution (+bias) fused with SiLU, forward and backward.
//
// This is the `conv2d` -> `act` pair in front of every SS2D core
// (models/fusion_vmamba.py:1198-1201 backbone, :594-601 deep fusion, :853-857 shallow fusion):
// nn.Conv2d(D, D, 3, padding=1, groups=D) followed by nn.SiLU.  It is a 9-tap stencil per (b, d)
// plane, i.e. HBM-bound: each plane is staged once in LDS with a zero halo, the pre-activation is
// recomputed in the backward instead of being saved, and the weight / bias gradients are reduced
// per wave in registers (shuffles) before one atomic per (wave, tap).
//
// A workgroup (256 threads) handles PP in {1,2,4} consecutive planes; each plane is owned by
// 256/PP >= 64 threads, so a wavefront never straddles two planes and the per-plane reductions
// stay inside a wave.
#include "xfm_common.hpp"

namespace xfm {

__device__ __forceinline__ float sigmoidf_fast(float z) { return 1.f / (1.f + __expf(-z)); }

template <typename T>
__global__ void __launch_bounds__(256) dwconv_fwd_kernel(const T *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ bias, T *__restrict__ y, int planes,
                                                         int D, int H, int W, int pp, int act) {
    extern __shared__ float smem[];
    const int L = H * W, PW = W + 2, PH = H + 2, psz = PH * PW;
    const int tpp = 256 / pp;                       // threads per plane
    const int sub = threadIdx.x / tpp, tl = threadIdx.x - sub * tpp;
    const int plane = blockIdx.x * pp + sub;
    const bool live = plane < planes;
    float *xp = smem + sub * psz;
    for (int e = tl; e < psz; e += tpp) xp[e] = 0.f;
    __syncthreads();
    if (live) {
        const T *xg = x + (int64_t)plane * L;
        for (int e = tl; e < L; e += tpp) {
            const int h = e / W, c = e - h * W;
            xp[(h + 1) * PW + c + 1] = ldf<T>(xg + e);
        }
    }
    __syncthreads();
    if (!live) return;
    const int d = plane % D;
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = w[d * 9 + i];
    const float b = bias ? bias[d] : 0.f;
    T *yg = y + (int64_t)plane * L;
    for (int e = tl; e < L; e += tpp) {
        const int h = e / W, c = e - h * W;
        const float *q = xp + h * PW + c;           // top-left of the 3x3 window in padded coords
        float z = b;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) z = fmaf(k[i * 3 + j], q[i * PW + j], z);
        stf<T>(yg + e, act ? z * sigmoidf_fast(z) : z);
    }
}

// dx = corr(dz, flip(w)), dw[i][j] = sum dz[h][w] * x[h+i-1][w+j-1], db = sum dz, with
// dz = dy * silu'(z), z recomputed from x.
template <typename T>
__global__ void __launch_bounds__(256) dwconv_bwd_kernel(const T *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ bias, const T *__restrict__ dy,
                                                         T *__restrict__ dx, float *__restrict__ dw,
                                                         float *__restrict__ dbias, int planes, int D, int H, int W,
                                                         int pp, int act) {
    extern __shared__ float smem[];
    const int L = H * W, PW = W + 2, PH = H + 2, psz = PH * PW;
    const int tpp = 256 / pp;
    const int sub = threadIdx.x / tpp, tl = threadIdx.x - sub * tpp;
    const int plane = blockIdx.x * pp + sub;
    const bool live = plane < planes;
    float *xp = smem + sub * 2 * psz;               // padded x
    float *zp = xp + psz;                           // padded dz
    for (int e = tl; e < 2 * psz; e += tpp) xp[e] = 0.f;
    __syncthreads();
    if (live) {
        const T *xg = x + (int64_t)plane * L;
        for (int e = tl; e < L; e += tpp) {
            const int h = e / W, c = e - h * W;
            xp[(h + 1) * PW + c + 1] = ldf<T>(xg + e);
        }
    }
    __syncthreads();
    const int d = live ? plane % D : 0;
    float k[9], acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        k[i] = w[d * 9 + i];
        acc[i] = 0.f;
    }
    float accb = 0.f;
    if (live) {
        const float b = bias ? bias[d] : 0.f;
        const T *gg = dy + (int64_t)plane * L;
        for (int e = tl; e < L; e += tpp) {
            const int h = e / W, c = e - h * W;
            const float *q = xp + h * PW + c;
            float xv[9];
            float z = b;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    xv[i * 3 + j] = q[i * PW + j];
                    z = fmaf(k[i * 3 + j], xv[i * 3 + j], z);
                }
            float g = ldf<T>(gg + e);
            if (act) {
                const float s = sigmoidf_fast(z);
                g *= s * fmaf(z, 1.f - s, 1.f);      // d silu(z)/dz = s * (1 + z * (1 - s))
            }
            zp[(h + 1) * PW + c + 1] = g;
#pragma unroll
            for (int i = 0; i < 9; ++i) acc[i] = fmaf(g, xv[i], acc[i]);
            accb += g;
        }
    }
    __syncthreads();
    if (live) {
        T *dxg = dx + (int64_t)plane * L;
        for (int e = tl; e < L; e += tpp) {
            const int h = e / W, c = e - h * W;
            const float *q = zp + h * PW + c;
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) v = fmaf(k[8 - (i * 3 + j)], q[i * PW + j], v);
            stf<T>(dxg + e, v);
        }
    }
    // per-wave reduction (a wave never straddles planes), then one atomic per tap
#pragma unroll
    for (int i = 0; i < 9; ++i)
        for (int off = 32; off > 0; off >>= 1) acc[i] += __shfl_xor(acc[i], off, 64);
    for (int off = 32; off > 0; off >>= 1) accb += __shfl_xor(accb, off, 64);
    if (live && (threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) atomicAdd(dw + d * 9 + i, acc[i]);
        if (dbias) atomicAdd(dbias + d, accb);
    }
}

static int pick_pp(int L) { return L >= 1024 ? 1 : (L >= 256 ? 2 : 4); }

template <typename T>
static int launch_dw(bool bwd, const void *x, const float *w, const float *bias, const void *dy, void *out, float *dw,
                     float *dbias, int B, int D, int H, int W, int act, hipStream_t s) {
    const int planes = B * D;
    const int pp = pick_pp(H * W);
    const size_t lds = (size_t)pp * (bwd ? 2 : 1) * (H + 2) * (W + 2) * sizeof(float);
    if (lds > 160 * 1024) return XFM_ELIMIT;
    const dim3 grid((planes + pp - 1) / pp);
    if (bwd)
        hipLaunchKernelGGL((dwconv_bwd_kernel<T>), grid, dim3(256), lds, s, (const T *)x, w, bias, (const T *)dy,
                           (T *)out, dw, dbias, planes, D, H, W, pp, act);
    else
        hipLaunchKernelGGL((dwconv_fwd_kernel<T>), grid, dim3(256), lds, s, (const T *)x, w, bias, (T *)out, planes, D,
                           H, W, pp, act);
    return check_launch();
}

}  // namespace xfm

extern "C" {

int xfm_dwconv3x3_fwd(const void *x, const float *weight, const float *bias, void *y, int B, int D, int H, int W,
                      int dtype, int silu, void *stream) {
    using namespace xfm;
    if (!x || !weight || !y || B <= 0 || D <= 0 || H <= 0 || W <= 0) return XFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case XFM_F32: return launch_dw<float>(false, x, weight, bias, nullptr, y, nullptr, nullptr, B, D, H, W, silu, s);
        case XFM_F16: return launch_dw<f16_t>(false, x, weight, bias, nullptr, y, nullptr, nullptr, B, D, H, W, silu, s);
        case XFM_BF16: return launch_dw<bf16_t>(false, x, weight, bias, nullptr, y, nullptr, nullptr, B, D, H, W, silu, s);
    }
    return XFM_EDTYPE;
}

int xfm_dwconv3x3_bwd(const void *x, const float *weight, const float *bias, const void *dy, void *dx, float *dweight,
                      float *dbias, int B, int D, int H, int W, int dtype, int silu, void *stream) {
    using namespace xfm;
    if (!x || !weight || !dy || !dx || !dweight || B <= 0 || D <= 0 || H <= 0 || W <= 0) return XFM_EINVAL;
    if (bias && !dbias) return XFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case XFM_F32: return launch_dw<float>(true, x, weight, bias, dy, dx, dweight, dbias, B, D, H, W, silu, s);
        case XFM_F16: return launch_dw<f16_t>(true, x, weight, bias, dy, dx, dweight, dbias, B, D, H, W, silu, s);
        case XFM_BF16: return launch_dw<bf16_t>(true, x, weight, bias, dy, dx, dweight, dbias, B, D, H, W, silu, s);
    }
    return XFM_EDTYPE;
}
}
