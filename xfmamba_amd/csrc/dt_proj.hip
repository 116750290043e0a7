// dt_proj of the SS2D core as a bandwidth-shaped kernel:  dts[b,k,d,l] = sum_r W[k,d,r] * xr[b,k,r,l]
// (reference: the grouped `einsum("b k r l, k d r -> b k d l")` of SS2Dv2.forward_corev2,
// models/fusion_vmamba.py:1154-1156).  The contraction length R = dt_rank is 6..24 while the output is the largest
// tensor of the block, (B, 4, D, L): as a library GEMM it runs at a fraction of HBM speed (K = 6 feeds no MFMA
// pipeline).  Here the (R x L-tile) slice of xr sits in LDS as fp32, a lane owns one 16-byte vector of the output
// and walks (d, chunk) pairs of the contiguous (D x L) matrix of its (b, k), and W rows stream through L1.
// HBM-bound: one write of the output; R*VEC FMAs per vector.
#include "xfm_common.hpp"

namespace xfm {

struct DtProjArgs {
    const void *xr;     // (B, 4, R, L)
    const float *w;     // (4, D, R) fp32
    const float *bias;  // (4*D) or null: epilogue dts = softplus(acc + bias) (threshold 20, as F.softplus)
    void *out;          // (B, 4, D, L)
    int D, R, L, TL, ntile, dsplit;
};

template <typename T, int VEC> __global__ __launch_bounds__(256) void dt_proj_fwd_kernel(DtProjArgs a) {
    extern __shared__ float xs[];                       // [R][TL]
    const int tile = blockIdx.x % a.ntile;
    const int ds = (blockIdx.x / a.ntile) % a.dsplit;
    const int bk = blockIdx.x / (a.ntile * a.dsplit);
    const int k = bk & 3;
    const int l0 = tile * a.TL;
    const int tl = min(a.TL, a.L - l0);                 // multiple of VEC by construction
    const T *xr = static_cast<const T *>(a.xr) + (int64_t)bk * a.R * a.L + l0;
    for (int e = threadIdx.x; e < a.R * tl; e += 256) {
        const int r = e / tl, c = e - r * tl;
        xs[r * a.TL + c] = ldf<T>(xr + (int64_t)r * a.L + c);
    }
    __syncthreads();
    const int cpr = tl / VEC;                           // chunks per row of this tile
    const int dper = (a.D + a.dsplit - 1) / a.dsplit;
    const int d0 = ds * dper, d1 = min(a.D, d0 + dper);
    T *out = static_cast<T *>(a.out) + (int64_t)bk * a.D * a.L + l0;
    const float *wk = a.w + (int64_t)k * a.D * a.R;
    for (int v = threadIdx.x; v < (d1 - d0) * cpr; v += 256) {
        const int dd = v / cpr, c = (v - dd * cpr) * VEC;
        const int d = d0 + dd;
        const float *wd = wk + (int64_t)d * a.R;
        float acc[VEC];
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc[i] = 0.f;
        for (int r = 0; r < a.R; ++r) {
            const float w = wd[r];
            const float *xv = xs + r * a.TL + c;
#pragma unroll
            for (int i = 0; i < VEC; i += 4) {
                const float4 q = *reinterpret_cast<const float4 *>(xv + i);
                acc[i] = fmaf(w, q.x, acc[i]);
                acc[i + 1] = fmaf(w, q.y, acc[i + 1]);
                acc[i + 2] = fmaf(w, q.z, acc[i + 2]);
                acc[i + 3] = fmaf(w, q.w, acc[i + 3]);
            }
        }
        if (a.bias) {
            const float bv = a.bias[k * a.D + d];
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc[i] = softplus20(acc[i] + bv);
        }
        T *o = out + (int64_t)d * a.L + c;
        if constexpr (VEC == Pack<T>::N) {
            Pack<T>::st(o, acc);
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) stf<T>(o + i, acc[i]);
        }
    }
}

template <typename T>
static int dt_proj_launch(const void *xr, const float *w, const float *bias, void *out, int B, int D, int R, int L,
                          hipStream_t s) {
    constexpr int VN = Pack<T>::N;                      // 8 (16-bit) or 4 (fp32) elements per 16-byte vector
    const int vec = (L % VN == 0) ? VN : 4;
    if (L % 4 != 0) return XFM_ELIMIT;
    DtProjArgs a{};
    a.xr = xr; a.w = w; a.bias = bias; a.out = out; a.D = D; a.R = R; a.L = L;
    // L-tile: whole rows of vectors, R*TL floats <= 40 KB of LDS
    int ntile = 1;
    while ((int64_t)R * ((L / vec + ntile - 1) / ntile) * vec * 4 > 40 * 1024) ++ntile;
    a.TL = ((L / vec + ntile - 1) / ntile) * vec;
    a.ntile = (L + a.TL - 1) / a.TL;
    int dsplit = 1;
    while ((int64_t)B * 4 * a.ntile * dsplit < 1024 && dsplit < D) dsplit *= 2;
    a.dsplit = dsplit;
    const size_t lds = (size_t)R * a.TL * sizeof(float);
    const dim3 grid((unsigned)((int64_t)B * 4 * a.ntile * dsplit));
    if (vec == VN) hipLaunchKernelGGL((dt_proj_fwd_kernel<T, VN>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((dt_proj_fwd_kernel<T, 4>), grid, dim3(256), lds, s, a);
    return check_launch();
}

}  // namespace xfm

extern "C" {

int xfm_ss2d_dt_proj_supported(int D, int R, int L) { return (L % 4 == 0 && R >= 1 && R <= 64 && D >= 1) ? 1 : 0; }

int xfm_ss2d_dt_proj_fwd(const void *xr, const float *weight, const float *softplus_bias, void *dts, int B, int D, int R,
                         int L, int dtype, void *stream) {
    using namespace xfm;
    if (!xr || !weight || !dts || B <= 0 || D <= 0 || R <= 0 || L <= 0) return XFM_EINVAL;
    if (!xfm_ss2d_dt_proj_supported(D, R, L)) return XFM_ELIMIT;
    hipStream_t s = (hipStream_t)stream;
    switch (dtype) {
        case XFM_F32: return dt_proj_launch<float>(xr, weight, softplus_bias, dts, B, D, R, L, s);
        case XFM_BF16: return dt_proj_launch<bf16_t>(xr, weight, softplus_bias, dts, B, D, R, L, s);
    }
    return XFM_EDTYPE;
}

}  // extern "C"
