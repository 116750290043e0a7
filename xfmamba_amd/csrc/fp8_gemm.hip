// fp8_gemm.hip -- BASELINE.json configs[4]: the SS2D x_proj / out_proj projections with fp8 (OCP e4m3fn) weights on the
// CDNA4 fp8 matrix cores (v_mfma_f32_32x32x16_fp8_fp8), scan kept in bf16.
//
//   y[b, l, m] = scale * sum_k Wq[m, k] * q(x[b, k, l])          x: (B, K, L) bf16 PLANES (the depthwise-conv / out_norm
//   output of an SS2D block, reference models/fusion_vmamba.py:1147-1150 x_proj, :1205 out_proj), y: (B, L, M) bf16 TOKENS.
// Wq is the weight quantised per tensor on the host side (scale = amax / 448); the activation is quantised in the kernel
// while it is staged: clamp to +-448, round to nearest even (v_cvt_pk_fp8_f32), no scale.  The non-scaled fp8 MFMA needs
// BOTH operands in fp8 and runs at the bf16 rate (MI355X_MICROARCH.md, Matrix cores), so this configuration buys weight
// bytes, not FLOP/s; these products are HBM-bound on the activation anyway.
#include "xfm_common.hpp"

namespace xfm {

typedef float gf32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t gu32x2_t __attribute__((ext_vector_type(2)));

struct Fp8GemmArgs {
    const uint16_t *x;     // (B, K, L) bf16
    const uint8_t *wq;     // (M, K) fp8 e4m3fn
    uint16_t *y;           // (B*L, M) bf16
    const float *scale;    // device scalar: weight scale (amax / 448)
    int B, K, L, M, T, mw;   // T = B*L tokens; mw = output columns per wave (multiple of 32)
};

__device__ __forceinline__ float fp8_clamp(float v) { return fminf(fmaxf(v, -448.f), 448.f); }

template <int MT>   // 32-column tiles per wave
__global__ void __launch_bounds__(256) fp8_planes_gemm_kernel(const Fp8GemmArgs a) {
    extern __shared__ uint8_t xq[];                                // [32 tokens][K + 8] fp8
    const int P = a.K + 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t0 = blockIdx.x * 32;
    {   // stage + quantise: thread = (token, group of 4 channels); a channel's 32 tokens are 64 contiguous bytes
        const int tok = threadIdx.x & 31, kg = threadIdx.x >> 5;
        const int t = min(t0 + tok, a.T - 1);
        const int b = t / a.L, l = t - b * a.L;
        const uint16_t *xp = a.x + (int64_t)b * a.K * a.L + l;
        for (int k0 = kg * 4; k0 < a.K; k0 += 32) {
            float f[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = fp8_clamp(__uint_as_float((uint32_t)xp[(int64_t)(k0 + j) * a.L] << 16));
            int pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], 0, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], pk, true);
            *reinterpret_cast<int *>(xq + tok * P + k0) = pk;
        }
    }
    __syncthreads();
    const int col = lane & 31, kb = lane >> 5;
    const int n0 = wave * a.mw;
    if (n0 >= a.M) return;
    gf32x16_t acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[m][j] = 0.f;
    const uint8_t *xrow = xq + col * P + 8 * kb;
    const int nks = a.K / 16;
    for (int s = 0; s < nks; ++s) {
        const long bf = *reinterpret_cast<const long *>(xrow + 16 * s);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int n = n0 + 32 * m + col;                        // A-operand row of this lane
            long af = 0;
            if (n < a.M) af = *reinterpret_cast<const long *>(a.wq + (int64_t)n * a.K + 16 * s + 8 * kb);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(af, bf, acc[m], 0, 0, 0);
        }
    }
    const int t = t0 + col;
    if (t >= a.T) return;
    const float sc = *a.scale;
    uint16_t *yrow = a.y + (int64_t)t * a.M;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + 32 * m + 8 * g + 4 * kb;             // rows n .. n+3 of D^T = four consecutive columns
            if (n + 3 < a.M) {
                gu32x2_t v;
                v[0] = pack_bf16x2(acc[m][4 * g] * sc, acc[m][4 * g + 1] * sc);
                v[1] = pack_bf16x2(acc[m][4 * g + 2] * sc, acc[m][4 * g + 3] * sc);
                *reinterpret_cast<gu32x2_t *>(yrow + n) = v;
            }
        }
}

}  // namespace xfm

extern "C" int xfm_fp8_planes_gemm_supported(int K, int M) { return (K % 16 == 0 && K >= 16 && K <= 4096 && M % 4 == 0 && M >= 4 && M <= 1024) ? 1 : 0; }

extern "C" int xfm_fp8_planes_gemm(const void *x_bf16, const void *wq_fp8, const float *scale, void *y_bf16, int B, int K, int L, int M,
                                   void *stream) {
    using namespace xfm;
    if (!x_bf16 || !wq_fp8 || !scale || !y_bf16 || B <= 0 || L <= 0) return XFM_EINVAL;
    if (!xfm_fp8_planes_gemm_supported(K, M)) return XFM_ELIMIT;
    Fp8GemmArgs a{};
    a.x = (const uint16_t *)x_bf16; a.wq = (const uint8_t *)wq_fp8; a.y = (uint16_t *)y_bf16; a.scale = scale;
    a.B = B; a.K = K; a.L = L; a.M = M; a.T = B * L;
    const int mt = ((M + 3) / 4 + 31) / 32;                          // 32-column tiles per wave (4 waves)
    a.mw = mt * 32;
    const size_t lds = (size_t)32 * (K + 8);
    const unsigned grid = (unsigned)((a.T + 31) / 32);
    hipStream_t s = (hipStream_t)stream;
    if (lds > 160 * 1024) return XFM_ELIMIT;
    if (lds > 64 * 1024) {                                           // K > 2040 (XFMamba-B stage 3: d_inner 2048)
        const void *fns[8] = {(const void *)fp8_planes_gemm_kernel<1>, (const void *)fp8_planes_gemm_kernel<2>,
                              (const void *)fp8_planes_gemm_kernel<3>, (const void *)fp8_planes_gemm_kernel<4>,
                              (const void *)fp8_planes_gemm_kernel<5>, (const void *)fp8_planes_gemm_kernel<6>,
                              (const void *)fp8_planes_gemm_kernel<7>, (const void *)fp8_planes_gemm_kernel<8>};
        if (mt < 1 || mt > 8) return XFM_ELIMIT;
        if (hipFuncSetAttribute(fns[mt - 1], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return XFM_ELAUNCH;
    }
    switch (mt) {
        case 1: hipLaunchKernelGGL(fp8_planes_gemm_kernel<1>, dim3(grid), dim3(256), lds, s, a); break;
        case 2: hipLaunchKernelGGL(fp8_planes_gemm_kernel<2>, dim3(grid), dim3(256), lds, s, a); break;
        case 3: hipLaunchKernelGGL(fp8_planes_gemm_kernel<3>, dim3(grid), dim3(256), lds, s, a); break;
        case 4: hipLaunchKernelGGL(fp8_planes_gemm_kernel<4>, dim3(grid), dim3(256), lds, s, a); break;
        case 5: hipLaunchKernelGGL(fp8_planes_gemm_kernel<5>, dim3(grid), dim3(256), lds, s, a); break;
        case 6: hipLaunchKernelGGL(fp8_planes_gemm_kernel<6>, dim3(grid), dim3(256), lds, s, a); break;
        case 7: hipLaunchKernelGGL(fp8_planes_gemm_kernel<7>, dim3(grid), dim3(256), lds, s, a); break;
        case 8: hipLaunchKernelGGL(fp8_planes_gemm_kernel<8>, dim3(grid), dim3(256), lds, s, a); break;
        default: return XFM_ELIMIT;
    }
    return check_launch();
}
