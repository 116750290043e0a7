// Device-side helpers shared by the gfx950 kernels of libxfm_hip.so.
// Written for CDNA4 only: 64-lane wavefronts, LDS tiles, no portability layer.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>

#include "../../include/xfm_hip.h"

namespace xfm {
// ---- host side: opt a kernel in to more than 64 KB of dynamic LDS, once per DEVICE -------------------------------------
// (hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute: a process-wide "done" flag would leave the second
//  GPU of a multi-device process with the 64 KB default.  `done` is a bit per device ordinal, updated atomically.)
struct LdsOptIn { unsigned long long done = 0; };
static inline bool lds_opt_in(LdsOptIn &st, const void *fn, size_t lds) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(&st.done, __ATOMIC_ACQUIRE) & bit) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
    __atomic_fetch_or(&st.done, bit, __ATOMIC_RELEASE);
    return true;
}


constexpr float kLog2e = 1.4426950408889634f;

using bf16_t = __hip_bfloat16;
using f16_t = __half;

// ---- scalar load / store with fp32 conversion -------------------------------------------------
template <typename T> __device__ __forceinline__ float ldf(const T *p);
template <> __device__ __forceinline__ float ldf<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ldf<f16_t>(const f16_t *p) { return __half2float(*p); }
template <> __device__ __forceinline__ float ldf<bf16_t>(const bf16_t *p) {
    return __uint_as_float(static_cast<uint32_t>(*reinterpret_cast<const uint16_t *>(p)) << 16);
}

template <typename T> __device__ __forceinline__ void stf(T *p, float v);
template <> __device__ __forceinline__ void stf<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<f16_t>(f16_t *p, float v) { *p = __float2half(v); }
template <> __device__ __forceinline__ void stf<bf16_t>(bf16_t *p, float v) { *p = __float2bfloat16(v); }  // RNE, NaN-safe cast

// two fp32 -> packed bf16 pair (round to nearest even) in ONE instruction: gfx950 has v_cvt_pk_bf16_f32
typedef __bf16 xfm_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float xfm_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const xfm_f32x2_t v = {lo, hi};
    const xfm_bf16x2_t r = __builtin_convertvector(v, xfm_bf16x2_t);
    return *reinterpret_cast<const uint32_t *>(&r);
}

// ---- 16-byte vector load / store with fp32 conversion -----------------------------------------
template <typename T> struct Pack;           // 16-byte vector of T  <->  fp32 lanes
template <> struct Pack<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void ld(const float *p, float *v) {
        const float4 r = *reinterpret_cast<const float4 *>(p);
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    }
    static __device__ __forceinline__ void st(float *p, const float *v) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct Pack<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void ld(const bf16_t *p, float *v) {
        const uint4 r = *reinterpret_cast<const uint4 *>(p);
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void st(bf16_t *p, const float *v) {
        uint4 r;
        r.x = pack_bf16x2(v[0], v[1]);
        r.y = pack_bf16x2(v[2], v[3]);
        r.z = pack_bf16x2(v[4], v[5]);
        r.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4 *>(p) = r;
    }
};

// ---- math ---------------------------------------------------------------------------------------
// torch.nn.functional.softplus with beta=1, threshold=20 (reference: models/csms6s.py:49-50,
// selective_scan_fwd_kernel.cuh:131-134).
// Hardware exp2/log2 (v_exp_f32 / v_log_f32, no denormal fix-ups: e^x underflowing to 0 is exact enough
// here) plus a 3-term series where 1+z would lose the small z (relative error < 1e-5).
// Optionally also returns d softplus / dx = sigmoid(x) = z / (1 + z) (one v_rcp_f32, no second exp).
__device__ __forceinline__ float softplus20_sig(float x, float &sig) {
    const float z = __builtin_amdgcn_exp2f(x * kLog2e);
    const float zp1 = 1.0f + z;
    const float series = z * fmaf(z, fmaf(z, 0.33333334f, -0.5f), 1.0f);   // log1p(z), |z| < 2^-5
    const float lg = __builtin_amdgcn_logf(zp1) * 0.6931471805599453f;
    const float sp = z < 0.03125f ? series : lg;
    const bool lin = x > 20.f;
    sig = lin ? 1.0f : z * __builtin_amdgcn_rcpf(zp1);
    return lin ? x : sp;
}
__device__ __forceinline__ float softplus20(float x) {
    float s;
    return softplus20_sig(x, s);
}
// The same function for results that are stored as 16-bit values: no log1p series (log2(1 + z) straight from v_log_f32 is off
// by ~6e-8 / z relative: far inside half a bf16 ulp for every step size above 1e-4) and no selects -- with r = x log2(e),
// log2(1 + 2^r) >= r and equals r in fp32 from r = 25 on, so max(log2(1 + 2^min(r, 64)), r) is the thresholded softplus
// (log1p(e^x) - x < 2.1e-9 beyond the threshold of 20) and never overflows.  9 issue slots instead of ~16.
// Below z = e^x = 2^-12 the sum 1 + z loses z's low bits (and all of z below 2^-24: the result would be 0 instead of e^x), so
// there log2(1 + z) is taken as z log2(e) (the next term, z / 2, is below a quarter of a bf16 ulp): relative accuracy holds
// for every step size, however far training drives it down.
__device__ __forceinline__ float softplus20_16bit(float x) {
    const float r = x * kLog2e;
    const float z = __builtin_amdgcn_exp2f(fminf(r, 64.f));
    const float t = z < 0.000244140625f ? z * kLog2e : __builtin_amdgcn_logf(1.0f + z);
    return fmaxf(t, r) * 0.6931471805599453f;
}

// exp(x * A) through the hardware exp2 (v_exp_f32); caller passes A pre-multiplied by log2(e).
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// Wave-level LDS hand-off: orders this wave's LDS writes before its later LDS reads (other lanes'
// data) without a workgroup barrier -- each wave owns a private LDS region.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- host-side error plumbing ------------------------------------------------------------------
void set_last_hip_error(hipError_t e);
int check_launch();
void prof_before_main(hipStream_t s);      // xfm_prof_main_kernel: events around an entry point's main kernel
void prof_after_main(hipStream_t s);

}  // namespace xfm
