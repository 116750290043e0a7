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

#include <cstdlib>
#include <type_traits>

namespace xfm {

// 1 / (1 + e^-z) on the raw v_exp_f32 / v_rcp_f32 (an IEEE division costs ~10 instructions: div_scale, div_fmas, div_fixup)
__device__ __forceinline__ float sigmoidf_fast(float z) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
}

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
    if (live) {                                                     // ONE atomic instruction: lane i adds total i
        const int ln = threadIdx.x & 63;
        float val = accb;
#pragma unroll
        for (int i = 0; i < 9; ++i) val = ln == i ? acc[i] : val;
        float *p = ln < 9 ? dw + d * 9 + ln : dbias + d;
        if (ln < 9 || (ln == 9 && dbias)) atomicAdd(p, val);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Fast path for maps whose rows are 7 vectors wide (W = 7 * VEC, VEC in {8,4,2,1}: the 56/28/14/7 maps of the trunk).
// A workgroup owns PP channels (one LDS plane each, with a zero halo that is written once) and walks a slice of the
// batch, so weights are loaded once and -- in the backward pass -- the 9 + 1 weight / bias gradient sums of a channel
// stay in registers across the whole slice: one atomic per tap per workgroup instead of one per plane and wavefront.
// TP = blockDim / PP threads share a plane; every thread moves whole 16-byte (VEC-element) vectors: one vector load,
// 3 x (1 vector + 2 scalar) LDS reads for the 3x3 window of VEC outputs, one vector store.
template <typename T> __device__ __forceinline__ float cvt16(uint16_t h);
template <> __device__ __forceinline__ float cvt16<bf16_t>(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
template <> __device__ __forceinline__ float cvt16<f16_t>(uint16_t h) { return __half2float(__ushort_as_half(h)); }
template <typename T> __device__ __forceinline__ uint16_t pack16(float f);
template <> __device__ __forceinline__ uint16_t pack16<bf16_t>(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
template <> __device__ __forceinline__ uint16_t pack16<f16_t>(float f) { return __half_as_ushort(__float2half(f)); }

template <typename T, int VEC> __device__ __forceinline__ void ld_vec(const T *p, float *v) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = reinterpret_cast<const float *>(p)[i];   // merged into dwordx2/x4 loads
    } else {
        uint16_t h[VEC];
        if constexpr (VEC == 8) *reinterpret_cast<uint4 *>(h) = *reinterpret_cast<const uint4 *>(p);
        else if constexpr (VEC == 4) *reinterpret_cast<uint2 *>(h) = *reinterpret_cast<const uint2 *>(p);
        else if constexpr (VEC == 2) *reinterpret_cast<uint32_t *>(h) = *reinterpret_cast<const uint32_t *>(p);
        else h[0] = *reinterpret_cast<const uint16_t *>(p);
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = cvt16<T>(h[i]);
    }
}
template <typename T, int VEC> __device__ __forceinline__ void st_vec(T *p, const float *v) {
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) reinterpret_cast<float *>(p)[i] = v[i];
    } else {
        uint16_t h[VEC];
        if constexpr (std::is_same<T, bf16_t>::value && VEC >= 2) {
#pragma unroll
            for (int i = 0; i < VEC; i += 2) *reinterpret_cast<uint32_t *>(h + i) = pack_bf16x2(v[i], v[i + 1]);
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) h[i] = pack16<T>(v[i]);
        }
        if constexpr (VEC == 8) *reinterpret_cast<uint4 *>(p) = *reinterpret_cast<const uint4 *>(h);
        else if constexpr (VEC == 4) *reinterpret_cast<uint2 *>(p) = *reinterpret_cast<const uint2 *>(h);
        else if constexpr (VEC == 2) *reinterpret_cast<uint32_t *>(p) = *reinterpret_cast<const uint32_t *>(h);
        else *reinterpret_cast<uint16_t *>(p) = h[0];
    }
}

struct Dw7Args {
    const void *x, *dy;
    const float *w, *bias;
    void *out;                       // y (fwd) / dx (bwd)
    float *dw, *dbias;
    int B, D, H, PP, TP, bsplit, act;
};

constexpr int kDwLeft = 4;           // interior starts 4 floats into a padded row: vector LDS accesses stay aligned
__host__ __device__ constexpr int dw7_pitch(int W) { return (W + kDwLeft + 1 + 3) & ~3; }

// window of row `row` for outputs c0 .. c0+VEC-1: q[0] = left neighbour, q[1..VEC] = the vector, q[VEC+1] = right
template <int VEC> __device__ __forceinline__ void ld_window(const float *row, int c0, float *q) {
    const float *m = row + kDwLeft + c0;
    q[0] = m[-1];
    if constexpr (VEC >= 4) {
#pragma unroll
        for (int i = 0; i < VEC; i += 4) {
            const float4 t = *reinterpret_cast<const float4 *>(m + i);
            q[1 + i] = t.x; q[2 + i] = t.y; q[3 + i] = t.z; q[4 + i] = t.w;
        }
    } else if constexpr (VEC == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(m);
        q[1] = t.x; q[2] = t.y;
    } else {
        q[1] = m[0];
    }
    q[VEC + 1] = m[VEC];
}

template <int VEC> __device__ __forceinline__ void st_row(float *dst, const float *v) {
    if constexpr (VEC >= 4) {
#pragma unroll
        for (int i = 0; i < VEC; i += 4) *reinterpret_cast<float4 *>(dst + i) = make_float4(v[i], v[i + 1], v[i + 2], v[i + 3]);
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<float2 *>(dst) = make_float2(v[0], v[1]);
    } else {
        dst[0] = v[0];
    }
}

// raw (unconverted) VEC-element global vectors held in registers between their load and their use
template <typename T, int VEC> struct DwRaw { static constexpr int NW = ((int)sizeof(T) * VEC + 3) / 4; };
template <typename T, int VEC> __device__ __forceinline__ void ld_raw(const T *p, uint32_t *w) {
    constexpr int NB = (int)sizeof(T) * VEC;
    if constexpr (NB == 32) {
        const uint4 t0 = reinterpret_cast<const uint4 *>(p)[0], t1 = reinterpret_cast<const uint4 *>(p)[1];
        w[0] = t0.x; w[1] = t0.y; w[2] = t0.z; w[3] = t0.w; w[4] = t1.x; w[5] = t1.y; w[6] = t1.z; w[7] = t1.w;
    } else if constexpr (NB == 16) {
        const uint4 t = *reinterpret_cast<const uint4 *>(p);
        w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
    } else if constexpr (NB == 8) {
        const uint2 t = *reinterpret_cast<const uint2 *>(p);
        w[0] = t.x; w[1] = t.y;
    } else if constexpr (NB == 4) {
        w[0] = *reinterpret_cast<const uint32_t *>(p);
    } else {
        w[0] = *reinterpret_cast<const uint16_t *>(p);
    }
}
template <typename T, int VEC> __device__ __forceinline__ void unpack_raw(const uint32_t *w, float *v) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
        if constexpr (sizeof(T) == 4) v[i] = __uint_as_float(w[i]);
        else v[i] = cvt16<T>((uint16_t)((w[i >> 1] >> (16 * (i & 1))) & 0xffffu));
    }
}

// vectors a thread moves per plane, at most: wave-private planes -> the 49*VEC vectors of a plane over 64 lanes;
// planes shared by TP = 256 / PP threads -> dw7_plan checks the bound
template <int VEC, bool WP, int LPR> struct Dw7Iters { static constexpr int value = WP ? (LPR * LPR * VEC + 63) / 64 : (VEC == 8 ? 4 : 7); };

// (LPR = vectors per row: 7 for the 224^2 maps, 6 for the 48 / 24 / 12 maps of 384^2 inputs)
template <typename T, int VEC, bool BWD, bool WP, int LPR = 7>
__global__ void __launch_bounds__(256) dwconv7_kernel(Dw7Args a) {
    constexpr int W = LPR * VEC, PITCH = dw7_pitch(W);
    constexpr int NIT = Dw7Iters<VEC, WP, LPR>::value, NW = DwRaw<T, VEC>::NW;
    extern __shared__ float smem[];
    const int H = a.H, PH = H + 2, L = H * W, psz = PH * PITCH;
    // WP (maps up to 28x28): ONE WAVE OWNS ONE CHANNEL -- its LDS planes are private, the three hand-offs per plane
    // (x staged -> dz written -> dx computed) are wave-level syncs and the four waves of a workgroup never wait for
    // each other.  Otherwise (56x56: a wave-private plane pair would leave one wave per SIMD, measured 1.5x slower)
    // TP = 256 / PP threads share a plane and the hand-offs are workgroup barriers.
    const int TP = WP ? 64 : a.TP, PP = WP ? 4 : a.PP;
    const int j = threadIdx.x / TP, tl = threadIdx.x - j * TP;
    const int ngrp = a.D / PP;
    const int grp = blockIdx.x % ngrp, sl = blockIdx.x / ngrp;
    const int d = grp * PP + j;
    auto sync = [&]() {
        if constexpr (WP) wave_sync();
        else __syncthreads();
    };
    const int b0 = (int)((int64_t)a.B * sl / a.bsplit), b1 = (int)((int64_t)a.B * (sl + 1) / a.bsplit);
    float *xs = smem + j * psz;                                    // padded x plane of this thread's channel
    float *zs = smem + (PP + j) * psz;                           // padded dz plane (backward only)
    // zero halo: only the cells the 3x3 windows read outside the map (top / bottom rows, one column left / right)
    for (int bufi = 0; bufi < (BWD ? 2 : 1); ++bufi) {
        float *pl = smem + (bufi * PP + j) * psz;
        for (int e = tl; e < 2 * PITCH; e += TP) pl[(e < PITCH ? 0 : (PH - 1) * PITCH) + (e < PITCH ? e : e - PITCH)] = 0.f;
        for (int e = tl; e < 2 * PH; e += TP) pl[(e >> 1) * PITCH + ((e & 1) ? kDwLeft + W : kDwLeft - 1)] = 0.f;
    }
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = a.w[d * 9 + i];
    const float bv = a.bias ? a.bias[d] : 0.f;
    float acc[9], accb = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = 0.f;
    const T *x = static_cast<const T *>(a.x), *dy = static_cast<const T *>(a.dy);
    T *out = static_cast<T *>(a.out);
    const int nvec = H * LPR;
    // Software pipeline over the planes of the batch slice: the x vectors of plane b+1 are requested as soon as plane
    // b's have been written to LDS, the dy vectors as soon as plane b's have been consumed -- the loads fly under the
    // stencil passes instead of each costing a round trip to HBM between two barriers.
    uint32_t xr[NIT][NW], gr[BWD ? NIT : 1][NW];
    int vh[NIT], vc[NIT];                                          // this thread's vectors: row, first column
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int v = tl + it * TP;
        vh[it] = v / LPR;
        vc[it] = (v - vh[it] * LPR) * VEC;
        if (v >= nvec) vh[it] = -1;
    }
    auto issue = [&](const T *src, uint32_t (&r)[NIT][NW], int b) {
        const int64_t po = ((int64_t)b * a.D + d) * L;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (vh[it] >= 0) ld_raw<T, VEC>(src + po + vh[it] * W + vc[it], r[it]);
    };
    if (b0 < b1) {
        issue(x, xr, b0);
        if constexpr (BWD) issue(dy, gr, b0);
    }
    for (int b = b0; b < b1; ++b) {
        const int64_t po = ((int64_t)b * a.D + d) * L;
        sync();                                                    // previous plane fully consumed (and halo zeroed)
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (vh[it] >= 0) {
                float t[VEC];
                unpack_raw<T, VEC>(xr[it], t);
                st_row<VEC>(xs + (vh[it] + 1) * PITCH + kDwLeft + vc[it], t);
            }
        sync();
        if (b + 1 < b1) issue(x, xr, b + 1);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (vh[it] < 0) continue;
            const int h = vh[it], c0 = vc[it];
            float q[3][VEC + 2];
#pragma unroll
            for (int r = 0; r < 3; ++r) ld_window<VEC>(xs + (h + r) * PITCH, c0, q[r]);
            float z[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float s = bv;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) s = fmaf(k[r * 3 + c], q[r][i + c], s);
                z[i] = s;
            }
            if constexpr (!BWD) {
                if (a.act) {
#pragma unroll
                    for (int i = 0; i < VEC; ++i) z[i] *= sigmoidf_fast(z[i]);
                }
                st_vec<T, VEC>(out + po + h * W + c0, z);
            } else {
                float g[VEC];
                unpack_raw<T, VEC>(gr[it], g);
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    if (a.act) {
                        const float s = sigmoidf_fast(z[i]);
                        g[i] *= s * fmaf(z[i], 1.f - s, 1.f);        // d silu(z)/dz = s * (1 + z * (1 - s))
                    }
                    accb += g[i];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) acc[r * 3 + c] = fmaf(g[i], q[r][i + c], acc[r * 3 + c]);
                }
                st_row<VEC>(zs + (h + 1) * PITCH + kDwLeft + c0, g);
            }
        }
        if constexpr (BWD) {
            if (b + 1 < b1) issue(dy, gr, b + 1);
            sync();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (vh[it] < 0) continue;
                const int h = vh[it], c0 = vc[it];
                float q[3][VEC + 2];
#pragma unroll
                for (int r = 0; r < 3; ++r) ld_window<VEC>(zs + (h + r) * PITCH, c0, q[r]);
                float o[VEC];
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    float s = 0.f;
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int c = 0; c < 3; ++c) s = fmaf(k[8 - (r * 3 + c)], q[r][i + c], s);
                    o[i] = s;
                }
                st_vec<T, VEC>(out + po + h * W + c0, o);
            }
        }
    }
    if constexpr (BWD && WP) {
        // fold the 64 partial sums of the channel: ten wave reductions, one atomic per tap per wave
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            float s = i < 9 ? acc[i] : accb;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (i < 9) acc[i] = s;
            else accb = s;
        }
        {                                                           // ONE atomic instruction: lane i adds total i
            float val = accb;
#pragma unroll
            for (int i = 0; i < 9; ++i) val = tl == i ? acc[i] : val;
            float *p = tl < 9 ? a.dw + d * 9 + tl : a.dbias + d;
            if (tl < 9 || (tl == 9 && a.dbias)) atomicAdd(p, val);
        }
    } else if constexpr (BWD) {
        // fold the TP partial sums of each channel through LDS (the plane buffers are free now)
        __syncthreads();
        float *red = smem;                                          // [PP][10][TP]
#pragma unroll
        for (int i = 0; i < 9; ++i) red[(j * 10 + i) * TP + tl] = acc[i];
        red[(j * 10 + 9) * TP + tl] = accb;
        __syncthreads();
        if (tl < 10) {
            float s = 0.f;
            for (int q = 0; q < TP; ++q) s += red[(j * 10 + tl) * TP + q];
            if (tl < 9) atomicAdd(a.dw + d * 9 + tl, s);
            else if (a.dbias) atomicAdd(a.dbias + d, s);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Register-stencil backward for the WIDE maps (56 x 56: VEC = 8, 28 x 28: VEC = 4; rows of 7 vectors).  The LDS forms
// above run these maps at ~1 TB/s: three LDS round trips and two or three hand-offs per plane.  Here nothing goes through
// LDS and nothing is synchronised: a lane owns one column strip (one 16- / 8-byte vector wide) of one plane and walks DOWN
// a band of RB rows with the 3 x (VEC + 2) windows of x and of dz rolling through its registers; the halo columns of a
// window are the edge elements of the neighbouring strips = the neighbouring lanes (one-lane DPP shifts).  A wave holds
// 8 planes x 7 strips (56 lanes): the same channel of 8 consecutive samples, so the weights are wave-uniform and the
// 9 + 1 weight / bias gradient sums fold across the wave at the end (one atomic per tap and wave).  The rows of a plane
// are cut into bands so that a launch has thousands of waves; a band recomputes dz for one row above and below it.
template <int CTRL, int ROW_MASK = 0xf> __device__ __forceinline__ float dw_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// sum over the 64 lanes on DPP adds (the total lands in lane 63): row-wise inclusive prefix (row_shr 1, 2, 4, 8), then the
// row totals carried across rows (row_bcast:15 into rows 1 / 3, row_bcast:31 into rows 2 / 3).  Ten of these are 60 VALU
// instructions; as __shfl_xor butterflies they were 60 ds_bpermute round trips at the end of every wave.
__device__ __forceinline__ float dw_wave_sum(float v) {
    v += dw_dpp<0x111>(v);
    v += dw_dpp<0x112>(v);
    v += dw_dpp<0x114>(v);
    v += dw_dpp<0x118>(v);
    v += dw_dpp<0x142, 0xa>(v);
    v += dw_dpp<0x143, 0xc>(v);
    return v;
}

// The ten wave totals of a wave (nine weight gradients, the bias gradient) leave as ONE atomic instruction -- lane i adds total
// i -- instead of ten single-lane ones: a CU retires about one atomic wave-instruction per 120 cycles whatever its lane count
// (MI355X_MICROARCH.md), and at 14 x 14 a CU's 24 waves x 10 instructions were 14 us of a 24 us launch.
__device__ __forceinline__ void dw_flush10(const float (&acc)[9], const float accb, const bool live, float *dw, float *dbias,
                                           const int d, const int lane) {
    float val = 0.f;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        float sm = i < 9 ? acc[i] : accb;
        if (!live) sm = 0.f;
        sm = dw_wave_sum(sm);
        const float tot = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sm), 63));
        val = lane == i ? tot : val;
    }
    float *p = lane < 9 ? dw + d * 9 + lane : dbias + d;
    if (lane < 9 || (lane == 9 && dbias)) atomicAdd(p, val);
}

template <typename T, int VEC, int RB, int LPR = 7>
__global__ void __launch_bounds__(256) dwconv_strip_bwd_kernel(const T *__restrict__ x, const float *__restrict__ w,
                                                               const float *__restrict__ bias, const T *__restrict__ dy,
                                                               T *__restrict__ dx, float *__restrict__ dw,
                                                               float *__restrict__ dbias, int B, int D, int H, int act) {
    constexpr int W = LPR * VEC, PPW = 64 / LPR > 8 ? 8 : 64 / LPR;    // strips per row, planes per wave
    const int lane = threadIdx.x & 63;
    const int nbands = H / RB, nbg = (B + PPW - 1) / PPW;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int band = gw % nbands, t1 = gw / nbands, bg = t1 % nbg, d = t1 / nbg;
    if (d >= D) return;                                            // (whole wave)
    const int pslot = lane / LPR, sp = lane - pslot * LPR;
    const int b = bg * PPW + pslot;
    const bool live = pslot < PPW && b < B;
    const int64_t po = live ? (((int64_t)b * D + d) * H) * W + sp * VEC : 0;
    const T *xg = x + po, *gg = dy + po;
    T *og = dx + po;
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = w[d * 9 + i];
    const float bv = bias ? bias[d] : 0.f;
    const int r0 = band * RB, r1 = r0 + RB;
    const bool hasl = sp > 0, hasr = sp < LPR - 1;
    // one row of a plane as VEC floats + the two halo columns from the neighbouring strips (zero outside the map)
    auto load_raw = [&](const T *base, int r, uint32_t (&raw)[DwRaw<T, VEC>::NW]) {
#pragma unroll
        for (int i = 0; i < DwRaw<T, VEC>::NW; ++i) raw[i] = 0u;
        if (live && r >= 0 && r < H) ld_raw<T, VEC>(base + (int64_t)r * W, raw);
    };
    auto widen = [&](const float (&v)[VEC], float (&q)[VEC + 2]) {
        const float l = dw_dpp<0x138>(v[VEC - 1]), rr = dw_dpp<0x130>(v[0]);     // wave_shr:1 / wave_shl:1
        q[0] = hasl ? l : 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) q[1 + i] = v[i];
        q[VEC + 1] = hasr ? rr : 0.f;
    };
    float xw[3][VEC + 2], gz[3][VEC + 2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < VEC + 2; ++c) gz[i][c] = 0.f;
    {
        uint32_t raw[DwRaw<T, VEC>::NW];
        float v[VEC];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            load_raw(xg, r0 - 2 + i, raw);
            unpack_raw<T, VEC>(raw, v);
            widen(v, xw[i]);
        }
    }
    float acc[9], accb = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = 0.f;
    // operand rows are requested one iteration ahead (two ahead measured the same: the kernel is VALU-bound, SQ counters)
    uint32_t xn[DwRaw<T, VEC>::NW], gn[DwRaw<T, VEC>::NW];
    load_raw(gg, r0 - 1, gn);
    load_raw(xg, r0 + 1, xn);
    // one row: R0 / R1 / R2 name the window slots holding rows t-1, t, t+1 (x) resp. t-2, t-1, t (dz) -- the loop below is
    // written out three times with the slots rotated, so the windows roll by renaming, not by 40 moves per row
    auto step = [&](auto r0_tag, auto r1_tag, auto r2_tag, const int t) {
        constexpr int R0 = decltype(r0_tag)::value, R1 = decltype(r1_tag)::value, R2 = decltype(r2_tag)::value;
        // ---- dz of row t from the x window (rows t-1, t, t+1 in slots R0, R1, R2) and dy row t
        float g[VEC];
        unpack_raw<T, VEC>(gn, g);
        uint32_t gnn[DwRaw<T, VEC>::NW], xnn[DwRaw<T, VEC>::NW];
        load_raw(gg, t + 1, gnn);                                  // operands of the next iteration
        load_raw(xg, t + 3, xnn);
        const bool inmap = t >= 0 && t < H, own = t >= r0 && t < r1;
        const float (*xr[3])[VEC + 2] = {&xw[R0], &xw[R1], &xw[R2]};
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            float z = bv;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) z = fmaf(k[i * 3 + j], (*xr[i])[c + j], z);
            if (act) {
                const float sg = sigmoidf_fast(z);
                g[c] *= sg * fmaf(z, 1.f - sg, 1.f);
            }
            if (!inmap) g[c] = 0.f;
            if (own) {
                accb += g[c];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) acc[i * 3 + j] = fmaf(g[c], (*xr[i])[c + j], acc[i * 3 + j]);
            }
        }
        // ---- dz row t goes to the slot of the oldest dz row (t-3); dx of row t-1 from dz rows t-2, t-1, t
        widen(g, gz[R0]);                                          // dz slots: rows t-2, t-1 in R1, R2, row t now in R0
        if (t - 1 >= r0) {                                         // (t - 1 < r1 by the loop bound)
            const float (*gr[3])[VEC + 2] = {&gz[R1], &gz[R2], &gz[R0]};
            float o[VEC];
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                float v = 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) v = fmaf(k[8 - (i * 3 + j)], (*gr[i])[c + j], v);
                o[c] = v;
            }
            if (live) st_vec<T, VEC>(og + (int64_t)(t - 1) * W, o);
        }
        // ---- x row t+2 replaces row t-1 (slot R0): the next iteration sees rows t, t+1, t+2 in R1, R2, R0
        {
            float v[VEC];
            unpack_raw<T, VEC>(xn, v);
            widen(v, xw[R0]);
        }
#pragma unroll
        for (int i = 0; i < DwRaw<T, VEC>::NW; ++i) {
            gn[i] = gnn[i];
            xn[i] = xnn[i];
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
#pragma unroll 1
    for (int t = r0 - 1; t <= r1; t += 3) {
        step(I0{}, I1{}, I2{}, t);
        if (t + 1 <= r1) step(I1{}, I2{}, I0{}, t + 1);
        if (t + 2 <= r1) step(I2{}, I0{}, I1{}, t + 2);
    }
    // ---- weight / bias gradient sums: all live lanes of the wave belong to channel d
    dw_flush10(acc, accb, live, dw, dbias, d, lane);
}

// forward of the same decomposition: y row t = silu(bias + 3 x 3 window of x rows t-1 .. t+1)
template <typename T, int VEC, int RB, int LPR = 7>
__global__ void __launch_bounds__(256) dwconv_strip_fwd_kernel(const T *__restrict__ x, const float *__restrict__ w,
                                                               const float *__restrict__ bias, T *__restrict__ y, int B, int D,
                                                               int H, int act) {
    constexpr int W = LPR * VEC, PPW = 64 / LPR > 8 ? 8 : 64 / LPR;    // strips per row, planes per wave
    const int lane = threadIdx.x & 63;
    const int nbands = H / RB, nbg = (B + PPW - 1) / PPW;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int band = gw % nbands, t1 = gw / nbands, bg = t1 % nbg, d = t1 / nbg;
    if (d >= D) return;
    const int pslot = lane / LPR, sp = lane - pslot * LPR;
    const int b = bg * PPW + pslot;
    const bool live = pslot < PPW && b < B;
    const int64_t po = live ? (((int64_t)b * D + d) * H) * W + sp * VEC : 0;
    const T *xg = x + po;
    T *og = y + po;
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = w[d * 9 + i];
    const float bv = bias ? bias[d] : 0.f;
    const int r0 = band * RB, r1 = r0 + RB;
    const bool hasl = sp > 0, hasr = sp < LPR - 1;
    auto load_raw = [&](int r, uint32_t (&raw)[DwRaw<T, VEC>::NW]) {
#pragma unroll
        for (int i = 0; i < DwRaw<T, VEC>::NW; ++i) raw[i] = 0u;
        if (live && r >= 0 && r < H) ld_raw<T, VEC>(xg + (int64_t)r * W, raw);
    };
    auto widen = [&](const uint32_t (&raw)[DwRaw<T, VEC>::NW], float (&q)[VEC + 2]) {
        float v[VEC];
        unpack_raw<T, VEC>(raw, v);
        const float l = dw_dpp<0x138>(v[VEC - 1]), rr = dw_dpp<0x130>(v[0]);
        q[0] = hasl ? l : 0.f;
#pragma unroll
        for (int i = 0; i < VEC; ++i) q[1 + i] = v[i];
        q[VEC + 1] = hasr ? rr : 0.f;
    };
    float xw[3][VEC + 2];
    uint32_t xn[DwRaw<T, VEC>::NW], xn2[DwRaw<T, VEC>::NW];
    {
        uint32_t raw[DwRaw<T, VEC>::NW];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            load_raw(r0 - 1 + i, raw);
            widen(raw, xw[i]);
        }
    }
    load_raw(r0 + 2, xn);
    load_raw(r0 + 3, xn2);
    auto step = [&](auto a_tag, auto b_tag, auto c_tag, const int t) {
        constexpr int R0 = decltype(a_tag)::value, R1 = decltype(b_tag)::value, R2 = decltype(c_tag)::value;
        uint32_t xnn[DwRaw<T, VEC>::NW];
        load_raw(t + 4, xnn);                                      // two rows ahead of the row consumed next
        const float (*xr[3])[VEC + 2] = {&xw[R0], &xw[R1], &xw[R2]};
        float o[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            float z = bv;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) z = fmaf(k[i * 3 + j], (*xr[i])[c + j], z);
            o[c] = act ? z * sigmoidf_fast(z) : z;
        }
        if (live) st_vec<T, VEC>(og + (int64_t)t * W, o);
        widen(xn, xw[R0]);                                         // row t + 2 replaces row t - 1
#pragma unroll
        for (int i = 0; i < DwRaw<T, VEC>::NW; ++i) {
            xn[i] = xn2[i];
            xn2[i] = xnn[i];
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
#pragma unroll 1
    for (int t = r0; t < r1; t += 3) {
        step(I0{}, I1{}, I2{}, t);
        if (t + 1 < r1) step(I1{}, I2{}, I0{}, t + 1);
        if (t + 2 < r1) step(I2{}, I0{}, I1{}, t + 2);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 14 x 14 (12 x 12) maps: a lane owns one ROW of one plane (28 / 24 bytes), the rows above and below are the neighbouring
// lanes (one-lane DPP shifts of the whole row), the columns left and right are its own registers.  No loop, no LDS: load the
// row, take the two neighbour rows, compute, store.  A wave holds 64 / H planes of one channel (H consecutive lanes each).
template <int HW> __device__ __forceinline__ void dwrow_load(const uint16_t *p, bool ok, float (&v)[HW + 2]) {
    if constexpr (HW % 2 != 0) {                       // 7 x 7: rows of 14 bytes, 2-byte aligned
        struct __attribute__((packed, aligned(2))) Raw2 { uint16_t h[HW]; };
        Raw2 r2;
#pragma unroll
        for (int i = 0; i < HW; ++i) r2.h[i] = 0;
        if (ok) r2 = *reinterpret_cast<const Raw2 *>(p);
        v[0] = 0.f;
#pragma unroll
        for (int i = 0; i < HW; ++i) v[1 + i] = __uint_as_float((uint32_t)r2.h[i] << 16);
        v[HW + 1] = 0.f;
        return;
    }
    struct __attribute__((packed, aligned(4))) Raw { uint32_t w[HW / 2]; };
    Raw r;
#pragma unroll
    for (int i = 0; i < HW / 2; ++i) r.w[i] = 0u;
    if (ok) r = *reinterpret_cast<const Raw *>(p);
    v[0] = 0.f;
#pragma unroll
    for (int i = 0; i < HW / 2; ++i) {
        v[1 + 2 * i] = __uint_as_float(r.w[i] << 16);
        v[2 + 2 * i] = __uint_as_float(r.w[i] & 0xffff0000u);
    }
    v[HW + 1] = 0.f;
}
template <int HW> __device__ __forceinline__ void dwrow_store(uint16_t *p, const float (&v)[HW]) {
    if constexpr (HW % 2 != 0) {
        struct __attribute__((packed, aligned(2))) Raw2 { uint16_t h[HW]; };
        Raw2 r2;
#pragma unroll
        for (int i = 0; i < HW; ++i) r2.h[i] = (uint16_t)(pack_bf16x2(v[i], 0.f) & 0xffffu);
        *reinterpret_cast<Raw2 *>(p) = r2;
        return;
    }
    struct __attribute__((packed, aligned(4))) Raw { uint32_t w[HW / 2]; };
    Raw r;
#pragma unroll
    for (int i = 0; i < HW / 2; ++i) r.w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<Raw *>(p) = r;
}
// rows r-1 / r+1 of the same plane = lanes -1 / +1 (zero at the map's edge)
template <int HW> __device__ __forceinline__ void dwrow_neigh(const float (&own)[HW + 2], bool hasu, bool hasd, float (&up)[HW + 2],
                                                              float (&dn)[HW + 2]) {
    up[0] = up[HW + 1] = dn[0] = dn[HW + 1] = 0.f;
#pragma unroll
    for (int c = 1; c <= HW; ++c) {
        const float u = dw_dpp<0x138>(own[c]), d = dw_dpp<0x130>(own[c]);
        up[c] = hasu ? u : 0.f;
        dn[c] = hasd ? d : 0.f;
    }
}

template <int HW, bool BWD>
__global__ void __launch_bounds__(256) dwconv_rowlane_kernel(const uint16_t *__restrict__ x, const float *__restrict__ w,
                                                             const float *__restrict__ bias, const uint16_t *__restrict__ dy,
                                                             uint16_t *__restrict__ out, float *__restrict__ dw,
                                                             float *__restrict__ dbias, int B, int D, int act) {
    constexpr int PPW = 64 / HW, L = HW * HW;
    const int lane = threadIdx.x & 63;
    const int nbg = (B + PPW - 1) / PPW;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int bg = gw % nbg, d = gw / nbg;
    if (d >= D) return;                                            // (whole wave)
    const int pslot = lane / HW, r = lane - pslot * HW;
    const int b = bg * PPW + pslot;
    const bool live = pslot < PPW && b < B;
    const int64_t po = live ? ((int64_t)b * D + d) * L + r * HW : 0;
    float k[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) k[i] = w[d * 9 + i];
    const float bv = bias ? bias[d] : 0.f;
    const bool hasu = r > 0, hasd = r < HW - 1;
    float X[3][HW + 2];
    dwrow_load<HW>(x + po, live, X[1]);
    float g[HW + 2];
    if constexpr (BWD) dwrow_load<HW>(dy + po, live, g);
    dwrow_neigh<HW>(X[1], hasu, hasd, X[0], X[2]);
    float z[HW];
#pragma unroll
    for (int c = 0; c < HW; ++c) {
        float s = bv;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) s = fmaf(k[i * 3 + j], X[i][c + j], s);
        z[c] = s;
    }
    if constexpr (!BWD) {
        if (act) {
#pragma unroll
            for (int c = 0; c < HW; ++c) z[c] *= sigmoidf_fast(z[c]);
        }
        if (live) dwrow_store<HW>(out + po, z);
    } else {
        float acc[9], accb = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[i] = 0.f;
        float G[3][HW + 2];
        G[1][0] = G[1][HW + 1] = 0.f;
#pragma unroll
        for (int c = 0; c < HW; ++c) {
            float gv = g[1 + c];
            if (act) {
                const float sg = sigmoidf_fast(z[c]);
                gv *= sg * fmaf(z[c], 1.f - sg, 1.f);
            }
            if (!live) gv = 0.f;
            G[1][1 + c] = gv;
            accb += gv;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) acc[i * 3 + j] = fmaf(gv, X[i][c + j], acc[i * 3 + j]);
        }
        dwrow_neigh<HW>(G[1], hasu, hasd, G[0], G[2]);
        float o[HW];
#pragma unroll
        for (int c = 0; c < HW; ++c) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) v = fmaf(k[8 - (i * 3 + j)], G[i][c + j], v);
            o[c] = v;
        }
        if (live) dwrow_store<HW>(out + po, o);
        dw_flush10(acc, accb, true, dw, dbias, d, lane);
    }
}

// plan of the fast path: PP channels per workgroup (TP = 256 / PP threads each), batch split into `bsplit` slices
static bool dw7_plan_shared(bool bwd, int B, int D, int H, int W, int lpr, int &vec, int &PP, int &bsplit, size_t &lds) {
    if (W % lpr != 0) return false;
    vec = W / lpr;
    if (vec != 1 && vec != 2 && vec != 4 && vec != 8) return false;
    if (!bwd && vec == 8) return false;          // measured: the one-plane-per-workgroup kernel is faster for the 56x56 forward
    const size_t psz = (size_t)(H + 2) * dw7_pitch(W) * sizeof(float);
    const int nvec = H * lpr;
    PP = 32;
    const int nit_max = vec == 8 ? 4 : 7;               // Dw7Iters: register-held vectors per thread
    while (PP > 1 && (256 / PP < 10 || nvec * PP > 1568 * 2 || (bwd ? 2 : 1) * PP * psz > 60 * 1024 || D % PP != 0 ||
                      (nvec + 256 / PP - 1) / (256 / PP) > nit_max))
        PP >>= 1;
    if (D % PP != 0 || (bwd ? 2 : 1) * PP * psz > 64 * 1024) return false;
    if ((nvec + 256 / PP - 1) / (256 / PP) > nit_max) return false;
    lds = (bwd ? 2 : 1) * PP * psz;
    if (bwd && lds < (size_t)PP * 10 * (256 / PP) * sizeof(float)) lds = (size_t)PP * 10 * (256 / PP) * sizeof(float);
    const int ngrp = D / PP;
    bsplit = 1;
    while (ngrp * bsplit < 1024 && bsplit < B) bsplit *= 2;
    if (const char *e = getenv("XFM_DW_BSPLIT")) bsplit = std::max(1, atoi(e));   // tuning hook
    if (bsplit > B) bsplit = B;
    return true;
}

// plan of the wave-private variant: four channels (waves) per workgroup, the batch split into `bsplit` slices so that the grid is
// about one round of resident waves -- each wave then walks several planes and the plane pipeline has something to hide
static bool dw7_plan_wave(bool bwd, int B, int D, int H, int W, int lpr, int &vec, int &PP, int &bsplit, size_t &lds) {
    if (W % lpr != 0 || H != W || D % 4 != 0) return false;
    vec = W / lpr;
    if (vec != 1 && vec != 2 && vec != 4 && vec != 8) return false;
    const size_t psz = (size_t)(H + 2) * dw7_pitch(W) * sizeof(float);
    PP = 4;
    lds = (bwd ? 2 : 1) * PP * psz;
    if (lds > 160 * 1024) return false;
    const size_t wg_per_cu = std::min<size_t>(8, (160 * 1024) / lds);
    const int64_t slots = (int64_t)256 * wg_per_cu * 4;                // resident waves
    bsplit = (int)((slots + D / 2) / D);
    if (bsplit < 1) bsplit = 1;
    if (bsplit > B) bsplit = B;
    return true;
}

// wave-private planes up to 14x14 (forward) / 28x28 (backward), shared planes above (measured both ways per shape)
static bool dw7_plan(bool bwd, int B, int D, int H, int W, int &vec, int &PP, int &bsplit, size_t &lds, bool &wp, int &lpr) {
    lpr = W % 7 == 0 ? 7 : 6;                             // rows of 7 vectors (224^2 inputs) or of 6 (384^2: 48 / 24 / 12)
    if (W % lpr != 0 || (lpr == 6 && W / 6 != 8 && W / 6 != 4 && W / 6 != 2)) return false;
    wp = H == W && (W / lpr <= 2 || (bwd && W / lpr == 4)) && D % 4 == 0;
    if (wp) return dw7_plan_wave(bwd, B, D, H, W, lpr, vec, PP, bsplit, lds);
    return dw7_plan_shared(bwd, B, D, H, W, lpr, vec, PP, bsplit, lds);
}

template <typename T, int VEC, bool BWD, bool WP, int LPR = 7> static int launch_dw7v(const Dw7Args &a, int grid, size_t lds, hipStream_t s) {
    auto fn = dwconv7_kernel<T, VEC, BWD, WP, LPR>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, s, a);
    return check_launch();
}

template <typename T, bool BWD> static int launch_dw7(int vec, bool wp, int lpr, const Dw7Args &a, int grid, size_t lds, hipStream_t s) {
    if (lpr == 6) {                                       // 48 x 48 (shared planes), 24 x 24 (wave-private in the backward), 12 x 12
        if (wp) {
            if (vec == 4) return BWD ? launch_dw7v<T, 4, BWD, true, 6>(a, grid, lds, s) : XFM_ELIMIT;
            if (vec == 2) return launch_dw7v<T, 2, BWD, true, 6>(a, grid, lds, s);
            return XFM_ELIMIT;
        }
        if (vec == 8) return launch_dw7v<T, 8, BWD, false, 6>(a, grid, lds, s);
        if (vec == 4) return launch_dw7v<T, 4, BWD, false, 6>(a, grid, lds, s);
        if (vec == 2) return launch_dw7v<T, 2, BWD, false, 6>(a, grid, lds, s);
        return XFM_ELIMIT;
    }
    if (wp) {
        switch (vec) {
            case 4: return BWD ? launch_dw7v<T, 4, BWD, true>(a, grid, lds, s) : XFM_ELIMIT;
            case 2: return launch_dw7v<T, 2, BWD, true>(a, grid, lds, s);
            case 1: return launch_dw7v<T, 1, BWD, true>(a, grid, lds, s);
        }
        return XFM_ELIMIT;
    }
    switch (vec) {
        case 8: return launch_dw7v<T, 8, BWD, false>(a, grid, lds, s);
        case 4: return launch_dw7v<T, 4, BWD, false>(a, grid, lds, s);
        case 2: return launch_dw7v<T, 2, BWD, false>(a, grid, lds, s);
    }
    return launch_dw7v<T, 1, BWD, false>(a, grid, lds, s);
}

static int pick_pp(int L) { return L >= 1024 ? 1 : (L >= 256 ? 2 : 4); }

template <typename T>
static int launch_dw(bool bwd, const void *x, const float *w, const float *bias, const void *dy, void *out, float *dw,
                     float *dbias, int B, int D, int H, int W, int act, hipStream_t s) {
    if (std::is_same<T, bf16_t>::value && H == W && (W == 14 || W == 12 || (W == 7 && !getenv("XFM_DWCONV_NO_ROWLANE7"))) &&
        !getenv("XFM_DWCONV_NO_ROWLANE") && !getenv("XFM_DWCONV_GENERIC")) {
        // 14 x 14 / 12 x 12 / 7 x 7: one lane per row (7 x 7: nine planes per wave; the wave-private plane kernel spent a whole
        // wave's instruction stream on 49 positions -- 37 us for the deep block's 14 MB backward)
        const int ppw = 64 / W;
        const int nwaves = D * ((B + ppw - 1) / ppw);
        const dim3 grid((nwaves + 3) / 4);
#define XFM_DW_ROW(HW)                                                                                                 \
    do {                                                                                                               \
        if (bwd)                                                                                                       \
            hipLaunchKernelGGL((dwconv_rowlane_kernel<HW, true>), grid, dim3(256), 0, s, (const uint16_t *)x, w, bias,  \
                               (const uint16_t *)dy, (uint16_t *)out, dw, dbias, B, D, act);                           \
        else                                                                                                           \
            hipLaunchKernelGGL((dwconv_rowlane_kernel<HW, false>), grid, dim3(256), 0, s, (const uint16_t *)x, w, bias, \
                               (const uint16_t *)nullptr, (uint16_t *)out, (float *)nullptr, (float *)nullptr, B, D, act); \
        return check_launch();                                                                                         \
    } while (0)
        if (W == 14) XFM_DW_ROW(14);
        else if (W == 12) XFM_DW_ROW(12);
        else XFM_DW_ROW(7);
#undef XFM_DW_ROW
    }
    // (14 x 14 maps measured no better on the strip kernels: forward 10.1 vs 11.6 us, backward 29.1 vs 24.7 us)
    if (sizeof(T) == 2 && H == W && !getenv("XFM_DWCONV_NO_STRIP") && !getenv("XFM_DWCONV_GENERIC")) {
        // wide maps: the register-stencil kernels.  (map, vector, band rows, strips per row); planes per wave = min(8, 64 / strips)
#define XFM_DW_STRIP(VEC, RB, LPR)                                                                                        \
    do {                                                                                                                  \
        constexpr int ppw = 64 / LPR > 8 ? 8 : 64 / LPR;                                                                  \
        const int nwaves = D * ((B + ppw - 1) / ppw) * (H / RB);                                                          \
        const dim3 grid((nwaves + 3) / 4);                                                                                \
        if (bwd)                                                                                                          \
            hipLaunchKernelGGL((dwconv_strip_bwd_kernel<T, VEC, RB, LPR>), grid, dim3(256), 0, s, (const T *)x, w, bias,  \
                               (const T *)dy, (T *)out, dw, dbias, B, D, H, act);                                         \
        else                                                                                                              \
            hipLaunchKernelGGL((dwconv_strip_fwd_kernel<T, VEC, RB, LPR>), grid, dim3(256), 0, s, (const T *)x, w, bias,  \
                               (T *)out, B, D, H, act);                                                                   \
        return check_launch();                                                                                            \
    } while (0)
        switch (W) {
            case 56: XFM_DW_STRIP(8, 14, 7);
            case 28: XFM_DW_STRIP(4, 14, 7);
            case 48: XFM_DW_STRIP(8, 12, 6);                      // XFMamba-B at 384 x 384
            case 24: XFM_DW_STRIP(4, 12, 6);
            case 96: XFM_DW_STRIP(8, 12, 12);
        }
#undef XFM_DW_STRIP
    }
    {
        int vec, PP, bsplit;
        size_t lds7;
        bool wp;
        int lpr;
        if (!getenv("XFM_DWCONV_GENERIC") && dw7_plan(bwd, B, D, H, W, vec, PP, bsplit, lds7, wp, lpr)) {
            Dw7Args a{};
            a.x = x; a.dy = dy; a.w = w; a.bias = bias; a.out = out; a.dw = dw; a.dbias = dbias;
            a.B = B; a.D = D; a.H = H; a.PP = PP; a.TP = 256 / PP; a.bsplit = bsplit; a.act = act;
            const int grid = (D / PP) * bsplit;
            return bwd ? launch_dw7<T, true>(vec, wp, lpr, a, grid, lds7, s) : launch_dw7<T, false>(vec, wp, lpr, a, grid, lds7, s);
        }
    }
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
