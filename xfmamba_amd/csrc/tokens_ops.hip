// Element-wise pieces of the Mlp on the token-major stream (models/fusion_vmamba.py:135-153:
// fc1 -> act -> drop -> fc2 -> drop) that sit between the library GEMMs:
//
//   bias_gelu fwd :  g  = gelu(z + b)                      (z = x @ W1^T from the GEMM, exact erf GELU = nn.GELU())
//   bias_gelu bwd :  dz = dg * gelu'(z + b),  db = column sums of dz      (one pass; no separate bias reduction)
//   colsum        :  out[c] = sum over rows of x[., c]      (fc2 / generic bias gradients)
//
// (rows, C) row-major; one thread owns one 16-byte channel group and walks rows, so the column sums stay in
// registers; a workgroup is NT = C / VEC threads wide and R rows deep.  Column sums leave the kernel as one
// partial row per workgroup and are folded by colsum_finish_kernel (deterministic, no atomics).
// HBM-bound: 2 tensor passes forward, 3 backward.
#include "xfm_common.hpp"

#include <algorithm>

namespace xfm {

constexpr float kInvSqrt2 = 0.70710678118654752f;
constexpr float kInvSqrt2Pi = 0.3989422804014327f;

// erf(x) by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. below fp32 resolution of the GELU it feeds) on the
// hardware exp2 / rcp; also returns E = exp(-x^2), which is the Gaussian density the GELU derivative needs.
__device__ __forceinline__ float erf_as(float x, float &E) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    E = __builtin_amdgcn_exp2f(-(x * x) * kLog2e);
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    const float r = fmaf(-(p * t), E, 1.0f);
    return copysignf(r, x);
}

struct TokArgs {
    const void *z;        // (rows, C)
    const float *bias;    // (C) or null
    const void *dg;       // (rows, C) backward only
    void *out;            // g (fwd) / dz (bwd); null for colsum
    float *part;          // (gridDim.x, C) partial column sums (bwd / colsum)
    long rows;
    int C, NT, R;         // NT = C / VEC threads per row, R rows per workgroup pass
};

// MODE 0: bias_gelu fwd, 1: bias_gelu bwd (+ column sums of dz), 2: column sums of z
template <typename T, int MODE> __global__ void tokens_kernel(TokArgs a) {
    constexpr int V = Pack<T>::N;
    extern __shared__ float red[];                 // (R, C) for the column sums
    const int t = threadIdx.x % a.NT, rs = threadIdx.x / a.NT;
    const int c0 = t * V;
    float b[V], acc[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        b[i] = (MODE != 2 && a.bias) ? a.bias[c0 + i] : 0.f;
        acc[i] = 0.f;
    }
    const T *z = static_cast<const T *>(a.z);
    const T *dg = static_cast<const T *>(a.dg);
    T *out = static_cast<T *>(a.out);
    // UN rows per trip, all loads issued before the maths: a thread keeps 2*UN 16-byte loads in flight
    constexpr int UN = 4;
    const long step = (long)gridDim.x * a.R;
    for (long r = (long)blockIdx.x * a.R + rs; r < a.rows; r += UN * step) {
        float v[UN][V], d[UN][V];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long ru = r + u * step;
            if (ru < a.rows) {
                Pack<T>::ld(z + ru * a.C + c0, v[u]);
                if (MODE == 1) Pack<T>::ld(dg + ru * a.C + c0, d[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long ru = r + u * step;
            if (ru >= a.rows) break;
            const long off = ru * a.C + c0;
            float o[V];
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float x = v[u][i] + b[i];
                    float E;
                    o[i] = 0.5f * x * (1.0f + erf_as(x * kInvSqrt2, E));
                }
                Pack<T>::st(out + off, o);
            } else if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float x = v[u][i] + b[i];
                    float E;
                    const float cdf = 0.5f * (1.0f + erf_as(x * kInvSqrt2, E));
                    const float pdf = kInvSqrt2Pi * E;
                    o[i] = d[u][i] * fmaf(x, pdf, cdf);
                    acc[i] += o[i];
                }
                Pack<T>::st(out + off, o);
            } else {
#pragma unroll
                for (int i = 0; i < V; ++i) acc[i] += v[u][i];
            }
        }
    }
    if (MODE != 0) {
#pragma unroll
        for (int i = 0; i < V; ++i) red[rs * a.C + c0 + i] = acc[i];
        __syncthreads();
        float *part = a.part + (long)blockIdx.x * a.C;
        for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
            float s = 0.f;
            for (int j = 0; j < a.R; ++j) s += red[j * a.C + c];
            part[c] = s;
        }
    }
}

// out[c] = sum_j part[j, c].  64 channels x 16 row slots per workgroup: the slots stride through the partial rows
// (coalesced 256-byte reads, independent loads in flight), then fold through LDS.
// ---- end-of-stage residual settle: out = x + scale[b] * (y + y_bias), emitted in the consumer's dtype -----------------
// (the last block of a trunk stage has no following LayerNorm to fold its residual add into; what follows is the
// downsample convolution, which under autocast reads bf16 -- as framework ops this was bias add, scale, add and the
// convolution's input cast: four passes over the stream instead of one)
template <typename To> __device__ __forceinline__ void settle_store8(To *p, const float (&v)[8]);
template <> __device__ __forceinline__ void settle_store8<float>(float *p, const float (&v)[8]) {
    *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4 *>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <> __device__ __forceinline__ void settle_store8<bf16_t>(bf16_t *p, const float (&v)[8]) {
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4 *>(p) = o;
}
__device__ __forceinline__ void settle_load8(const float *p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void settle_load8(const bf16_t *p, float (&v)[8]) {
    const uint4 r = *reinterpret_cast<const uint4 *>(p);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v[2 * q] = __uint_as_float(w[q] << 16);
        v[2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u);
    }
}

// one thread = 8 consecutive channels of one row; rows = B * rows_per_sample, C % 8 == 0
template <typename Ty, typename To>
__global__ void __launch_bounds__(256) settle_fwd_kernel(const float *__restrict__ x, const Ty *__restrict__ y,
                                                         const float *__restrict__ scale, const float *__restrict__ yb,
                                                         To *__restrict__ out, long long nvec, int C8, int rows_per_sample) {
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.x * 256) {
        const long long row = v / C8;
        const int c = (int)(v - row * C8) * 8;
        const float s = scale ? scale[row / rows_per_sample] : 1.f;
        float xv[8], yv[8], o[8];
        settle_load8(x + v * 8, xv);
        settle_load8(y + v * 8, yv);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = fmaf(s, yv[q] + (yb ? yb[c + q] : 0.f), xv[q]);
        settle_store8<To>(out + v * 8, o);
    }
}

// dx = dout (fp32), dy = scale[b] * dout (y's dtype)
template <typename Ty, typename To>
__global__ void __launch_bounds__(256) settle_bwd_kernel(const To *__restrict__ dout, const float *__restrict__ scale,
                                                         float *__restrict__ dx, Ty *__restrict__ dy, long long nvec, int C8,
                                                         int rows_per_sample) {
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.x * 256) {
        const long long row = v / C8;
        const float s = scale ? scale[row / rows_per_sample] : 1.f;
        float g[8], o[8];
        settle_load8(dout + v * 8, g);
        settle_store8<float>(dx + v * 8, g);
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = s * g[q];
        settle_store8<Ty>(dy + v * 8, o);
    }
}

// many producers' partial rows in one launch (xfm_partial_sums_multi): 64 columns x 16 row slots per workgroup
__global__ __launch_bounds__(1024) void partial_sums_multi_kernel(const int64_t *__restrict__ jobs, const int *__restrict__ blocks) {
    __shared__ float red[16][64];
    const int e = blocks[blockIdx.x];
    const int64_t *jb = jobs + 6 * (e & 0xffff);
    const float *part = reinterpret_cast<const float *>(jb[0]);
    const int nblk = (int)(jb[4] & 0xffffffff), C = (int)(jb[4] >> 32), nparts = (int)jb[5];
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int i = (e >> 16) * 64 + lane;
    const int W = nparts * C;
    float s0 = 0.f, s1 = 0.f;
    if (i < W) {
        int j = slot;
        for (; j + 16 < nblk; j += 32) {
            s0 += part[(long)j * W + i];
            s1 += part[(long)(j + 16) * W + i];
        }
        if (j < nblk) s0 += part[(long)j * W + i];
    }
    red[slot][lane] = s0 + s1;
    __syncthreads();
    if (slot == 0 && i < W) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][lane];
        const int k = i / C;
        float *out = reinterpret_cast<float *>(jb[1 + k]);
        if (out) out[i - k * C] = s;
    }
}

__global__ __launch_bounds__(1024) void colsum_finish_kernel(const float *part, float *out, int nblk, int C) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        int j = slot;
        for (; j + 16 < nblk; j += 32) {
            s0 += part[(long)j * C + c];
            s1 += part[(long)(j + 16) * C + c];
        }
        if (j < nblk) s0 += part[(long)j * C + c];
    }
    red[slot][lane] = s0 + s1;
    __syncthreads();
    if (slot == 0 && c < C) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red[k][lane];
        out[c] = s;
    }
}

struct TokShape {
    int NT, R, threads, nblk;
};

static bool tok_shape(long rows, int C, int vec, bool reduce, TokShape &s) {
    if (rows <= 0 || C <= 0 || C % vec != 0) return false;
    s.NT = C / vec;
    if (s.NT > 1024) return false;
    s.R = 256 / s.NT;
    if (s.R < 1) s.R = 1;
    s.threads = s.NT * s.R;
    if (reduce && (long)s.R * C * 4 > 64 * 1024) return false;
    long nb = (rows + s.R - 1) / s.R;
    const long cap = reduce ? 512 : 4096;
    s.nblk = (int)(nb > cap ? cap : nb);
    return true;
}

template <typename T, int MODE> static int tok_launch(const TokArgs &a0, hipStream_t s) {
    TokShape sh;
    if (!tok_shape(a0.rows, a0.C, Pack<T>::N, MODE != 0, sh)) return XFM_ELIMIT;
    TokArgs a = a0;
    a.NT = sh.NT;
    a.R = sh.R;
    const size_t lds = MODE != 0 ? (size_t)sh.R * a.C * sizeof(float) : 0;
    hipLaunchKernelGGL((tokens_kernel<T, MODE>), dim3(sh.nblk), dim3(sh.threads), lds, s, a);
    return check_launch();
}

// ---- tokens <-> planes for short maps (7 x 7: 49 positions): (B, R, C) <-> (B, C, R) with the whole position axis R <= 64 in a
// tile of 64 channels.  The plane-major side of such a tile is ONE contiguous run of 64 R elements (rows of 98 bytes follow each
// other), so both sides move as 16-byte vectors; the tile turns in LDS (rows of 33 dwords: the 2-byte accesses of the flat run
// walk r fastest, 33 r + c / 2 spreads them over the banks).  The framework's strided copy runs 15-30 us on these 5-14 MB
// tensors; this is a plain HBM stream.
template <bool TOK2PL, bool ACC = false>
__global__ __launch_bounds__(256) void transpose_short_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, int R,
                                                              int C) {
    __shared__ uint32_t tile[64 * 33];
    uint16_t *t16 = reinterpret_cast<uint16_t *>(tile);                 // element (r, c) at r * 66 + c
    const int b = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x;
    const int n8 = 8 * R;                                               // 16-byte vectors of the tile on either side
    const uint16_t *pl_in = src + ((int64_t)b * C + c0) * R;
    uint16_t *pl_out = dst + ((int64_t)b * C + c0) * R;
    if (TOK2PL) {
        for (int i = tid; i < n8; i += 256) {
            const int r = i >> 3, j = i & 7;
            const uint4 v = *reinterpret_cast<const uint4 *>(src + ((int64_t)b * R + r) * C + c0 + j * 8);
            uint32_t *p = tile + r * 33 + j * 4;
            p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
        }
        __syncthreads();
        for (int i = tid; i < n8; i += 256) {
            const int f = i * 8;
            int c = f / R, r = f - c * R;
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t lo = t16[r * 66 + c];
                if (++r == R) { r = 0; ++c; }
                const uint32_t hi = t16[r * 66 + c];
                if (++r == R) { r = 0; ++c; }
                w[k] = lo | (hi << 16);
            }
            if (ACC) {                                   // dst += (bf16 pairs widened, added in fp32, rounded once)
                const uint4 o = *reinterpret_cast<const uint4 *>(pl_out + f);
                const uint32_t ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    w[k] = pack_bf16x2(__uint_as_float(w[k] << 16) + __uint_as_float(ov[k] << 16),
                                       __uint_as_float(w[k] & 0xffff0000u) + __uint_as_float(ov[k] & 0xffff0000u));
            }
            *reinterpret_cast<uint4 *>(pl_out + f) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    } else {
        for (int i = tid; i < n8; i += 256) {
            const int f = i * 8;
            const uint4 v = *reinterpret_cast<const uint4 *>(pl_in + f);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            int c = f / R, r = f - c * R;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                t16[r * 66 + c] = (uint16_t)(w[k] & 0xffffu);
                if (++r == R) { r = 0; ++c; }
                t16[r * 66 + c] = (uint16_t)(w[k] >> 16);
                if (++r == R) { r = 0; ++c; }
            }
        }
        __syncthreads();
        for (int i = tid; i < n8; i += 256) {
            const int r = i >> 3, j = i & 7;
            const uint32_t *p = tile + r * 33 + j * 4;
            *reinterpret_cast<uint4 *>(dst + ((int64_t)b * R + r) * C + c0 + j * 8) = make_uint4(p[0], p[1], p[2], p[3]);
        }
    }
}

// ---- tokens -> planes with the squeeze pooling: planes (B, C, R) = t^T and pooled (B, C) = mean over the R positions (bf16, fp32
// sums) -- `self.avg_pool(xp)` of the shallow block read off the tile the transpose holds anyway; backward: d t[b, r, c] =
// d planes[b, c, r] + d pooled[b, c] / R (one fp32 add, one rounding).
__global__ __launch_bounds__(256) void pooled_transpose_fwd_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst,
                                                                   uint16_t *__restrict__ pooled, int R, int C) {
    __shared__ uint32_t tile[64 * 33];
    uint16_t *t16 = reinterpret_cast<uint16_t *>(tile);
    const int b = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x, n8 = 8 * R;
    uint16_t *pl_out = dst + ((int64_t)b * C + c0) * R;
    for (int i = tid; i < n8; i += 256) {
        const int r = i >> 3, j = i & 7;
        const uint4 v = *reinterpret_cast<const uint4 *>(src + ((int64_t)b * R + r) * C + c0 + j * 8);
        uint32_t *p = tile + r * 33 + j * 4;
        p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    }
    __syncthreads();
    if (tid < 64) {
        float a = 0.f;
        for (int r = 0; r < R; ++r) a += __uint_as_float((uint32_t)t16[r * 66 + tid] << 16);
        pooled[(int64_t)b * C + c0 + tid] = (uint16_t)(pack_bf16x2(a / (float)R, 0.f) & 0xffffu);
    }
    for (int i = tid; i < n8; i += 256) {
        const int f = i * 8;
        int c = f / R, r = f - c * R;
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t lo = t16[r * 66 + c];
            if (++r == R) { r = 0; ++c; }
            const uint32_t hi = t16[r * 66 + c];
            if (++r == R) { r = 0; ++c; }
            w[k] = lo | (hi << 16);
        }
        *reinterpret_cast<uint4 *>(pl_out + f) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

__global__ __launch_bounds__(256) void pooled_transpose_bwd_kernel(const uint16_t *__restrict__ dpl, const uint16_t *__restrict__ dpool,
                                                                   uint16_t *__restrict__ dt, int R, int C) {
    __shared__ uint32_t tile[64 * 33];
    uint16_t *t16 = reinterpret_cast<uint16_t *>(tile);
    const int b = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x, n8 = 8 * R;
    const uint16_t *pl_in = dpl + ((int64_t)b * C + c0) * R;
    for (int i = tid; i < n8; i += 256) {
        const int f = i * 8;
        const uint4 v = *reinterpret_cast<const uint4 *>(pl_in + f);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        int c = f / R, r = f - c * R;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            t16[r * 66 + c] = (uint16_t)(w[k] & 0xffffu);
            if (++r == R) { r = 0; ++c; }
            t16[r * 66 + c] = (uint16_t)(w[k] >> 16);
            if (++r == R) { r = 0; ++c; }
        }
    }
    __syncthreads();
    const float invR = 1.f / (float)R;
    for (int i = tid; i < n8; i += 256) {
        const int r = i >> 3, j = i & 7;
        const uint32_t *p = tile + r * 33 + j * 4;
        const uint4 gq = *reinterpret_cast<const uint4 *>(dpool + (int64_t)b * C + c0 + j * 8);
        const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = pack_bf16x2(fmaf(__uint_as_float(gw[k] << 16), invR, __uint_as_float(p[k] << 16)),
                               fmaf(__uint_as_float(gw[k] & 0xffff0000u), invR, __uint_as_float(p[k] & 0xffff0000u)));
        *reinterpret_cast<uint4 *>(dt + ((int64_t)b * R + r) * C + c0 + j * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ---- gate (.) map with the layout change: out (B, R, C) tokens = yy (B, C, R) planes * gate (B, C), bf16 (the product rounded
// once, as the framework's bf16 multiply), and its backward: d yy = g^T * gate (planes), d gate[b, c] = sum_r g[b, r, c] yy[b, c, r]
// (fp32 sums).  Same tiling as transpose_short_kernel: a workgroup holds all R positions of 64 channels of one sample, so the
// per-channel sum over the positions is local (LDS float adds, at most two per thread).
__global__ __launch_bounds__(256) void gated_transpose_fwd_kernel(const uint16_t *__restrict__ yy, const uint16_t *__restrict__ gate,
                                                                  uint16_t *__restrict__ out, int R, int C) {
    __shared__ uint32_t tile[64 * 33];
    uint16_t *t16 = reinterpret_cast<uint16_t *>(tile);
    const int b = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x, n8 = 8 * R;
    const uint16_t *pl_in = yy + ((int64_t)b * C + c0) * R;
    for (int i = tid; i < n8; i += 256) {
        const int f = i * 8;
        const uint4 v = *reinterpret_cast<const uint4 *>(pl_in + f);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        int c = f / R, r = f - c * R;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            t16[r * 66 + c] = (uint16_t)(w[k] & 0xffffu);
            if (++r == R) { r = 0; ++c; }
            t16[r * 66 + c] = (uint16_t)(w[k] >> 16);
            if (++r == R) { r = 0; ++c; }
        }
    }
    __syncthreads();
    for (int i = tid; i < n8; i += 256) {
        const int r = i >> 3, j = i & 7;
        const uint32_t *p = tile + r * 33 + j * 4;
        const uint4 gq = *reinterpret_cast<const uint4 *>(gate + (int64_t)b * C + c0 + j * 8);
        const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            o[k] = pack_bf16x2(__uint_as_float(p[k] << 16) * __uint_as_float(gw[k] << 16),
                               __uint_as_float(p[k] & 0xffff0000u) * __uint_as_float(gw[k] & 0xffff0000u));
        *reinterpret_cast<uint4 *>(out + ((int64_t)b * R + r) * C + c0 + j * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

__global__ __launch_bounds__(256) void gated_transpose_bwd_kernel(const uint16_t *__restrict__ g, const uint16_t *__restrict__ yy,
                                                                  const uint16_t *__restrict__ gate, uint16_t *__restrict__ dyy,
                                                                  uint16_t *__restrict__ dgate, int R, int C) {
    __shared__ uint32_t tile[64 * 33];
    __shared__ float dg[64];
    uint16_t *t16 = reinterpret_cast<uint16_t *>(tile);
    const int b = blockIdx.y, c0 = blockIdx.x * 64, tid = threadIdx.x, n8 = 8 * R;
    if (tid < 64) dg[tid] = 0.f;
    for (int i = tid; i < n8; i += 256) {
        const int r = i >> 3, j = i & 7;
        const uint4 v = *reinterpret_cast<const uint4 *>(g + ((int64_t)b * R + r) * C + c0 + j * 8);
        uint32_t *p = tile + r * 33 + j * 4;
        p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
    }
    __syncthreads();
    const int64_t po = ((int64_t)b * C + c0) * R;
    for (int i = tid; i < n8; i += 256) {
        const int f = i * 8;
        const uint4 yq = *reinterpret_cast<const uint4 *>(yy + po + f);
        const uint32_t yw[4] = {yq.x, yq.y, yq.z, yq.w};
        int c = f / R, r = f - c * R;
        const int cfirst = c;
        float acc0 = 0.f, acc1 = 0.f;                   // this thread's share of d gate for channel cfirst / cfirst + 1
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float e[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float gv = __uint_as_float((uint32_t)t16[r * 66 + c] << 16);
                const float yv = h == 0 ? __uint_as_float(yw[k] << 16) : __uint_as_float(yw[k] & 0xffff0000u);
                const float gt = __uint_as_float((uint32_t)gate[(int64_t)b * C + c0 + c] << 16);
                e[h] = gv * gt;
                if (c == cfirst) acc0 = fmaf(gv, yv, acc0);
                else acc1 = fmaf(gv, yv, acc1);
                if (++r == R) { r = 0; ++c; }
            }
            o[k] = pack_bf16x2(e[0], e[1]);
        }
        *reinterpret_cast<uint4 *>(dyy + po + f) = make_uint4(o[0], o[1], o[2], o[3]);
        atomicAdd(&dg[cfirst], acc0);
        if (cfirst + 1 < 64 && acc1 != 0.f) atomicAdd(&dg[cfirst + 1], acc1);
    }
    __syncthreads();
    if (tid < 64) dgate[(int64_t)b * C + c0 + tid] = (uint16_t)(pack_bf16x2(dg[tid], 0.f) & 0xffffu);
}

// ---- the deep fusion block's input assembly: [view 1 | view 2 | (view 1 + view 2) / 2] (reference models/fusion_vmamba.py:
// Cross_SS2Dv5.forward: x_fuse = (x + x2) / 2, one in_proj_sec over the three streams) written in the GEMM's dtype by one
// kernel, and its gradient (d view k = g_k + g_fuse / 2) by another -- instead of mean + cat + cast and their backward chain.
template <typename Ty>
__global__ void __launch_bounds__(256) views_avg_stack_fwd_kernel(const float *__restrict__ n, Ty *__restrict__ out, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float a[4], b[4], m[4];
        Pack<float>::ld(n + i * 4, a);
        Pack<float>::ld(n + (nvec + i) * 4, b);
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = (a[k] + b[k]) * 0.5f;
        if constexpr (sizeof(Ty) == 4) {
            float *o = reinterpret_cast<float *>(out);
            Pack<float>::st(o + i * 4, a);
            Pack<float>::st(o + (nvec + i) * 4, b);
            Pack<float>::st(o + (2 * nvec + i) * 4, m);
        } else {
            uint16_t *o = reinterpret_cast<uint16_t *>(out);
            *reinterpret_cast<uint2 *>(o + i * 4) = make_uint2(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]));
            *reinterpret_cast<uint2 *>(o + (nvec + i) * 4) = make_uint2(pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3]));
            *reinterpret_cast<uint2 *>(o + (2 * nvec + i) * 4) = make_uint2(pack_bf16x2(m[0], m[1]), pack_bf16x2(m[2], m[3]));
        }
    }
}

template <typename Ty>
__global__ void __launch_bounds__(256) views_avg_stack_bwd_kernel(const Ty *__restrict__ g, float *__restrict__ dn, int64_t nvec) {
    auto ld = [&](int64_t e, float (&v)[4]) {
        if constexpr (sizeof(Ty) == 4) {
            Pack<float>::ld(reinterpret_cast<const float *>(g) + e * 4, v);
        } else {
            const uint2 w = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(g) + e * 4);
            v[0] = __uint_as_float(w.x << 16); v[1] = __uint_as_float(w.x & 0xffff0000u);
            v[2] = __uint_as_float(w.y << 16); v[3] = __uint_as_float(w.y & 0xffff0000u);
        }
    };
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float a[4], b[4], m[4];
        ld(i, a);
        ld(nvec + i, b);
        ld(2 * nvec + i, m);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[k] = fmaf(m[k], 0.5f, a[k]);
            b[k] = fmaf(m[k], 0.5f, b[k]);
        }
        Pack<float>::st(dn + i * 4, a);
        Pack<float>::st(dn + (nvec + i) * 4, b);
    }
}

}  // namespace xfm

extern "C" {

int xfm_colsum_blocks(long long rows, int C, int dtype) {
    xfm::TokShape sh;
    if (!xfm::tok_shape(rows, C, dtype == XFM_F32 ? 4 : 8, true, sh)) return 0;
    return sh.nblk;
}

int xfm_bias_gelu_fwd(const void *z, const float *bias, void *g, long long rows, int C, int dtype, void *stream) {
    using namespace xfm;
    if (!z || !g || rows <= 0 || C <= 0) return XFM_EINVAL;
    TokArgs a{};
    a.z = z; a.bias = bias; a.out = g; a.rows = rows; a.C = C;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == XFM_F32) return tok_launch<float, 0>(a, s);
    if (dtype == XFM_BF16) return tok_launch<bf16_t, 0>(a, s);
    return XFM_EDTYPE;
}

int xfm_bias_gelu_bwd(const void *z, const float *bias, const void *dg, void *dz, float *dbias, float *workspace,
                      long long rows, int C, int dtype, void *stream) {
    using namespace xfm;
    if (!z || !dg || !dz || !workspace || rows <= 0 || C <= 0) return XFM_EINVAL;
    TokArgs a{};
    a.z = z; a.bias = bias; a.dg = dg; a.out = dz; a.part = workspace; a.rows = rows; a.C = C;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (dtype == XFM_F32) rc = tok_launch<float, 1>(a, s);
    else if (dtype == XFM_BF16) rc = tok_launch<bf16_t, 1>(a, s);
    else return XFM_EDTYPE;
    if (rc != XFM_OK) return rc;
    if (!dbias) return XFM_OK;                       // the caller folds the partial rows itself (xfm_partial_sums_multi)
    const int nblk = xfm_colsum_blocks(rows, C, dtype);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 63) / 64), dim3(1024), 0, s, workspace, dbias, nblk, C);
    return check_launch();
}

int xfm_colsum(const void *x, float *out, float *workspace, long long rows, int C, int dtype, void *stream) {
    using namespace xfm;
    if (!x || !workspace || rows <= 0 || C <= 0) return XFM_EINVAL;
    TokArgs a{};
    a.z = x; a.part = workspace; a.rows = rows; a.C = C;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (dtype == XFM_F32) rc = tok_launch<float, 2>(a, s);
    else if (dtype == XFM_BF16) rc = tok_launch<bf16_t, 2>(a, s);
    else return XFM_EDTYPE;
    if (rc != XFM_OK) return rc;
    if (!out) return XFM_OK;                         // partial rows only (xfm_partial_sums_multi folds them later)
    const int nblk = xfm_colsum_blocks(rows, C, dtype);
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 63) / 64), dim3(1024), 0, s, workspace, out, nblk, C);
    return check_launch();
}

int xfm_residual_settle_fwd(const float *x, const void *y, const float *scale, const float *y_bias, void *out, int B,
                            int rows_per_sample, int C, int y_dtype, int out_dtype, void *stream) {
    using namespace xfm;
    if (!x || !y || !out || B <= 0 || rows_per_sample <= 0 || C <= 0) return XFM_EINVAL;
    if (C % 8) return XFM_ELIMIT;
    const long long nvec = (long long)B * rows_per_sample * (C / 8);
    const unsigned grid = (unsigned)std::min<long long>((nvec + 255) / 256, 256 * 16);
    hipStream_t s = (hipStream_t)stream;
#define XFM_SETTLE_F(TY, TO)                                                                                        \
    hipLaunchKernelGGL((settle_fwd_kernel<TY, TO>), dim3(grid), dim3(256), 0, s, x, (const TY *)y, scale, y_bias, \
                       (TO *)out, nvec, C / 8, rows_per_sample)
    if (y_dtype == XFM_BF16 && out_dtype == XFM_BF16) XFM_SETTLE_F(bf16_t, bf16_t);
    else if (y_dtype == XFM_BF16 && out_dtype == XFM_F32) XFM_SETTLE_F(bf16_t, float);
    else if (y_dtype == XFM_F32 && out_dtype == XFM_F32) XFM_SETTLE_F(float, float);
    else if (y_dtype == XFM_F32 && out_dtype == XFM_BF16) XFM_SETTLE_F(float, bf16_t);
    else return XFM_EDTYPE;
#undef XFM_SETTLE_F
    return check_launch();
}

int xfm_residual_settle_bwd(const void *dout, const float *scale, float *dx, void *dy, int B, int rows_per_sample, int C,
                            int y_dtype, int out_dtype, void *stream) {
    using namespace xfm;
    if (!dout || !dx || !dy || B <= 0 || rows_per_sample <= 0 || C <= 0) return XFM_EINVAL;
    if (C % 8) return XFM_ELIMIT;
    const long long nvec = (long long)B * rows_per_sample * (C / 8);
    const unsigned grid = (unsigned)std::min<long long>((nvec + 255) / 256, 256 * 16);
    hipStream_t s = (hipStream_t)stream;
#define XFM_SETTLE_B(TY, TO)                                                                                        \
    hipLaunchKernelGGL((settle_bwd_kernel<TY, TO>), dim3(grid), dim3(256), 0, s, (const TO *)dout, scale, dx, (TY *)dy, \
                       nvec, C / 8, rows_per_sample)
    if (y_dtype == XFM_BF16 && out_dtype == XFM_BF16) XFM_SETTLE_B(bf16_t, bf16_t);
    else if (y_dtype == XFM_BF16 && out_dtype == XFM_F32) XFM_SETTLE_B(bf16_t, float);
    else if (y_dtype == XFM_F32 && out_dtype == XFM_F32) XFM_SETTLE_B(float, float);
    else if (y_dtype == XFM_F32 && out_dtype == XFM_BF16) XFM_SETTLE_B(float, bf16_t);
    else return XFM_EDTYPE;
#undef XFM_SETTLE_B
    return check_launch();
}

// Fold the per-workgroup partial rows of MANY column-sum producers in one launch.  Every producer of this library that
// ends in a small "finish" kernel (row-LayerNorm dw / db / d pre_bias, bias+GELU's bias gradient, xfm_colsum) can leave its
// partial rows in its workspace instead (pass a null result pointer); a training step has ~50 of them, each a 6-7 us
// kernel of a few workgroups whose results nobody reads before the optimizer.  jobs: device array of 6 int64 per job
// {part, out0, out1, out2 (device addresses, outs may be 0), nblk | C << 32, nparts}: out_k[c] = sum_j part[j * nparts * C +
// k * C + c].  blocks: device int32 array, one entry per workgroup: job | (64-column block << 16).
int xfm_partial_sums_multi(const void *jobs, const void *blocks, int nblocks, void *stream) {
    using namespace xfm;
    if (!jobs || !blocks || nblocks <= 0) return XFM_EINVAL;
    hipLaunchKernelGGL(partial_sums_multi_kernel, dim3((unsigned)nblocks), dim3(1024), 0, (hipStream_t)stream,
                       (const int64_t *)jobs, (const int *)blocks);
    return check_launch();
}

/* (B, R, C) tokens <-> (B, C, R) planes of 2-byte elements for short maps: R <= 64 positions, C % 64 == 0, 16-byte aligned
 * tensors.  tokens_to_planes != 0: src is (B, R, C), dst (B, C, R); else the reverse.  Replaces the framework's
 * permute + contiguous around the 7 x 7 SS2D blocks (reference models/fusion_vmamba.py:594-601, 853-857: the NHWC <-> NCHW
 * permutes around conv2d). */
int xfm_transpose_short_supported(int R, int C) { return (R > 0 && R <= 64 && C > 0 && C % 64 == 0) ? 1 : 0; }

int xfm_transpose_short(const void *src, void *dst, int B, int R, int C, int tokens_to_planes, void *stream) {
    using namespace xfm;
    if (!src || !dst || B <= 0) return XFM_EINVAL;
    if (!xfm_transpose_short_supported(R, C)) return XFM_ELIMIT;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return XFM_EINVAL;
    const dim3 grid((unsigned)(C / 64), (unsigned)B), block(256);
    if (tokens_to_planes)
        hipLaunchKernelGGL(transpose_short_kernel<true>, grid, block, 0, (hipStream_t)stream, (const uint16_t *)src, (uint16_t *)dst, R, C);
    else
        hipLaunchKernelGGL(transpose_short_kernel<false>, grid, block, 0, (hipStream_t)stream, (const uint16_t *)src, (uint16_t *)dst, R, C);
    return check_launch();
}

/* dst (B, C, R) bf16 planes += src (B, R, C) bf16 tokens, transposed: the accumulating half of a gradient that arrives
 * token-major for a plane-major tensor (x_proj's data gradient at 7 x 7: dx += (d x_dbl . Wx)^T). */
int xfm_transpose_short_add_bf16(const void *src, void *dst, int B, int R, int C, void *stream) {
    using namespace xfm;
    if (!src || !dst || B <= 0) return XFM_EINVAL;
    if (!xfm_transpose_short_supported(R, C)) return XFM_ELIMIT;
    if (((uintptr_t)src | (uintptr_t)dst) & 15) return XFM_EINVAL;
    hipLaunchKernelGGL((transpose_short_kernel<true, true>), dim3((unsigned)(C / 64), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)src, (uint16_t *)dst, R, C);
    return check_launch();
}

/* out (3, M) = [n[0] | n[1] | (n[0] + n[1]) / 2] for n (2, M) fp32, M % 4 == 0, in out_dtype (fp32 / bf16): the three streams of
 * Cross_SS2Dv5 (reference models/fusion_vmamba.py: x_fuse = (x + x2) / 2 ahead of in_proj_sec); _bwd: dn (2, M) fp32 from
 * g (3, M): dn[k] = g[k] + g[2] / 2. */
int xfm_views_avg_stack_fwd(const float *n, void *out, long long M, int out_dtype, void *stream) {
    using namespace xfm;
    if (!n || !out || M <= 0 || M % 4) return XFM_EINVAL;
    if (((uintptr_t)n | (uintptr_t)out) & 15) return XFM_EINVAL;
    const int64_t nvec = M / 4;
    const unsigned grid = (unsigned)std::min<int64_t>((nvec + 255) / 256, 256 * 8);
    if (out_dtype == XFM_F32)
        hipLaunchKernelGGL((views_avg_stack_fwd_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, n, (float *)out, nvec);
    else if (out_dtype == XFM_BF16)
        hipLaunchKernelGGL((views_avg_stack_fwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream, n, (bf16_t *)out, nvec);
    else return XFM_EDTYPE;
    return check_launch();
}

int xfm_views_avg_stack_bwd(const void *g, float *dn, long long M, int g_dtype, void *stream) {
    using namespace xfm;
    if (!g || !dn || M <= 0 || M % 4) return XFM_EINVAL;
    if (((uintptr_t)g | (uintptr_t)dn) & 15) return XFM_EINVAL;
    const int64_t nvec = M / 4;
    const unsigned grid = (unsigned)std::min<int64_t>((nvec + 255) / 256, 256 * 8);
    if (g_dtype == XFM_F32)
        hipLaunchKernelGGL((views_avg_stack_bwd_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float *)g, dn, nvec);
    else if (g_dtype == XFM_BF16)
        hipLaunchKernelGGL((views_avg_stack_bwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t *)g, dn, nvec);
    else return XFM_EDTYPE;
    return check_launch();
}

/* out (B, R, C) tokens = yy (B, C, R) planes * gate (B, C), bf16: `y * gate` of ShallowFuse_SS2Dv4.forward (reference
 * models/fusion_vmamba.py:870-871) with the permute for out_proj folded in; _bwd: d yy (B, C, R) = g^T * gate and
 * d gate (B, C) = sum_r g[b, r, c] * yy[b, c, r] from g (B, R, C).  Shapes of xfm_transpose_short_supported(R, C). */
int xfm_gated_transpose_fwd(const void *yy, const void *gate, void *out, int B, int R, int C, void *stream) {
    using namespace xfm;
    if (!yy || !gate || !out || B <= 0) return XFM_EINVAL;
    if (!xfm_transpose_short_supported(R, C) || R < 8) return XFM_ELIMIT;   // (R >= 8: a thread's 8 elements span <= 2 channels)
    if (((uintptr_t)yy | (uintptr_t)gate | (uintptr_t)out) & 15) return XFM_EINVAL;
    hipLaunchKernelGGL(gated_transpose_fwd_kernel, dim3((unsigned)(C / 64), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)yy, (const uint16_t *)gate, (uint16_t *)out, R, C);
    return check_launch();
}

int xfm_gated_transpose_bwd(const void *g, const void *yy, const void *gate, void *dyy, void *dgate, int B, int R, int C,
                            void *stream) {
    using namespace xfm;
    if (!g || !yy || !gate || !dyy || !dgate || B <= 0) return XFM_EINVAL;
    if (!xfm_transpose_short_supported(R, C) || R < 8) return XFM_ELIMIT;
    if (((uintptr_t)g | (uintptr_t)yy | (uintptr_t)dyy) & 15) return XFM_EINVAL;
    hipLaunchKernelGGL(gated_transpose_bwd_kernel, dim3((unsigned)(C / 64), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)g, (const uint16_t *)yy, (const uint16_t *)gate, (uint16_t *)dyy, (uint16_t *)dgate, R, C);
    return check_launch();
}

/* planes (B, C, R) = t (B, R, C)^T together with pooled (B, C) = mean_r t[b, r, c], bf16: the tokens -> planes move ahead of
 * conv2d and `self.avg_pool(xp)` of ShallowFuse_SS2Dv4.forward (reference models/fusion_vmamba.py:853-871) in one kernel;
 * _bwd: d t (B, R, C) = d planes^T + d pooled / R.  Shapes of xfm_transpose_short_supported(R, C). */
int xfm_pooled_transpose_fwd(const void *t, void *planes, void *pooled, int B, int R, int C, void *stream) {
    using namespace xfm;
    if (!t || !planes || !pooled || B <= 0) return XFM_EINVAL;
    if (!xfm_transpose_short_supported(R, C)) return XFM_ELIMIT;
    if (((uintptr_t)t | (uintptr_t)planes) & 15) return XFM_EINVAL;
    hipLaunchKernelGGL(pooled_transpose_fwd_kernel, dim3((unsigned)(C / 64), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)t, (uint16_t *)planes, (uint16_t *)pooled, R, C);
    return check_launch();
}

int xfm_pooled_transpose_bwd(const void *dplanes, const void *dpooled, void *dt, int B, int R, int C, void *stream) {
    using namespace xfm;
    if (!dplanes || !dpooled || !dt || B <= 0) return XFM_EINVAL;
    if (!xfm_transpose_short_supported(R, C)) return XFM_ELIMIT;
    if (((uintptr_t)dplanes | (uintptr_t)dpooled | (uintptr_t)dt) & 15) return XFM_EINVAL;
    hipLaunchKernelGGL(pooled_transpose_bwd_kernel, dim3((unsigned)(C / 64), (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)dplanes, (const uint16_t *)dpooled, (uint16_t *)dt, R, C);
    return check_launch();
}

}  // extern "C"
