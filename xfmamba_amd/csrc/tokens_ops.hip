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

}  // extern "C"
