// Residual add + DropPath scale + LayerNorm over the channels of a TOKEN-MAJOR (B, H*W, C) stream, one kernel.
//
//     x_new = x + scale[b] * y          (the `x = x + self.drop_path(branch)` of VSSBlock._forward,
//     h     = LayerNorm_C(x_new) * w + b       models/fusion_vmamba.py:1325-1337, fused with the norm that follows)
//
// The residual stream x is fp32; y (branch output), h (input of the next GEMM) and their gradients are in the GEMM
// dtype.  A row of C = 4*G*NV channels is owned by G lanes holding NV float4 each, so a wavefront works on 64/G rows
// with fully coalesced 16-byte accesses and the row statistics are reduced with cross-lane shuffles only.
// HBM-bound: fwd reads 4+2 B and writes 4+2 B per element, bwd reads 4+4+2 B and writes 4+2 B.
#include "xfm_common.hpp"

namespace xfm {

template <typename T> struct Vec4IO;
template <> struct Vec4IO<float> {
    static __device__ __forceinline__ float4 ld(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ void st(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
};
template <> struct Vec4IO<bf16_t> {
    static __device__ __forceinline__ float4 ld(const bf16_t *p) {
        const uint2 r = *reinterpret_cast<const uint2 *>(p);
        return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16),
                           __uint_as_float(r.y & 0xffff0000u));
    }
    static __device__ __forceinline__ void st(bf16_t *p, float4 v) {
        uint2 r;
        r.x = pack_bf16x2(v.x, v.y);
        r.y = pack_bf16x2(v.z, v.w);
        *reinterpret_cast<uint2 *>(p) = r;
    }
};

// Sum over the G lanes of a row group, returned in all of them -- on the vector ALU only: quad permutes, the two mirrors inside a
// row of 16 lanes, then gfx950's row / half swaps (v_permlane16_swap, v_permlane32_swap).  (As __shfl_xor every step is a
// ds_bpermute round trip through the LDS pipeline: 2 log2(G) dependent round trips per row in the forward pass, with one
// row group per wave in flight.)
template <int CTRL> __device__ __forceinline__ float rowln_dpp(const float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int G> __device__ __forceinline__ float group_sum(float v) {
    static_assert(G >= 4 && G <= 64, "row groups of 4 ... 64 lanes");
    v += rowln_dpp<0xB1>(v);                           // quad_perm [1, 0, 3, 2]
    v += rowln_dpp<0x4E>(v);                           // quad_perm [2, 3, 0, 1]
    if constexpr (G >= 8) v += rowln_dpp<0x141>(v);    // row_half_mirror: lane i <- lane 7 - i of its half row
    if constexpr (G >= 16) v += rowln_dpp<0x140>(v);   // row_mirror: lane i <- lane 15 - i of its row
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    if constexpr (G >= 32) {                           // rows 0 <-> 1, 2 <-> 3
        const uint32_t b = __float_as_uint(v);
        const u32x2_t r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    if constexpr (G >= 64) {                           // halves
        const uint32_t b = __float_as_uint(v);
        const u32x2_t r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return v;
}

struct RowLnArgs {
    const void *x;          // (rows, C) residual stream in (fp32; bf16 allowed when y is null)
    const void *y;          // (rows, C) branch output or null
    const float *scale;     // (B) per-sample DropPath factor or null
    const float *w, *b;     // (C); b may be null
    const float *pre_bias;  // (C) or null: y == null: added to x before the norm (a convolution's bias);
                            // y != null: added to y inside the scaled sum (the bias of the linear layer that made y)
    int nparts;             // 2 (dw, db) or 3 (+ d pre_bias) partial rows per workgroup
    float *x_new;           // (rows, C) residual stream out (null when y is null: x passes through)
    void *h;                // (rows, C) normalised output
    float *mean, *rstd;     // (rows)
    // backward
    const void *dh;         // (rows, C)
    const float *dres;      // (rows, C) gradient arriving on x_new from later consumers, or null
    void *dx;               // (rows, C) gradient of x (and of x_new), in x's dtype
    void *dy;               // (rows, C) gradient of y, or null
    float *part;            // (nblk, 2, C) per-workgroup partial dw / db
    int rows, rows_per_sample, C;
    float eps;
    int act;                // 1: h = gelu(LayerNorm(x)) (exact erf form) -- the patch embedding's norm -> GELU (reference
                            // models/fusion_vmamba.py:1504-1518); the backward pass recomputes LayerNorm(x) from x, mean, rstd,
                            // w and b (needs `b`) and multiplies dh by gelu' first
};

// erf(x) by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7) on the hardware exp2 / rcp, E = exp(-x^2): as csrc/tokens_ops.hip
__device__ __forceinline__ float rowln_erf(const float x, float &E) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    E = __builtin_amdgcn_exp2f(-(x * x) * 1.4426950408889634f);
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    const float r = fmaf(-(p * t), E, 1.0f);
    return copysignf(r, x);
}
__device__ __forceinline__ float rowln_gelu(const float x) {
    float E;
    return 0.5f * x * (1.0f + rowln_erf(x * 0.70710678118654752f, E));
}
// gelu'(x) = Phi(x) + x phi(x)
__device__ __forceinline__ float rowln_gelu_grad(const float x) {
    float E;
    const float cdf = 0.5f * (1.0f + rowln_erf(x * 0.70710678118654752f, E));
    return fmaf(x, 0.3989422804014327f * E, cdf);
}

template <typename Tx, typename Ty, int G, int NV> __global__ __launch_bounds__(256) void rowln_fwd_kernel(RowLnArgs a) {
    constexpr int RPW = 64 / G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % G, rsub = lane / G;
    const int C = a.C;
    float4 w[NV], bb[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = (k * G + sub) * 4;
        w[k] = Vec4IO<float>::ld(a.w + c);
        bb[k] = a.b ? Vec4IO<float>::ld(a.b + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const Ty *y = static_cast<const Ty *>(a.y);
    const Tx *x = static_cast<const Tx *>(a.x);
    Ty *h = static_cast<Ty *>(a.h);
    const float invC = 1.0f / (float)C;
    float4 pb[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k)
        pb[k] = a.pre_bias ? Vec4IO<float>::ld(a.pre_bias + (k * G + sub) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (long rg = (long)blockIdx.x * 4 + wave; rg * RPW < a.rows; rg += (long)gridDim.x * 4) {
        const long row = rg * RPW + rsub;
        const bool live = row < a.rows;
        const long r = live ? row : a.rows - 1;
        float4 v[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            v[k] = Vec4IO<Tx>::ld(x + r * C + (k * G + sub) * 4);
            if (!y) { v[k].x += pb[k].x; v[k].y += pb[k].y; v[k].z += pb[k].z; v[k].w += pb[k].w; }
        }
        if (y) {
            const float s = a.scale ? a.scale[r / a.rows_per_sample] : 1.0f;
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const float4 t = Vec4IO<Ty>::ld(y + r * C + (k * G + sub) * 4);
                v[k].x = fmaf(s, t.x + pb[k].x, v[k].x);
                v[k].y = fmaf(s, t.y + pb[k].y, v[k].y);
                v[k].z = fmaf(s, t.z + pb[k].z, v[k].z);
                v[k].w = fmaf(s, t.w + pb[k].w, v[k].w);
                if (live) Vec4IO<float>::st(a.x_new + r * C + (k * G + sub) * 4, v[k]);
            }
        }
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) s1 += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        const float mu = group_sum<G>(s1) * invC;
        float s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            v[k].x -= mu; v[k].y -= mu; v[k].z -= mu; v[k].w -= mu;
            s2 += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
        }
        const float rs = rsqrtf(group_sum<G>(s2) * invC + a.eps);
        if (live) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                float4 o;
                o.x = fmaf(v[k].x * rs, w[k].x, bb[k].x);
                o.y = fmaf(v[k].y * rs, w[k].y, bb[k].y);
                o.z = fmaf(v[k].z * rs, w[k].z, bb[k].z);
                o.w = fmaf(v[k].w * rs, w[k].w, bb[k].w);
                if (a.act) {
                    o.x = rowln_gelu(o.x); o.y = rowln_gelu(o.y); o.z = rowln_gelu(o.z); o.w = rowln_gelu(o.w);
                }
                Vec4IO<Ty>::st(h + r * C + (k * G + sub) * 4, o);
            }
            if (sub == 0) {
                a.mean[r] = mu;
                a.rstd[r] = rs;
            }
        }
    }
}

// dx = rstd * (g - mean_C(g) - xhat * mean_C(g * xhat)) + dres,  g = dh * w;  dy = scale[b] * dx.
// dw / db column sums: per-lane accumulators over the rows a workgroup walks, folded across the workgroup through
// LDS and written as one partial row per workgroup (summed by rowln_wb_kernel: deterministic, no atomics).
template <typename Tx, typename Ty, int G, int NV> __global__ __launch_bounds__(256) void rowln_bwd_kernel(RowLnArgs a) {
    constexpr int RPW = 64 / G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % G, rsub = lane / G;
    const int C = a.C;
    float4 w[NV], aw[NV], ab[NV], ap[NV], pb[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        w[k] = Vec4IO<float>::ld(a.w + (k * G + sub) * 4);
        // (with a residual add the saved x_new already contains the bias of y: nothing to re-add)
        pb[k] = (a.pre_bias && !a.dy) ? Vec4IO<float>::ld(a.pre_bias + (k * G + sub) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        aw[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        ap[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const Ty *dh = static_cast<const Ty *>(a.dh);
    Ty *dy = static_cast<Ty *>(a.dy);
    const Tx *xin = static_cast<const Tx *>(a.x);      // x_new saved by the forward
    Tx *dxo = static_cast<Tx *>(a.dx);
    const float invC = 1.0f / (float)C;
    for (long rg = (long)blockIdx.x * 4 + wave; rg * RPW < a.rows; rg += (long)gridDim.x * 4) {
        const long row = rg * RPW + rsub;
        const bool live = row < a.rows;
        const long r = live ? row : a.rows - 1;
        const float mu = a.mean[r], rs = a.rstd[r];
        float4 xh[NV], g[NV], e[NV];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k)                    // (the residual gradient is requested with the other operands)
            e[k] = a.dres ? Vec4IO<float>::ld(a.dres + r * C + (k * G + sub) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const long off = r * C + (k * G + sub) * 4;
            const float4 xv = Vec4IO<Tx>::ld(xin + off);
            float4 d = Vec4IO<Ty>::ld(dh + off);
            if (!live) d = make_float4(0.f, 0.f, 0.f, 0.f);
            xh[k] = make_float4((xv.x + pb[k].x - mu) * rs, (xv.y + pb[k].y - mu) * rs, (xv.z + pb[k].z - mu) * rs,
                                (xv.w + pb[k].w - mu) * rs);
            if (a.act) {                                 // dh arrives on gelu(LayerNorm(x)): through the activation first
                const float4 bv = a.b ? Vec4IO<float>::ld(a.b + (k * G + sub) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                d.x *= rowln_gelu_grad(fmaf(xh[k].x, w[k].x, bv.x));
                d.y *= rowln_gelu_grad(fmaf(xh[k].y, w[k].y, bv.y));
                d.z *= rowln_gelu_grad(fmaf(xh[k].z, w[k].z, bv.z));
                d.w *= rowln_gelu_grad(fmaf(xh[k].w, w[k].w, bv.w));
            }
            aw[k].x = fmaf(d.x, xh[k].x, aw[k].x); aw[k].y = fmaf(d.y, xh[k].y, aw[k].y);
            aw[k].z = fmaf(d.z, xh[k].z, aw[k].z); aw[k].w = fmaf(d.w, xh[k].w, aw[k].w);
            ab[k].x += d.x; ab[k].y += d.y; ab[k].z += d.z; ab[k].w += d.w;
            g[k] = make_float4(d.x * w[k].x, d.y * w[k].y, d.z * w[k].z, d.w * w[k].w);
            c1 += (g[k].x + g[k].y) + (g[k].z + g[k].w);
            c2 += (g[k].x * xh[k].x + g[k].y * xh[k].y) + (g[k].z * xh[k].z + g[k].w * xh[k].w);
        }
        c1 = group_sum<G>(c1) * invC;
        c2 = group_sum<G>(c2) * invC;
        const float s = (dy && a.scale) ? a.scale[r / a.rows_per_sample] : 1.0f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const long off = r * C + (k * G + sub) * 4;
            float4 o;
            o.x = rs * (g[k].x - c1 - xh[k].x * c2);
            o.y = rs * (g[k].y - c1 - xh[k].y * c2);
            o.z = rs * (g[k].z - c1 - xh[k].z * c2);
            o.w = rs * (g[k].w - c1 - xh[k].w * c2);
            o.x += e[k].x; o.y += e[k].y; o.z += e[k].z; o.w += e[k].w;
            if (live) {
                ap[k].x = fmaf(o.x, s, ap[k].x); ap[k].y = fmaf(o.y, s, ap[k].y);       // d pre_bias: sum of dx, or of
                ap[k].z = fmaf(o.z, s, ap[k].z); ap[k].w = fmaf(o.w, s, ap[k].w);       // dy = s * dx with a residual add
                Vec4IO<Tx>::st(dxo + off, o);
                if (dy) Vec4IO<Ty>::st(dy + off, make_float4(o.x * s, o.y * s, o.z * s, o.w * s));
            }
        }
    }
    // fold the RPW row slots of the wave (lanes with equal `sub`), then the 4 waves through LDS
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma unroll
        for (int m = G; m < 64; m <<= 1) {
            aw[k].x += __shfl_xor(aw[k].x, m, 64); aw[k].y += __shfl_xor(aw[k].y, m, 64);
            aw[k].z += __shfl_xor(aw[k].z, m, 64); aw[k].w += __shfl_xor(aw[k].w, m, 64);
            ab[k].x += __shfl_xor(ab[k].x, m, 64); ab[k].y += __shfl_xor(ab[k].y, m, 64);
            ab[k].z += __shfl_xor(ab[k].z, m, 64); ab[k].w += __shfl_xor(ab[k].w, m, 64);
            ap[k].x += __shfl_xor(ap[k].x, m, 64); ap[k].y += __shfl_xor(ap[k].y, m, 64);
            ap[k].z += __shfl_xor(ap[k].z, m, 64); ap[k].w += __shfl_xor(ap[k].w, m, 64);
        }
    }
    __shared__ float red[4][3][4 * G * NV];         // [wave][dw|db|dpre][channel], C = 4*G*NV
    if (rsub == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = (k * G + sub) * 4;
            *reinterpret_cast<float4 *>(&red[wave][0][c]) = aw[k];
            *reinterpret_cast<float4 *>(&red[wave][1][c]) = ab[k];
            *reinterpret_cast<float4 *>(&red[wave][2][c]) = ap[k];
        }
    }
    __syncthreads();
    float *part = a.part + (long)blockIdx.x * a.nparts * C;
    for (int i = threadIdx.x; i < a.nparts * C; i += 256) {
        const int which = i / C, c = i - which * C;
        part[i] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    }
}

// dw[c] = sum over workgroups of part[., 0, c]; db and d pre_bias likewise.  64 columns (of the nparts*C) x 16 row
// slots per workgroup.
__global__ __launch_bounds__(1024) void rowln_wb_kernel(const float *part, float *dw, float *db, float *dpre, int nblk,
                                                         int C, int nparts) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, slot = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
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
        if (i < C) dw[i] = s;
        else if (i < 2 * C) { if (db) db[i - C] = s; }
        else if (dpre) dpre[i - 2 * C] = s;
    }
}

static bool pick_shape(int C, int &G, int &NV) {
    for (int nv : {3, 4}) {
        for (int g : {4, 8, 16, 32, 64}) {
            if (4 * g * nv == C) {
                G = g;
                NV = nv;
                return true;
            }
        }
    }
    return false;
}

static int fwd_blocks(int rows, int G) {
    const long rgs = ((long)rows * G + 63) / 64;
    long nb = (rgs + 3) / 4;
    if (nb > 8192) nb = 8192;
    return (int)(nb < 1 ? 1 : nb);
}

template <typename Tx, typename Ty, int NV>
static int launch(bool bwd, int G, const RowLnArgs &a, int nblk, hipStream_t s) {
#define XFM_ROWLN_CASE(GG)                                                                                   \
    case GG:                                                                                                   \
        if (bwd) hipLaunchKernelGGL((rowln_bwd_kernel<Tx, Ty, GG, NV>), dim3(nblk), dim3(256), 0, s, a);        \
        else hipLaunchKernelGGL((rowln_fwd_kernel<Tx, Ty, GG, NV>), dim3(nblk), dim3(256), 0, s, a);            \
        break;
    switch (G) {
        XFM_ROWLN_CASE(4)
        XFM_ROWLN_CASE(8)
        XFM_ROWLN_CASE(16)
        XFM_ROWLN_CASE(32)
        XFM_ROWLN_CASE(64)
    default: return XFM_ELIMIT;
    }
#undef XFM_ROWLN_CASE
    return check_launch();
}

static int launch_any(bool bwd, int x_dtype, int dtype, int G, int NV, const RowLnArgs &a, int nblk, hipStream_t s) {
#define XFM_ROWLN_DT(TX, TY) (NV == 3 ? launch<TX, TY, 3>(bwd, G, a, nblk, s) : launch<TX, TY, 4>(bwd, G, a, nblk, s))
    if (x_dtype == XFM_F32 && dtype == XFM_F32) return XFM_ROWLN_DT(float, float);
    if (x_dtype == XFM_F32 && dtype == XFM_BF16) return XFM_ROWLN_DT(float, bf16_t);
    if (x_dtype == XFM_BF16 && dtype == XFM_F32) return XFM_ROWLN_DT(bf16_t, float);
    if (x_dtype == XFM_BF16 && dtype == XFM_BF16) return XFM_ROWLN_DT(bf16_t, bf16_t);
#undef XFM_ROWLN_DT
    return XFM_EDTYPE;
}

}  // namespace xfm

extern "C" {

int xfm_add_layernorm_rows_supported(int C) {
    int G, NV;
    return xfm::pick_shape(C, G, NV) ? 1 : 0;
}

int xfm_add_layernorm_rows_bwd_blocks(int rows, int C) {
    int G, NV;
    if (!xfm::pick_shape(C, G, NV) || rows <= 0) return 0;
    const int nb = xfm::fwd_blocks(rows, G);
    // resident workgroups: the kernel walks its rows with a grid stride and leaves one partial row set per workgroup.
    // Measured in the step (same box, samples/s): 512 -> 2111 / 2117, 768 -> 2115 / 2112, 1024 -> 2098 / 2098.
    static const int cap_env = getenv("XFM_ROWLN_BWD_BLOCKS") ? atoi(getenv("XFM_ROWLN_BWD_BLOCKS")) : 0;
    const int cap = cap_env > 0 ? cap_env : 512;
    return nb > cap ? cap : nb;
}

int xfm_add_layernorm_rows_fwd(const void *x, const void *y, const float *scale, const float *pre_bias,
                               const float *weight, const float *bias, float *x_new, void *h, float *mean, float *rstd, int B, int rows_per_sample, int C,
                               float eps, int x_dtype, int dtype, void *stream) {
    using namespace xfm;
    if (!x || !weight || !h || !mean || !rstd || B <= 0 || rows_per_sample <= 0 || C <= 0) return XFM_EINVAL;
    if (y && (!x_new || x_dtype != XFM_F32)) return XFM_EINVAL;
    int G, NV;
    if (!pick_shape(C, G, NV)) return XFM_ELIMIT;
    RowLnArgs a{};
    a.x = x; a.y = y; a.scale = scale; a.pre_bias = pre_bias; a.w = weight; a.b = bias; a.x_new = x_new; a.h = h; a.mean = mean; a.rstd = rstd;
    a.rows = B * rows_per_sample; a.rows_per_sample = rows_per_sample; a.C = C; a.eps = eps;
    const int nblk = fwd_blocks(a.rows, G);
    return launch_any(false, x_dtype, dtype, G, NV, a, nblk, (hipStream_t)stream);
}

int xfm_add_layernorm_rows_bwd(const void *x_new, const float *pre_bias, const float *weight, const void *dh,
                               const float *dres, const float *mean, const float *rstd, const float *scale, void *dx,
                               void *dy, float *dweight, float *dbias, float *dpre_bias, float *workspace, int B, int rows_per_sample, int C,
                               int x_dtype, int dtype, void *stream) {
    using namespace xfm;
    if (!x_new || !weight || !dh || !mean || !rstd || !dx || !workspace || B <= 0 || rows_per_sample <= 0)
        return XFM_EINVAL;
    int G, NV;
    if (!pick_shape(C, G, NV)) return XFM_ELIMIT;
    RowLnArgs a{};
    // (dweight == null: the partial rows stay in the workspace -- ALWAYS three parts wide then -- for xfm_partial_sums_multi)
    a.x = x_new; a.pre_bias = pre_bias; a.nparts = (dpre_bias || (!dweight && pre_bias)) ? 3 : 2; a.w = weight; a.dh = dh; a.dres = dres; a.mean = const_cast<float *>(mean); a.rstd = const_cast<float *>(rstd); a.scale = scale; a.dx = dx;
    a.dy = dy; a.part = workspace;
    a.rows = B * rows_per_sample; a.rows_per_sample = rows_per_sample; a.C = C;
    const int nblk = xfm_add_layernorm_rows_bwd_blocks(a.rows, C);
    hipStream_t s = (hipStream_t)stream;
    if ((dy || dres) && x_dtype != XFM_F32) return XFM_EINVAL;
    const int rc = launch_any(true, x_dtype, dtype, G, NV, a, nblk, s);
    if (rc != XFM_OK) return rc;
    if (!dweight) return XFM_OK;
    hipLaunchKernelGGL(rowln_wb_kernel, dim3((a.nparts * C + 63) / 64), dim3(1024), 0, s, workspace, dweight, dbias,
                       dpre_bias, nblk, C, a.nparts);
    return check_launch();
}

/* h = gelu(LayerNorm(x + pre_bias)) and its backward pass on (rows, C) rows: xfm_add_layernorm_rows_fwd / _bwd without a residual
 * branch and with the exact-erf GELU inside (reference models/fusion_vmamba.py:1504-1518: norm -> GELU of the patch embedding).
 * The backward pass takes the LayerNorm bias as well (it recomputes the pre-activation); everything else as the plain entries. */
int xfm_layernorm_rows_gelu_fwd(const void *x, const float *pre_bias, const float *weight, const float *bias, void *h, float *mean,
                                float *rstd, int rows, int C, float eps, int x_dtype, int dtype, void *stream) {
    using namespace xfm;
    if (!x || !weight || !h || !mean || !rstd || rows <= 0 || C <= 0) return XFM_EINVAL;
    int G, NV;
    if (!pick_shape(C, G, NV)) return XFM_ELIMIT;
    RowLnArgs a{};
    a.x = x; a.pre_bias = pre_bias; a.w = weight; a.b = bias; a.h = h; a.mean = mean; a.rstd = rstd;
    a.rows = rows; a.rows_per_sample = rows; a.C = C; a.eps = eps; a.act = 1;
    return launch_any(false, x_dtype, dtype, G, NV, a, fwd_blocks(a.rows, G), (hipStream_t)stream);
}

int xfm_layernorm_rows_gelu_bwd(const void *x, const float *pre_bias, const float *weight, const float *bias, const void *dh,
                                const float *mean, const float *rstd, void *dx, float *dweight, float *dbias, float *dpre_bias,
                                float *workspace, int rows, int C, int x_dtype, int dtype, void *stream) {
    using namespace xfm;
    if (!x || !weight || !dh || !mean || !rstd || !dx || !workspace || rows <= 0) return XFM_EINVAL;
    int G, NV;
    if (!pick_shape(C, G, NV)) return XFM_ELIMIT;
    RowLnArgs a{};
    a.x = x; a.pre_bias = pre_bias; a.nparts = (dpre_bias || (!dweight && pre_bias)) ? 3 : 2; a.w = weight; a.b = bias; a.dh = dh;
    a.mean = const_cast<float *>(mean); a.rstd = const_cast<float *>(rstd); a.dx = dx; a.part = workspace;
    a.rows = rows; a.rows_per_sample = rows; a.C = C; a.act = 1;
    const int nblk = xfm_add_layernorm_rows_bwd_blocks(a.rows, C);
    hipStream_t s = (hipStream_t)stream;
    const int rc = launch_any(true, x_dtype, dtype, G, NV, a, nblk, s);
    if (rc != XFM_OK) return rc;
    if (!dweight) return XFM_OK;
    hipLaunchKernelGGL(rowln_wb_kernel, dim3((a.nparts * C + 63) / 64), dim3(1024), 0, s, workspace, dweight, dbias, dpre_bias, nblk,
                       C, a.nparts);
    return check_launch();
}

}  // extern "C"
