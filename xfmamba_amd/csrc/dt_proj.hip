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

// MFMA variant (bf16, D % 32 == 0): per (b, k) the product is a (D x R) . (R x L) GEMM with a tiny contraction length; a
// wavefront owns 32 positions, keeps the xr fragment(s) of `v_mfma_f32_32x32x16_bf16` in registers for the whole
// channel loop and issues ONE MFMA (R <= 16) or two (R <= 32) per 32 x 32 output tile, so the VALU only runs the
// bias + softplus epilogue.  Operand maps (cdna_hip_programming.md): lane l (r = l & 31, h = l >> 5) holds
// A[row r][k = 8h + j], B[k = 8h + j][col r]; D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h.
typedef __bf16 xfm_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float xfm_f32x16_t __attribute__((ext_vector_type(16)));

struct DtProjMfmaArgs {
    const bf16_t *xr;     // (B, 4, R, L)
    const bf16_t *w;      // (4, D, R) bf16 (padded to RP = 16 * KS with zeros while it is staged in LDS)
    const float *bias;    // (4*D) or null
    bf16_t *out;          // (B, 4, D, L)
    int D, R, L, RP, ltiles, dsplit, wide;
};

template <int KS> __global__ __launch_bounds__(256) void dt_proj_mfma_kernel(DtProjMfmaArgs a) {
    constexpr int RP = 16 * KS, P = RP + 8;                       // LDS row pitch: 16-byte aligned, off the bank period
    extern __shared__ uint16_t wl[];                               // [channels of this workgroup][P] zero-padded weights
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lt = (blockIdx.x % a.ltiles) * 4 + wave;
    const int ds = (blockIdx.x / a.ltiles) % a.dsplit;
    const int bk = blockIdx.x / (a.ltiles * a.dsplit);
    const int k = bk & 3;
    const int dper = (a.D / 32 + a.dsplit - 1) / a.dsplit;
    const int t0 = ds * dper, t1 = min(a.D / 32, t0 + dper);
    {
        const uint16_t *wg = reinterpret_cast<const uint16_t *>(a.w) + ((int64_t)k * a.D + t0 * 32) * a.R;
        const int nd = (t1 - t0) * 32;
        for (int e = threadIdx.x; e < nd * RP; e += 256) {
            const int dd = e / RP, r = e - dd * RP;
            wl[dd * P + r] = r < a.R ? wg[dd * a.R + r] : (uint16_t)0;
        }
    }
    __syncthreads();
    const int c = lane & 31, h = lane >> 5;
    const int pos = lt * 32 + c;
    const bool valid = pos < a.L;
    if (lt * 32 >= a.L) return;                                   // (whole wave; no barrier below)
    const uint16_t *xr = reinterpret_cast<const uint16_t *>(a.xr) + (int64_t)bk * a.R * a.L;
    xfm_bf16x8_t bf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        uint16_t t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 16 * s + 8 * h + j;
            t[j] = (valid && r < a.R) ? xr[(int64_t)r * a.L + pos] : (uint16_t)0;
        }
        bf[s] = *reinterpret_cast<const xfm_bf16x8_t *>(t);
    }
    uint16_t *out = reinterpret_cast<uint16_t *>(a.out) + (int64_t)bk * a.D * a.L;
    if (a.wide) {
        // Rows are whole 8-position runs: form the product TRANSPOSED (D[pos][d]: the lane owns channel d0 + c and four
        // consecutive positions per register group), pack to bf16 and let two v_permlane32_swap per pair of groups give
        // every lane 8 consecutive positions -- 16-byte stores instead of sixteen 2-byte ones per 32 x 32 block.
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        const int pa = lt * 32 + 8 * h, pb = pa + 16;               // the lane's two runs
        for (int dt = t0; dt < t1; ++dt) {
            const int d0 = dt * 32;
            xfm_f32x16_t acc;
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const xfm_bf16x8_t af = *reinterpret_cast<const xfm_bf16x8_t *>(wl + ((dt - t0) * 32 + c) * P + 16 * s + 8 * h);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[s], af, acc, 0, 0, 0);
            }
            const bool act = a.bias != nullptr;
            const float bv = act ? a.bias[k * a.D + d0 + c] : 0.f;
            uint32_t pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float y[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] = act ? softplus20_16bit(acc[4 * g + i] + bv) : acc[4 * g + i];
                pk[g][0] = pack_bf16x2(y[0], y[1]);
                pk[g][1] = pack_bf16x2(y[2], y[3]);
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const u32x2_t r = __builtin_amdgcn_permlane32_swap(pk[g][q], pk[g + 1][q], false, false);
                    pk[g][q] = r[0];
                    pk[g + 1][q] = r[1];
                }
            u32x4_t v0, v1;
            v0[0] = pk[0][0]; v0[1] = pk[0][1]; v0[2] = pk[1][0]; v0[3] = pk[1][1];
            v1[0] = pk[2][0]; v1[1] = pk[2][1]; v1[2] = pk[3][0]; v1[3] = pk[3][1];
            uint16_t *row = out + (int64_t)(d0 + c) * a.L;
            if (pa < a.L) *reinterpret_cast<u32x4_t *>(row + pa) = v0;
            if (pb < a.L) *reinterpret_cast<u32x4_t *>(row + pb) = v1;
        }
        return;
    }
    for (int dt = t0; dt < t1; ++dt) {
        const int d0 = dt * 32;
        xfm_f32x16_t acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const xfm_bf16x8_t af = *reinterpret_cast<const xfm_bf16x8_t *>(wl + ((dt - t0) * 32 + c) * P + 16 * s + 8 * h);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[s], acc, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
            const int row0 = (v & 3) + 8 * (v >> 2) + 4 * h;        // rows of regs v and v + 1 are adjacent channels
            float y0 = acc[v], y1 = acc[v + 1];
            if (a.bias) {
                y0 = softplus20_16bit(y0 + a.bias[k * a.D + d0 + row0]);
                y1 = softplus20_16bit(y1 + a.bias[k * a.D + d0 + row0 + 1]);
            }
            if (valid) {
                const uint32_t pk = pack_bf16x2(y0, y1);
                out[(int64_t)(d0 + row0) * a.L + pos] = (uint16_t)(pk & 0xffffu);
                out[(int64_t)(d0 + row0 + 1) * a.L + pos] = (uint16_t)(pk >> 16);
            }
        }
    }
}

static int dt_proj_mfma(const void *xr, const bf16_t *w_padded, const float *bias, void *out, int B, int D, int R, int RP,
                        int L, hipStream_t s) {
    DtProjMfmaArgs a{};
    a.xr = static_cast<const bf16_t *>(xr); a.w = w_padded; a.bias = bias; a.out = static_cast<bf16_t *>(out);
    a.D = D; a.R = R; a.L = L; a.RP = RP;
    a.wide = (L % 8 == 0 && !getenv("XFM_DTPROJ_NARROW")) ? 1 : 0;
    a.ltiles = ((L + 31) / 32 + 3) / 4;                           // 4 waves = 4 position tiles per workgroup
    int dsplit = 1;
    while ((int64_t)B * 4 * a.ltiles * dsplit < 2048 && dsplit * 2 <= D / 32) dsplit *= 2;
    a.dsplit = dsplit;
    const dim3 grid((unsigned)((int64_t)B * 4 * a.ltiles * dsplit));
    const int dper = (D / 32 + dsplit - 1) / dsplit;
    const size_t lds = (size_t)dper * 32 * (RP + 8) * sizeof(uint16_t);
    if (lds > 64 * 1024) return XFM_ELIMIT;
    if (RP == 16) hipLaunchKernelGGL((dt_proj_mfma_kernel<1>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((dt_proj_mfma_kernel<2>), grid, dim3(256), lds, s, a);
    return check_launch();
}

// ---- backward of dt_proj on MFMA (bf16) ----------------------------------------------------------------------------
//   dxr[b,k,r,l] = sum_d W[k,d,r] * ddts[b,k,d,l]            (R x D) . (D x L): contraction over the channels
//   dW[k,d,r]   += sum_{b,l} ddts[b,k,d,l] * xr[b,k,r,l]     (D x L) . (L x R): contraction over batch and positions
// Both read the big ddts tensor exactly once.  dxr: a wavefront owns 32 positions and walks the channels 16 at a time
// (W^T staged in LDS, zero-padded to 32 rows).  dW: a wavefront owns 32 channels of one (b, k), walks the positions 16
// at a time and adds its 32 x R tile to the fp32 result with atomics (contiguous 4 R-byte runs per channel).
struct DtProjBwdArgs {
    const bf16_t *ddts;   // (B, 4, D, L)
    const bf16_t *xr;     // (B, 4, R, L)
    const bf16_t *w;      // (4, D, R)
    bf16_t *dxr;          // (B, 4, R, L)
    float *dw;            // (4, D, R) fp32, ZEROED by the caller
    int D, R, L, ltiles;
};

__global__ __launch_bounds__(256) void dt_proj_bwd_dx_kernel(DtProjBwdArgs a) {
    extern __shared__ uint16_t wt[];                               // W^T of this k: [32 rows r][D + 8]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bk = blockIdx.x / a.ltiles, k = bk & 3;
    const int lt = (blockIdx.x % a.ltiles) * 4 + wave;
    const int P = a.D + 8;
    {
        const uint16_t *wg = reinterpret_cast<const uint16_t *>(a.w) + (int64_t)k * a.D * a.R;
        for (int e = threadIdx.x; e < 32 * a.D; e += 256) {
            const int r = e / a.D, d = e - r * a.D;
            wt[r * P + d] = r < a.R ? wg[d * a.R + r] : (uint16_t)0;
        }
    }
    __syncthreads();
    if (lt * 32 >= a.L) return;
    const int c = lane & 31, h = lane >> 5;
    const int pos = lt * 32 + c;
    const bool valid = pos < a.L;
    const uint16_t *g = reinterpret_cast<const uint16_t *>(a.ddts) + (int64_t)bk * a.D * a.L;
    xfm_f32x16_t acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    // (channels 64 at a time: the 32 two-byte loads of four MFMA steps are in flight together; D % 32 == 0, so a
    //  trailing half block is handled by the step guard)
#pragma unroll 4
    for (int d0 = 0; d0 < a.D; d0 += 16) {
        uint16_t t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = valid ? g[(int64_t)(d0 + 8 * h + j) * a.L + pos] : (uint16_t)0;
        const xfm_bf16x8_t bfr = *reinterpret_cast<const xfm_bf16x8_t *>(t);
        const xfm_bf16x8_t afr = *reinterpret_cast<const xfm_bf16x8_t *>(wt + c * P + d0 + 8 * h);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, acc, 0, 0, 0);
    }
    uint16_t *o = reinterpret_cast<uint16_t *>(a.dxr) + (int64_t)bk * a.R * a.L;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int r = (v & 3) + 8 * (v >> 2) + 4 * h;
        if (valid && r < a.R) o[(int64_t)r * a.L + pos] = (uint16_t)(pack_bf16x2(acc[v], 0.f) & 0xffffu);
    }
}

__global__ __launch_bounds__(256) void dt_proj_bwd_dw_kernel(DtProjBwdArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dtiles = a.D / 32;
    const int tile = blockIdx.x * 4 + wave;                        // (bk, d-tile)
    if (tile >= a.ltiles) return;                                  // here ltiles = B * 4 * dtiles (total tiles)
    const int bk = tile / dtiles, d0 = (tile - bk * dtiles) * 32, k = bk & 3;
    const int c = lane & 31, h = lane >> 5;
    const uint16_t *g = reinterpret_cast<const uint16_t *>(a.ddts) + ((int64_t)bk * a.D + d0 + c) * a.L;   // row of channel d0+c
    const uint16_t *x = reinterpret_cast<const uint16_t *>(a.xr) + ((int64_t)bk * a.R + c) * a.L;          // row of rank c
    const bool rv = c < a.R;
    xfm_f32x16_t acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll 4
    for (int l0 = 0; l0 < a.L; l0 += 16) {
        uint16_t ta[8], tb[8];
#pragma unroll
        for (int j = 0; j < 8; j += 4) {                           // 8-byte loads: rows are 8-byte aligned (L % 4 == 0)
            const int l = l0 + 8 * h + j;
            uint2 va = make_uint2(0u, 0u), vb = make_uint2(0u, 0u);
            if (l < a.L) {
                va = *reinterpret_cast<const uint2 *>(g + l);
                if (rv) vb = *reinterpret_cast<const uint2 *>(x + l);
            }
            *reinterpret_cast<uint2 *>(ta + j) = va;
            *reinterpret_cast<uint2 *>(tb + j) = vb;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const xfm_bf16x8_t *>(ta),
                                                      *reinterpret_cast<const xfm_bf16x8_t *>(tb), acc, 0, 0, 0);
    }
    if (rv) {
        float *dw = a.dw + ((int64_t)k * a.D + d0) * a.R + c;      // column r = c of this tile
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int row = (v & 3) + 8 * (v >> 2) + 4 * h;
            atomicAdd(dw + (int64_t)row * a.R, acc[v]);
        }
    }
}

// ---- both products in ONE pass over ddts ----------------------------------------------------------------------------
// A workgroup owns one (b, k) slab of ddts -- D rows of L positions, 154 MB per launch at the 56 x 56 stage in all -- and
// walks it in tiles of 128 positions that arrive by LDS-direct loads (rows of 256 bytes, the chunk-XOR image of
// csrc/wgrad_gemm.hip; two tiles ring).  The tile is read twice from LDS: transposed (ds_read_b64_tr_b16: column = position,
// contraction over the channel rows) for d xr = W^T . ddts, and along its rows (lane = channel, eight consecutive positions) for
// dW += ddts . xr^T, whose accumulators live in registers across the whole slab (one set of atomics per workgroup at the end).
// The two separate kernels read ddts with 2-byte / 8-byte per-lane loads (32 different lines per instruction): 32 + 55 us at
// 64 x 4 x 96 x 3136; this one is bound by the stream itself.
constexpr int kDtmTL = 128;                                // positions per tile
__device__ uint4 dtm_zero_page[4];

__device__ __forceinline__ int dtm_off(const int row, const int ch) {      // 16-byte chunk ch (0..15) of row `row`
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}

// NSV: the d xr stores of a wave and tile that have a live lane.  Register v of the accumulator holds rank rows
// (v & 3) + 8 (v >> 2) + 4 h: increasing in v, so the registers with a row below R are the first NSV = dtm_nsv(R).  The counted
// vmcnt waits below are built on EXACTLY this many stores per tile: a store whose EXEC mask is empty is not a memory operation
// and is not counted.  (Rounds 4-5 issued all 16 under lane masks and counted 16: with R = 6 twelve of them were empty, the
// waits for a tile's LDS-direct loads returned with up to 24 of them still in flight, and under memory contention -- two
// processes on one GPU -- the MFMAs read tiles that had not landed: NaN gradients in one run out of three of the two-rank
// bench flow.)
static int dtm_nsv(const int R) {
    int n = 0;
    for (int v = 0; v < 16; ++v) n += ((v & 3) + 8 * (v >> 2)) < R ? 1 : 0;
    return n;
}

template <int DB, int NSV>                                 // DB = D / 32
__global__ void __launch_bounds__(256, 1) dt_proj_bwd_merged_kernel(const DtProjBwdArgs a) {
    constexpr int D = 32 * DB;
    constexpr int TILE = D * 256, XT = 32 * 256;           // bytes: ddts tile, xr tile (32 rank rows)
    constexpr int P = D + 8;                               // row pitch of W^T (halfwords)
    constexpr int NBUF = DB <= 4 ? 3 : 2;                  // tiles in the ring (3 x 40 KB at 128 channels; 2 x 56 KB at 192)
    extern __shared__ __align__(16) uint8_t dl[];          // NBUF x (ddts tile | xr tile) | W^T [32][D + 8]
    uint16_t *wt = reinterpret_cast<uint16_t *>(dl + NBUF * (TILE + XT));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bk = blockIdx.x, k = bk & 3;
    const int R = a.R, L = a.L;
    {
        const uint16_t *wg = reinterpret_cast<const uint16_t *>(a.w) + (int64_t)k * D * R;
        for (int e = tid; e < 32 * D; e += 256) {
            const int r = e / D, d = e - r * D;
            wt[r * P + d] = r < R ? wg[d * R + r] : (uint16_t)0;
        }
    }
    const uint16_t *g = reinterpret_cast<const uint16_t *>(a.ddts) + (int64_t)bk * D * L;
    const uint16_t *x = reinterpret_cast<const uint16_t *>(a.xr) + (int64_t)bk * R * L;
    const uint16_t *zp = reinterpret_cast<const uint16_t *>(dtm_zero_page);
    const int ntile = (L + kDtmTL - 1) / kDtmTL;
    // LDS-direct fill of tile t into buffer t & 1: instruction i of a wave covers rows 4 j .. 4 j + 3 (1 KB), j = wave + 4 i;
    // lane l: row 4 j + (l >> 4), LDS position l & 15 = chunk (l & 15) ^ swz(row) of that row; chunks past L read zeros
    auto issue = [&](const int t) {
        uint8_t *dst = dl + (t % NBUF) * (TILE + XT);
        const int l0 = t * kDtmTL;
#pragma unroll
        for (int i = 0; i < D / 16; ++i) {
            const int j = wave + 4 * i, row = 4 * j + (lane >> 4);
            const int ch = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
            const uint16_t *src = (l0 + 8 * ch < L) ? g + (int64_t)row * L + l0 + 8 * ch : zp;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + j * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                      // xr: 32 rows (ranks >= R: zeros)
            const int j = wave + 4 * i, row = 4 * j + (lane >> 4);
            const int ch = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
            const uint16_t *src = (row < R && l0 + 8 * ch < L) ? x + (int64_t)row * L + l0 + 8 * ch : zp;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + TILE + j * 1024), 16, 0, 0);
        }
    };
    constexpr int NLD = D / 16 + 2;                        // loads per wave and tile
    // stores per wave and tile.  (The asm block of a store ANDs the lane mask into EXEC -- s_and_b64 writes SCC -- and until
    // round 5 did not say so: where the compiler had placed the loop's exit compare in front of the stores, the branch behind them
    // took the store's SCC.  That was the NSV = 4 instance of the two-tile ring, which wrote two tiles per slab and left; every
    // other instance happened to compare after the stores.)
    constexpr int NST = NSV;
    static_assert(NLD < 32, "vmcnt budget");
    xfm_f32x16_t dwacc[DB];
#pragma unroll
    for (int b = 0; b < DB; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) dwacc[b][v] = 0.f;
    const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)dl;
    const int c = lane & 31, h = lane >> 5;
    // transposed fragment addresses (tile buffer 0, channel rows 0..15; k16-step s adds 4096 s bytes): wave w owns positions 32 w ..
    uint32_t trlo, trhi;
    {
        const int gq = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
        const int c0 = (32 * wave + 16 * (gq & 1)) >> 3, r0 = 8 * (gq >> 1);
        trlo = base + dtm_off(r0 + q, c0 + (p >> 1)) + 8 * (p & 1);
        trhi = base + dtm_off(r0 + 4 + q, c0 + (p >> 1)) + 8 * (p & 1);
    }
    typedef __bf16 dtm_bf16x4_t __attribute__((ext_vector_type(4)));
    uint16_t *o = reinterpret_cast<uint16_t *>(a.dxr) + (int64_t)bk * R * L;
    issue(0);
    if (NBUF == 3 && ntile > 1) issue(1);
    for (int t = 0; t < ntile; ++t) {
        // this wave's share of tile t has landed.  Operations retire in issue order; younger than tile t's loads are, in a
        // three-tile ring, the 16 d xr stores of tiles t - 2 and t - 1 and the loads of tile t + 1 (none of which is waited for),
        // in a two-tile ring the 16 stores of tile t - 1.
        // (The first two trips have fewer younger operations: no stores yet at t = 0, one tile's at t = 1.  Until round 5 they
        //  used the steady-state count as well -- i.e. did not wait for tiles 0 and 1 at all: unnoticed on an idle GPU, where
        //  the loads land during the prologue, wrong d xr in 1-10 % of the launches with a second process on the same GPU.)
        if (NBUF == 3 && t + 1 < ntile) {
            if (t >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD + 2 * NSV) : "memory");   // + the stores of two tiles
            else if (t == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD + NSV) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        } else if (NBUF == 2 && t > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");                 // the previous tile's stores
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (first trip: W^T too) everyone's share landed, tile t - 1 is free.  A BARE barrier: __syncthreads() carries a fence
        // that the compiler turns into s_waitcnt vmcnt(0) for LDS-direct loads -- the ring would run with nothing in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (t + NBUF - 1 < ntile) issue(t + NBUF - 1);
        const uint32_t so = (t % NBUF) * (TILE + XT);
        // ---- d xr[r][l] for positions 32 wave .. + 31 of the tile: contraction over the channels; all fragment reads of (up to)
        // six k16-steps are requested before the first MFMA waits
        xfm_f32x16_t acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
        constexpr int KS = D / 16, KH = KS > 8 ? KS / 2 : KS;
#pragma unroll
        for (int s0 = 0; s0 < KS; s0 += KH) {
            dtm_bf16x4_t lo[KH], hi[KH];
            xfm_bf16x8_t afr[KH];
#pragma unroll
            for (int q = 0; q < KH; ++q) {
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[q]) : "v"(trlo + so + 4096 * (s0 + q)) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi[q]) : "v"(trhi + so + 4096 * (s0 + q)) : "memory");
                afr[q] = *reinterpret_cast<const xfm_bf16x8_t *>(wt + c * P + 16 * (s0 + q) + 8 * h);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < KH; ++q) {
                asm volatile("" : "+v"(lo[q]), "+v"(hi[q]));
                const xfm_bf16x8_t bfr = __builtin_shufflevector(lo[q], hi[q], 0, 1, 2, 3, 4, 5, 6, 7);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[q], bfr, acc, 0, 0, 0);
            }
        }
        {
            // NSV stores per tile, ALWAYS issued (lane mask in EXEC; every one of them has live lanes in every tile but the
            // last): the counted wait above needs a fixed number of vector-memory operations per tile
            const int pos = t * kDtmTL + 32 * wave + c;
#pragma unroll
            for (int v = 0; v < NST; ++v) {
                const int r = (v & 3) + 8 * (v >> 2) + 4 * h;
                const bool live = pos < L && r < R;
                const uint64_t m = __builtin_amdgcn_ballot_w64(live);
                const uint32_t val = pack_bf16x2(acc[v], 0.f);
                const uint16_t *ptr = o + (live ? (int64_t)r * L + pos : 0);
                uint64_t sv;
                asm volatile("s_mov_b64 %[sv], exec\n\t"
                             "s_and_b64 exec, exec, %[m]\n\t"
                             "global_store_short %[p], %[v], off\n\t"
                             "s_mov_b64 exec, %[sv]"
                             : [sv] "=&s"(sv) : [m] "s"(m), [p] "v"(ptr), [v] "v"(val) : "memory", "scc");
            }
        }
        // ---- dW[d][r] += sum over the tile's positions: wave w takes k16-steps 2 w, 2 w + 1 (positions 32 w .. + 31)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int ch = 2 * (2 * wave + s2) + h;        // 16-byte chunk of 8 positions
            const xfm_bf16x8_t xfr = *reinterpret_cast<const xfm_bf16x8_t *>(dl + so + TILE + dtm_off(c, ch));
#pragma unroll
            for (int b = 0; b < DB; ++b) {
                const xfm_bf16x8_t gfr = *reinterpret_cast<const xfm_bf16x8_t *>(dl + so + dtm_off(32 * b + c, ch));
                dwacc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gfr, xfr, dwacc[b], 0, 0, 0);
            }
        }
    }
    // ---- the four waves' partial dW (disjoint position ranges) are summed in LDS and leave as D * R / 64 full-wave atomic
    // instructions per workgroup: a CU retires about one atomic wave-instruction per 120 cycles whatever its lane count, and
    // 4 waves x 16 DB instructions of R lanes each were 11 us at the end of every workgroup
    __syncthreads();                                       // (all tiles consumed: the ring is free)
    float *red = reinterpret_cast<float *>(dl);            // [D][R]
    for (int e = tid; e < D * R; e += 256) red[e] = 0.f;
    __syncthreads();
    if (c < R) {
#pragma unroll
        for (int b = 0; b < DB; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = 32 * b + (v & 3) + 8 * (v >> 2) + 4 * h;
                atomicAdd(red + row * R + c, dwacc[b][v]);  // (LDS)
            }
    }
    __syncthreads();
    float *dw = a.dw + (int64_t)k * D * R;
    for (int e = tid; e < D * R; e += 256) atomicAdd(dw + e, red[e]);
}

template <int DB, int NSV> static int dt_proj_bwd_merged_launch_nsv(const DtProjBwdArgs &a, int B, hipStream_t s) {
    constexpr int D = 32 * DB;
    const size_t lds = (size_t)(DB <= 4 ? 3 : 2) * (D * 256 + 32 * 256) + (size_t)32 * (D + 8) * 2;
    auto fn = dt_proj_bwd_merged_kernel<DB, NSV>;
    static xfm::LdsOptIn opted;
    if (lds > 64 * 1024) {
        if (!xfm::lds_opt_in(opted, (const void *)fn, lds)) return XFM_ELAUNCH;
    }
    hipLaunchKernelGGL(fn, dim3((unsigned)(B * 4)), dim3(256), lds, s, a);
    return check_launch();
}

// (built for the store counts of whole register groups: R in 4..8, 12..16, 20..24, 28..32; XFM_ELIMIT otherwise -- the caller
//  then runs the two separate kernels)
template <int DB> static int dt_proj_bwd_merged_launch(const DtProjBwdArgs &a, int B, hipStream_t s) {
    switch (dtm_nsv(a.R)) {
        case 4: return dt_proj_bwd_merged_launch_nsv<DB, 4>(a, B, s);
        case 8: return dt_proj_bwd_merged_launch_nsv<DB, 8>(a, B, s);
        case 12: return dt_proj_bwd_merged_launch_nsv<DB, 12>(a, B, s);
        case 16: return dt_proj_bwd_merged_launch_nsv<DB, 16>(a, B, s);
    }
    return XFM_ELIMIT;
}

}  // namespace xfm

extern "C" {

/* bf16 MFMA path: weight_bf16 is the (4, D, R) weight in bf16; returns the padded contraction length (16 / 32) or 0. */
int xfm_ss2d_dt_proj_mfma_rp(int D, int R, int L) {
    if (D % 32 != 0 || R < 1 || R > 32 || L < 1) return 0;
    return R <= 16 ? 16 : 32;
}

int xfm_ss2d_dt_proj_fwd_mfma(const void *xr, const void *weight_bf16, const float *softplus_bias, void *dts, int B, int D,
                              int R, int L, void *stream) {
    using namespace xfm;
    if (!xr || !weight_bf16 || !dts || B <= 0) return XFM_EINVAL;
    const int RP = xfm_ss2d_dt_proj_mfma_rp(D, R, L);
    if (!RP) return XFM_ELIMIT;
    return dt_proj_mfma(xr, static_cast<const bf16_t *>(weight_bf16), softplus_bias, dts, B, D, R, RP, L, (hipStream_t)stream);
}

/* bf16 backward of dt_proj on MFMA: dxr (B,4,R,L) bf16 and dweight (4,D,R) fp32 (ZEROED by the caller, accumulated with
 * atomics); D % 32 == 0, D <= 1024, R <= 32, L % 4 == 0. */
int xfm_ss2d_dt_proj_bwd_mfma(const void *ddts, const void *xr, const void *weight_bf16, void *dxr, float *dweight, int B,
                              int D, int R, int L, void *stream) {
    using namespace xfm;
    if (!ddts || !xr || !weight_bf16 || !dxr || !dweight || B <= 0) return XFM_EINVAL;
    if (!xfm_ss2d_dt_proj_mfma_rp(D, R, L) || L % 4 != 0 || D > 1024) return XFM_ELIMIT;
    hipStream_t s = (hipStream_t)stream;
    DtProjBwdArgs a{};
    a.ddts = static_cast<const bf16_t *>(ddts); a.xr = static_cast<const bf16_t *>(xr);
    a.w = static_cast<const bf16_t *>(weight_bf16); a.dxr = static_cast<bf16_t *>(dxr); a.dw = dweight;
    a.D = D; a.R = R; a.L = L;
    // one pass over ddts for both products where a (b, k) slab per workgroup fills the chip and the ring fits LDS
    static const bool merged_on = [] { const char *e = getenv("XFM_DTPROJ_MERGED"); return !e || atoi(e) != 0; }();
    if (merged_on && B * 4 >= 128 && L % 8 == 0 && (((uintptr_t)ddts | (uintptr_t)xr) & 15) == 0) {
        int rc = XFM_ELIMIT;
        if (D == 96) rc = dt_proj_bwd_merged_launch<3>(a, B, s);
        else if (D == 128) rc = dt_proj_bwd_merged_launch<4>(a, B, s);
        else if (D == 192) rc = dt_proj_bwd_merged_launch<6>(a, B, s);   // (256 channels: the two-tile ring is 164 KB)
        if (rc != XFM_ELIMIT) return rc;
    }
    a.ltiles = ((L + 31) / 32 + 3) / 4;
    const size_t lds = (size_t)32 * (D + 8) * sizeof(uint16_t);
    if (lds > 64 * 1024) {                                        // D = 1024 (XFMamba-B stage 2): 66 048 B, opt in
        static xfm::LdsOptIn once;
        if (!xfm::lds_opt_in(once, (const void *)dt_proj_bwd_dx_kernel, 160 * 1024)) return XFM_ELAUNCH;
    }
    hipLaunchKernelGGL(dt_proj_bwd_dx_kernel, dim3((unsigned)((int64_t)B * 4 * a.ltiles)), dim3(256), lds, s, a);
    int rc = check_launch();
    if (rc != XFM_OK) return rc;
    a.ltiles = B * 4 * (D / 32);                                  // dw kernel: total (b, k, channel-tile) count
    hipLaunchKernelGGL(dt_proj_bwd_dw_kernel, dim3((unsigned)((a.ltiles + 3) / 4)), dim3(256), 0, s, a);
    return check_launch();
}

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
