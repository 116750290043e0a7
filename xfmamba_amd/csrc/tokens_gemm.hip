// tokens_gemm.hip -- Y[T, OUT] = X[T, CON] . Wl[OUT, CON]^T (+ bias) on the token-major stream, bf16 in / bf16 out,
// fp32 accumulation on MFMA, for the SKINNY products of the 56x56 stage (CON, OUT in {96, 384}; T = B*H*W ~ 2e5).
//
// These products are HBM-bound (the weight is 72 KB; X is read once and Y written once), but a general GEMM library
// tiles them for compute.  Here the whole weight lives in LDS for the lifetime of a persistent workgroup, a wavefront
// owns 32 consecutive rows at a time, the A fragments of its next row tile are requested before the current tile's
// MFMAs, and nothing but X and Y touches HBM.  The same kernel serves the backward data product (dX = dY . W) by staging
// the weight transposed.
//
// v_mfma_f32_32x32x16_bf16 operand layout (lane l: r = l & 31, h = l >> 5): A[r][8h + j], B[8h + j][r];
// D: column l & 31, row (reg & 3) + 8 (reg >> 2) + 4 h.  The product is formed TRANSPOSED, D[n][t] with A = weight rows
// and B = token rows, so that a lane (token t = l & 31) ends up with runs of four consecutive output channels per
// register group; after the bf16 pack two v_permlane32_swap per pair of groups hand each lane eight consecutive
// channels, i.e. 16-byte stores (2-byte stores, one channel per lane, ran the wide-output shapes 1.6x SLOWER than the
// library).
#include "xfm_common.hpp"

#include <algorithm>
#include <cstdlib>

namespace xfm {

typedef __bf16 tg_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float tg_f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t tg_u32x4_t __attribute__((ext_vector_type(4)));

struct TokGemmArgs {
    const uint16_t *x;      // (T, CON) bf16
    const uint16_t *w;      // weight bf16: (OUT, CON) row-major, or (CON, OUT) when wt != 0
    const float *bias;      // (OUT) or null
    uint16_t *y;            // (T, OUT) bf16
    int64_t T;
    int wt;
    int L;                  // layout-changing variants: tokens per sample (planes are (B, C, L)); L % 8 == 0 && (B*L) % 32 == 0
    int con_r, out_r;       // proj_gemm_kernel: real contraction / output widths (<= the kernel's CON / OUT: the weight is zero-
                            // padded in LDS, plane rows past con_r read as zero, channels past out_r are not stored); 0: CON / OUT
};

template <int CON, int OUT, int OB>       // OB: output columns processed per pass (accumulators OB/32 x 16 registers)
__global__ void __launch_bounds__(512) tokens_gemm_kernel(const TokGemmArgs a) {
    constexpr int P = CON + 8;             // LDS row pitch (halfwords): 16-byte rows, conflict-free 16-byte column reads
    constexpr int KS = CON / 16, NB = OB / 32, NPASS = OUT / OB;
    static_assert(CON % 16 == 0 && OUT % OB == 0 && OB % 32 == 0, "shape");
    extern __shared__ __align__(16) uint16_t wl[];              // [OUT][P] weight, then OUT floats of bias
    float *bl = reinterpret_cast<float *>(wl + OUT * P);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int n = threadIdx.x; n < OUT; n += 512) bl[n] = a.bias ? a.bias[n] : 0.f;
    // ---- stage the weight once per workgroup
    if (!a.wt) {
        constexpr int VPR = CON / 8;                               // 16-byte vectors per row
        for (int v = threadIdx.x; v < OUT * VPR; v += 512) {
            const int n = v / VPR, q = v - n * VPR;
            *reinterpret_cast<tg_u32x4_t *>(wl + n * P + 8 * q) = reinterpret_cast<const tg_u32x4_t *>(a.w)[v];
        }
    } else {                                                       // w is (CON, OUT): transpose while staging
        for (int e = threadIdx.x; e < CON * OUT; e += 512) {
            const int k = e / OUT, n = e - k * OUT;
            wl[n * P + k] = a.w[e];
        }
    }
    __syncthreads();
    const int c = lane & 31, h = lane >> 5;
    const int64_t ntiles = (a.T + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * 8;
    int64_t tile = (int64_t)blockIdx.x * 8 + wave;
    // A fragments of the NEXT tile are requested one tile ahead while they fit the register budget of two waves per
    // SIMD (CON <= 128); longer rows are loaded at the top of their tile and the other resident waves cover the latency
    constexpr bool PREF = KS <= 8;
    constexpr int NA = PREF ? KS : 1;
    tg_u32x4_t an[NA];
    auto src_of = [&](int64_t tl) {
        int64_t row = tl * 32 + c;
        if (row >= a.T) row = a.T - 1;
        return reinterpret_cast<const tg_u32x4_t *>(a.x + row * CON + 8 * h);   // [2 s]: columns 16 s + 8 h .. + 7
    };
    if (PREF && tile < ntiles) {
        const tg_u32x4_t *src = src_of(tile);
#pragma unroll
        for (int s = 0; s < NA; ++s) an[s] = src[2 * s];
    }
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    auto store = [&](const tg_f32x16_t (&acc)[NB], int ps, int64_t t0) {
        const int64_t row = t0 + c;
        uint16_t *yr = a.y + row * OUT + ps * OB + 8 * h;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            uint32_t pk[4][2];                                     // group g: channels 32 b + 8 g + 4 h .. + 3
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4 *>(bl + ps * OB + 32 * b + 8 * g + 4 * h);
                pk[g][0] = pack_bf16x2(acc[b][4 * g] + bv.x, acc[b][4 * g + 1] + bv.y);
                pk[g][1] = pack_bf16x2(acc[b][4 * g + 2] + bv.z, acc[b][4 * g + 3] + bv.w);
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2)
#pragma unroll
                for (int q = 0; q < 2; ++q) {                      // lower half gets channels +4..7 of group g, upper +0..3 of g+1
                    const u32x2_t r = __builtin_amdgcn_permlane32_swap(pk[g][q], pk[g + 1][q], false, false);
                    pk[g][q] = r[0];
                    pk[g + 1][q] = r[1];
                }
            if (row < a.T) {
                tg_u32x4_t v0, v1;
                v0[0] = pk[0][0]; v0[1] = pk[0][1]; v0[2] = pk[1][0]; v0[3] = pk[1][1];      // channels 32 b + 8 h .. + 7
                v1[0] = pk[2][0]; v1[1] = pk[2][1]; v1[2] = pk[3][0]; v1[3] = pk[3][1];      // channels 32 b + 16 + 8 h .. + 7
                *reinterpret_cast<tg_u32x4_t *>(yr + 32 * b) = v0;
                *reinterpret_cast<tg_u32x4_t *>(yr + 32 * b + 16) = v1;
            }
        }
    };
    for (; tile < ntiles; tile += stride) {
        const int64_t t0 = tile * 32;
        if constexpr (PREF) {
            tg_u32x4_t af[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) af[s] = an[s];
            if (tile + stride < ntiles) {
                const tg_u32x4_t *src = src_of(tile + stride);
#pragma unroll
                for (int s = 0; s < NA; ++s) an[s] = src[2 * s];
            }
#pragma unroll 1
            for (int ps = 0; ps < NPASS; ++ps) {
                tg_f32x16_t acc[NB];
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int v = 0; v < 16; ++v) acc[b][v] = 0.f;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const tg_bf16x8_t afr = __builtin_bit_cast(tg_bf16x8_t, af[s]);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const tg_bf16x8_t bfr =
                            *reinterpret_cast<const tg_bf16x8_t *>(wl + (ps * OB + b * 32 + c) * P + 16 * s + 8 * h);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr, afr, acc[b], 0, 0, 0);
                    }
                }
                store(acc, ps, t0);
            }
        } else {
            // long rows: the contraction runs in groups of KG k-steps, each group's A fragments loaded just before it
            static_assert(PREF || NPASS == 1, "long contraction: one pass over the outputs");
            constexpr int KG = 8;
            static_assert(KS % KG == 0, "k-step groups");
            const tg_u32x4_t *src = src_of(tile);
            tg_f32x16_t acc[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[b][v] = 0.f;
#pragma unroll 1
            for (int s0 = 0; s0 < KS; s0 += KG) {
                tg_u32x4_t af[KG];
#pragma unroll
                for (int s = 0; s < KG; ++s) af[s] = src[2 * (s0 + s)];
#pragma unroll
                for (int s = 0; s < KG; ++s) {
                    const tg_bf16x8_t afr = __builtin_bit_cast(tg_bf16x8_t, af[s]);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const tg_bf16x8_t bfr =
                            *reinterpret_cast<const tg_bf16x8_t *>(wl + (b * 32 + c) * P + 16 * (s0 + s) + 8 * h);
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr, afr, acc[b], 0, 0, 0);
                    }
                }
            }
            store(acc, 0, t0);
        }
    }
}

// ---- layout-changing projections: token-major in -> plane-major out (IN_PL = false) or the reverse.  Same weight
// residency and epilogue; L % 8 == 0 and (B * L) % 32 == 0.
//   tokens -> planes: D[t][n] (A = token rows, B = weight): a lane owns channel n and, after the swap, 8 consecutive tokens
//   planes -> tokens: D[n][t] (A = weight, B[k][t] gathered from the planes with 2-byte loads, 64 bytes per wave and k)
// OUT_PL: plane-major output (lane owns a channel, D[t][n]); otherwise token rows (D[n][t]).  ACC (plane output):
// y += product, the existing bf16 values widened, added in fp32 and rounded once (x_proj backward adds into dx).
template <int CON, int OUT, bool IN_PL, bool OUT_PL, bool ACC>
__global__ void __launch_bounds__(512) proj_gemm_kernel(const TokGemmArgs a) {
    constexpr int P = CON + 8;
    constexpr int KS = CON / 16, NB = OUT / 32;
    static_assert(CON % 16 == 0 && OUT % 32 == 0 && KS <= 12 && NB <= 6, "shape");
    static_assert(IN_PL || OUT_PL, "token rows on both sides: tokens_gemm_kernel");
    static_assert(!ACC || OUT_PL, "accumulation is implemented for plane-major outputs");
    constexpr int KH = KS > 6 ? KS / 2 : KS;                      // plane gathers: k-steps fetched together (48 loads)
    static_assert(KS % KH == 0, "k halves");
    extern __shared__ __align__(16) uint16_t wl[];
    float *bl = reinterpret_cast<float *>(wl + OUT * P);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int con_r = a.con_r > 0 ? a.con_r : CON, out_r = a.out_r > 0 ? a.out_r : OUT;
    const bool padded = con_r != CON || out_r != OUT;
    for (int n = threadIdx.x; n < OUT; n += 512) bl[n] = (a.bias && n < out_r) ? a.bias[n] : 0.f;
    if (padded) {
        // (widths that are not whole MFMA tiles: x_proj of the 28 x 28 stage, 192 <-> 4 x 14 = 56 rows; the weight is (out_r,
        //  con_r) row-major, or (con_r, out_r) when wt, and sits zero-padded to (OUT, CON) in LDS)
        for (int e = threadIdx.x; e < CON * OUT; e += 512) {
            const int n = e / CON, k = e - n * CON;
            const bool in = n < out_r && k < con_r;
            wl[n * P + k] = in ? (a.wt ? a.w[k * out_r + n] : a.w[n * con_r + k]) : (uint16_t)0;
        }
    } else if (!a.wt) {
        constexpr int VPR = CON / 8;
        for (int v = threadIdx.x; v < OUT * VPR; v += 512) {
            const int n = v / VPR, q = v - n * VPR;
            *reinterpret_cast<tg_u32x4_t *>(wl + n * P + 8 * q) = reinterpret_cast<const tg_u32x4_t *>(a.w)[v];
        }
    } else {
        for (int e = threadIdx.x; e < CON * OUT; e += 512) {
            const int k = e / OUT, n = e - k * OUT;
            wl[n * P + k] = a.w[e];
        }
    }
    __syncthreads();
    const int c = lane & 31, h = lane >> 5;
    const int64_t ntiles = a.T / 32;
    const int64_t stride = (int64_t)gridDim.x * 8;
    int64_t tile = (int64_t)blockIdx.x * 8 + wave;
    const int L = a.L;
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    // Samples may end inside a 32-token tile (L % 8 == 0 only): a lane works out the sample of ITS token (gathers) or of
    // its 8-token output run (plane stores) itself.
    tg_u32x4_t an[IN_PL ? KH : KS];                                // token rows: the next tile's fragments; planes: one k half
    auto fetch_tokens = [&](int64_t tl) {
        const tg_u32x4_t *src = reinterpret_cast<const tg_u32x4_t *>(a.x + (tl * 32 + c) * CON + 8 * h);
#pragma unroll
        for (int s = 0; s < KS; ++s) an[s] = src[2 * s];
    };
    auto fetch_planes = [&](int64_t tl, int s0) {                  // k-steps s0 .. s0 + KH - 1 of tile tl
        const int64_t t = tl * 32 + c, bi = t / L;
        const uint16_t *pp = a.x + (bi * con_r + 16 * s0 + 8 * h) * L + (t - bi * L);
#pragma unroll
        for (int s = 0; s < KH; ++s) {
            uint16_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (padded) {                                                    // (rows past the real width: the zero padding)
                    const int k = 16 * (s0 + s) + 8 * h + j;
                    v[j] = k < con_r ? pp[(int64_t)j * L] : (uint16_t)0;
                } else {
                    v[j] = pp[(int64_t)j * L];
                }
            }
            pp += 16 * (int64_t)L;
            asm volatile("" : "+v"(pp));                                         // one running address, not 48 of them
#pragma unroll
            for (int q = 0; q < 4; ++q) an[s][q] = (uint32_t)v[2 * q] | ((uint32_t)v[2 * q + 1] << 16);
        }
    };
    constexpr bool PREFT = KS <= 8;                                // token rows one tile ahead while the registers allow
    if constexpr (!IN_PL && PREFT) {
        if (tile < ntiles) fetch_tokens(tile);
    }
    for (; tile < ntiles; tile += stride) {
        const int64_t t0 = tile * 32;
        tg_f32x16_t acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[b][v] = 0.f;
        if constexpr (!IN_PL) {
            tg_u32x4_t af[PREFT ? KS : 1];
            if constexpr (PREFT) {
#pragma unroll
                for (int s = 0; s < KS; ++s) af[s] = an[s];
                if (tile + stride < ntiles) fetch_tokens(tile + stride);
            } else {
                fetch_tokens(tile);
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const tg_bf16x8_t xfr = __builtin_bit_cast(tg_bf16x8_t, PREFT ? af[PREFT ? s : 0] : an[s]);
                if (s % 4 == 3) __builtin_amdgcn_sched_barrier(0);     // keep the weight-fragment reads from piling up
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const tg_bf16x8_t wfr = *reinterpret_cast<const tg_bf16x8_t *>(wl + (b * 32 + c) * P + 16 * s + 8 * h);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xfr, wfr, acc[b], 0, 0, 0);                  // D[t][n] (tokens in: planes out)
                }
            }
        } else {
            // (plane gathers are 48 two-byte loads per lane and k half: issued at the top of their own half -- holding
            //  more in flight spills at two waves per SIMD; the other resident waves cover the latency)
#pragma unroll 1
            for (int s0 = 0; s0 < KS; s0 += KH) {
                fetch_planes(tile, s0);
#pragma unroll
                for (int s = 0; s < KH; ++s) {
                    const tg_bf16x8_t xfr = __builtin_bit_cast(tg_bf16x8_t, an[s]);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const tg_bf16x8_t wfr =
                            *reinterpret_cast<const tg_bf16x8_t *>(wl + (b * 32 + c) * P + 16 * (s0 + s) + 8 * h);
                        if constexpr (OUT_PL) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xfr, wfr, acc[b], 0, 0, 0);      // D[t][n]
                        else acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr, xfr, acc[b], 0, 0, 0);                 // D[n][t]
                    }
                }
            }
        }
        // plane stores: the lane's two 8-token runs start at tokens t0 + 8 h and t0 + 16 + 8 h
        const int64_t ta = t0 + 8 * h, tb = ta + 16, ba = ta / L, bb2 = tb / L;
        const int64_t oa = ba * out_r * L + (ta - ba * L), ob = bb2 * out_r * L + (tb - bb2 * L);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            uint32_t pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 bv;
                if constexpr (!OUT_PL) bv = *reinterpret_cast<const float4 *>(bl + 32 * b + 8 * g + 4 * h);   // per channel
                else bv.x = bv.y = bv.z = bv.w = bl[32 * b + c];                                              // lane's channel
                if constexpr (ACC) {                   // the four tokens t0 + 8 g + 4 h .. + 3 of channel 32 b + c, before the swap
                    const int64_t tg = t0 + 8 * g + 4 * h, bg = tg / L;
                    const int cc = 32 * b + c < out_r ? 32 * b + c : out_r - 1;         // (padding channels: read anything valid)
                    const uint2 old = *reinterpret_cast<const uint2 *>(a.y + (bg * out_r + cc) * L + (tg - bg * L));
                    bv.x += __uint_as_float(old.x << 16);
                    bv.y += __uint_as_float(old.x & 0xffff0000u);
                    bv.z += __uint_as_float(old.y << 16);
                    bv.w += __uint_as_float(old.y & 0xffff0000u);
                }
                pk[g][0] = pack_bf16x2(acc[b][4 * g] + bv.x, acc[b][4 * g + 1] + bv.y);
                pk[g][1] = pack_bf16x2(acc[b][4 * g + 2] + bv.z, acc[b][4 * g + 3] + bv.w);
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const u32x2_t r = __builtin_amdgcn_permlane32_swap(pk[g][q], pk[g + 1][q], false, false);
                    pk[g][q] = r[0];
                    pk[g + 1][q] = r[1];
                }
            tg_u32x4_t v0, v1;
            v0[0] = pk[0][0]; v0[1] = pk[0][1]; v0[2] = pk[1][0]; v0[3] = pk[1][1];
            v1[0] = pk[2][0]; v1[1] = pk[2][1]; v1[2] = pk[3][0]; v1[3] = pk[3][1];
            if constexpr (!OUT_PL) {                                                              // token row, 8 channels
                uint16_t *dst = a.y + (t0 + c) * OUT + 32 * b + 8 * h;
                *reinterpret_cast<tg_u32x4_t *>(dst) = v0;
                *reinterpret_cast<tg_u32x4_t *>(dst + 16) = v1;
            } else if (32 * b + c < out_r) {                                                      // channel plane, 8 tokens
                *reinterpret_cast<tg_u32x4_t *>(a.y + oa + (int64_t)(32 * b + c) * L) = v0;
                *reinterpret_cast<tg_u32x4_t *>(a.y + ob + (int64_t)(32 * b + c) * L) = v1;
            }
        }
    }
}

template <int CON, int OUT, bool IN_PL, bool OUT_PL = !IN_PL, bool ACC = false>
static int proj_gemm_launch(const TokGemmArgs &a, hipStream_t s) {
    const size_t lds = (size_t)OUT * (CON + 8) * sizeof(uint16_t) + (size_t)OUT * sizeof(float);
    auto fn = proj_gemm_kernel<CON, OUT, IN_PL, OUT_PL, ACC>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int64_t ntiles = a.T / 32;
    int grid = (int)std::min<int64_t>((ntiles + 7) / 8, 512);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds, s, a);
    return check_launch();
}

// ---------------------------------------------------------------------------------------------------------------------
// Chunked form for the WIDE side of the Mlp (fc1 forward x W1^T and fc2's data gradient dy W2: contraction over the
// stream width C, output over the hidden width 4 C) at the 28 x 28 / 14 x 14 / 7 x 7 stages, with the GELU fused in.
//
// The weight no longer fits LDS as a whole (4 C x C: 1.2 MB at C = 384), so a workgroup owns a CHUNK of OCH output
// columns ([OCH][CON + 8] bf16, ~100 KB) for its whole life and walks token tiles; the token rows are re-read once per
// chunk from L2 (C is the narrow side: 9.6 MB at C = 384).  No staging pipeline, no barrier in the main loop: the only
// traffic of a tile is its A fragments (global, requested a k-group ahead) and 16-byte LDS reads of the resident weight.
// The library runs these products at 15 % of the bf16 MFMA peak (39 us at 12544 x 384 x 1536) and leaves bias + GELU to
// a separate pass over the hidden activation (reference models/fusion_vmamba.py:135-153: fc1 -> act -> fc2).  Epilogues:
//   EPI 0: y = x W^T + bias
//   EPI 1: z = x W^T (bf16, kept for the backward pass) and g = gelu(z + bias)            (fc1 + GELU, forward)
//   EPI 2: dz = bf16(dy W) * gelu'(z + bias)                                              (fc2 data gradient + GELU')
// both exactly what the unfused chain computes (z / dg rounded to bf16 before the activation is applied).
struct TokGemm2Args {
    const uint16_t *x;      // (T, CON) bf16
    const uint16_t *w;      // (OUT, CON) row-major, or (CON, OUT) when wt != 0
    const float *bias;      // (OUT) or null
    uint16_t *y;            // (T, OUT): EPI 0 result / EPI 1 z / EPI 2 dz
    uint16_t *y2;           // (T, OUT): EPI 1 g
    const uint16_t *zin;    // (T, OUT): EPI 2 the forward pass's z
    int64_t T;
    int OUT, wt, wgs_per_chunk;
    float *colpart;         // tiled form, EPI 2: (ceil(T / 128), OUT) per-tile column sums of dz (the bias gradient's partial rows) or null
};

constexpr float kTgInvSqrt2 = 0.70710678118654752f, kTgInvSqrt2Pi = 0.3989422804014327f;
// erf by Abramowitz & Stegun 7.1.26 on the hardware exp2 / rcp (|error| <= 1.5e-7), as csrc/tokens_ops.hip; E = exp(-x^2)
__device__ __forceinline__ float tg_erf(float x, float &E) {
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

// timing-only switches (build with -DXFM_GEMM2_TIMING, then XFM_GEMM2_DBG=<bits>: 2 every tile reads the same rows, 4 no
// epilogue, 8 no MFMA, 16 no token-row loads, 32 staging only; the tiled form: XFM_G3_DBG=<bits> 4 no epilogue stores, 8 no MFMA,
// 16 no operand loads, 64 no GELU arithmetic, 128 no epilogue); the production build compiles them away.  Measured with them
// at 12544 x 384 -> 1536 (45 us): staging 4.5 us, the tile loop without MFMAs / loads / stores another 10, MFMAs + token rows
// +9, the GELU epilogue and its two 38.5 MB stores +20.
__device__ __forceinline__ bool tg2_dbg(const TokGemm2Args &a, const int bit) {
#ifdef XFM_GEMM2_TIMING
    return (a.wt & bit) != 0;
#else
    return false;
#endif
}

template <int CON, int OCH, int EPI, int NT>
__global__ void __launch_bounds__(NT) tokens_gemm2_kernel(const TokGemm2Args a) {
    constexpr int P = CON + 8;             // LDS row pitch (halfwords): 16-byte rows, conflict-free 16-byte column reads
    constexpr int KS = CON / 16, NB = OCH / 32, KG = KS % 8 == 0 ? 8 : 6, NG = KS / KG;
    static_assert(KS % KG == 0 && OCH % 32 == 0, "shape");
    extern __shared__ __align__(16) uint16_t wl[];              // [OCH][P] weight chunk, OCH floats of bias, the waves' output images
    float *bl = reinterpret_cast<float *>(wl + OCH * P);
    constexpr int SP = OCH + 8;                                  // row pitch of an output image (halfwords)
    uint16_t *stage = reinterpret_cast<uint16_t *>(bl + OCH);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int chunk = blockIdx.x / a.wgs_per_chunk, wg = blockIdx.x - chunk * a.wgs_per_chunk;
    const int n0 = chunk * OCH;
    for (int n = threadIdx.x; n < OCH; n += NT) bl[n] = a.bias ? a.bias[n0 + n] : 0.f;
    // stage the weight chunk: ALL of a thread's loads are requested before the first LDS store (a load / store loop exposed
    // one L2 round trip per 16 bytes: 12 of them, most of the kernel's time at these sizes)
    constexpr int NVS = OCH * CON / 8 / NT;                         // 16-byte vectors per thread
    static_assert(OCH * CON % (8 * NT) == 0, "staging vectors");
    tg_u32x4_t sv[NVS];
    if (!(a.wt & 1)) {
        constexpr int VPR = CON / 8;                               // 16-byte vectors per row
        const tg_u32x4_t *src = reinterpret_cast<const tg_u32x4_t *>(a.w + (int64_t)n0 * CON);
#pragma unroll
        for (int i = 0; i < NVS; ++i) sv[i] = src[threadIdx.x + i * NT];
#pragma unroll
        for (int i = 0; i < NVS; ++i) {
            const int v = threadIdx.x + i * NT, n = v / VPR, q = v - n * VPR;
            *reinterpret_cast<tg_u32x4_t *>(wl + n * P + 8 * q) = sv[i];
        }
    } else {                                                       // w is (CON, OUT): transpose while staging
        constexpr int VPK = OCH / 8;                                // vectors of 8 output columns per k
#pragma unroll
        for (int i = 0; i < NVS; ++i) {
            const int v = threadIdx.x + i * NT, k = v / VPK, n = 8 * (v - k * VPK);
            sv[i] = *reinterpret_cast<const tg_u32x4_t *>(a.w + (int64_t)k * a.OUT + n0 + n);
        }
#pragma unroll
        for (int i = 0; i < NVS; ++i) {
            const int v = threadIdx.x + i * NT, k = v / VPK, n = 8 * (v - k * VPK);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                wl[(n + 2 * q) * P + k] = (uint16_t)(sv[i][q] & 0xffffu);
                wl[(n + 2 * q + 1) * P + k] = (uint16_t)(sv[i][q] >> 16);
            }
        }
    }
    __syncthreads();
    if (tg2_dbg(a, 32)) return;                                    // (timing switch: staging only)
    const int c = lane & 31, h = lane >> 5;
    const int64_t ntiles = (a.T + 31) / 32;
    const int64_t stride = (int64_t)a.wgs_per_chunk * (NT / 64);
    int64_t tile = (int64_t)wg * (NT / 64) + wave;
    auto src_of = [&](int64_t tl) {
        int64_t row = tl * 32 + c;
        if (row >= a.T) row = a.T - 1;
        if (tg2_dbg(a, 2)) row = c;                                // (timing switch: every tile reads the same 32 rows)
        return reinterpret_cast<const tg_u32x4_t *>(a.x + row * CON + 8 * h);   // [2 s]: columns 16 s + 8 h .. + 7
    };
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    // A fragments travel one k-group (KG k-steps) ahead of their MFMAs, across tile boundaries
    tg_u32x4_t nx[KG];
    if (tile < ntiles) {
        const tg_u32x4_t *src = src_of(tile);
#pragma unroll
        for (int s = 0; s < KG; ++s) nx[s] = src[2 * s];
    }
    // weight fragments travel ONE k-step ahead of their MFMAs, also across k-groups and tiles (the weight does not depend on
    // the tile: the step after a tile's last one reads the fragments of step 0 again); two register sets, KS even
    static_assert(KS % 2 == 0, "fragment ping-pong");
    tg_bf16x8_t wb[2][NB];
    const uint16_t *wrow = wl + c * P + 8 * h;
#pragma unroll
    for (int b = 0; b < NB; ++b) wb[0][b] = *reinterpret_cast<const tg_bf16x8_t *>(wrow + b * 32 * P);
    for (; tile < ntiles; tile += stride) {
        const int64_t t0 = tile * 32;
        tg_f32x16_t acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[b][v] = 0.f;
        tg_u32x4_t af[KG];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int g = ks / KG, sgi = ks % KG;
            if (sgi == 0) {
#pragma unroll
                for (int q = 0; q < KG; ++q) af[q] = nx[q];
                if (tg2_dbg(a, 16)) {                              // (timing switch: no token-row loads)
                } else if (g + 1 < NG) {
                    const tg_u32x4_t *src = src_of(tile);
#pragma unroll
                    for (int q = 0; q < KG; ++q) nx[q] = src[2 * ((g + 1) * KG + q)];
                } else if (tile + stride < ntiles) {
                    const tg_u32x4_t *src = src_of(tile + stride);
#pragma unroll
                    for (int q = 0; q < KG; ++q) nx[q] = src[2 * q];
                }
            }
            const int kn = (ks + 1) % KS;
#pragma unroll
            for (int b = 0; b < NB; ++b) wb[(ks + 1) & 1][b] = *reinterpret_cast<const tg_bf16x8_t *>(wrow + b * 32 * P + 16 * kn);
            const tg_bf16x8_t afr = __builtin_bit_cast(tg_bf16x8_t, af[sgi]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (tg2_dbg(a, 8)) {                               // (timing switch: no matrix instruction)
                    asm volatile("" :: "v"(wb[ks & 1][b]), "v"(afr));
                } else {
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[ks & 1][b], afr, acc[b], 0, 0, 0);   // D[n][t]
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue: D[n][t] -> a lane (token t0 + c) ends up with runs of eight consecutive output channels.  Written
        // straight to memory that is 64 different 128-byte lines per store instruction (rows are OUT * 2 bytes apart): the
        // address path takes 64 cycles for each, and the loads of z in the backward form are the same shape.  So the tile goes
        // through a wave-private LDS image [32 tokens][OCH + 8] first and leaves in whole rows: 16 (8) lanes per row, four
        // (eight) rows per instruction.
        uint16_t *stg = stage + wave * (32 * SP);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            __builtin_amdgcn_sched_barrier(0);                     // one accumulator at a time (register pressure)
            uint32_t pk[4][2];                                     // group g: channels 32 b + 8 g + 4 h .. + 3
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 bv = {0.f, 0.f, 0.f, 0.f};
                if constexpr (EPI == 0) bv = *reinterpret_cast<const float4 *>(bl + 32 * b + 8 * g + 4 * h);
                pk[g][0] = pack_bf16x2(acc[b][4 * g] + bv.x, acc[b][4 * g + 1] + bv.y);
                pk[g][1] = pack_bf16x2(acc[b][4 * g + 2] + bv.z, acc[b][4 * g + 3] + bv.w);
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2)
#pragma unroll
                for (int q = 0; q < 2; ++q) {                      // lower half gets channels +4..7 of group g, upper +0..3 of g+1
                    const u32x2_t r = __builtin_amdgcn_permlane32_swap(pk[g][q], pk[g + 1][q], false, false);
                    pk[g][q] = r[0];
                    pk[g + 1][q] = r[1];
                }
            tg_u32x4_t v0, v1;
            v0[0] = pk[0][0]; v0[1] = pk[0][1]; v0[2] = pk[1][0]; v0[3] = pk[1][1];      // channels 32 b + 8 h .. + 7
            v1[0] = pk[2][0]; v1[1] = pk[2][1]; v1[2] = pk[3][0]; v1[3] = pk[3][1];      // channels 32 b + 16 + 8 h .. + 7
            *reinterpret_cast<tg_u32x4_t *>(stg + c * SP + 32 * b + 8 * h) = v0;
            *reinterpret_cast<tg_u32x4_t *>(stg + c * SP + 32 * b + 16 + 8 * h) = v1;
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // (same wave, other lanes: LDS order, no barrier needed)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int CPR = OCH / 8, RPI = 64 / CPR, NIT = 32 / RPI;   // 16-byte chunks per row, rows per instruction
        const int ck = lane % CPR, rl = lane / CPR;
        float bbv[8];
        if constexpr (EPI != 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bbv[e] = bl[8 * ck + e];
        }
        tg_u32x4_t zi[EPI == 2 ? NIT : 1];
        if constexpr (EPI == 2) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                int64_t row = t0 + RPI * i + rl;
                if (row >= a.T) row = a.T - 1;
                zi[i] = *reinterpret_cast<const tg_u32x4_t *>(a.zin + row * a.OUT + n0 + 8 * ck);
            }
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            const int r = RPI * i + rl;
            const int64_t row = t0 + r;
            const tg_u32x4_t v = *reinterpret_cast<const tg_u32x4_t *>(stg + r * SP + 8 * ck);
            if (row >= a.T || tg2_dbg(a, 4)) continue;
            const int64_t off = row * a.OUT + n0 + 8 * ck;
            if constexpr (EPI == 0) {
                *reinterpret_cast<tg_u32x4_t *>(a.y + off) = v;
            } else {
                tg_u32x4_t o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float r2[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const uint32_t wv = EPI == 1 ? v[q] : zi[EPI == 2 ? i : 0][q];
                        const float zf = (e ? __uint_as_float(wv & 0xffff0000u) : __uint_as_float(wv << 16)) + bbv[2 * q + e];
                        float E;
                        const float cdf = 0.5f * (1.0f + tg_erf(zf * kTgInvSqrt2, E));
                        if constexpr (EPI == 1) {
                            r2[e] = zf * cdf;
                        } else {
                            const float dgf = e ? __uint_as_float(v[q] & 0xffff0000u) : __uint_as_float(v[q] << 16);
                            r2[e] = dgf * fmaf(zf, kTgInvSqrt2Pi * E, cdf);
                        }
                    }
                    o[q] = pack_bf16x2(r2[0], r2[1]);
                }
                if constexpr (EPI == 1) {
                    *reinterpret_cast<tg_u32x4_t *>(a.y + off) = v;          // z
                    *reinterpret_cast<tg_u32x4_t *>(a.y2 + off) = o;         // g
                } else {
                    *reinterpret_cast<tg_u32x4_t *>(a.y + off) = o;          // dz
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the next tile's image is written after these reads
    }
}

template <int CON, int OCH, int EPI, int NT>
static int tokens_gemm2_launch_nt(TokGemm2Args a, hipStream_t s) {
    const size_t lds = (size_t)OCH * (CON + 8) * sizeof(uint16_t) + (size_t)OCH * sizeof(float) +
                       (size_t)(NT / 64) * 32 * (OCH + 8) * sizeof(uint16_t);
    auto fn = tokens_gemm2_kernel<CON, OCH, EPI, NT>;
    static LdsOptIn opted;
    if (lds > 64 * 1024 && !lds_opt_in(opted, reinterpret_cast<const void *>(fn), lds)) return XFM_ELAUNCH;
    const int chunks = a.OUT / OCH;
    const int64_t ntiles = (a.T + 31) / 32;
    // one workgroup per CU (the weight chunk takes ~100 KB of LDS): the CUs are split evenly over the chunks
    int wgs = std::max(1, 256 / chunks);
    wgs = (int)std::min<int64_t>(wgs, (ntiles + NT / 64 - 1) / (NT / 64));
    a.wgs_per_chunk = wgs;
    hipLaunchKernelGGL(fn, dim3(chunks * wgs), dim3(NT), lds, s, a);
    return check_launch();
}

template <int CON, int OCH, int EPI>
static int tokens_gemm2_launch(const TokGemm2Args &a, hipStream_t s) {
    // waves per workgroup (tuning hook XFM_GEMM2_NT: 256 or 512 threads).  Four waves: the 32-row x 128-byte pieces the waves
    // have in flight (4 KB each) stay inside the 32 KB L1, so a line of the token rows is fetched from L2 once, not once per
    // k-step that touches it
    static const int nt = [] { const char *e = getenv("XFM_GEMM2_NT"); return e ? atoi(e) : 256; }();
    constexpr size_t lds512 = (size_t)OCH * (CON + 8) * 2 + (size_t)OCH * 4 + (size_t)8 * 32 * (OCH + 8) * 2;
    if constexpr (lds512 <= 160 * 1024) {                        // (eight output images next to the weight chunk)
        if (nt == 512) return tokens_gemm2_launch_nt<CON, OCH, EPI, 512>(a, s);
    }
    return tokens_gemm2_launch_nt<CON, OCH, EPI, 256>(a, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// Third form of the Mlp's wide products (same C entry, same epilogues): a workgroup owns a 128-token x 128-output tile and
// rings 64-wide k-stages of BOTH operands through LDS by LDS-direct loads (global_load_lds_dwordx4: rows arrive as whole
// 128-byte segments -- the second form reads its token rows straight into MFMA operands, 32 different lines per load
// instruction, and serialises rows / MFMAs / erf arithmetic / stores inside each wave).  Two workgroups per CU (64 KB
// each): one's epilogue runs under the other's k-loop.  Lessons of csrc/wgrad_gemm.hip apply: a BARE s_barrier with a
// counted vmcnt (a __syncthreads fence would drain the loads in flight), the next stage's loads issued in parts between
// the MFMA groups, per-lane constant fragment addresses.
//   stage image, operand with k contiguous per row (token rows of x; weight rows when the weight is (OUT, CON)):
//     [128 rows][64 k] = rows of 128 bytes, 16-byte chunk c of row r at position c ^ ((r >> 1) & 7): the ds_read_b128 of an
//     MFMA operand (lane = row, 8 k-values) is conflict-free; the XOR is applied on the global side of the LDS-direct load.
//   weight given as (CON, OUT) (fc2's weight for its data gradient): [64 k][128 n] = rows of 256 bytes in the chunk-XOR
//     image of wgrad_gemm.hip, operands by ds_read_b64_tr_b16.
//   accumulators D[n][t] (weight rows = MFMA rows): a lane ends up with 4 consecutive channels of ONE token per group, the
//     tile leaves through an LDS image [128 tokens][128 + 8] as whole rows (see the second form's epilogue).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kG3Stage = 32768, kG3Half = 16384;       // bytes of one stage (x tile | weight tile)
// contraction widths that are not whole 64-wide stages (96 channels at trunk stage 0): the 16-byte chunks of the last stage that
// lie past the row read this page instead -- zeros in both operands
__device__ uint4 g3_zero_page[4];

__device__ __forceinline__ int g3_off(const int r, const int c) { return 128 * r + 16 * (c ^ ((r >> 1) & 7)); }
__device__ __forceinline__ int g3_tok_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

template <int OFF> __device__ __forceinline__ void g3_read16(tg_u32x4_t &d, const uint32_t ad) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"(OFF) : "memory");
}
typedef __bf16 g3_bf16x4_t __attribute__((ext_vector_type(4)));
template <int OFF> __device__ __forceinline__ void g3_read_tr(g3_bf16x4_t &d, const uint32_t ad) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(ad), "n"(OFF) : "memory");
}

// NW waves per workgroup: 4 (a wave owns 64 outputs x 64 tokens) or 8 (32 outputs x 64 tokens: twice the waves per SIMD behind the
// same 64 KB ring, half the accumulators per wave)
template <int EPI, bool WT, int NW>
__global__ void __launch_bounds__(64 * NW, 2) tokens_gemm3_kernel(const TokGemm2Args a, const int CON, const int ntn, const int ntiles) {
    extern __shared__ __align__(16) uint8_t g3_lds[];      // 2 stages (64 KB) | bias (512 B); the output image overlays the stages
    float *bl = reinterpret_cast<float *>(g3_lds + 2 * kG3Stage);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-major order with the output tile fastest: the n-tiles of one token tile are neighbours on one XCD (its x rows are
    // fetched from HBM once and re-read from that L2)
    // A workgroup walks the tiles slot, slot + nslot, .. of its XCD's range (grid <= 2 workgroups per CU: the stores of one tile
    // drain under the next tile's k-loop instead of holding a finished workgroup's LDS and registers until they are acknowledged).
    const int per_x = (ntiles + 7) >> 3, nslot = (int)(gridDim.x >> 3), slot0 = (int)(blockIdx.x >> 3);
    for (int slot = slot0; slot < per_x; slot += nslot) {
    const int vid = (blockIdx.x & 7) * per_x + slot;
    if (vid >= ntiles) break;
    if (slot != slot0) __syncthreads();                    // the previous tile's image / bias / column-sum reads are done
    const int tn = vid % ntn, tm = vid / ntn;
    const int64_t t0 = (int64_t)tm * 128;
    const int n0 = tn * 128;
    if (tid < 128) bl[tid] = (a.bias && n0 + tid < a.OUT) ? a.bias[n0 + tid] : 0.f;
    constexpr int NI = NW == 4 ? 2 : 1;                    // 32-output blocks of a wave
    constexpr int NP = 16 / NW;                            // 1 KB pieces of a stage (per operand) a wave requests
    const int wm = wave >> 1, wn = wave & 1;               // wave -> outputs 32 NI wm .., tokens 64 wn ..
    const int ow = 32 * NI * wm;
    // ---- global side of the LDS-direct loads (lane constants; k0 is added per stage through the scalar base)
    const uint16_t *gx[NP], *gw[NP];
    int kch[NP];                                           // first k of this lane's chunk inside a stage (k-contiguous images)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = 8 * (NP * wave + i) + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
        kch[i] = 8 * ch;
        int64_t tr = t0 + row;
        if (tr >= a.T) tr = a.T - 1;
        gx[i] = a.x + tr * CON + 8 * ch;
        // (OUT need not be whole tiles: rows / column chunks past it read clamped addresses, and are never stored)
        if constexpr (!WT) {
            gw[i] = a.w + (int64_t)min(n0 + row, a.OUT - 1) * CON + 8 * ch;
        } else {                                           // (CON, OUT): rows 4 (4 wave + i) .. + 3 of the k-major image
            const int kr = 4 * (NP * wave + i) + (lane >> 4);
            const int cw = (lane & 15) ^ (((kr & 3) << 2) | ((kr >> 2) & 3));
            gw[i] = a.w + (int64_t)kr * a.OUT + min(n0 + 8 * cw, a.OUT - 8);
        }
    }
    const int NST = (CON + 63) / 64;
    const bool ragged = (CON & 63) != 0;                   // (uniform)
    const uint16_t *zp = reinterpret_cast<const uint16_t *>(g3_zero_page);
    auto issue_part = [&](const int st, const int i) {
        if (tg2_dbg(a, 16 << 8)) return;                   // (timing switch: no operand loads)
        uint8_t *dst = g3_lds + (st & 1) * kG3Stage + (NP * wave + i) * 1024;
        const bool last = ragged && st == NST - 1;
        const uint16_t *px = (last && 64 * st + kch[i] >= CON) ? zp : gx[i] + 64 * st;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)px,
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        const uint16_t *pw = WT ? gw[i] + (int64_t)64 * st * a.OUT : gw[i] + 64 * st;
        if (last) {
            if constexpr (!WT) {
                if (64 * st + kch[i] >= CON) pw = zp;
            } else {                                       // k-major weight: whole rows past CON
                if (64 * st + 4 * (NP * wave + i) + (lane >> 4) >= CON) pw = zp;
            }
        }
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)pw,
                                         (__attribute__((address_space(3))) void *)(dst + kG3Half), 16, 0, 0);
    };
    // ---- fragment addresses (stage 0; stage s & 1 adds kG3Stage)
    const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)g3_lds;
    const int c = lane & 31, kb = lane >> 5;
    uint32_t ax[2][4], aw[NI][4];                          // [32-row block][k16-step]; WT: aw[i][0 / 1] = lo / hi of step 0
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            ax[j][s4] = base + g3_off(wn * 64 + j * 32 + c, 2 * s4 + kb);
            if constexpr (!WT) {
                if (j < NI) aw[j % NI][s4] = base + kG3Half + g3_off(ow + j * 32 + c, 2 * s4 + kb);
            }
        }
    if constexpr (WT) {
        const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
        const int r0 = 8 * (g >> 1);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int ct = ow + i * 32, c0 = (ct + 16 * (g & 1)) >> 3;
            aw[i][0] = base + kG3Half + g3_tok_off(r0 + q, c0 + (p >> 1)) + 8 * (p & 1);
            aw[i][1] = base + kG3Half + g3_tok_off(r0 + 4 + q, c0 + (p >> 1)) + 8 * (p & 1);
            aw[i][2] = aw[i][3] = 0;
        }
    }
    tg_f32x16_t acc[NI][2];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) issue_part(0, i);
    for (int st = 0; st < NST; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of stage st (nothing else is in flight)
        __builtin_amdgcn_s_barrier();                      // everyone's share landed; everyone is done with stage st - 1
        const bool more = st + 1 < NST;
        const uint32_t so = (st & 1) * kG3Stage;
        tg_u32x4_t xf[2][2], wf[2][NI];                    // [ring][block]
        g3_bf16x4_t wlo[2][NI], whi[2][NI];                // (k-major weight: the two transposed halves of a fragment)
        auto frags = [&](const int ring, auto sc) {
            constexpr int S = decltype(sc)::value;
            g3_read16<0>(xf[ring][0], ax[0][S] + so);
            g3_read16<0>(xf[ring][1], ax[1][S] + so);
            if constexpr (!WT) {
#pragma unroll
                for (int i = 0; i < NI; ++i) g3_read16<0>(wf[ring][i], aw[i][S] + so);
            } else {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    g3_read_tr<4096 * S>(wlo[ring][i], aw[i][0] + so);
                    g3_read_tr<4096 * S>(whi[ring][i], aw[i][1] + so);
                }
            }
        };
        constexpr int NRD = 2 + (WT ? 2 * NI : NI);        // LDS reads of one k16-step
        auto k16 = [&](auto sc) {
            constexpr int S = decltype(sc)::value, r = S & 1;
            if constexpr (S + 1 < 4) frags(r ^ 1, std::integral_constant<int, S + 1>{});
            if constexpr (S < NP) {
                if (more) issue_part(st + 1, S);
            }
            if constexpr (S + 1 < 4) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NRD) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            tg_bf16x8_t wop[NI];
            asm volatile("" : "+v"(xf[r][0]), "+v"(xf[r][1]));
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                if constexpr (!WT) {
                    asm volatile("" : "+v"(wf[r][i]));
                    wop[i] = __builtin_bit_cast(tg_bf16x8_t, wf[r][i]);
                } else {
                    asm volatile("" : "+v"(wlo[r][i]), "+v"(whi[r][i]));
                    wop[i] = __builtin_shufflevector(wlo[r][i], whi[r][i], 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
            if (tg2_dbg(a, 8 << 8)) {                         // (timing switch: no matrix instruction)
#pragma unroll
                for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(wop[i]));
                asm volatile("" ::"v"(xf[r][0]), "v"(xf[r][1]));
                return;
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wop[i], __builtin_bit_cast(tg_bf16x8_t, xf[r][j]), acc[i][j], 0, 0, 0);
        };
        frags(0, std::integral_constant<int, 0>{});
        k16(std::integral_constant<int, 0>{});
        k16(std::integral_constant<int, 1>{});
        k16(std::integral_constant<int, 2>{});
        k16(std::integral_constant<int, 3>{});
    }
    __builtin_amdgcn_s_barrier();                          // the stage buffers become the output image
    if (tg2_dbg(a, 128 << 8)) {                            // (timing switch: no epilogue at all)
        if (acc[0][0][0] + acc[NI - 1][1][3] == 123.456f) a.y[tid] = 1;
        continue;
    }
    // ---- epilogue: D[n][t] -> image [128 tokens][136] (bf16), then whole rows
    constexpr int SP = 136;
    uint16_t *img = reinterpret_cast<uint16_t *>(g3_lds);
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const int h = kb;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint32_t pk[4][2];                             // group g: channels 8 g + 4 h .. + 3 of the 32-block, token c
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 bv = {0.f, 0.f, 0.f, 0.f};
                if constexpr (EPI == 0) bv = *reinterpret_cast<const float4 *>(bl + ow + i * 32 + 8 * g + 4 * h);
                pk[g][0] = pack_bf16x2(acc[i][j][4 * g] + bv.x, acc[i][j][4 * g + 1] + bv.y);
                pk[g][1] = pack_bf16x2(acc[i][j][4 * g + 2] + bv.z, acc[i][j][4 * g + 3] + bv.w);
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2)
#pragma unroll
                for (int q = 0; q < 2; ++q) {              // lower half gets channels +4..7 of group g, upper +0..3 of g+1
                    const u32x2_t r = __builtin_amdgcn_permlane32_swap(pk[g][q], pk[g + 1][q], false, false);
                    pk[g][q] = r[0];
                    pk[g + 1][q] = r[1];
                }
            tg_u32x4_t v0, v1;
            v0[0] = pk[0][0]; v0[1] = pk[0][1]; v0[2] = pk[1][0]; v0[3] = pk[1][1];      // channels 8 h .. + 7
            v1[0] = pk[2][0]; v1[1] = pk[2][1]; v1[2] = pk[3][0]; v1[3] = pk[3][1];      // channels 16 + 8 h .. + 7
            uint16_t *row = img + (wn * 64 + j * 32 + c) * SP + ow + i * 32;
            *reinterpret_cast<tg_u32x4_t *>(row + 8 * h) = v0;
            *reinterpret_cast<tg_u32x4_t *>(row + 16 + 8 * h) = v1;
        }
    __syncthreads();
    const int ck = tid & 15, rl = tid >> 4;                // 16 chunks of 8 channels per row, RP rows per pass
    constexpr int RP = 4 * NW, NPS = 128 / RP;
    float bbv[8];
    if constexpr (EPI != 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bbv[e] = bl[8 * ck + e];
    }
    tg_u32x4_t zi[EPI == 2 ? NPS : 1];
    const bool col_ok = n0 + 8 * ck < a.OUT;               // (OUT % 8 == 0)
    if constexpr (EPI == 2) {
#pragma unroll
        for (int i = 0; i < NPS; ++i) {
            int64_t row = t0 + RP * i + rl;
            if (row >= a.T) row = a.T - 1;
            zi[i] = *reinterpret_cast<const tg_u32x4_t *>(a.zin + row * a.OUT + min(n0 + 8 * ck, a.OUT - 8));
        }
    }
    float csum[8];                                         // EPI 2: column sums of this thread's rows (dz as stored: bf16)
#pragma unroll
    for (int e = 0; e < 8; ++e) csum[e] = 0.f;
#pragma unroll
    for (int i = 0; i < NPS; ++i) {
        const int r = RP * i + rl;
        const int64_t row = t0 + r;
        const tg_u32x4_t v = *reinterpret_cast<const tg_u32x4_t *>(img + r * SP + 8 * ck);
        if (row >= a.T || !col_ok) continue;
        const int64_t off = row * a.OUT + n0 + 8 * ck;
        if (tg2_dbg(a, 64 << 8)) {                         // (timing switch: stores without the GELU arithmetic)
            *reinterpret_cast<tg_u32x4_t *>(a.y + off) = v;
            if constexpr (EPI == 1) *reinterpret_cast<tg_u32x4_t *>(a.y2 + off) = v;
            continue;
        }
        if constexpr (EPI == 0) {
            *reinterpret_cast<tg_u32x4_t *>(a.y + off) = v;
        } else {
            tg_u32x4_t o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float r2[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const uint32_t wv = EPI == 1 ? v[q] : zi[EPI == 2 ? i : 0][q];
                    const float zf = (e ? __uint_as_float(wv & 0xffff0000u) : __uint_as_float(wv << 16)) + bbv[2 * q + e];
                    float E;
                    const float cdf = 0.5f * (1.0f + tg_erf(zf * kTgInvSqrt2, E));
                    if constexpr (EPI == 1) {
                        r2[e] = zf * cdf;
                    } else {
                        const float dgf = e ? __uint_as_float(v[q] & 0xffff0000u) : __uint_as_float(v[q] << 16);
                        r2[e] = dgf * fmaf(zf, kTgInvSqrt2Pi * E, cdf);
                    }
                }
                o[q] = pack_bf16x2(r2[0], r2[1]);
            }
            if (tg2_dbg(a, 4 << 8)) {                          // (timing switch: the arithmetic without the stores)
                if (o[0] == 0x12345678u) a.y[off] = 1;
                continue;
            }
            if constexpr (EPI == 1) {
                *reinterpret_cast<tg_u32x4_t *>(a.y + off) = v;          // z
                *reinterpret_cast<tg_u32x4_t *>(a.y2 + off) = o;         // g
            } else {
                *reinterpret_cast<tg_u32x4_t *>(a.y + off) = o;          // dz
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    csum[2 * q] += __uint_as_float(o[q] << 16);
                    csum[2 * q + 1] += __uint_as_float(o[q] & 0xffff0000u);
                }
            }
        }
    }
    if constexpr (EPI == 2) {
        // the bias gradient's partial row of this tile: the 16 row-lanes of a column chunk meet in LDS (past the image)
        if (a.colpart) {
            float *red = reinterpret_cast<float *>(g3_lds + 40960);          // [RP][128]
            *reinterpret_cast<float4 *>(red + rl * 128 + 8 * ck) = make_float4(csum[0], csum[1], csum[2], csum[3]);
            *reinterpret_cast<float4 *>(red + rl * 128 + 8 * ck + 4) = make_float4(csum[4], csum[5], csum[6], csum[7]);
            __syncthreads();
            if (tid < 128 && n0 + tid < a.OUT) {
                float s = 0.f;
#pragma unroll
                for (int q = 0; q < RP; ++q) s += red[q * 128 + tid];
                a.colpart[(int64_t)tm * a.OUT + n0 + tid] = s;
            }
        }
    }
    }   // tiles of this workgroup
}

template <int EPI, bool WT, int NW>
static int tokens_gemm3_launch_nw(const TokGemm2Args &a, int con, hipStream_t s) {
    const size_t lds = 2 * kG3Stage + 512;
    auto fn = tokens_gemm3_kernel<EPI, WT, NW>;
    static LdsOptIn opted;
    if (!lds_opt_in(opted, reinterpret_cast<const void *>(fn), lds)) return XFM_ELAUNCH;
    const int ntn = (a.OUT + 127) / 128;
    const int64_t ntm = (a.T + 127) / 128;
    const int64_t ntiles = ntm * ntn;
    if (ntiles > (1 << 30)) return XFM_ELIMIT;
    // (measured, 12544 x 384 -> 1536 with the GELU epilogue: 37.6 us one tile per workgroup, 36.0 walking, 33.6 walking with eight
    //  waves; with three k-stages per tile -- 50176 x 192 -> 768 -- walking costs 2 us: there the epilogues are most of a tile and
    //  a fresh workgroup's k-loop overlaps them better)
    static const int env_persist = [] { const char *e = getenv("XFM_G3_PERSIST"); return e ? atoi(e) : -1; }();
    const bool persist = env_persist >= 0 ? env_persist != 0 : con >= 384;
    unsigned grid = (unsigned)((ntiles + 7) / 8 * 8);
    if (persist && grid > 512u) grid = 512u;                        // two workgroups per CU, each walks its share of the tiles
    hipLaunchKernelGGL(fn, dim3(grid), dim3(64 * NW), lds, s, a, con, ntn, (int)ntiles);
    return check_launch();
}

template <int EPI, bool WT>
static int tokens_gemm3_launch(const TokGemm2Args &a, int con, hipStream_t s) {
    // eight waves (32 outputs x 64 tokens each, ~90 registers: four waves per SIMD behind the same two 64 KB rings) pay where the
    // output side is wide -- 5 ... 10 % at out >= 768 --, not for the narrow products (768 -> 384: 16.7 vs 17.5 us)
    static const int env_nw = [] { const char *e = getenv("XFM_G3_NW"); return e ? atoi(e) : 0; }();
    const int nw = env_nw ? env_nw : (a.OUT >= 768 ? 8 : 4);
    return nw == 8 ? tokens_gemm3_launch_nw<EPI, WT, 8>(a, con, s) : tokens_gemm3_launch_nw<EPI, WT, 4>(a, con, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// The tiled form for the LAYOUT-CHANGING projections of an SS2D block (in_proj: tokens -> planes, out_proj: planes -> tokens,
// and their data gradients; reference models/fusion_vmamba.py:1190-1206 `Linear2d` / permutes): y[b] = W x[b] with one side
// token-major (B, L, C) and the other plane-major (B, C, L).  Tiles are cut PER SAMPLE (a plane row of L positions is
// contiguous only inside its sample): sample b, positions 128 lt .. + 127 (the last tile of a sample is ragged: 196 = 128 +
// 68), 128 output channels.
//   PIN (planes in, tokens out): the x tile is staged k-major -- [64 k][128 positions], rows of 256 bytes in the chunk-XOR
//     image -- and read with ds_read_b64_tr_b16 (column = position); the 16-byte chunks that start past L read a zero page
//     (a chunk that straddles L brings the next channel row's first positions into columns that are never stored).
//   !PIN (tokens in, planes out): x as in the token -> token kernel; the accumulators D[n][t] (lane = position) leave by
//     2-byte stores straight from registers: for one channel 32 lanes = 64 contiguous bytes of its plane row.
// ---------------------------------------------------------------------------------------------------------------------
struct ProjTiledArgs {
    const uint16_t *x, *w;
    const float *bias;
    uint16_t *y;
    int B, L, CON, OUT, ltiles, ntn, ntiles;
    int accumulate;         // planes out: y += W x (the existing bf16 values are widened, added in fp32 and rounded once)
    int img;                // planes out, not accumulating: leave through the LDS image (XFM_PROJ_IMG=0: A/B switch)
};

template <bool PIN, bool WT, int NW>
__global__ void __launch_bounds__(64 * NW, 2) proj_tiled_kernel(const ProjTiledArgs a) {
    extern __shared__ __align__(16) uint8_t g3_lds[];      // 2 stages (64 KB) | bias (512 B); the token-major output image overlays
    float *bl = reinterpret_cast<float *>(g3_lds + 2 * kG3Stage);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int vid = (blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (vid >= a.ntiles) return;
    const int tn = vid % a.ntn, tm = vid / a.ntn;
    const int b = tm / a.ltiles, l0 = (tm - b * a.ltiles) * 128;
    const int n0 = tn * 128, L = a.L, CON = a.CON;
    if (tid < 128) bl[tid] = a.bias ? a.bias[n0 + tid] : 0.f;
    constexpr int NI = NW == 4 ? 2 : 1;                    // 32-output blocks of a wave (NW: as tokens_gemm3_kernel)
    constexpr int NP = 16 / NW;                            // 1 KB pieces of a stage (per operand) a wave requests
    const int wm = wave >> 1, wn = wave & 1;               // wave -> outputs 32 NI wm .., positions 64 wn ..
    const int ow = 32 * NI * wm;
    const uint16_t *zp = reinterpret_cast<const uint16_t *>(g3_zero_page);
    // ---- global side of the LDS-direct loads
    const uint16_t *gx[NP], *gw[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        if constexpr (!PIN) {                              // x tile [128 positions][64 k]
            const int row = 8 * (NP * wave + i) + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
            const int l = min(l0 + row, L - 1);
            gx[i] = a.x + ((int64_t)b * L + l) * CON + 8 * ch;
        } else {                                           // x tile [64 k][128 positions]: rows 4 (4 wave + i) .. + 3
            const int kr = 4 * (NP * wave + i) + (lane >> 4);
            const int cw = (lane & 15) ^ (((kr & 3) << 2) | ((kr >> 2) & 3));
            // a chunk that straddles L is read 8 - Lrem % 8 positions EARLIER (inside the row: nothing is read past the tensor);
            // its columns then hold positions Lrem - 8 .. Lrem - 1, which the epilogue's row map undoes
            const int lrem = L - l0, lc = 8 * cw;
            const int ls = lc + 8 <= lrem ? lc : lrem - 8;
            gx[i] = lc < lrem ? a.x + ((int64_t)b * CON + kr) * L + l0 + ls : nullptr;
        }
        if constexpr (!WT) {
            const int row = 8 * (NP * wave + i) + (lane >> 3), ch = (lane & 7) ^ ((row >> 1) & 7);
            gw[i] = a.w + (int64_t)(n0 + row) * CON + 8 * ch;
        } else {
            const int kr = 4 * (NP * wave + i) + (lane >> 4);
            const int cw = (lane & 15) ^ (((kr & 3) << 2) | ((kr >> 2) & 3));
            gw[i] = a.w + (int64_t)kr * a.OUT + n0 + 8 * cw;
        }
    }
    const int NST = CON / 64;
    auto issue_part = [&](const int st, const int i) {
        uint8_t *dst = g3_lds + (st & 1) * kG3Stage + (NP * wave + i) * 1024;
        const uint16_t *px;
        if constexpr (!PIN) px = gx[i] + 64 * st;
        else px = gx[i] ? gx[i] + (int64_t)64 * st * L : zp;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)px,
                                         (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        const uint16_t *pw = WT ? gw[i] + (int64_t)64 * st * a.OUT : gw[i] + 64 * st;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)pw,
                                         (__attribute__((address_space(3))) void *)(dst + kG3Half), 16, 0, 0);
    };
    // ---- fragment addresses (stage buffer 0)
    const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint8_t *)g3_lds;
    const int c = lane & 31, kb = lane >> 5;
    uint32_t ax[2][4], aw[NI][4];                          // k-contiguous images: [32-row block][k16-step]; k-major: [block][lo, hi]
    {
        const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
        const int r0 = 8 * (g >> 1);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if constexpr (!PIN) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) ax[j][s4] = base + g3_off(wn * 64 + j * 32 + c, 2 * s4 + kb);
            } else {
                const int ct = wn * 64 + j * 32, c0 = (ct + 16 * (g & 1)) >> 3;
                ax[j][0] = base + g3_tok_off(r0 + q, c0 + (p >> 1)) + 8 * (p & 1);
                ax[j][1] = base + g3_tok_off(r0 + 4 + q, c0 + (p >> 1)) + 8 * (p & 1);
                ax[j][2] = ax[j][3] = 0;
            }
            if (j < NI) {
                const int jw = j % NI;
                if constexpr (!WT) {
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) aw[jw][s4] = base + kG3Half + g3_off(ow + j * 32 + c, 2 * s4 + kb);
                } else {
                    const int ct = ow + j * 32, c0 = (ct + 16 * (g & 1)) >> 3;
                    aw[jw][0] = base + kG3Half + g3_tok_off(r0 + q, c0 + (p >> 1)) + 8 * (p & 1);
                    aw[jw][1] = base + kG3Half + g3_tok_off(r0 + 4 + q, c0 + (p >> 1)) + 8 * (p & 1);
                    aw[jw][2] = aw[jw][3] = 0;
                }
            }
        }
    }
    tg_f32x16_t acc[NI][2];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
#pragma unroll
    for (int i = 0; i < NP; ++i) issue_part(0, i);
    for (int st = 0; st < NST; ++st) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool more = st + 1 < NST;
        const uint32_t so = (st & 1) * kG3Stage;
        tg_u32x4_t xf[2][2], wf[2][NI];
        g3_bf16x4_t xlo[2][2], xhi[2][2], wlo[2][NI], whi[2][NI];
        auto frags = [&](const int ring, auto sc) {
            constexpr int S = decltype(sc)::value;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (!PIN) g3_read16<0>(xf[ring][j], ax[j][S] + so);
                else {
                    g3_read_tr<4096 * S>(xlo[ring][j], ax[j][0] + so);
                    g3_read_tr<4096 * S>(xhi[ring][j], ax[j][1] + so);
                }
                if (j < NI) {
                    if constexpr (!WT) g3_read16<0>(wf[ring][j % NI], aw[j % NI][S] + so);
                    else {
                        g3_read_tr<4096 * S>(wlo[ring][j % NI], aw[j % NI][0] + so);
                        g3_read_tr<4096 * S>(whi[ring][j % NI], aw[j % NI][1] + so);
                    }
                }
            }
        };
        constexpr int NRD = (PIN ? 4 : 2) + (WT ? 2 * NI : NI);  // LDS reads of one k16-step
        auto k16 = [&](auto sc) {
            constexpr int S = decltype(sc)::value, r = S & 1;
            if constexpr (S + 1 < 4) frags(r ^ 1, std::integral_constant<int, S + 1>{});
            if constexpr (S < NP) {
                if (more) issue_part(st + 1, S);
            }
            if constexpr (S + 1 < 4) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(NRD) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            tg_bf16x8_t wop[NI], xop[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (!PIN) {
                    asm volatile("" : "+v"(xf[r][j]));
                    xop[j] = __builtin_bit_cast(tg_bf16x8_t, xf[r][j]);
                } else {
                    asm volatile("" : "+v"(xlo[r][j]), "+v"(xhi[r][j]));
                    xop[j] = __builtin_shufflevector(xlo[r][j], xhi[r][j], 0, 1, 2, 3, 4, 5, 6, 7);
                }
                if (j < NI) {
                    if constexpr (!WT) {
                        asm volatile("" : "+v"(wf[r][j % NI]));
                        wop[j % NI] = __builtin_bit_cast(tg_bf16x8_t, wf[r][j % NI]);
                    } else {
                        asm volatile("" : "+v"(wlo[r][j % NI]), "+v"(whi[r][j % NI]));
                        wop[j % NI] = __builtin_shufflevector(wlo[r][j % NI], whi[r][j % NI], 0, 1, 2, 3, 4, 5, 6, 7);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wop[i], xop[j], acc[i][j], 0, 0, 0);
        };
        frags(0, std::integral_constant<int, 0>{});
        k16(std::integral_constant<int, 0>{});
        k16(std::integral_constant<int, 1>{});
        k16(std::integral_constant<int, 2>{});
        k16(std::integral_constant<int, 3>{});
    }
    const int h = kb;
    if constexpr (!PIN) {
        // ---- planes out: D[n][position]: for one register 32 lanes hold 32 consecutive positions of one channel's plane row
        __syncthreads();                                   // (bias; everyone is done with the stage buffers)
        if (!a.accumulate && a.img) {
            // through an image [128 channels][128 positions + 8] in the (now free) stage buffers: 2-byte LDS writes (consecutive
            // lanes = consecutive addresses), then 8-byte stores -- four positions of a plane row per lane, a row's 32 pieces on
            // consecutive lanes -- instead of 64 two-byte global stores per thread
            constexpr int SP2 = 136;
            uint16_t *img = reinterpret_cast<uint16_t *>(g3_lds);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int pl = wn * 64 + j * 32 + c;
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int nl = ow + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                        img[nl * SP2 + pl] = (uint16_t)(pack_bf16x2(acc[i][j][v] + bl[nl], 0.f) & 0xffffu);
                    }
                }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 64 / NW; ++it) {
                const int item = tid + 64 * NW * it, row = item >> 5, ch = item & 31;
                if (l0 + 4 * ch < L)                       // (L % 4 == 0)
                    *reinterpret_cast<uint2 *>(a.y + ((int64_t)b * a.OUT + n0 + row) * L + l0 + 4 * ch) =
                        *reinterpret_cast<const uint2 *>(img + row * SP2 + 4 * ch);
            }
            return;
        }
        if (a.accumulate && a.img) {
            // y += W x: the fp32 tile through an image [128 channels][128 positions] (64 KB = the stage ring), then per lane four
            // positions of a plane row: the existing bf16 values widened, added in fp32, rounded once, one 8-byte store
            float *imgf = reinterpret_cast<float *>(g3_lds);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int pl = wn * 64 + j * 32 + c;
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int nl = ow + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                        imgf[nl * 128 + (pl ^ ((nl & 7) << 2))] = acc[i][j][v] + bl[nl];      // (row-dependent column swizzle: see the reads)
                    }
                }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 64 / NW; ++it) {
                const int item = tid + 64 * NW * it, row = item >> 5, ch = item & 31;
                if (l0 + 4 * ch < L) {
                    uint2 *dst = reinterpret_cast<uint2 *>(a.y + ((int64_t)b * a.OUT + n0 + row) * L + l0 + 4 * ch);
                    const uint2 old = *dst;
                    const float4 t = *reinterpret_cast<const float4 *>(imgf + row * 128 + ((4 * ch) ^ ((row & 7) << 2)));
                    const float o0 = t.x + __uint_as_float(old.x << 16), o1 = t.y + __uint_as_float(old.x & 0xffff0000u);
                    const float o2 = t.z + __uint_as_float(old.y << 16), o3 = t.w + __uint_as_float(old.y & 0xffff0000u);
                    *dst = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int l = l0 + wn * 64 + j * 32 + c;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int nl = ow + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                    if (l < L) {
                        uint16_t *dst = a.y + ((int64_t)b * a.OUT + n0 + nl) * L + l;
                        float val = acc[i][j][v] + bl[nl];
                        if (a.accumulate) val += __uint_as_float((uint32_t)*dst << 16);
                        *dst = (uint16_t)(pack_bf16x2(val, 0.f) & 0xffffu);
                    }
                }
            }
    } else {
        // ---- tokens out: through the image [128 positions][136], whole rows (as the token -> token kernel)
        __builtin_amdgcn_s_barrier();
        constexpr int SP = 136;
        uint16_t *img = reinterpret_cast<uint16_t *>(g3_lds);
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint32_t pk[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4 *>(bl + ow + i * 32 + 8 * g + 4 * h);
                    pk[g][0] = pack_bf16x2(acc[i][j][4 * g] + bv.x, acc[i][j][4 * g + 1] + bv.y);
                    pk[g][1] = pack_bf16x2(acc[i][j][4 * g + 2] + bv.z, acc[i][j][4 * g + 3] + bv.w);
                }
#pragma unroll
                for (int g = 0; g < 4; g += 2)
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const u32x2_t r = __builtin_amdgcn_permlane32_swap(pk[g][q], pk[g + 1][q], false, false);
                        pk[g][q] = r[0];
                        pk[g + 1][q] = r[1];
                    }
                tg_u32x4_t v0, v1;
                v0[0] = pk[0][0]; v0[1] = pk[0][1]; v0[2] = pk[1][0]; v0[3] = pk[1][1];
                v1[0] = pk[2][0]; v1[1] = pk[2][1]; v1[2] = pk[3][0]; v1[3] = pk[3][1];
                uint16_t *row = img + (wn * 64 + j * 32 + c) * SP + ow + i * 32;
                *reinterpret_cast<tg_u32x4_t *>(row + 8 * h) = v0;
                *reinterpret_cast<tg_u32x4_t *>(row + 16 + 8 * h) = v1;
            }
        __syncthreads();
        const int ck = tid & 15, rl = tid >> 4;
        const int lrem = L - l0, tail0 = lrem & ~7, shift = (8 - (lrem & 7)) & 7;     // (see the straddling chunk above)
#pragma unroll
        for (int i = 0; i < 32 / NW; ++i) {
            const int r = 4 * NW * i + rl;
            const int lr = (shift && r >= tail0) ? r - shift : r;               // position this image row holds
            if (lr < 0 || lr >= lrem || r >= tail0 + 8) continue;
            const tg_u32x4_t v = *reinterpret_cast<const tg_u32x4_t *>(img + r * SP + 8 * ck);
            *reinterpret_cast<tg_u32x4_t *>(a.y + ((int64_t)b * L + l0 + lr) * a.OUT + n0 + 8 * ck) = v;
        }
    }
}

template <bool PIN, bool WT, int NW> static int proj_tiled_launch_nw(ProjTiledArgs a, hipStream_t s) {
    const size_t lds = 2 * kG3Stage + 512;
    auto fn = proj_tiled_kernel<PIN, WT, NW>;
    static LdsOptIn opted;
    if (!lds_opt_in(opted, reinterpret_cast<const void *>(fn), lds)) return XFM_ELAUNCH;
    a.ltiles = (a.L + 127) / 128;
    a.ntn = a.OUT / 128;
    a.ntiles = a.B * a.ltiles * a.ntn;
    static const int img_on = [] { const char *e = getenv("XFM_PROJ_IMG"); return (!e || atoi(e) != 0) ? 1 : 0; }();
    a.img = img_on;
    hipLaunchKernelGGL(fn, dim3((unsigned)((a.ntiles + 7) / 8 * 8)), dim3(64 * NW), lds, s, a);
    return check_launch();
}

template <bool PIN, bool WT> static int proj_tiled_launch(ProjTiledArgs a, hipStream_t s) {
    // eight waves (as tokens_gemm3_kernel): the 14 x 14 projections are 384 tiles -- one or two workgroups per CU --, so four-wave
    // workgroups leave most SIMDs with a single wave; step 13.29 -> 13.25 ms same-box
    static const int nw = [] { const char *e = getenv("XFM_PROJ_NW"); return e ? atoi(e) : 8; }();
    return nw == 8 ? proj_tiled_launch_nw<PIN, WT, 8>(a, s) : proj_tiled_launch_nw<PIN, WT, 4>(a, s);
}

int tokens_gemm2_form() {
    static const int form = [] { const char *e = getenv("XFM_GEMM2_FORM"); return e ? atoi(e) : 3; }();
    return form;
}

static int tokens_gemm3(const TokGemm2Args &a, int con, int epi, hipStream_t s) {
    const bool wt = (a.wt & 1) != 0;
    if (epi == 0) return wt ? tokens_gemm3_launch<0, true>(a, con, s) : tokens_gemm3_launch<0, false>(a, con, s);
    if (epi == 1) return wt ? tokens_gemm3_launch<1, true>(a, con, s) : tokens_gemm3_launch<1, false>(a, con, s);
    return wt ? tokens_gemm3_launch<2, true>(a, con, s) : tokens_gemm3_launch<2, false>(a, con, s);
}

// y (T, out) = x (T, con) . W^T through the tiled form, no bias: W as (out, con), or -- wt -- as (con, out).  con, out % 8 == 0,
// 16-byte aligned operands.  (xfm namespace, not part of the C ABI: csrc/conv_tok.hip)
int tokens_gemm3_plain(const void *x, const void *w, void *y, long long T, int con, int out, bool wt, hipStream_t s) {
    if (con % 8 || out % 8 || con < 8 || out < 8 || T <= 0 || ((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15))
        return XFM_ELIMIT;
    TokGemm2Args a{};
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(w);
    a.y = static_cast<uint16_t *>(y);
    a.T = T;
    a.OUT = out;
    a.wt = wt ? 1 : 0;
    a.wgs_per_chunk = 1;
    return tokens_gemm3(a, con, 0, s);
}

template <int CON, int OCH>
static int tokens_gemm2_epi(const TokGemm2Args &a, int epi, hipStream_t s) {
    if (epi == 0) return tokens_gemm2_launch<CON, OCH, 0>(a, s);
    if (epi == 1) return tokens_gemm2_launch<CON, OCH, 1>(a, s);
    return tokens_gemm2_launch<CON, OCH, 2>(a, s);
}

template <int CON, int OUT, int OB>
static int tokens_gemm_launch(const TokGemmArgs &a, hipStream_t s) {
    const size_t lds = (size_t)OUT * (CON + 8) * sizeof(uint16_t) + (size_t)OUT * sizeof(float);
    auto fn = tokens_gemm_kernel<CON, OUT, OB>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int64_t ntiles = (a.T + 31) / 32;
    int grid = (int)std::min<int64_t>((ntiles + 7) / 8, 512);       // persistent: up to two workgroups per CU
    hipLaunchKernelGGL(fn, dim3(grid), dim3(512), lds, s, a);
    return check_launch();
}

}  // namespace xfm

extern "C" {

int xfm_tokens_gemm_supported(int con, int out) {
    return ((con == 96 && out == 384) || (con == 384 && out == 96) || (con == 96 && out == 96) || (con == 96 && out == 192)) ? 1 : 0;
}

int xfm_tokens_gemm(const void *x, const void *weight_bf16, const float *bias, void *y, long long T, int con, int out,
                    int weight_transposed, void *stream) {
    using namespace xfm;
    if (!x || !weight_bf16 || !y || T <= 0) return XFM_EINVAL;
    TokGemmArgs a{};
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(weight_bf16);
    a.bias = bias;
    a.y = static_cast<uint16_t *>(y);
    a.T = T;
    a.wt = weight_transposed;
    a.L = 0;
    hipStream_t s = (hipStream_t)stream;
    if (con == 96 && out == 384) return tokens_gemm_launch<96, 384, 192>(a, s);
    if (con == 384 && out == 96) return tokens_gemm_launch<384, 96, 96>(a, s);
    if (con == 96 && out == 96) return tokens_gemm_launch<96, 96, 96>(a, s);
    if (con == 96 && out == 192) return tokens_gemm_launch<96, 192, 96>(a, s);
    return XFM_ELIMIT;
}

/* Chunked form (weight chunk resident in LDS, GELU epilogues): con -> out = 4 con at the later trunk stages. */
int xfm_tokens_gemm2_supported(int con, int out) {
    // the chunked form: con -> out = 4 con at the later trunk stages; the tiled form (LDS-direct, 128 x 128 tiles): any
    // con % 64 == 0, out % 128 == 0
    if ((con == 192 && out == 768) || (con == 384 && out == 1536) || (con == 768 && out == 3072)) return 1;
    return (xfm::tokens_gemm2_form() == 3 && con >= 64 && con % 32 == 0 && out >= 128 && out % 128 == 0) ? 1 : 0;
}

int xfm_tokens_gemm2(const void *x, const void *weight_bf16, const float *bias, void *y, void *y2, const void *zin, long long T,
                     int con, int out, int weight_transposed, int epilogue, void *stream) {
    using namespace xfm;
    if (!x || !weight_bf16 || !y || T <= 0 || epilogue < 0 || epilogue > 2) return XFM_EINVAL;
    if ((epilogue == 1 && !y2) || (epilogue == 2 && !zin)) return XFM_EINVAL;
    if (!xfm_tokens_gemm2_supported(con, out)) return XFM_ELIMIT;
    TokGemm2Args a{};
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(weight_bf16);
    a.bias = bias;
    a.y = static_cast<uint16_t *>(y);
    a.y2 = static_cast<uint16_t *>(y2);
    a.zin = static_cast<const uint16_t *>(zin);
    a.T = T;
    a.OUT = out;
    a.wt = weight_transposed ? 1 : 0;
#ifdef XFM_GEMM2_TIMING
    static const int dbg = [] { const char *e = getenv("XFM_GEMM2_DBG"); return e ? atoi(e) : 0; }();   // timing switches: 2, 4
    a.wt |= dbg & 62;
    static const int dbg3 = [] { const char *e = getenv("XFM_G3_DBG"); return e ? atoi(e) : 0; }();   // tiled form: bits << 8
    a.wt |= dbg3 << 8;
#else
    constexpr int dbg = 0;                                        // (the timing switches exist in -DXFM_GEMM2_TIMING builds only)
#endif
    a.wgs_per_chunk = 1;
    hipStream_t s = (hipStream_t)stream;
    const int form = tokens_gemm2_form();
    // (192 -> 768 forward: three k-stages per tile, the chunked form is 2 us faster there -- 59.0 vs 60.8)
    if (form == 3 && !(con == 192 && out == 768 && epilogue == 1) && con % 32 == 0 && out % 128 == 0 && !(dbg & 62) && ((uintptr_t)x & 15) == 0 && ((uintptr_t)weight_bf16 & 15) == 0)
        return tokens_gemm3(a, con, epilogue, s);
    if (con == 192 && out == 768) return tokens_gemm2_epi<192, 128>(a, epilogue, s);
    if (con == 384 && out == 1536) return tokens_gemm2_epi<384, 128>(a, epilogue, s);
    if (con == 768 && out == 3072) return tokens_gemm2_epi<768, 64>(a, epilogue, s);
    return XFM_ELIMIT;
}

/* xfm_tokens_gemm2 with epilogue 2 (dz = (x W) gelu'(z + b)) that ALSO leaves the column sums of dz -- fc1's bias gradient --
 * as partial rows: colpart (xfm_tokens_gemm2_parts_blocks(T, con, out), out) fp32, one row per 128-token tile, every
 * element written (fold them with xfm_partial_sums_multi or a sum over rows).  0 blocks: not available for this shape. */
int xfm_tokens_gemm2_parts_blocks(long long T, int con, int out) {
    if (xfm::tokens_gemm2_form() != 3 || con % 32 != 0 || con < 64 || out % 128 != 0 || T <= 0) return 0;
    return (int)((T + 127) / 128);
}

int xfm_tokens_gemm2_parts(const void *x, const void *weight_bf16, const float *bias, void *dz, const void *zin, float *colpart,
                           long long T, int con, int out, int weight_transposed, void *stream) {
    using namespace xfm;
    if (!x || !weight_bf16 || !dz || !zin || !colpart || T <= 0) return XFM_EINVAL;
    if (!xfm_tokens_gemm2_parts_blocks(T, con, out) || ((uintptr_t)x & 15) || ((uintptr_t)weight_bf16 & 15)) return XFM_ELIMIT;
    TokGemm2Args a{};
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(weight_bf16);
    a.bias = bias;
    a.y = static_cast<uint16_t *>(dz);
    a.zin = static_cast<const uint16_t *>(zin);
    a.T = T;
    a.OUT = out;
    a.wt = weight_transposed ? 1 : 0;
    a.colpart = colpart;
    return tokens_gemm3(a, con, 2, (hipStream_t)stream);
}

static int proj_tiled_ok(int con, int out, int L) {
    static const bool on = [] { const char *e = getenv("XFM_PROJ_TILED"); return !e || atoi(e) != 0; }();
    return on && con >= 64 && con % 64 == 0 && out % 128 == 0 && L >= 64 && L % 4 == 0;
}

int xfm_proj_gemm_supported(int con, int out, int L) {
    if (((con == 96 && out == 96) || (con == 192 && out == 192)) && L > 0 && L % 8 == 0) return 1;
    return proj_tiled_ok(con, out, L);
}

int xfm_proj_gemm(const void *x, const void *weight_bf16, const float *bias, void *y, int B, int L, int con, int out,
                  int in_planes, int weight_transposed, void *stream) {
    using namespace xfm;
    if (!x || !weight_bf16 || !y || B <= 0) return XFM_EINVAL;
    if (!xfm_proj_gemm_supported(con, out, L) || ((int64_t)B * L) % 32 != 0) return XFM_ELIMIT;
    hipStream_t s = (hipStream_t)stream;
    if (!((con == 96 && out == 96) || (con == 192 && out == 192))) {      // the tiled form (csrc: proj_tiled_kernel)
        if (((uintptr_t)x | (uintptr_t)weight_bf16 | (uintptr_t)y) & 15) return XFM_EINVAL;
        ProjTiledArgs t{};
        t.x = static_cast<const uint16_t *>(x); t.w = static_cast<const uint16_t *>(weight_bf16); t.bias = bias;
        t.y = static_cast<uint16_t *>(y); t.B = B; t.L = L; t.CON = con; t.OUT = out;
        if (in_planes) return weight_transposed ? proj_tiled_launch<true, true>(t, s) : proj_tiled_launch<true, false>(t, s);
        return weight_transposed ? proj_tiled_launch<false, true>(t, s) : proj_tiled_launch<false, false>(t, s);
    }
    TokGemmArgs a{};
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(weight_bf16);
    a.bias = bias;
    a.y = static_cast<uint16_t *>(y);
    a.T = (int64_t)B * L;
    a.wt = weight_transposed;
    a.L = L;
    if (con == 96) return in_planes ? proj_gemm_launch<96, 96, true>(a, s) : proj_gemm_launch<96, 96, false>(a, s);
    return in_planes ? proj_gemm_launch<192, 192, true>(a, s) : proj_gemm_launch<192, 192, false>(a, s);
}

/* tokens -> planes with accumulation: y (B, out, L) += W . x (B, L, con) -- the x_proj data gradient of a channel-lane SS2D
 * block, dx += Wx^T . d x_dbl^T (reference models/fusion_vmamba.py:1150-1152 through autograd), on the tiled form only
 * (xfm_proj_gemm_supported(con, out, L) with con % 64 == 0, out % 128 == 0). */
int xfm_proj_gemm_accumulate(const void *x, const void *weight_bf16, void *y, int B, int L, int con, int out,
                             int weight_transposed, void *stream) {
    using namespace xfm;
    if (!x || !weight_bf16 || !y || B <= 0) return XFM_EINVAL;
    if (!proj_tiled_ok(con, out, L)) return XFM_ELIMIT;
    if (((uintptr_t)x | (uintptr_t)weight_bf16) & 15) return XFM_EINVAL;
    ProjTiledArgs t{};
    t.x = static_cast<const uint16_t *>(x); t.w = static_cast<const uint16_t *>(weight_bf16);
    t.y = static_cast<uint16_t *>(y); t.B = B; t.L = L; t.CON = con; t.OUT = out; t.accumulate = 1;
    hipStream_t s = (hipStream_t)stream;
    return weight_transposed ? proj_tiled_launch<false, true>(t, s) : proj_tiled_launch<false, false>(t, s);
}

/* planes -> planes: y (B, out, L) (+)= W . x (B, con, L); (con, out) = (96, 32): x_proj of the 56x56 stage on the natural
 * map; (32, 96) with accumulate != 0: its backward, dx += W^T . d x_dbl. */
int xfm_planes_gemm_supported(int con, int out, int L) {
    // x_proj of XFMamba-T's 56 x 56 stage (96 <-> 4 x 8 rows) and of its 28 x 28 stage (192 <-> 4 x 14 = 56 rows: padded to 64
    // inside the kernel)
    return (((con == 96 && out == 32) || (con == 32 && out == 96) || (con == 192 && out == 56) || (con == 56 && out == 192)) &&
            L > 0 && L % 8 == 0) ? 1 : 0;
}

int xfm_planes_gemm(const void *x, const void *weight_bf16, const float *bias, void *y, int B, int L, int con, int out,
                    int weight_transposed, int accumulate, void *stream) {
    using namespace xfm;
    if (!x || !weight_bf16 || !y || B <= 0) return XFM_EINVAL;
    if (!xfm_planes_gemm_supported(con, out, L) || ((int64_t)B * L) % 32 != 0) return XFM_ELIMIT;
    TokGemmArgs a{};
    a.x = static_cast<const uint16_t *>(x);
    a.w = static_cast<const uint16_t *>(weight_bf16);
    a.bias = bias;
    a.y = static_cast<uint16_t *>(y);
    a.T = (int64_t)B * L;
    a.wt = weight_transposed;
    a.L = L;
    a.con_r = con; a.out_r = out;
    hipStream_t s = (hipStream_t)stream;
    if (con == 96) return accumulate ? proj_gemm_launch<96, 32, true, true, true>(a, s) : proj_gemm_launch<96, 32, true, true, false>(a, s);
    if (con == 192) return accumulate ? proj_gemm_launch<192, 64, true, true, true>(a, s) : proj_gemm_launch<192, 64, true, true, false>(a, s);
    if (con == 56) return accumulate ? proj_gemm_launch<64, 192, true, true, true>(a, s) : proj_gemm_launch<64, 192, true, true, false>(a, s);
    return accumulate ? proj_gemm_launch<32, 96, true, true, true>(a, s) : proj_gemm_launch<32, 96, true, true, false>(a, s);
}
}
