// tile_gemm.hip -- K-looped MFMA tile GEMM on the token-major stream, bf16 in / bf16 out, fp32 accumulation:
//
//     Y[t][n] = sum_k A[t][k] * Wm[n][k] (+ bias[n])          t: tokens (rows of the stream), n: output channels
//
// for the products whose weight does not fit LDS whole (tokens_gemm.hip keeps the weight resident and covers the
// skinny 56x56-stage shapes): the Mlp GEMMs of the later stages and -- with a gathering A loader -- the 3x3 stride-2
// convolutions of the patch embedding / downsampling layers as IMPLICIT GEMMs on the (B, H, W, C) stream (no im2col
// buffer, no NCHW<->NHWC transposes, bias and output layout in the epilogue).
//
// Tile: 128 tokens x 128 channels per workgroup of four waves (each 64 x 64 = 2 x 2 v_mfma_f32_32x32x16_bf16 tiles),
// K in steps of 64.  Both operand tiles are staged k-contiguous in LDS (row pitch 72 halfwords: the 16-byte fragment
// reads of 32 consecutive rows fall on distinct banks), double buffered: the global loads of step i+1 are in flight
// during the MFMAs of step i and one workgroup barrier per step separates the buffers.  73.7 KB of LDS per workgroup
// -> two workgroups (8 waves) per CU.  The product is formed transposed, D[n][t] (A operand = weight rows, B operand =
// token rows), so that after the bf16 pack and two v_permlane32_swap a lane holds 8 consecutive channels of its token:
// 16-byte stores (same epilogue as tokens_gemm.hip).
//
// Workgroup order: ids go round-robin over the 8 XCDs, so id -> (xcd, j) with j enumerating (token block, channel
// block) channel-fastest INSIDE an XCD: the channel blocks that re-read one token block run back to back on one L2.
//
// A "K segment" generalises the row addressing (dense rows have one segment):
//   convolution forward:  segment = tap (kh, kw), length C_in:  A[t][tap][c] = x[b][2 ho + kh - 1][2 wo + kw - 1][c]
//   convolution backward (data), one launch per input-pixel parity class (h & 1, w & 1): the taps that reach the class
//   (1, 2, 2 or 4 of the 9) are the segments, length C_out: A[t][i][n] = dy[b][i_h + dh_i][i_w + dw_i][n]; the weight
//   rows are read from a (9, C_in, C_out) copy and the rows of Y scatter to the class's pixels.
#include "xfm_common.hpp"

#include <algorithm>

namespace xfm {

typedef __bf16 tl_bf16x8_t __attribute__((ext_vector_type(8)));
typedef float tl_f32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t tl_u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t tl_u32x2_t __attribute__((ext_vector_type(2)));

constexpr int kTlBM = 128, kTlBN = 128, kTlBK = 64, kTlP = kTlBK + 8;
constexpr int kTlMaxSeg = 9;

struct TileGemmArgs {
    const uint16_t *a;       // activations: (T, K) dense, or a (B, AH, AW, AC) map
    const uint16_t *w;       // weight rows: row n at w + n * wrow + wseg[i] + k_in_segment
    const float *bias;       // (N) or null
    uint16_t *y;             // (T, N), or a (B, OH, OW, N) map written through the row scatter
    int64_t T;               // rows of this launch
    int N, K;                // output channels, contraction length (nseg * seglen)
    int mode;                // 0 dense, 1 convolution gather (A rows from a map; Y rows dense or scattered)
    int nseg, seglen;
    int wrow;
    int wseg[kTlMaxSeg];
    int seg_dh[kTlMaxSeg], seg_dw[kTlMaxSeg];
    int RH, RW;              // row grid: t = (b * RH + r_h) * RW + r_w
    int AH, AW, asy;         // A map size and row-grid stride: pixel (asy * r_h + seg_dh, asy * r_w + seg_dw)
    int scatter, OH, OW, osy, oph, opw;   // Y row = ((b * OH + osy * r_h + oph) * OW + osy * r_w + opw) * N when scatter
};

__device__ __forceinline__ uint32_t tl_pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const bf2 v = __builtin_convertvector(f2{lo, hi}, bf2);
    return __builtin_bit_cast(uint32_t, v);
}

__global__ void __launch_bounds__(256, 2) tile_gemm_kernel(const TileGemmArgs a) {
    extern __shared__ __align__(16) uint16_t tl_lds[];
    uint16_t *Xs = tl_lds;                               // [2][BM][P] token rows
    uint16_t *Ws = tl_lds + 2 * kTlBM * kTlP;            // [2][BN][P] weight rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- workgroup -> (token block, channel block), XCD aware
    const int nbn = (a.N + kTlBN - 1) / kTlBN;
    const int64_t nbm = (a.T + kTlBM - 1) / kTlBM;
    int64_t mb;
    int nb;
    {
        const int64_t id = blockIdx.x;
        if ((nbm & 7) == 0) {
            const int xcd = (int)(id & 7);
            const int64_t j = id >> 3, mloc = j / nbn;
            mb = mloc * 8 + xcd;
            nb = (int)(j - mloc * nbn);
        } else {                                          // token blocks not a multiple of the XCD count: plain order
            mb = id / nbn;
            nb = (int)(id - mb * nbn);
        }
    }
    const int64_t t0 = mb * kTlBM;
    const int n0 = nb * kTlBN;
    // ---- loader roles: 4 rows (r = (tid >> 3) + 32 v) x one 16-byte k-vector (kv = tid & 7) of each operand tile
    const int lr = tid >> 3, kv = tid & 7;
    int64_t arow[4];                                     // dense: element offset of the row; map: packed (b, r_h, r_w)
    int ah[4], aw_[4];
    bool aok[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int64_t t = t0 + lr + 32 * v;
        aok[v] = t < a.T;
        const int64_t tc = aok[v] ? t : 0;
        if (a.mode == 0) {
            arow[v] = tc * a.K;
            ah[v] = aw_[v] = 0;
        } else {
            const int64_t q = tc / a.RW;
            const int rw = (int)(tc - q * a.RW);
            const int64_t b = q / a.RH;
            const int rh = (int)(q - b * a.RH);
            arow[v] = b * a.AH * a.AW;                   // pixel index of (b, 0, 0)
            ah[v] = a.asy * rh;
            aw_[v] = a.asy * rw;
        }
    }
    const int nkt = (a.K + kTlBK - 1) / kTlBK;
    tl_u32x4_t xa[4], wa[4];
    auto load_tiles = [&](const int kt) {
        const int k = kt * kTlBK + 8 * kv;
        const bool kok = k < a.K;
        int seg = 0, kin = k;
        if (a.nseg > 1) {
            seg = k / a.seglen;
            kin = k - seg * a.seglen;
        }
        seg = kok ? seg : 0;
        const int woff = a.wseg[seg] + kin;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int n = n0 + lr + 32 * v;
            wa[v] = tl_u32x4_t{0, 0, 0, 0};
            if (kok && n < a.N) wa[v] = *reinterpret_cast<const tl_u32x4_t *>(a.w + (int64_t)n * a.wrow + woff);
        }
        if (a.mode == 0) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                xa[v] = tl_u32x4_t{0, 0, 0, 0};
                if (kok && aok[v]) xa[v] = *reinterpret_cast<const tl_u32x4_t *>(a.a + arow[v] + k);
            }
        } else {
            const int dh = a.seg_dh[seg], dw = a.seg_dw[seg];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int ph = ah[v] + dh, pw = aw_[v] + dw;
                xa[v] = tl_u32x4_t{0, 0, 0, 0};
                if (kok && aok[v] && ph >= 0 && ph < a.AH && pw >= 0 && pw < a.AW)
                    xa[v] = *reinterpret_cast<const tl_u32x4_t *>(a.a + (arow[v] + (int64_t)ph * a.AW + pw) * a.seglen + kin);
            }
        }
    };
    auto store_tiles = [&](const int buf) {
        uint16_t *xs = Xs + buf * kTlBM * kTlP, *ws = Ws + buf * kTlBN * kTlP;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            *reinterpret_cast<tl_u32x4_t *>(xs + (lr + 32 * v) * kTlP + 8 * kv) = xa[v];
            *reinterpret_cast<tl_u32x4_t *>(ws + (lr + 32 * v) * kTlP + 8 * kv) = wa[v];
        }
    };
    // ---- MFMA roles: wave -> 64 channels x 64 tokens
    const int wn = wave >> 1, wm = wave & 1;
    const int c = lane & 31, h = lane >> 5;
    tl_f32x16_t acc[2][2];                               // [channel tile][token tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tiles(kt + 1);
        const uint16_t *xs = Xs + buf * kTlBM * kTlP + (wm * 64 + c) * kTlP + 8 * h;
        const uint16_t *ws = Ws + buf * kTlBN * kTlP + (wn * 64 + c) * kTlP + 8 * h;
#pragma unroll
        for (int s = 0; s < kTlBK / 16; ++s) {
            tl_bf16x8_t wf[2], xf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                wf[i] = *reinterpret_cast<const tl_bf16x8_t *>(ws + i * 32 * kTlP + 16 * s);
                xf[i] = *reinterpret_cast<const tl_bf16x8_t *>(xs + i * 32 * kTlP + 16 * s);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nkt) store_tiles(buf ^ 1);
        __syncthreads();
    }
    // ---- epilogue: D[n][t]: lane = token (c), registers = channels (reg & 3) + 8 (reg >> 2) + 4 h of the 32-channel tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t t = t0 + wm * 64 + j * 32 + c;
        const bool tok = t < a.T;
        int64_t yoff = t * a.N;
        if (a.scatter && tok) {
            const int64_t q = t / a.RW;
            const int rw = (int)(t - q * a.RW);
            const int64_t b = q / a.RH;
            const int rh = (int)(q - b * a.RH);
            yoff = ((b * a.OH + a.osy * rh + a.oph) * a.OW + a.osy * rw + a.opw) * (int64_t)a.N;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int nt = n0 + wn * 64 + i * 32;         // first channel of this 32-channel tile
            uint32_t pk[4][2];                            // group g: channels nt + 8 g + 4 h .. + 3
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (a.bias) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = nt + 8 * g + 4 * h + q;
                        bv[q] = n < a.N ? a.bias[n] : 0.f;
                    }
                }
                pk[g][0] = tl_pack_bf16x2(acc[i][j][4 * g] + bv[0], acc[i][j][4 * g + 1] + bv[1]);
                pk[g][1] = tl_pack_bf16x2(acc[i][j][4 * g + 2] + bv[2], acc[i][j][4 * g + 3] + bv[3]);
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2)
#pragma unroll
                for (int q = 0; q < 2; ++q) {             // lower half: channels +4..7 of group g; upper: +0..3 of g + 1
                    const tl_u32x2_t r = __builtin_amdgcn_permlane32_swap(pk[g][q], pk[g + 1][q], false, false);
                    pk[g][q] = r[0];
                    pk[g + 1][q] = r[1];
                }
            if (tok) {
                uint16_t *yr = a.y + yoff + nt + 8 * h;   // channels nt + 8 h .. + 7 and nt + 16 + 8 h .. + 7
                if (nt + 8 * h < a.N) *reinterpret_cast<tl_u32x4_t *>(yr) = tl_u32x4_t{pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
                if (nt + 16 + 8 * h < a.N)
                    *reinterpret_cast<tl_u32x4_t *>(yr + 16) = tl_u32x4_t{pk[2][0], pk[2][1], pk[3][0], pk[3][1]};
            }
        }
    }
}

static int tile_launch(const TileGemmArgs &a, hipStream_t s) {
    if (a.T <= 0) return XFM_OK;
    const int64_t nbm = (a.T + kTlBM - 1) / kTlBM;
    const int nbn = (a.N + kTlBN - 1) / kTlBN;
    const size_t lds = (size_t)2 * (kTlBM + kTlBN) * kTlP * sizeof(uint16_t);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)tile_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL(tile_gemm_kernel, dim3((unsigned)(nbm * nbn)), dim3(256), lds, s, a);
    return check_launch();
}

static bool tile_ptr_ok(const void *p) { return p && ((uintptr_t)p & 15) == 0; }

// The gathered taps of the 3x3 / stride-2 convolution as a matrix, col[t][(kh, kw, c)] (zeros outside the map): the
// token-major operand of the weight-gradient product dW[n][(kh, kw, c)] = sum_t dy[t][n] col[t][(kh, kw, c)].
__global__ void __launch_bounds__(256) im2col3x3s2_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ col,
                                                          const int64_t nvec, const int H, const int W, const int C,
                                                          const int Ho, const int Wo) {
    const int vpt = C / 8;                                // 16-byte vectors per tap
    for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
        const int64_t tt = v / vpt;                       // (token, tap)
        const int cv = (int)(v - tt * vpt);
        const int64_t t = tt / 9;
        const int tap = (int)(tt - t * 9);
        const int64_t q = t / Wo;
        const int wo = (int)(t - q * Wo);
        const int64_t b = q / Ho;
        const int ho = (int)(q - b * Ho);
        const int ph = 2 * ho + tap / 3 - 1, pw = 2 * wo + tap % 3 - 1;
        tl_u32x4_t val{0, 0, 0, 0};
        if (ph >= 0 && ph < H && pw >= 0 && pw < W)
            val = *reinterpret_cast<const tl_u32x4_t *>(x + ((b * H + ph) * W + pw) * C + 8 * cv);
        reinterpret_cast<tl_u32x4_t *>(col)[v] = val;
    }
}

}  // namespace xfm

extern "C" {

int xfm_tile_gemm(const void *x, const void *w, const float *bias, void *y, int64_t T, int K, int N, void *stream) {
    using namespace xfm;
    if (!tile_ptr_ok(x) || !tile_ptr_ok(w) || !tile_ptr_ok(y) || T < 0 || K <= 0 || N <= 0) return XFM_EINVAL;
    if (K % 8 || N % 8) return XFM_ELIMIT;
    TileGemmArgs a{};
    a.a = (const uint16_t *)x; a.w = (const uint16_t *)w; a.bias = bias; a.y = (uint16_t *)y;
    a.T = T; a.N = N; a.K = K; a.mode = 0; a.nseg = 1; a.seglen = K; a.wrow = K;
    return tile_launch(a, (hipStream_t)stream);
}

int xfm_conv3x3s2_fwd(const void *x, const void *w9, const float *bias, void *y, int batch, int H, int W, int C, int N,
                      void *stream) {
    using namespace xfm;
    if (!tile_ptr_ok(x) || !tile_ptr_ok(w9) || !tile_ptr_ok(y) || batch <= 0 || H <= 0 || W <= 0) return XFM_EINVAL;
    if (C % 8 || N % 8) return XFM_ELIMIT;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    TileGemmArgs a{};
    a.a = (const uint16_t *)x; a.w = (const uint16_t *)w9; a.bias = bias; a.y = (uint16_t *)y;
    a.T = (int64_t)batch * Ho * Wo; a.N = N; a.K = 9 * C; a.mode = 1; a.nseg = 9; a.seglen = C; a.wrow = 9 * C;
    for (int i = 0; i < 9; ++i) {
        a.wseg[i] = i * C;
        a.seg_dh[i] = i / 3 - 1;
        a.seg_dw[i] = i % 3 - 1;
    }
    a.RH = Ho; a.RW = Wo; a.AH = H; a.AW = W; a.asy = 2;
    return tile_launch(a, (hipStream_t)stream);
}

int xfm_im2col3x3s2(const void *x, void *col, int batch, int H, int W, int C, void *stream) {
    using namespace xfm;
    if (!tile_ptr_ok(x) || !tile_ptr_ok(col) || batch <= 0 || H <= 0 || W <= 0) return XFM_EINVAL;
    if (C % 8) return XFM_ELIMIT;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int64_t nvec = (int64_t)batch * Ho * Wo * 9 * (C / 8);
    const unsigned grid = (unsigned)std::min<int64_t>((nvec + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(im2col3x3s2_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t *)x, (uint16_t *)col,
                       nvec, H, W, C, Ho, Wo);
    return check_launch();
}

int xfm_conv3x3s2_dgrad(const void *dy, const void *wt, void *dx, int batch, int H, int W, int C, int N, void *stream) {
    using namespace xfm;
    if (!tile_ptr_ok(dy) || !tile_ptr_ok(wt) || !tile_ptr_ok(dx) || batch <= 0 || H <= 0 || W <= 0) return XFM_EINVAL;
    if (C % 8 || N % 8 || H % 2 || W % 2) return XFM_ELIMIT;
    const int Ho = H / 2, Wo = W / 2;
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
            TileGemmArgs a{};
            a.a = (const uint16_t *)dy; a.w = (const uint16_t *)wt; a.bias = nullptr; a.y = (uint16_t *)dx;
            a.T = (int64_t)batch * Ho * Wo; a.N = C; a.mode = 1; a.seglen = N; a.wrow = N;
            // input pixel (2 i + ph): tap kh reaches it from output row ho with 2 ho + kh - 1 = 2 i + ph
            //   ph = 0: kh = 1, ho = i;   ph = 1: kh = 0, ho = i + 1  and  kh = 2, ho = i
            int nh = 0, khs[2], dhs[2], nw = 0, kws[2], dws[2];
            if (ph == 0) { khs[nh] = 1; dhs[nh++] = 0; } else { khs[nh] = 0; dhs[nh++] = 1; khs[nh] = 2; dhs[nh++] = 0; }
            if (pw == 0) { kws[nw] = 1; dws[nw++] = 0; } else { kws[nw] = 0; dws[nw++] = 1; kws[nw] = 2; dws[nw++] = 0; }
            int ns = 0;
            for (int i = 0; i < nh; ++i)
                for (int j = 0; j < nw; ++j) {
                    a.wseg[ns] = (khs[i] * 3 + kws[j]) * C * N;
                    a.seg_dh[ns] = dhs[i];
                    a.seg_dw[ns] = dws[j];
                    ++ns;
                }
            a.nseg = ns; a.K = ns * N;
            a.RH = Ho; a.RW = Wo; a.AH = Ho; a.AW = Wo; a.asy = 1;
            a.scatter = 1; a.OH = H; a.OW = W; a.osy = 2; a.oph = ph; a.opw = pw;
            const int rc = tile_launch(a, (hipStream_t)stream);
            if (rc) return rc;
        }
    return XFM_OK;
}

}  // extern "C"
