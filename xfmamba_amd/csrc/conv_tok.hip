// conv_tok.hip -- the 3 x 3, stride-2, padding-1 convolutions of the trunk (patch embedding, second convolution; the three
// downsample layers: reference models/fusion_vmamba.py:1504-1518, 1531-1538) on TOKEN-MAJOR maps, on this library's own MFMA
// GEMM kernels instead of the convolution library:
//   forward      col (T, 9 C) = the 3 x 3 neighbourhoods of x (B, H, W, C) as rows, taps (kh, kw) outer, channels inner
//                (zeros where a tap falls into the padding);  y (T, O) = col . W^T  with W as (O, 3, 3, C) -- the
//                channels_last memory of the (O, C, 3, 3) parameter -- through the tiled token GEMM (tokens_gemm3)
//   data grad    dcol (T, 9 C) = dy (T, O) . W  (the same kernel, weight read k-major);  dx = the taps of dcol gathered back:
//                an input pixel (ih, iw) belongs to at most four windows (kh = ih + 1 - 2 oh, kw likewise)
//   weight grad  dW (O, 9 C) += dy^T . col  through xfm_wgrad (token x token, LDS-direct) -- the forward's col rows are kept
// T = B . H/2 . W/2.  The rows move 9/4 of the input once per direction (43 MB at 28 x 28 x 192 -> 14 x 14 x 384, batch 64: ~12 us
// of the ~40 us a pass takes); what they buy is that every FLOP of these layers runs on the kernels the Mlp products use and
// nothing in a step depends on a convolution library's solver search.  H, W even, C % 8 == 0, O % 8 == 0, bf16.
#include <algorithm>
#include <cstdlib>

#include "xfm_common.hpp"

extern "C" int xfm_wgrad(const void *a, const void *b, float *dw, int M, int N, int batch, int L, int64_t a_bs, int64_t b_bs,
                         int a_planes, int b_planes, void *stream);

namespace xfm {

int tokens_gemm3_plain(const void *x, const void *w, void *y, long long T, int con, int out, bool wt, hipStream_t s);

struct ConvTokArgs {
    const uint4 *src;
    uint4 *dst;
    int B, H, W, OH, OW, C8;     // C8: 16-byte chunks per pixel
    long long n;                 // threads with work
};

// col[t][tap][c8] = x[n, 2 oh + kh - 1, 2 ow + kw - 1, c8] or zero: one 16-byte chunk per thread, consecutive threads along
// a col row (whole-row coalesced writes; reads contiguous inside a tap)
__global__ void __launch_bounds__(256) conv_im2col_kernel(const ConvTokArgs a) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.n) return;
    const int K8 = 9 * a.C8;
    const long long t = e / K8;
    const int r = (int)(e - t * K8), tap = r / a.C8, c8 = r - tap * a.C8;
    const int kh = tap / 3, kw = tap - 3 * kh;
    const int ow = (int)(t % a.OW);
    const long long q = t / a.OW;
    const int oh = (int)(q % a.OH);
    const long long n = q / a.OH;
    const int ih = 2 * oh + kh - 1, iw = 2 * ow + kw - 1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (ih >= 0 && iw >= 0 && ih < a.H && iw < a.W) v = a.src[((n * a.H + ih) * a.W + iw) * a.C8 + c8];
    a.dst[e] = v;
}

__device__ __forceinline__ void conv_acc8(float (&s)[8], const uint4 v) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        s[2 * q] += __uint_as_float(w[q] << 16);
        s[2 * q + 1] += __uint_as_float(w[q] & 0xffff0000u);
    }
}

// dx[n, ih, iw, c8] = sum over the windows the pixel belongs to of dcol[t(n, oh, ow)][kh, kw][c8]: fp32 sums, one rounding
__global__ void __launch_bounds__(256) conv_col2im_kernel(const ConvTokArgs a) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.n) return;
    const int c8 = (int)(e % a.C8);
    const long long p = e / a.C8;
    const int iw = (int)(p % a.W);
    const long long q = p / a.W;
    const int ih = (int)(q % a.H);
    const long long n = q / a.H;
    const int K8 = 9 * a.C8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int th = ih + 1 - kh;
        if ((th & 1) || th < 0 || (th >> 1) >= a.OH) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int tw = iw + 1 - kw;
            if ((tw & 1) || tw < 0 || (tw >> 1) >= a.OW) continue;
            const long long t = (n * a.OH + (th >> 1)) * a.OW + (tw >> 1);
            conv_acc8(s, a.src[t * K8 + (kh * 3 + kw) * a.C8 + c8]);
        }
    }
    a.dst[e] = make_uint4(pack_bf16x2(s[0], s[1]), pack_bf16x2(s[2], s[3]), pack_bf16x2(s[4], s[5]), pack_bf16x2(s[6], s[7]));
}

// ---------------------------------------------------------------------------------------------------------------------
// The FIRST convolution of the patch embedding on a one-channel image.  The model replicates its single input channel
// (reference net_fusionmamba.py:88-104: x.expand(-1, 3, -1, -1)), so conv(x3, W)[o] = conv(x1, sum_c W[o, c]) and
// dW[o, c] = dW9[o] for every c: 9 taps instead of 27, a 6 MB image instead of 19 MB, and no replicated copy.
// Lane = (output pixel, group of 8 output channels): the lanes of a wave cover consecutive 16-byte pieces of y / dy -- every
// load and store instruction of a wave is one contiguous KB -- and a lane keeps its group's 8 x 9 weights (forward) or sums
// (weight gradient) in registers; its group is fixed because every stride below is a multiple of the group count.
// ---------------------------------------------------------------------------------------------------------------------
struct GrayArgs {
    const uint16_t *x;           // (B, H, W) bf16
    const uint16_t *w;           // (O, CI, 3, 3) bf16 (forward): the parameter's shadow; the kernel sums it over the CI replicas
    int CI;
    uint16_t *y;                 // (B, OH, OW, O) bf16: y (forward) / dy (weight gradient)
    float *dw9;                  // (O, 9) fp32 (weight gradient; atomics)
    int B, H, W, OH, OW, O, OG;  // OG = O / 8
    long long n;                 // pixels * OG
};

// the 3 x 3 window of output pixel (oh, ow) of one image: rows as two aligned dwords each (columns 2 ow - 2 .. 2 ow + 1), the
// last three halfwords are the taps; zero outside the image
__device__ __forceinline__ void gray_window(const uint16_t *img, const int W, const int oh, const int ow, float (&xv)[9]) {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = 2 * oh + kh - 1;
        uint32_t d0 = 0, d1 = 0;
        if (ih >= 0) {                                               // (ih <= H - 1 always: H even)
            const uint32_t *row = reinterpret_cast<const uint32_t *>(img + ih * W);
            d1 = row[ow];                                            // columns 2 ow, 2 ow + 1
            if (ow > 0) d0 = row[ow - 1];                            // column 2 ow - 1 in its high half
        }
        xv[3 * kh] = __uint_as_float(d0 & 0xffff0000u);
        xv[3 * kh + 1] = __uint_as_float(d1 << 16);
        xv[3 * kh + 2] = __uint_as_float(d1 & 0xffff0000u);
    }
}

// the six dwords of the window of output pixel (row r = (n, oh), column ow): clamped, unconditional loads (they stay in flight),
// zeroed where the window leaves the image
__device__ __forceinline__ void gray_win_load(const GrayArgs &a, const int r, const int ow, uint32_t (&d)[6]) {
    const int n = r / a.OH, oh = r - n * a.OH;
    const uint16_t *img = a.x + (long long)n * a.H * a.W;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = 2 * oh + kh - 1;
        const uint32_t *row = reinterpret_cast<const uint32_t *>(img + (ih < 0 ? 0 : ih) * a.W);
        const uint32_t d1 = row[ow], d0 = row[ow > 0 ? ow - 1 : 0];
        d[2 * kh] = (ih >= 0 && ow > 0) ? d0 : 0u;
        d[2 * kh + 1] = ih >= 0 ? d1 : 0u;
    }
}

// A workgroup walks output ROWS (n, oh); thread = (column slot, channel group): og = tid % OG, columns tid / OG + k (192 / OG) --
// no division in the loops, and the 192 lanes' 16-byte pieces of a pass are 3 KB of consecutive bytes of the row.
__global__ void __launch_bounds__(192) conv_gray_fwd_kernel(const GrayArgs a) {
    const int og = threadIdx.x % a.OG, ow0 = threadIdx.x / a.OG, step = 192 / a.OG;
    // the weight summed over its CI input channels, once per workgroup through LDS (per thread from global memory the 72 CI
    // two-byte loads of every wave were most of a 115 us launch)
    __shared__ float w9[24 * 72];                                    // (O, 9), O <= 192
    for (int j = threadIdx.x; j < a.O * 9; j += 192) {
        const int o = j / 9, k = j - 9 * o;
        float t = 0.f;
        for (int ci = 0; ci < a.CI; ++ci) t += __uint_as_float((uint32_t)a.w[(o * a.CI + ci) * 9 + k] << 16);
        w9[j] = t;
    }
    __syncthreads();
    float w[8][9];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int k = 0; k < 9; ++k) w[c][k] = w9[(og * 8 + c) * 9 + k];
    const int rows = a.B * a.OH;
    // one visit of look-ahead, as in the weight-gradient kernel below: the next window's six dwords are requested before this one
    // is multiplied (a thread makes ~28 visits; without it every visit's loads were in the open: 37 us)
    int r = blockIdx.x, ow = ow0;
    uint32_t cur[6], nxt[6];
    bool live = r < rows && ow < a.OW;
    if (live) gray_win_load(a, r, ow, cur);
    while (live) {
        int r2 = r, ow2 = ow + step;
        if (ow2 >= a.OW) {
            ow2 = ow0;
            r2 = r + gridDim.x;
        }
        const bool live2 = r2 < rows;
        if (live2) gray_win_load(a, r2, ow2, nxt);
        float xv[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            xv[3 * kh] = __uint_as_float(cur[2 * kh] & 0xffff0000u);
            xv[3 * kh + 1] = __uint_as_float(cur[2 * kh + 1] << 16);
            xv[3 * kh + 2] = __uint_as_float(cur[2 * kh + 1] & 0xffff0000u);
        }
        float s[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) t = fmaf(w[c][k], xv[k], t);
            s[c] = t;
        }
        (reinterpret_cast<uint4 *>(a.y) + (long long)r * a.OW * a.OG)[ow * a.OG + og] =
            make_uint4(pack_bf16x2(s[0], s[1]), pack_bf16x2(s[2], s[3]), pack_bf16x2(s[4], s[5]), pack_bf16x2(s[6], s[7]));
#pragma unroll
        for (int q = 0; q < 6; ++q) cur[q] = nxt[q];
        r = r2; ow = ow2; live = live2;
    }
}

constexpr int kGrayRep = 32;

// dweight9 (O, 9) += the kGrayRep replicas the workgroups of conv_gray_wgrad_kernel added into
__global__ void __launch_bounds__(256) conv_gray_fold_kernel(const float *rep, float *dw9, const int n) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    float s = 0.f;
    for (int q = 0; q < kGrayRep; ++q) s += rep[q * n + j];
    dw9[j] += s;
}

// raw operands of one (row, column) visit of a thread: the window's six dwords and the lane's 16 bytes of dy
struct GrayOps { uint32_t d[6]; uint4 g; };

__device__ __forceinline__ void gray_ops_load(const GrayArgs &a, const int r, const int ow, const int og, GrayOps &o) {
    const int n = r / a.OH, oh = r - n * a.OH;
    const uint16_t *img = a.x + (long long)n * a.H * a.W;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = 2 * oh + kh - 1;
        const uint32_t *row = reinterpret_cast<const uint32_t *>(img + (ih < 0 ? 0 : ih) * a.W);
        const uint32_t d1 = row[ow], d0 = row[ow > 0 ? ow - 1 : 0];        // (clamped, unconditional: the loads stay in flight)
        o.d[2 * kh] = (ih >= 0 && ow > 0) ? d0 : 0u;
        o.d[2 * kh + 1] = ih >= 0 ? d1 : 0u;
    }
    o.g = (reinterpret_cast<const uint4 *>(a.y) + (long long)r * a.OW * a.OG)[ow * a.OG + og];
}

__global__ void __launch_bounds__(192) conv_gray_wgrad_kernel(const GrayArgs a) {
    __shared__ float red[192 * 19];                                  // [thread][18 sums], pitch 19: conflict-free column reads
    const int og = threadIdx.x % a.OG, ow0 = threadIdx.x / a.OG, step = 192 / a.OG;
    float acc[8][9];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[c][k] = 0.f;
    const int rows = a.B * a.OH;
    // one visit ahead: a thread has few partners on its SIMD (110 registers, one round of workgroups), so the operands of visit
    // i + 1 are requested before visit i is multiplied
    int r = blockIdx.x, ow = ow0;
    GrayOps cur, nxt;
    bool live = r < rows && ow < a.OW;
    if (live) gray_ops_load(a, r, ow, og, cur);
    while (live) {
        int r2 = r, ow2 = ow + step;
        if (ow2 >= a.OW) {
            ow2 = ow0;
            r2 = r + gridDim.x;
        }
        const bool live2 = r2 < rows;
        if (live2) gray_ops_load(a, r2, ow2, og, nxt);
        float xv[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            xv[3 * kh] = __uint_as_float(cur.d[2 * kh] & 0xffff0000u);
            xv[3 * kh + 1] = __uint_as_float(cur.d[2 * kh + 1] << 16);
            xv[3 * kh + 2] = __uint_as_float(cur.d[2 * kh + 1] & 0xffff0000u);
        }
        const uint32_t gw[4] = {cur.g.x, cur.g.y, cur.g.z, cur.g.w};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float gv = (c & 1) ? __uint_as_float(gw[c >> 1] & 0xffff0000u) : __uint_as_float(gw[c >> 1] << 16);
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[c][k] = fmaf(gv, xv[k], acc[c][k]);
        }
        cur = nxt;
        r = r2; ow = ow2; live = live2;
    }
    // sums of a channel group over the 192 / OG threads that own it, two channels (18 sums) per pass: 14 KB of LDS, so that the
    // register count, not this array, decides how many waves share a SIMD (with all 72 sums in one 56 KB array: two workgroups
    // per CU, 1.5 waves per SIMD, and every visit's load latency in the open)
#pragma unroll
    for (int c2 = 0; c2 < 4; ++c2) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 9; ++k) red[threadIdx.x * 19 + c * 9 + k] = acc[2 * c2 + c][k];
        __syncthreads();
        for (int j = threadIdx.x; j < a.OG * 18; j += 192) {
            const int g2 = j / 18, r2 = j - g2 * 18;
            float sum = 0.f;
            for (int q = 0; q < step; ++q) sum += red[(g2 + q * a.OG) * 19 + r2];
            // one of kGrayRep replicas: adds to ONE address retire one by one (~45 ns each measured: 2048 workgroups on a single
            // (O, 9) array spent 90 us there)
            atomicAdd(a.dw9 + (blockIdx.x % kGrayRep) * (a.O * 9) + g2 * 72 + c2 * 18 + r2, sum);
        }
        __syncthreads();
    }
}

static int conv_gray_ok(int B, int H, int W, int O) {
    return B > 0 && H >= 2 && W >= 2 && !(H & 1) && !(W & 1) && O >= 8 && O <= 192 && O % 8 == 0 && 192 % (O / 8) == 0 &&
           (long long)B * H * W < (1ll << 31);            // (O <= 192: the forward stages O x 9 summed weights in w9[24 * 72])
}

static GrayArgs gray_args(const void *x, int B, int H, int W, int O) {
    GrayArgs a{};
    a.x = static_cast<const uint16_t *>(x);
    a.B = B; a.H = H; a.W = W; a.OH = H / 2; a.OW = W / 2; a.O = O; a.OG = O / 8;
    a.n = (long long)B * a.OH * a.OW * a.OG;
    return a;
}

static int conv_tok_ok(int B, int H, int W, int C, int O) {
    return B > 0 && H >= 2 && W >= 2 && !(H & 1) && !(W & 1) && C >= 8 && C % 8 == 0 && O >= 8 && O % 8 == 0 &&
           (long long)B * H * W * C < (1ll << 40);
}

static ConvTokArgs conv_args(const void *src, void *dst, int B, int H, int W, int C) {
    ConvTokArgs a{};
    a.src = static_cast<const uint4 *>(src);
    a.dst = static_cast<uint4 *>(dst);
    a.B = B; a.H = H; a.W = W; a.OH = H / 2; a.OW = W / 2; a.C8 = C / 8;
    return a;
}

}  // namespace xfm

extern "C" {

int xfm_conv3x3s2_tokens_supported(int C, int O, int H, int W) { return xfm::conv_tok_ok(1, H, W, C, O); }

int xfm_conv3x3s2_tokens_fwd(const void *x, const void *weight, void *col, void *y, int B, int H, int W, int C, int O, void *stream) {
    using namespace xfm;
    if (!x || !weight || !col || !y) return XFM_EINVAL;
    if (!conv_tok_ok(B, H, W, C, O)) return XFM_ELIMIT;
    if (((uintptr_t)x | (uintptr_t)weight | (uintptr_t)col | (uintptr_t)y) & 15) return XFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ConvTokArgs a = conv_args(x, col, B, H, W, C);
    const long long T = (long long)B * a.OH * a.OW;
    a.n = T * 9 * a.C8;
    hipLaunchKernelGGL(conv_im2col_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a);
    const int rc = check_launch();
    if (rc != XFM_OK) return rc;
    return tokens_gemm3_plain(col, weight, y, T, 9 * C, O, false, s);
}

int xfm_conv3x3s2_tokens_bwd_data(const void *dy, const void *weight, void *dcol, void *dx, int B, int H, int W, int C, int O,
                                  void *stream) {
    using namespace xfm;
    if (!dy || !weight || !dcol || !dx) return XFM_EINVAL;
    if (!conv_tok_ok(B, H, W, C, O)) return XFM_ELIMIT;
    if (((uintptr_t)dy | (uintptr_t)weight | (uintptr_t)dcol | (uintptr_t)dx) & 15) return XFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const long long T = (long long)B * (H / 2) * (W / 2);
    const int rc = tokens_gemm3_plain(dy, weight, dcol, T, O, 9 * C, true, s);      // the (O, 9 C) weight is k-major for this product
    if (rc != XFM_OK) return rc;
    ConvTokArgs a = conv_args(dcol, dx, B, H, W, C);
    a.n = (long long)B * H * W * a.C8;
    hipLaunchKernelGGL(conv_col2im_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, s, a);
    return check_launch();
}

int xfm_conv3x3s2_gray_supported(int O, int H, int W) { return xfm::conv_gray_ok(1, H, W, O); }

int xfm_conv3x3s2_gray_fwd(const void *x, const void *weight, void *y, int B, int H, int W, int CI, int O, void *stream) {
    using namespace xfm;
    if (!x || !weight || !y || CI < 1) return XFM_EINVAL;
    if (!conv_gray_ok(B, H, W, O)) return XFM_ELIMIT;
    if (((uintptr_t)x & 3) || ((uintptr_t)y & 15) || ((uintptr_t)weight & 1)) return XFM_EINVAL;
    GrayArgs a = gray_args(x, B, H, W, O);
    a.w = static_cast<const uint16_t *>(weight);
    a.CI = CI;
    a.y = static_cast<uint16_t *>(y);
    static const int env_fwgs = [] { const char *e = getenv("XFM_GRAY_FWD_WGS"); return e ? atoi(e) : 1024; }();   // tuning hook, read once
    const long long blocks = std::min<long long>((long long)B * a.OH, env_fwgs);
    hipLaunchKernelGGL(conv_gray_fwd_kernel, dim3((unsigned)blocks), dim3(192), 0, (hipStream_t)stream, a);
    return check_launch();
}

int xfm_conv3x3s2_gray_ws_floats(int O) { return xfm::kGrayRep * O * 9; }

int xfm_conv3x3s2_gray_bwd_weight(const void *dy, const void *x, float *dw9, float *ws, int B, int H, int W, int O, void *stream) {
    using namespace xfm;
    if (!dy || !x || !dw9 || !ws) return XFM_EINVAL;
    if (!conv_gray_ok(B, H, W, O)) return XFM_ELIMIT;
    if (((uintptr_t)x & 3) || ((uintptr_t)dy & 15)) return XFM_EINVAL;
    GrayArgs a = gray_args(x, B, H, W, O);
    a.y = const_cast<uint16_t *>(static_cast<const uint16_t *>(dy));
    a.dw9 = ws;
    static const int env_wgs = [] { const char *e = getenv("XFM_GRAY_WGS"); return e ? atoi(e) : 1024; }();   // tuning hook, read once (256: 79 us, 512: 51, 1024: 36, 2048: 41)
    const long long blocks = std::min<long long>((long long)B * a.OH, env_wgs);
    hipLaunchKernelGGL(conv_gray_wgrad_kernel, dim3((unsigned)blocks), dim3(192), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(conv_gray_fold_kernel, dim3((unsigned)((O * 9 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ws, dw9, O * 9);
    return check_launch();
}

int xfm_conv3x3s2_tokens_bwd_weight(const void *dy, const void *col, float *dw, int B, int H, int W, int C, int O, void *stream) {
    using namespace xfm;
    if (!dy || !col || !dw) return XFM_EINVAL;
    if (!conv_tok_ok(B, H, W, C, O)) return XFM_ELIMIT;
    const long long T = (long long)B * (H / 2) * (W / 2);
    if (T > 0x7fffffffll) return XFM_ELIMIT;
    return xfm_wgrad(dy, col, dw, O, 9 * C, 1, (int)T, 0, 0, 0, 0, stream);
}

}  // extern "C"
