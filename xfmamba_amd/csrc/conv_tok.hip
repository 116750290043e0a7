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

int xfm_conv3x3s2_tokens_bwd_weight(const void *dy, const void *col, float *dw, int B, int H, int W, int C, int O, void *stream) {
    using namespace xfm;
    if (!dy || !col || !dw) return XFM_EINVAL;
    if (!conv_tok_ok(B, H, W, C, O)) return XFM_ELIMIT;
    const long long T = (long long)B * (H / 2) * (W / 2);
    if (T > 0x7fffffffll) return XFM_ELIMIT;
    return xfm_wgrad(dy, col, dw, O, 9 * C, 1, (int)T, 0, 0, 0, 0, stream);
}

}  // extern "C"
