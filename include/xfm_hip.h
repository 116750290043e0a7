/*
 * xfm_hip.h -- C ABI of libxfm_hip.so, the MI355X (gfx950) kernels of the XFMamba hot path.
 *
 * Drop-in boundary.  Every entry point takes plain device pointers,
 * sizes, element strides and a hipStream_t passed as void*; nothing allocates, nothing
 * synchronises, all launches are asynchronous on the given stream and graph-capturable.
 * Return value: 0 on success, a negative XFM_E* code otherwise (xfm_strerror() names it).
 * The caller (xfmamba_amd/csms6s.py, csm.py, fusion_vmamba.py) owns allocation and autograd.
 *
 * What each entry point replaces in the reference (paths relative to XZheng0427/XFMamba):
 *
 *   xfm_selective_scan_fwd  <- pybind `fwd(u, delta, A, B, C, D?, delta_bias?, delta_softplus, nrows)`
 *                              models/selective_scan/csrc/selective_scan/selective_scan.cpp:165-249,
 *                              reached from models/csms6s.py:81-85 (SelectiveScanCuda.forward)
 *   xfm_selective_scan_bwd  <- pybind `bwd(u, delta, A, B, C, D?, delta_bias?, dout, x?, delta_softplus, nrows)`
 *                              selective_scan.cpp:251-362, reached from models/csms6s.py:96-108
 *   xfm_scan_plan           <- the chunk-count rule `n_chunks = (seqlen + 2048 - 1) / 2048`
 *                              selective_scan.cpp:225-228 (sizes the saved-state tensor `x`)
 *   xfm_cross_scan          <- CrossScanTritonF.forward / triton_cross_scan_flex,
 *                              models/csm_triton.py:403-430, 278-400 (scans=0, channel-first)
 *   xfm_cross_merge         <- CrossMergeTritonF.forward, models/csm_triton.py:456-483
 *   xfm_swap_scan           <- SwappingScan_multiview.forward, models/fusion_vmamba.py:189-213
 *   xfm_dwconv3x3_fwd/_bwd  <- `self.conv2d` (nn.Conv2d(D, D, 3, padding=1, groups=D)) followed by `self.act`
 *                              (nn.SiLU) in front of every SS2D core: models/fusion_vmamba.py:1198-1201,
 *                              :594-601, :853-857 (MIOpen depthwise conv + elementwise kernels upstream)
 *   xfm_layernorm2d_fwd/_bwd <- LayerNorm2d.forward = permute -> F.layer_norm -> permute on NCHW maps,
 *                              models/fusion_vmamba.py:52-57 (block norms, out_norm of SS2Dv2, patch-embed/downsample norms)
 *   xfm_ss2d_fwd/_bwd       <- the fused body of SS2Dv2.forward_corev2, models/fusion_vmamba.py:1145-1174
 *                              (cross_scan_fn -> selective_scan_fn -> cross_merge_fn in one kernel)
 *
 * Contracts shared with the reference FFI (selective_scan.cpp:173-221): u, delta, B, C (and
 * dout unless `out` is fp32) share one dtype in {fp32, fp16, bf16}; A, D, delta_bias are fp32;
 * the last (sequence) dimension of every tensor has stride 1; dim % n_groups == 0.
 */
#ifndef XFM_HIP_H
#define XFM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XFM_ABI_VERSION 2      /* 2: xfm_ss2d_params_t.bc_f32 (round 6); layouts of round 5's y_tokens / x_tokens / xrt fields */

enum { XFM_F32 = 0, XFM_F16 = 1, XFM_BF16 = 2 };

enum {
    XFM_OK = 0,
    XFM_EINVAL = -1,      /* bad size / null pointer / dim % n_groups != 0 */
    XFM_EDTYPE = -2,      /* unsupported dtype combination */
    XFM_ELIMIT = -3,      /* size outside what the kernels support (dstate > 256, H*W too large for LDS ...) */
    XFM_ELAUNCH = -4      /* hipLaunchKernel failed; hipGetLastError() text via xfm_last_hip_error() */
};

/* Work decomposition chosen for a scan of this shape.  Forward and backward use the same plan,
 * so `n_chunks` sizes the chunk-state tensor x: (batch, dim, n_chunks, dstate) fp32. */
typedef struct {
    int lanes_per_row;  /* 1..64 lanes of a 64-wide wavefront cooperate on one (batch, dim) row */
    int items;          /* sequence elements per lane per chunk (compile-time variant) */
    int n_chunks;       /* ceil(seqlen / (lanes_per_row * items)) */
} xfm_scan_plan_t;

typedef struct {
    int batch, dim, seqlen, dstate, n_groups;
    int delta_softplus;   /* bool */
    int in_dtype;         /* of u, delta, B, C (and du, ddelta, and dout/out unless out_dtype says fp32) */
    int out_dtype;        /* of out / dout: XFM_F32 ("oflex", csms6s.py:68) or == in_dtype */
    /* inputs (borrowed, never written) */
    const void *u, *delta;          /* (batch, dim, seqlen) */
    const float *A;                 /* (dim, dstate), row stride A_d_stride, dstate stride 1 */
    const void *B, *C;              /* (batch, n_groups, dstate, seqlen) */
    const float *D, *delta_bias;    /* (dim) or NULL */
    int64_t u_batch_stride, u_d_stride;
    int64_t delta_batch_stride, delta_d_stride;
    int64_t A_d_stride;
    int64_t B_batch_stride, B_group_stride, B_dstate_stride;
    int64_t C_batch_stride, C_group_stride, C_dstate_stride;
    /* forward output / backward input */
    void *out;                      /* fwd: (batch, dim, seqlen) out_dtype;  bwd: unused */
    int64_t out_batch_stride, out_d_stride;
    float *x;                       /* (batch, dim, n_chunks, dstate) fp32 contiguous; may be NULL iff n_chunks == 1 */
    /* backward only */
    const void *dout;               /* (batch, dim, seqlen) out_dtype */
    int64_t dout_batch_stride, dout_d_stride;
    void *du, *ddelta;              /* (batch, dim, seqlen) in_dtype, contiguous */
    float *dA;                      /* (dim, dstate) fp32 contiguous, ZEROED by the caller (accumulated) */
    float *dB, *dC;                 /* (batch, n_groups, dstate, seqlen) fp32 contiguous, ZEROED by the caller */
    float *dD, *ddelta_bias;        /* (dim) fp32 ZEROED by the caller, or NULL */
} xfm_scan_params_t;

int xfm_abi_version(void);
const char *xfm_strerror(int code);
const char *xfm_last_hip_error(void);
/* Profiling hook (no reference counterpart; bench.py's per-kernel timer uses it): hands two hipEvent_t to the library.  The next
 * entry point of THIS thread that launches a main kernel followed by a small finishing kernel (xfm_ss2d_bwd_ws on the wide-map
 * kernels: the scan, then the sum of the workgroups' partial dB / dC rows) records `start_event` right before and `stop_event`
 * right after the main kernel on its stream and forgets the pair.  Returns 1 if a pair handed over earlier was still pending (no
 * such launch happened since) -- pass NULL, NULL to ask and clear. */
int xfm_prof_main_kernel(void *start_event, void *stop_event);

int xfm_scan_plan(int batch, int dim, int seqlen, int dstate, int n_groups, xfm_scan_plan_t *plan);
int xfm_selective_scan_fwd(const xfm_scan_params_t *p, void *stream);
int xfm_selective_scan_bwd(const xfm_scan_params_t *p, void *stream);

/* x: (B, C, H, W) contiguous -> y: (B, 4, C, H*W) contiguous, same dtype.
 * Route k: 0 row-major, 1 column-major, 2 = reverse of 0, 3 = reverse of 1. */
int xfm_cross_scan(const void *x, void *y, int B, int C, int H, int W, int dtype, void *stream);
/* y: (B, 4, C, H*W) contiguous (in_dtype) -> x: (B, C, H*W) contiguous (out_dtype);
 * x = y0 + flip(y2) + T^-1(y1 + flip(y3)), accumulated in fp32. */
int xfm_cross_merge(const void *y, void *x, int B, int C, int H, int W, int in_dtype, int out_dtype, void *stream);
/* x, x2: (B, C, L) contiguous -> out: (B, 2, C, L): out[:,0] = x with even channels from x2,
 * out[:,1] = x2 with even channels from x. */
int xfm_swap_scan(const void *x, const void *x2, void *out, int B, int C, int L, int dtype, void *stream);

/* Route split of the x_proj output and its adjoint.  x_proj of all four routes is ONE dense GEMM on the map in natural
 * order (replaces the per-route `einsum("b k d l, k c d -> b k c l")` on the cross-scanned copies,
 * models/fusion_vmamba.py:1150-1153); xd: (B, 4, R+2N, H*W) holds route k = 2*rev + col in channels [k*(R+2N), ...).
 * split: xr (B,4,R,L) = dt_proj input, Bs / Cs (B,4,N,L), planes of the column routes (k odd) transposed to column-major
 * -- the layout contract of xfm_ss2d_fwd.  merge: gradient of xd from dxr (`dtype`) and the fp32 dBs / dCs accumulators
 * of xfm_ss2d_bwd. */
int xfm_ss2d_route_split(const void *xd, void *xr, void *Bs, void *Cs, int B, int R, int N, int H, int W, int dtype,
                         void *stream);
/* xfm_ss2d_route_split that writes the B / C rows a second time as fp32 (Bs32 / Cs32: (B, 4, N, H*W) fp32, per-route order) */
int xfm_ss2d_route_split_bc32(const void *xd, void *xr, void *Bs, void *Cs, float *Bs32, float *Cs32, int B, int R, int N, int H,
                              int W, int dtype, void *stream);
int xfm_ss2d_route_merge(const void *dxr, const float *dBs, const float *dCs, void *dxd, int B, int R, int N, int H,
                         int W, int dtype, void *stream);

/* dt_proj of the SS2D core: dts[b,k,d,l] = sum_r weight[k,d,r] * xr[b,k,r,l] (the grouped
 * `einsum("b k r l, k d r -> b k d l")` of forward_corev2, models/fusion_vmamba.py:1154-1156).  xr (B,4,R,L) and
 * dts (B,4,D,L) in `dtype` (XFM_F32 / XFM_BF16), weight (4,D,R) fp32; L % 4 == 0, R <= 64.
 * softplus_bias (4*D fp32) != NULL: the epilogue stores softplus(dts + bias) (threshold 20) -- the activated step
 * size, for xfm_ss2d_fwd/_bwd called with delta_softplus = 2. */
int xfm_ss2d_dt_proj_supported(int D, int R, int L);
/* MFMA variant for bf16 I/O (D % 32 == 0, R <= 32; xfm_ss2d_dt_proj_mfma_rp() returns the padded contraction length
 * 16 / 32, or 0 when the shape is not covered): weight_bf16 is the (4, D, R) weight in bf16 (what autocast feeds the
 * reference's einsum); same result contract as xfm_ss2d_dt_proj_fwd. */
int xfm_ss2d_dt_proj_mfma_rp(int D, int R, int L);
/* Backward of dt_proj (bf16, same shape limits, L % 4 == 0): dxr[b,k,r,l] = sum_d weight[k,d,r] * ddts[b,k,d,l] and
 * dweight[k,d,r] += sum_{b,l} ddts[b,k,d,l] * xr[b,k,r,l] (fp32, ZEROED by the caller; atomics); ddts is read once by each. */
int xfm_ss2d_dt_proj_bwd_mfma(const void *ddts, const void *xr, const void *weight_bf16, void *dxr, float *dweight, int B,
                              int D, int R, int L, void *stream);
int xfm_ss2d_dt_proj_fwd_mfma(const void *xr, const void *weight_bf16, const float *softplus_bias, void *dts, int B, int D,
                              int R, int L, void *stream);
int xfm_ss2d_dt_proj_fwd(const void *xr, const float *weight, const float *softplus_bias, void *dts, int B, int D, int R,
                         int L, int dtype, void *stream);

/* Depthwise 3x3 convolution, padding 1, stride 1, optional bias, optionally fused with SiLU.
 * x, y, dy, dx: (B, D, H, W) contiguous in `dtype`; weight: (D, 1, 3, 3) fp32; bias: (D) fp32 or NULL.
 * Backward recomputes the pre-activation; dweight (D*9) and dbias (D) are fp32 and must be ZEROED. */
int xfm_dwconv3x3_fwd(const void *x, const float *weight, const float *bias, void *y, int B, int D, int H, int W,
                      int dtype, int silu, void *stream);
int xfm_dwconv3x3_bwd(const void *x, const float *weight, const float *bias, const void *dy, void *dx, float *dweight,
                      float *dbias, int B, int D, int H, int W, int dtype, int silu, void *stream);

/*
 * The same operator on TOKEN-MAJOR maps x (B, H, W, C) bf16 (csrc/dwconv_tok.hip; 14 x 14 and 7 x 7 maps, C % 8 == 0): with the
 * depthwise stage token-major the SS2D block of the short-map stages never leaves the token layout (xfm_ss2dc_fwd/_bwd with
 * x_tokens / y_tokens).  weight (C, 1, 3, 3) = (C, 9) fp32, bias (C) fp32 or NULL; y = silu(conv(x) + bias) bf16.
 * Backward: dz_ws (B, H, W, C) bf16 and part_ws (B * H, 10, C) fp32 are caller workspaces (the gradient of the pre-activation
 * and one partial row of the weight / bias sums per map row, folded by a small second kernel); dweight (C, 9) and dbias (C, or
 * NULL) fp32 ZEROED by the caller (the fold adds), dx (B, H, W, C) bf16.  All pointers 16-byte aligned.
 */
int xfm_dwconv3x3_tokens_supported(int H, int W, int C);
int xfm_dwconv3x3_tokens_fwd(const void *x, const float *weight, const float *bias, void *y, int B, int H, int W, int C, void *stream);
int xfm_dwconv3x3_tokens_bwd(const void *x, const float *weight, const float *bias, const void *dy, void *dz_ws, void *dx,
                             float *part_ws, float *dweight, float *dbias, int B, int H, int W, int C, void *stream);

/*
 * 3 x 3, stride-2, padding-1 convolution on TOKEN-MAJOR maps (csrc/conv_tok.hip): the second convolution of the patch embedding
 * and the three downsample layers of the trunk (reference models/fusion_vmamba.py:1504-1518 `_make_patch_embed_v2`, :1531-1538
 * `_make_downsample_v3`; there `nn.Conv2d(dim, out_dim, 3, 2, 1)` on (B, C, H, W)), without bias (the caller folds it into the
 * LayerNorm that follows).  x (B, H, W, C), y (B, H/2, W/2, O), weight (O, 3, 3, C) -- the channels_last memory of the
 * (O, C, 3, 3) parameter --, all bf16; H, W even, C % 8 == 0, O % 8 == 0, pointers 16-byte aligned.
 *   _fwd         fills col (B H/2 W/2, 9 C) bf16 (the 3 x 3 neighbourhoods as rows: keep it for _bwd_weight) and y = col . W^T
 *   _bwd_data    dcol (B H/2 W/2, 9 C) bf16 is a workspace; dx (B, H, W, C) bf16
 *   _bwd_weight  dweight (O, 3, 3, C) fp32 ZEROED by the caller (fp32 atomics add into it)
 */
int xfm_conv3x3s2_tokens_supported(int C, int O, int H, int W);
int xfm_conv3x3s2_tokens_fwd(const void *x, const void *weight, void *col, void *y, int B, int H, int W, int C, int O, void *stream);
int xfm_conv3x3s2_tokens_bwd_data(const void *dy, const void *weight, void *dcol, void *dx, int B, int H, int W, int C, int O,
                                  void *stream);
int xfm_conv3x3s2_tokens_bwd_weight(const void *dy, const void *col, float *dweight, int B, int H, int W, int C, int O, void *stream);
/* The same weight gradient straight from the input map x (B, H, W, C) -- no `col` rows -- for layers whose forward pass ran
 * elsewhere (csrc/wgrad_gemm.hip: the token x token kernel gathers the window taps itself).  _supported: B H/2 W/2 a multiple of 64
 * and at least 2048, H/2 >= 64 / (W/2) + 2. */
int xfm_conv3x3s2_tokens_bwd_weight_x_supported(int B, int H, int W, int C, int O);
/* The FIRST convolution of the patch embedding when its CI input channels are replicas of ONE channel (reference
 * net_fusionmamba.py:88-104 `x.expand(-1, 3, -1, -1)` in front of models/fusion_vmamba.py:1504-1518): x (B, H, W) bf16 is that
 * channel, weight (O, CI, 3, 3) bf16 the parameter's shadow (summed over CI inside), y / dy (B, H/2, W/2, O) bf16;
 * dweight9 (O, 9) fp32 ZEROED by the caller -- the gradient of every one of the CI input-channel slices of the parameter --
 * and ws: xfm_conv3x3s2_gray_ws_floats(O) fp32, ZEROED by the caller (replicas of the sums: adds to one address retire one by one).
 * O % 8 == 0 with 192 % (O / 8) == 0, H, W even; no data gradient (the image is an input). */
int xfm_conv3x3s2_gray_supported(int O, int H, int W);
int xfm_conv3x3s2_gray_ws_floats(int O);
int xfm_conv3x3s2_gray_fwd(const void *x, const void *weight, void *y, int B, int H, int W, int CI, int O, void *stream);
int xfm_conv3x3s2_gray_bwd_weight(const void *dy, const void *x, float *dweight9, float *ws, int B, int H, int W, int O,
                                  void *stream);
int xfm_conv3x3s2_tokens_bwd_weight_x(const void *dy, const void *x, float *dweight, int B, int H, int W, int C, int O,
                                      void *stream);

/* LayerNorm over C of x (B, C, L) [NCHW with L = H*W], eps inside the rsqrt, affine weight/bias (C) fp32 (bias may
 * be NULL).  y may be a narrower dtype than x (the consumer GEMM's).  mean / rstd: (B, L) fp32, written by fwd and
 * read by bwd.  bwd: dx in x_dtype; dweight / dbias fp32, ZEROED by the caller (dbias may be NULL). */
int xfm_layernorm2d_fwd(const void *x, const float *weight, const float *bias, void *y, float *mean, float *rstd, int B,
                        int C, int L, float eps, int x_dtype, int y_dtype, void *stream);
int xfm_layernorm2d_bwd(const void *x, const float *weight, const void *dy, const float *mean, const float *rstd,
                        void *dx, float *dweight, float *dbias, int B, int C, int L, int x_dtype, int y_dtype,
                        void *stream);
/* The backward with the weight / bias gradient left as PARTIAL ROWS by the dx kernel (no second pass over x and dy): parts
 * (xfm_layernorm2d_bwd_parts_blocks(B, C, L, x_dtype, y_dtype), 2, C) fp32, row pair j = [sum dy * xhat | sum dy] over the
 * positions of workgroup j, every element written (fold with xfm_partial_sums_multi or a sum over j).  0 blocks: the kernel
 * chosen for the shape cannot -- use xfm_layernorm2d_bwd. */
int xfm_layernorm2d_bwd_parts_blocks(int B, int C, int L, int x_dtype, int y_dtype);
int xfm_layernorm2d_bwd_parts(const void *x, const float *weight, const void *dy, const float *mean, const float *rstd,
                              void *dx, float *parts, int B, int C, int L, int x_dtype, int y_dtype, void *stream);

/* LayerNorm2d on 7 x 7 maps with wide rows (16 <= L <= 51, C % 64 == 0, C >= 512), the slab form: workgroups own a (sample,
 * 64-channel slab) and read it as one coalesced run; two kernels with a workspace of xfm_layernorm2d_ws_floats(B, C, L) fp32
 * values between them (0: not covered, use the entries above).  Results as xfm_layernorm2d_fwd / xfm_layernorm2d_bwd_parts;
 * the partial rows of the backward are (B, 2, C), one row pair per sample (parts may be NULL: no weight / bias gradient). */
int xfm_layernorm2d_ws_floats(int B, int C, int L);
int xfm_layernorm2d_fwd_ws(const void *x, const float *weight, const float *bias, void *y, float *mean, float *rstd,
                           float *workspace, int B, int C, int L, float eps, int x_dtype, int y_dtype, void *stream);
/* Backward: workspace size and number of partial row pairs of xfm_layernorm2d_bwd_parts_ws -- the slab form above, or the split
 * form for 14 x 14 maps with 384 channels (nothing cached, two coalesced passes; one row pair per 64 positions); 0: not covered. */
int xfm_layernorm2d_bwd_ws_floats(int B, int C, int L);
int xfm_layernorm2d_bwd_ws_blocks(int B, int C, int L);
int xfm_layernorm2d_bwd_parts_ws(const void *x, const float *weight, const void *dy, const float *mean, const float *rstd,
                                 void *dx, float *parts, float *workspace, int B, int C, int L, int x_dtype, int y_dtype,
                                 void *stream);

/* Residual add + DropPath scale + LayerNorm over C of a TOKEN-MAJOR stream (rows = B*rows_per_sample tokens of C
 * channels, contiguous), one pass:   x_new = x + scale[b]*y ;  h = LayerNorm_C(x_new)*weight + bias.
 * Replaces `x = x + self.drop_path(branch(x))` followed by the next `self.norm2(x)` / `self.norm(x)` of
 * VSSBlock._forward (models/fusion_vmamba.py:1325-1337).  x_new, mean, rstd, weight, bias fp32; y, h (and dh, dy)
 * in `dtype` (XFM_F32 or XFM_BF16).  y == NULL: plain LayerNorm of x (x_new unused; x and dx in `x_dtype`, fp32 or
 * bf16 -- e.g. a convolution's bf16 output); with y, x must be fp32.  scale: (B) fp32 or NULL (= 1).
 * Supported C: 4*G*NV with G in {4,8,16,32,64}, NV in {3,4} (48..768 by doubling, 64..1024 by doubling); otherwise
 * XFM_ELIMIT (`xfm_add_layernorm_rows_supported` tells).
 * bwd: dh = gradient of h, dres = gradient arriving on x_new from its other consumers (fp32, or NULL);
 * dx = gradient of x (== gradient of x_new), dy = scale[b]*dx in `dtype` (NULL when the forward had no y);
 * dweight / dbias (C) fp32 are OVERWRITTEN (dbias may be NULL); workspace: 3*C*xfm_add_layernorm_rows_bwd_blocks()
 * floats.  pre_bias (C, fp32): with y == NULL it is added to x before the norm -- the bias of the convolution that
 * produced x (patch-embed / downsample convs, models/fusion_vmamba.py:1504-1538); with y it is added to y inside the
 * scaled sum, x_new = x + scale[b]*(y + pre_bias) -- the bias of the linear layer that produced y (Mlp.fc2).  Its
 * gradient dpre_bias (the column sums of dx, resp. of dy) comes out of the same backward pass.  x_new: the forward's x_new (or its x when y was NULL). */
int xfm_add_layernorm_rows_supported(int C);
int xfm_add_layernorm_rows_bwd_blocks(int rows, int C);
int xfm_add_layernorm_rows_fwd(const void *x, const void *y, const float *scale, const float *pre_bias,
                               const float *weight, const float *bias, float *x_new, void *h, float *mean, float *rstd, int B, int rows_per_sample, int C,
                               float eps, int x_dtype, int dtype, void *stream);
int xfm_add_layernorm_rows_bwd(const void *x_new, const float *pre_bias, const float *weight, const void *dh,
                               const float *dres, const float *mean, const float *rstd, const float *scale, void *dx,
                               void *dy, float *dweight, float *dbias, float *dpre_bias, float *workspace, int B, int rows_per_sample, int C,
                               int x_dtype, int dtype, void *stream);

/* The same row LayerNorm without a residual branch and with the exact-erf GELU INSIDE: h = gelu(LayerNorm(x + pre_bias)) --
 * norm -> GELU of the patch embedding (reference models/fusion_vmamba.py:1504-1518).  The backward pass recomputes the
 * pre-activation from x, mean, rstd, weight and bias (so it takes `bias` too) and multiplies dh by gelu' first; workspace,
 * dweight / dbias / dpre_bias (NULL: partial rows left for xfm_partial_sums_multi) as xfm_add_layernorm_rows_bwd. */
int xfm_layernorm_rows_gelu_fwd(const void *x, const float *pre_bias, const float *weight, const float *bias, void *h, float *mean,
                                float *rstd, int rows, int C, float eps, int x_dtype, int dtype, void *stream);
int xfm_layernorm_rows_gelu_bwd(const void *x, const float *pre_bias, const float *weight, const float *bias, const void *dh,
                                const float *mean, const float *rstd, void *dx, float *dweight, float *dbias, float *dpre_bias,
                                float *workspace, int rows, int C, int x_dtype, int dtype, void *stream);

/* Element-wise pieces of the Mlp (models/fusion_vmamba.py:135-153, fc1 -> GELU -> fc2) between the library GEMMs, on
 * (rows, C) row-major token-major activations in `dtype` (XFM_F32 / XFM_BF16; C % 4 resp. % 8 == 0, C <= 8192):
 *   xfm_bias_gelu_fwd: g = gelu(z + bias)  (exact erf GELU = nn.GELU(); bias (C) fp32 or NULL)
 *   xfm_bias_gelu_bwd: dz = dg * gelu'(z + bias) and dbias[c] = sum_rows dz[., c]  (fp32, OVERWRITTEN) in one pass
 *   xfm_colsum:        out[c] = sum_rows x[., c]  (fp32, OVERWRITTEN) -- bias gradient of a plain linear layer
 * workspace: C * xfm_colsum_blocks(rows, C, dtype) floats. */
int xfm_colsum_blocks(long long rows, int C, int dtype);
int xfm_bias_gelu_fwd(const void *z, const float *bias, void *g, long long rows, int C, int dtype, void *stream);
int xfm_bias_gelu_bwd(const void *z, const float *bias, const void *dg, void *dz, float *dbias, float *workspace,
                      long long rows, int C, int dtype, void *stream);
int xfm_colsum(const void *x, float *out, float *workspace, long long rows, int C, int dtype, void *stream);

/*
 * End-of-stage residual settle on the token-major stream (reference models/fusion_vmamba.py:1325-1337: the last
 * `x = x + self.drop_path(self.mlp(self.norm2(x)))` of a stage, whose result feeds the downsample convolution):
 *   fwd: out[r, c] = x[r, c] + scale[b(r)] * (y[r, c] + y_bias[c])   x fp32, y / out bf16 or fp32 (out = the consumer's dtype)
 *   bwd: dx = dout (fp32), dy = scale[b] * dout (y's dtype)            (d y_bias = column sum of dy: xfm_colsum)
 * rows r = b * rows_per_sample + i; scale (B) and y_bias (C) may be NULL; C % 8 == 0.
 */
int xfm_residual_settle_fwd(const float *x, const void *y, const float *scale, const float *y_bias, void *out, int B,
                            int rows_per_sample, int C, int y_dtype, int out_dtype, void *stream);
int xfm_residual_settle_bwd(const void *dout, const float *scale, float *dx, void *dy, int B, int rows_per_sample, int C,
                            int y_dtype, int out_dtype, void *stream);

/*
 * Deferred column sums.  xfm_add_layernorm_rows_bwd (dweight == NULL), xfm_bias_gelu_bwd (dbias == NULL) and xfm_colsum
 * (out == NULL) then only leave their per-workgroup partial rows in `workspace` -- layout [block][parts][C] with
 * parts = 2 (dw, db) or 3 (+ d pre_bias, whenever pre_bias is given) for the LayerNorm, 1 for the other two, `block`
 * counts from xfm_add_layernorm_rows_bwd_blocks / xfm_colsum_blocks -- and ONE xfm_partial_sums_multi launch folds the
 * partial rows of any number of such producers (the parameter gradients of autograd of nn.LayerNorm / nn.Linear biases,
 * reference models/fusion_vmamba.py:135-153, 1325-1337: nothing reads them before the optimizer).
 *   jobs:   device array, 6 int64 per job {part, out0, out1, out2 (device addresses; outs may be 0), nblk | (int64)C << 32,
 *           parts}:  out_k[c] = sum over j < nblk of part[(j * parts + k) * C + c]
 *   blocks: device int32 array, one entry per workgroup: job index | (64-column block of its parts * C columns) << 16
 */
int xfm_partial_sums_multi(const void *jobs, const void *blocks, int nblocks, void *stream);

/* tokens -> planes ahead of conv2d together with the squeeze pooling of ShallowFuse_SS2Dv4.forward (reference
 * models/fusion_vmamba.py:853-871: `self.avg_pool(xp)`): planes (B, C, R) = t (B, R, C)^T, pooled (B, C) = mean_r t[b, r, c], bf16;
 * _bwd: d t (B, R, C) = d planes^T + d pooled / R.  R <= 64, C % 64 == 0, 16-byte aligned. */
int xfm_pooled_transpose_fwd(const void *t, void *planes, void *pooled, int B, int R, int C, void *stream);
int xfm_pooled_transpose_bwd(const void *dplanes, const void *dpooled, void *dt, int B, int R, int C, void *stream);

/* Squeeze gate (.) map with the layout change for out_proj: out (B, R, C) tokens = yy (B, C, R) planes * gate (B, C), bf16
 * (`y * gate` of ShallowFuse_SS2Dv4.forward, reference models/fusion_vmamba.py:870-871); _bwd from g (B, R, C):
 * d yy (B, C, R) = g^T * gate, d gate (B, C) = sum_r g[b, r, c] * yy[b, c, r].  8 <= R <= 64, C % 64 == 0, 16-byte aligned. */
int xfm_gated_transpose_fwd(const void *yy, const void *gate, void *out, int B, int R, int C, void *stream);
int xfm_gated_transpose_bwd(const void *g, const void *yy, const void *gate, void *dyy, void *dgate, int B, int R, int C,
                            void *stream);

/* The deep fusion block's three streams from its two normalised views: out (3, M) = [n[0] | n[1] | (n[0] + n[1]) / 2] for
 * n (2, M) fp32 (M % 4 == 0), in out_dtype (fp32 / bf16) -- `x_fuse = (x + x2) / 2` ahead of in_proj_sec (reference
 * models/fusion_vmamba.py, Cross_SS2Dv5.forward) with the cat and the GEMM's cast in one kernel; _bwd: dn[k] = g[k] + g[2] / 2. */
int xfm_views_avg_stack_fwd(const float *n, void *out, long long M, int out_dtype, void *stream);
int xfm_views_avg_stack_bwd(const void *g, float *dn, long long M, int g_dtype, void *stream);

/* Training-mode BatchNorm2d of the shallow fusion block on the token-major stream, all views in one call: the SAME nn.BatchNorm2d
 * applied to view 1, then to view 2 (reference models/fusion_vmamba.py:906-907).  x (V, N, C) fp32 with N = B*H*W rows per view;
 * batch statistics per view (biased variance in y, as F.batch_norm), running statistics updated view after view with
 * `momentum` and the unbiased variance (NULL: not tracked); y in y_dtype (fp32 / bf16: the following GEMM's dtype); mean / rstd
 * (V, C) are kept for the backward pass.  Backward: dx (V, N, C) fp32, dgamma / dbeta (C) = totals over the views (every element
 * written; NULL: not wanted).  workspace: xfm_bn_tokens_ws_floats(V, N, C) fp32 values.  C % 64 == 0, V <= 8. */
int xfm_bn_tokens_supported(int V, int N, int C);
int xfm_bn_tokens_ws_floats(int V, int N, int C);
int xfm_bn_tokens_fwd(const float *x, const float *gamma, const float *beta, float *running_mean, float *running_var,
                      float momentum, float eps, void *y, float *mean, float *rstd, float *workspace, int V, int N, int C,
                      int y_dtype, void *stream);
int xfm_bn_tokens_bwd(const float *x, const void *dy, const float *gamma, const float *mean, const float *rstd, float *dx,
                      float *dgamma, float *dbeta, float *workspace, int V, int N, int C, int dy_dtype, void *stream);

/* (B, R, C) tokens <-> (B, C, R) planes of 2-byte elements on short maps (R <= 64 positions, C % 64 == 0, 16-byte aligned
 * tensors): the NHWC <-> NCHW permutes around the 7 x 7 SS2D blocks (reference models/fusion_vmamba.py:594-601, 853-857) as
 * one streaming kernel.  tokens_to_planes != 0: src (B, R, C) -> dst (B, C, R); 0: src (B, C, R) -> dst (B, R, C). */
int xfm_transpose_short_supported(int R, int C);
int xfm_transpose_short(const void *src, void *dst, int B, int R, int C, int tokens_to_planes, void *stream);
/* dst (B, C, R) bf16 planes += transpose of src (B, R, C) bf16 tokens (fp32 add, one rounding): x_proj's data gradient at 7 x 7. */
int xfm_transpose_short_add_bf16(const void *src, void *dst, int B, int R, int C, void *stream);

/*
 * Skinny token-major linear layer on MFMA (csrc/tokens_gemm.hip):  y[T, out] = x[T, con] . W^T (+ bias), bf16 in / out,
 * fp32 accumulation -- `F.linear` of Mlp.fc1 / fc2 (reference models/fusion_vmamba.py:135-153) and its backward data
 * product at the 56x56 stage, where the product is HBM-bound and the whole weight fits in LDS.
 *   weight_bf16: (out, con) row-major when weight_transposed == 0 (forward: the Linear weight itself);
 *                (con, out) row-major when weight_transposed != 0 (backward: dx = dy . W reads the same Linear weight).
 *   bias: (out) fp32 or NULL.   xfm_tokens_gemm_supported(con, out) tells which (con, out) pairs are built.
 */
int xfm_tokens_gemm_supported(int con, int out);
int xfm_tokens_gemm(const void *x, const void *weight_bf16, const float *bias, void *y, long long T, int con, int out,
                    int weight_transposed, void *stream);
/*
 * The WIDE side of the Mlp at the later trunk stages (con -> out = 4 con: fc1's forward product and fc2's data gradient,
 * reference models/fusion_vmamba.py:135-153) with the GELU fused into the product's epilogue; a workgroup keeps a chunk of
 * output columns of the weight in LDS and walks the token rows.  epilogue:
 *   0: y = x W^T + bias
 *   1: y = z = x W^T (bf16, no bias: what the backward pass keeps) and y2 = gelu(z + bias), exact erf GELU (nn.GELU())
 *   2: y = dz = bf16(x W^T) * gelu'(zin + bias)   (x = dy of fc2, weight = fc2's (con, out) weight with weight_transposed = 1)
 * -- the values of the unfused chain (z / dg rounded to bf16 before the activation).  xfm_tokens_gemm2_supported(con, out).
 */
int xfm_tokens_gemm2_supported(int con, int out);
int xfm_tokens_gemm2(const void *x, const void *weight_bf16, const float *bias, void *y, void *y2, const void *zin, long long T,
                     int con, int out, int weight_transposed, int epilogue, void *stream);
/* The same with epilogue 2 that ALSO leaves the column sums of dz (fc1's bias gradient, reference models/fusion_vmamba.py:135-153
 * through autograd) as partial rows: colpart (xfm_tokens_gemm2_parts_blocks(T, con, out), out) fp32, one row per 128-token tile,
 * every element written; 0 blocks = not available for the shape (use xfm_tokens_gemm2 + xfm_colsum). */
int xfm_tokens_gemm2_parts_blocks(long long T, int con, int out);
int xfm_tokens_gemm2_parts(const void *x, const void *weight_bf16, const float *bias, void *dz, const void *zin, float *colpart,
                           long long T, int con, int out, int weight_transposed, void *stream);

/*
 * The same kernel family for the layout-changing 1x1 projections of an SS2D block (in_proj: tokens -> planes, out_proj:
 * planes -> tokens; reference Linear2d = F.conv2d with a 1x1 kernel, models/fusion_vmamba.py:42-45, :1190-1206) and their
 * backward data products:  in_planes == 0: x (B, L, con) tokens -> y (B, out, L) planes;  in_planes != 0: x (B, con, L)
 * planes -> y (B, L, out) tokens.  weight / weight_transposed / bias as in xfm_tokens_gemm.  L % 8 == 0, (B * L) % 32 == 0;
 * con = out = 96 or 192.
 */
int xfm_proj_gemm_supported(int con, int out, int L);
int xfm_proj_gemm(const void *x, const void *weight_bf16, const float *bias, void *y, int B, int L, int con, int out,
                  int in_planes, int weight_transposed, void *stream);
/* tokens -> planes with accumulation, y (B, out, L) += W . x (B, L, con): the x_proj data gradient of a channel-lane SS2D
 * block (dx += Wx^T . d x_dbl^T); tiled form only: con % 64 == 0, out % 128 == 0, L % 4 == 0, L >= 64. */
int xfm_proj_gemm_accumulate(const void *x, const void *weight_bf16, void *y, int B, int L, int con, int out,
                             int weight_transposed, void *stream);

/*
 * Plane-major on both sides: y (B, out, L) = W . x (B, con, L), or y += W . x when accumulate != 0 (the existing bf16
 * values are widened, added in fp32 and rounded once).  (con, out) = (96, 32): the x_proj of the four routes evaluated
 * on the natural map at the 56x56 stage (reference `einsum("b k d l, k c d -> b k c l", xs, x_proj_weight)`,
 * models/fusion_vmamba.py:1150-1152); (32, 96) with weight_transposed and accumulate: its backward, dx += W^T . d x_dbl.
 */
int xfm_planes_gemm_supported(int con, int out, int L);
int xfm_planes_gemm(const void *x, const void *weight_bf16, const float *bias, void *y, int B, int L, int con, int out,
                    int weight_transposed, int accumulate, void *stream);

/*
 * Fused SS2D core: y[b,d,p] = sum_k scan_k(...)[b,d,.] gathered back to position p, i.e.
 * cross-scan + 4-route selective scan + cross-merge in ONE kernel; the (B,4,D,L) scan inputs /
 * fp32 scan outputs of the unfused chain never reach HBM.
 *
 * Layout contract.  x, y, dy, dx are feature maps in natural row-major order (p = h*W + w).
 * The per-route tensors dts, Bs, Cs (and their gradients) are stored, per route k, in the order in
 * which the route's FORWARD sibling walks the map:
 *     k = 0, 2 : row-major    index t = h*W + w      (route 2 scans this sequence backwards)
 *     k = 1, 3 : column-major index t = w*H + h      (route 3 scans this sequence backwards)
 * so every route streams its operands contiguously.  (The host produces routes 1/3 in column-major
 * order for free by transposing the small x_proj output before dt_proj; xfmamba_amd/fusion_vmamba.py.)
 */
typedef struct {
    int batch, d_inner, H, W, dstate;
    int delta_softplus;    /* 0: step = dts + bias; 1: step = softplus(dts + bias) (models/csms6s.py:49-50); 2: dts already
                            * holds softplus(raw + bias) (xfm_ss2d_dt_proj_fwd epilogue): delta_bias is not read, and
                            * ddts / ddelta_bias are still the gradients of the RAW pre-activation / of the bias;
                            * 3: dt_proj INSIDE the kernel (models/fusion_vmamba.py:1147-1150):
                            * step = softplus(dt_w . xrt + bias) formed per position, dts is NOT read (may be NULL) and the
                            * (batch, 4, d_inner, L) step sizes never reach HBM; ddts is still written (gradient of the raw
                            * pre-activation, the operand of xfm_ss2d_dt_proj_bwd_mfma).  Shapes: xfm_ss2d_dtfused_rank() */
    int in_dtype;          /* of x, dts, Bs, Cs, dx, ddts */
    int out_dtype;         /* of y / dy (fp32 = "oflex") */
    const void *x;         /* (batch, d_inner, H*W) */
    const void *dts;       /* (batch, 4, d_inner, H*W)   delta pre-activation, per-route order */
    const void *Bs, *Cs;   /* (batch, 4, dstate, H*W)    per-route order */
    const float *A;        /* (4*d_inner, dstate) */
    const float *D, *delta_bias; /* (4*d_inner) */
    void *y;               /* (batch, d_inner, H*W) */
    float *chk;            /* (batch, 4, d_inner, n_chunks, dstate) fp32 chunk states written by fwd, read by
                              bwd; may be NULL iff xfm_ss2d_plan() says n_chunks == 1 */
    /* backward */
    const void *dy;        /* (batch, d_inner, H*W) out_dtype */
    void *dx;              /* (batch, d_inner, H*W) in_dtype */
    void *ddts;            /* (batch, 4, d_inner, H*W) in_dtype, per-route order */
    float *dBs, *dCs;      /* (batch, 4, dstate, H*W) fp32 ZEROED, per-route order */
    float *dA;             /* (4*d_inner, dstate) fp32 ZEROED */
    float *dD, *ddelta_bias; /* (4*d_inner) fp32 ZEROED */
    /* delta_softplus == 3 only (ignored otherwise) */
    const void *xrt;       /* the dt_proj input rows of x_proj's output in the blocked form xfm_ss2d_xr_rows writes:
                              (batch, 4, ceil(H*W / 512), dt_rank_p, 64, 8) in_dtype, per-route order */
    const void *dt_w;      /* (4, d_inner, dt_rank_p) in_dtype: dt_projs_weight rows, zero-padded to dt_rank_p */
    int dt_rank_p;         /* xfm_ss2d_dtfused_rank(...) */
    int bc_f32;            /* 1: Bs32 / Cs32 below hold the B / C rows as fp32 too (xfm_ss2d_route_split_bc32); forward AND backward of
                            * such a call are served by csrc/ss2d_w.hpp (xfm_ss2d_bc_f32() says for which shapes), whose
                            * backward reads the fp32 rows; chk then holds one state per 8-position chunk (xfm_ss2d_plan) */
    const float *Bs32, *Cs32;   /* (batch, 4, dstate, H*W) fp32, per-route order; read only when bc_f32 */
} xfm_ss2d_params_t;

/* 1 when the fused core wants the fp32 copies of the B / C rows for this shape in the activated-step-size mode
 * (delta_softplus == 2): bf16 I/O, d_state 1, the wide square maps csrc/ss2d_w.hpp covers (56, 28, 48, 24).  The caller then
 * produces the rows with xfm_ss2d_route_split_bc32 and sets bc_f32 / Bs32 / Cs32 in the forward and the backward call; with
 * bc_f32 == 0 the same call is served by the kernels of csrc/ss2d_l3.hip. */
int xfm_ss2d_bc_f32(int batch, int d_inner, int H, int W, int dstate, int in_dtype);

/* dt_proj inside the fused SS2D core (delta_softplus == 3): the padded rank to lay xrt / dt_w out with (dt_rank rounded up
 * to even), or 0 when this shape / dtype has no such kernel (bf16 I/O, d_state 1, the wide square maps of csrc/ss2d_l3.hip,
 * dt_rank <= 16) -- the caller then materialises dts (xfm_ss2d_dt_proj_fwd*) and uses mode 2. */
int xfm_ss2d_dtfused_rank(int batch, int d_inner, int H, int W, int dstate, int dt_rank, int in_dtype);
/* xr (B4, R, L) -> xrt (B4, ceil(L / 512), Rp, 64, 8): the copy of the (small) dt_proj input rows that the mode-3 kernels read,
 * blocked by their chunk geometry: a chunk row is 512 route positions = 64 chunks of 8; piece j (8 values = 16 bytes) of chunk
 * c holds values 8 j .. 8 j + 7 of the chunk's 8 positions x Rp ranks, position-major (value p * Rp + r), so that every load
 * instruction of a wave (lane = chunk) is one contiguous KB.  Positions past L and ranks R .. Rp - 1 are zero.  B4 = batch * 4
 * routes, rows already in per-route order (xfm_ss2d_route_split); 16-bit dtypes, L % 8 == 0, Rp 6 or 12. */
int xfm_ss2d_xr_rows(const void *xr, void *xrt, long long B4, int R, int Rp, int L, int dtype, void *stream);

/* The chunking (and so the size of chk) depends on the I/O dtype: 16-byte vectors per lane where rows allow. */
int xfm_ss2d_plan(int batch, int d_inner, int H, int W, int dstate, int in_dtype, xfm_scan_plan_t *plan);
int xfm_ss2d_fwd(const xfm_ss2d_params_t *p, void *stream);
int xfm_ss2d_bwd(const xfm_ss2d_params_t *p, void *stream);
/* The same backward with a caller-owned scratch buffer.  Where the kernel family supports it (the wide-map kernels of
 * csrc/ss2d_l3.hip) each workgroup then STORES its partial dB / dC sums there and a second small kernel sums them into dBs /
 * dCs (which need not be zeroed then) -- plain stores and one streaming pass instead of float atomics, which the chip
 * retires at ~1.3 TB/s (51 MB per 56 x 56 launch).  xfm_ss2d_bwd_ws_bytes: the size to pass (0: no use for a workspace;
 * xfm_ss2d_bwd_ws then behaves as xfm_ss2d_bwd).  Same adjoint as selective_scan_bwd_kernel.cuh:141-273. */
size_t xfm_ss2d_bwd_ws_bytes(const xfm_ss2d_params_t *p);
int xfm_ss2d_bwd_ws(const xfm_ss2d_params_t *p, void *workspace, size_t workspace_bytes, void *stream);

/*
 * "Channel-lane" fused SS2D core for short square maps (5x5 ... 14x14; csrc/ss2d_chan.hip): x_proj output ->
 * dt_proj (MFMA, in-kernel) -> softplus -> n_routes-way selective scan -> cross-merge, forward and backward, so that the
 * (B,4,D,L) step-size tensor of `forward_corev2` (reference models/fusion_vmamba.py:1150-1172; fusion blocks :490-540,
 * :813-833) never exists.  One lane owns one channel and walks the sequence; the two halves of a wavefront are two samples.
 *   x      (batch, d_inner, L) bf16 planes, natural order (the depthwise-conv + SiLU output).
 *   xdbl   (batch, L, n_routes*C2p) bf16 TOKEN-MAJOR x_proj output evaluated on the natural map.  Route k owns columns
 *          [k*C2p, (k+1)*C2p), C2p = Rp8 + (dstate == 1 ? 8 : 2*dstate), Rp8 = dt_rank rounded up to 8:
 *          [0, dt_rank) dt_proj input, [Rp8, Rp8+dstate) B, then C at Rp8+1 (dstate 1) or Rp8+dstate; other columns 0.
 *   wdt    (n_sets, d_inner, Rp8) bf16 dt_proj weight (reference dt_projs_weight (K, D, R)), zero columns beyond dt_rank
 *          (none when dt_rank % 8 == 0: the cast weight itself).  n_sets = 4 with n_routes == 4 (route k uses set k); with n_routes == 1 (one forward
 *          row-major route per sample: the two views of the shallow swap block as 2B samples) sample sb uses set sb / wdiv.
 *   A (n_sets*d_inner, dstate), D / delta_bias (n_sets*d_inner) fp32.
 *   c_mod > 0: sample sb reads its C columns from sample c_off + sb % c_mod (deep fusion block: the view streams read
 *          their state through the fused stream's C, reference :536-538, :567-569) and its dC is added there.
 *   y      (batch, d_inner, L) fp32 = sum over the routes, gathered back to natural order (n_routes == 1: that route).
 *   chk    (batch, n_routes, xfm_ss2dc_nsteps(H, W, dstate), dstate, d_inner) fp32 workspace written by fwd, read by bwd.
 * backward: dy (batch, d_inner, L) fp32 -> dx (batch, d_inner, L) bf16; ddts (batch, n_routes, L, d_inner) bf16 = gradient
 *   of the RAW step size (before bias + softplus), natural position order, channel fastest; dBC (batch, n_routes, 2,
 *   dstate, L) fp32 ZEROED: gradients of the B (index 0) and C (index 1) columns, natural order; dA, dD, ddelta_bias fp32
 *   ZEROED.  The dt_proj / x_proj weight gradients and d xdbl follow from ddts and dBC by dense products (host side).
 * Supported: H == W in {7, 12, 14} with dstate 1 (n_routes 4); H == W in {5, 7, 12} with dstate 16 (n_routes 4 or 1);
 * d_inner % 32 == 0; dt_rank <= 64.  xfm_ss2dc_supported() tells.
 */
typedef struct {
    int batch, d_inner, H, W, dstate, dt_rank, n_routes;
    int c_mod, c_off, wdiv;
    int y_tokens;          /* != 0: y and dy are TOKEN-MAJOR (batch, L, d_inner) fp32 -- what a row LayerNorm / token GEMM behind the
                              scan reads and hands back (out_norm + out_proj, models/fusion_vmamba.py:1186-1205), so that neither is a
                              layout-changing operator; only where xfm_ss2dc_ytokens_supported() says so, XFM_ELIMIT otherwise */
    int x_tokens;          /* != 0: x and dx are TOKEN-MAJOR (batch, L, d_inner) bf16 as well (a token-major depthwise convolution in
                              front of the scan: in_proj / x_proj and their gradients are plain token GEMMs then); same condition */
    const void *x, *xdbl, *wdt;
    const void *zeros;     /* >= 256 bytes of zeros, 16-byte aligned (k-slots of the sibling route in the stacked product) */
    const float *A, *D, *delta_bias;
    void *y;
    float *chk;
    const void *dy;
    void *dx, *ddts;
    float *dBC, *dA, *dD, *ddelta_bias;
} xfm_ss2dc_params_t;
int xfm_ss2dc_supported(int H, int W, int dstate, int n_routes, int d_inner, int dt_rank);
int xfm_ss2dc_ytokens_supported(int H, int W, int dstate, int n_routes);
int xfm_ss2dc_nsteps(int H, int W, int dstate);
int xfm_ss2dc_fwd(const xfm_ss2dc_params_t *p, void *stream);
int xfm_ss2dc_bwd(const xfm_ss2dc_params_t *p, void *stream);
/* The two dense products behind xfm_ss2dc_bwd (n_routes == 4), both on MFMA, each reading ddts once (the backward of the
 * dt_proj einsum, reference models/fusion_vmamba.py:1154-1156, in the token-major layout):
 *   dxdbl (batch, L, 4*C2p) bf16, every column written: [0, Rp8) = ddts . W_dt (contraction over the channels; wdt is the
 *         (4, d_inner, Rp8) bf16 weight xfm_ss2dc_fwd/_bwd take), the B / C columns from dBC
 *         (batch, 4, 2, dstate, L) fp32, zeros elsewhere;
 *   dwdt  (4, d_inner, dt_rank) fp32 ZEROED += sum over batch and positions of ddts x (dt_proj input columns of xdbl). */
int xfm_ss2dc_post(const void *ddts, const void *xdbl, const void *wdt, const float *dBC, void *dxdbl, float *dwdt,
                   int batch, int d_inner, int L, int dt_rank, int dstate, void *stream);

/*
 * BASELINE.json configs[4]: x_proj / out_proj of an SS2D block with fp8 (OCP e4m3fn) weights on the CDNA4 fp8 matrix
 * cores (reference call sites models/fusion_vmamba.py:1147-1150 and :1205, there plain F.conv1d / F.conv2d):
 *   y[b, l, m] = scale * sum_k wq[m, k] * q(x[b, k, l])    x (B, K, L) bf16 planes -> y (B, L, M) bf16 tokens
 * wq: (M, K) fp8 e4m3fn, quantised per tensor by the caller (scale = amax / 448, a DEVICE scalar); q(): clamp to +-448 and round to
 * nearest even to e4m3fn inside the kernel (no activation scale).  K % 16 == 0, M % 4 == 0, M <= 1024.
 */
int xfm_fp8_planes_gemm_supported(int K, int M);
int xfm_fp8_planes_gemm(const void *x_bf16, const void *wq_fp8, const float *scale, void *y_bf16, int B, int K, int L, int M,
                        void *stream);

/*
 * The optimizer step of the reference loop (torch.optim.Adam(lr, weight_decay), 1_train_model.py:141) for all parameters
 * in ONE multi-tensor launch, fused with the refresh of the bf16 weight shadows of the mixed-precision forward:
 *   g' = g + wd p;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * p/g/m/v/shadow_ptrs: device arrays of int64 device addresses (fp32 tensors; shadow = bf16 copy of p or 0); numel int64;
 * chunks (int64): tensor index | (chunk index inside the tensor << 32) for each of the nchunks workgroups (chunk elements
 * each, a multiple of 1024); step: device fp32 scalar t, read then incremented (graph-capturable).
 */
int xfm_adam_multi(const void *p_ptrs, const void *g_ptrs, const void *m_ptrs, const void *v_ptrs, const void *shadow_ptrs,
                   const void *numel, const void *chunks, int nchunks, int chunk, float *step,
                   float lr, float beta1, float beta2, float eps, float weight_decay, void *stream);
/*
 * The same update with g * grad_scale in place of g, for data-parallel runs that hand in the all-reduced SUM of the ranks'
 * gradients (grad_scale = 1 / world).  A numel entry with bit 62 set marks a bf16 gradient tensor (the wire bucket of
 * xfmamba_amd/dp.py read in place: no widening copy).  xfm_adam_multi is this entry with grad_scale = 1.
 */
int xfm_adam_multi_scaled(const void *p_ptrs, const void *g_ptrs, const void *m_ptrs, const void *v_ptrs,
                          const void *shadow_ptrs, const void *numel, const void *chunks, int nchunks, int chunk, float *step,
                          float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale, void *stream);

/*
 * Token-contracting product on the matrix cores (csrc/wgrad_gemm.hip): the weight gradient of a channel projection,
 *   dw (M, N) fp32 += sum over samples b and tokens l of a[b, l, m] * b[b, l, n]
 * Each operand is token-major (batch, L, C) or plane-major (batch, C, L) (x_planes != 0), bf16, with sample stride
 * a_bs / b_bs in elements; dw is ACCUMULATED into (zero it for a plain product).  Token-major operands need C % 8 == 0,
 * plane-major ones L % 4 == 0; 16-byte aligned bases.  Replaces the autograd weight gradients of nn.Linear / 1x1
 * nn.Conv2d (in_proj, out_proj, x_proj, Mlp fc1 / fc2: models/fusion_vmamba.py:42-45,135-153,1090-1109).
 */
int xfm_wgrad_supported(int M, int N, int L, int a_planes, int b_planes);
int xfm_wgrad(const void *a, const void *b, float *dw, int M, int N, int batch, int L, int64_t a_bs, int64_t b_bs,
              int a_planes, int b_planes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* XFM_HIP_H */
