// dwconv_tok.hip -- depthwise 3 x 3 convolution + SiLU on TOKEN-MAJOR maps (B, H, W, C) bf16, for the short maps of the trunk
// (14 x 14 and 7 x 7; reference models/fusion_vmamba.py:1198-1201: nn.Conv2d(D, D, 3, padding=1, groups=D) followed by nn.SiLU,
// there on NCHW maps).  With the depthwise stage token-major the SS2D block of these stages never leaves the token layout:
// in_proj / x_proj / out_proj and their data gradients are plain token GEMMs, every weight gradient is tokens x tokens, out_norm
// is the row LayerNorm (the plane-major kernels of dwconv.hip stay for the wide maps, where the scan wants planes).
//
// Channels are independent, so the stencil is per-lane arithmetic with nothing to stage or synchronise; neighbouring lanes are
// neighbouring channel groups, so a wave's loads and stores are contiguous runs, and the eight neighbours of a position are other
// threads' positions and come out of L1 / L2.
//   dwconv_tok_stencil_kernel: ONE thread per output vector (8 channels of one position: 600 k threads at 14 x 14 x 384, batch
//     64), the (9, C) weights in LDS for the life of a workgroup that walks several 256-vector slabs.
//       MODE 0  y  = silu(conv(x) + bias)
//       MODE 1  dz = dy * silu'(conv(x) + bias)
//       MODE 2  dx = conv of dz with the flipped taps (the transposed convolution)
//     (The first version gave a thread a whole map ROW to walk with the window in registers: 672 waves for 1024 SIMDs, 14 serial
//      steps each exposing an L2 round trip -- 28 us forward, 72 us backward against 19 / 25 us of the plane-major kernels.)
//   dwconv_tok_kernel<HW, 3, 4>: the weight / bias gradient sums need a walk (sums over positions stay in registers): a thread
//     owns 4 channels of ONE map row, walks it with the 3 x 3 x 4 window of x and the row's dz, and leaves one partial row per
//     (sample, map row): part (B * H, 10, C) fp32, folded by dwconv_tok_reduce_kernel.
// Roofline: HBM (2 / 3 + 2 tensor passes of 2 B per element).
#include "xfm_common.hpp"

// channels per thread of the LDS-tile kernels: 8 at 14 x 14 (forward 14.3 vs 16.6 us, backward 27.0 vs 29.4 us at batch 64 x 384
// channels), 4 at 7 x 7 (8.8 vs 9.9, 16.4 vs 30.0 us at 768 channels: 324 registers + AGPRs at 8)
#define DWT_CH (HW == 14 ? 8 : 4)

namespace xfm {

// CH bf16 channels (CH = 8: one 16-byte vector, CH = 4: 8 bytes) <-> fp32
template <int CH> __device__ __forceinline__ void dwt_load(const uint16_t *p, float (&o)[CH]) {
    uint32_t w[CH / 2];
    if constexpr (CH == 8) {
        const uint4 v = *reinterpret_cast<const uint4 *>(p);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
        const uint2 v = *reinterpret_cast<const uint2 *>(p);
        w[0] = v.x; w[1] = v.y;
    }
#pragma unroll
    for (int i = 0; i < CH / 2; ++i) {
        o[2 * i] = __uint_as_float(w[i] << 16);
        o[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
template <int CH> __device__ __forceinline__ void dwt_store(uint16_t *p, const float (&v)[CH]) {
    if constexpr (CH == 8)
        *reinterpret_cast<uint4 *>(p) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
    else
        *reinterpret_cast<uint2 *>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}

// (CH channels per thread: 8 in the stencil-only modes; 4 in MODE 1, whose 9 x CH partial sums next to the 9 x CH weights and
//  the 9 x CH window do not fit the register file at 8)
template <int HW, int MODE, int CH>
__global__ void __launch_bounds__(256) dwconv_tok_kernel(const uint16_t *__restrict__ x, const float *__restrict__ w,
                                                         const float *__restrict__ bias, const uint16_t *__restrict__ dy,
                                                         uint16_t *__restrict__ out, float *__restrict__ part, const int B,
                                                         const int C) {
    const int G = C / CH;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (int64_t)B * HW * G) return;
    const int g = (int)(gid % G);
    const int r = (int)(gid / G), h = r % HW, b = r / HW;
    float wt[9][CH], bs[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t][j] = w[(CH * g + j) * 9 + (MODE == 2 ? 8 - t : t)];
        bs[j] = (MODE != 2 && MODE != 3 && bias) ? bias[CH * g + j] : 0.f;
    }
    // the three map rows this thread reads (null: outside the map)
    const uint16_t *rows[3];
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
        const int hh = h + dh - 1;
        rows[dh] = (hh >= 0 && hh < HW) ? x + (((int64_t)b * HW + hh) * HW) * C + CH * g : nullptr;
    }
    const int64_t orow = (((int64_t)b * HW + h) * HW) * C + CH * g;
    float win[3][3][CH];                                   // [map row][column slot][channel]; slot of column c = (c + 3) % 3
    auto load_col = [&](const int c, const int slot) {
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            float (&o)[CH] = win[dh][slot];
            if (rows[dh] != nullptr && c >= 0 && c < HW) {
                dwt_load<CH>(rows[dh] + (int64_t)c * C, o);
            } else {
#pragma unroll
                for (int j = 0; j < CH; ++j) o[j] = 0.f;
            }
        }
    };
    float pw[(MODE == 1 || MODE == 3) ? 9 : 1][CH], pb[CH];
    if constexpr (MODE == 1 || MODE == 3) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            pb[j] = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) pw[t][j] = 0.f;
        }
    }
    load_col(-1, 2);
    load_col(0, 0);
    // the walk: written out three columns at a time (the period of the window's slots, so the rotation is renaming) inside a
    // ROLLED loop -- fully unrolled the compiler hoisted all 3 x HW loads to the top: 292 registers + 36 AGPRs in MODE 1
#pragma unroll 1
    for (int w3 = 0; w3 < HW; w3 += 3) {
#pragma unroll
    for (int ww = 0; ww < 3; ++ww) {
        const int wq = w3 + ww;
        if (wq >= HW) break;
        load_col(wq + 1, (ww + 1) % 3);
        float gy[CH];
        if constexpr (MODE == 1 || MODE == 3) dwt_load<CH>(dy + orow + (int64_t)wq * C, gy);
        if constexpr (MODE == 3) {                       // dy IS the pre-activation gradient dz: sums only
#pragma unroll
            for (int j = 0; j < CH; ++j) pb[j] += gy[j];
#pragma unroll
            for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                for (int dw = 0; dw < 3; ++dw) {
                    const float (&v)[CH] = win[dh][(ww + dw + 2) % 3];
#pragma unroll
                    for (int j = 0; j < CH; ++j) pw[dh * 3 + dw][j] = fmaf(gy[j], v[j], pw[dh * 3 + dw][j]);
                }
            continue;
        }
        float acc[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[j] = bs[j];
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const float (&v)[CH] = win[dh][(ww + dw + 2) % 3];
#pragma unroll
                for (int j = 0; j < CH; ++j) acc[j] = fmaf(wt[dh * 3 + dw][j], v[j], acc[j]);
            }
        float o[CH];
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < CH; ++j) o[j] = acc[j] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-acc[j] * kLog2e));
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-acc[j] * kLog2e));
                o[j] = gy[j] * sg * fmaf(acc[j], 1.f - sg, 1.f);
                pb[j] += o[j];
            }
#pragma unroll
            for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                for (int dw = 0; dw < 3; ++dw) {
                    const float (&v)[CH] = win[dh][(ww + dw + 2) % 3];
#pragma unroll
                    for (int j = 0; j < CH; ++j) pw[dh * 3 + dw][j] = fmaf(o[j], v[j], pw[dh * 3 + dw][j]);
                }
        } else {
#pragma unroll
            for (int j = 0; j < CH; ++j) o[j] = acc[j];
        }
        dwt_store<CH>(out + orow + (int64_t)wq * C, o);
    }
    }
    if constexpr (MODE == 1 || MODE == 3) {
        float *p = part + ((int64_t)r * 10) * C + CH * g;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int q = 0; q < CH / 4; ++q)
                *reinterpret_cast<float4 *>(p + (int64_t)t * C + 4 * q) = make_float4(pw[t][4 * q], pw[t][4 * q + 1], pw[t][4 * q + 2], pw[t][4 * q + 3]);
#pragma unroll
        for (int q = 0; q < CH / 4; ++q)
            *reinterpret_cast<float4 *>(p + (int64_t)9 * C + 4 * q) = make_float4(pb[4 * q], pb[4 * q + 1], pb[4 * q + 2], pb[4 * q + 3]);
    }
}

// ---- one thread per output vector ---------------------------------------------------------------------------------------
template <int HW, int MODE>
__global__ void __launch_bounds__(256) dwconv_tok_stencil_kernel(const uint16_t *__restrict__ x, const float *__restrict__ w,
                                                                 const float *__restrict__ bias, const uint16_t *__restrict__ dy,
                                                                 uint16_t *__restrict__ out, const int B, const int C, const int slabs) {
    extern __shared__ float wl[];                                    // [9 taps][C] (+ [C] bias): tap t of channel c at t * C + c
    const int G = C >> 3;
    {   // weights (C, 9) -> LDS (9, C): all of a thread's 16-byte loads in flight before the first LDS write (a load / scatter
        // loop exposed ~14 round trips per workgroup: the first version of this kernel ran 38 us on 9.6 MB)
        constexpr int KW = 7;                                        // 9 C / 4 float4s over 256 threads, C <= 768 ... 1536: two rounds
        const int n4 = 9 * C / 4;
        for (int base = 0; base < n4; base += KW * 256) {
            float4 r[KW];
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const int e4 = base + threadIdx.x + 256 * k;
                if (e4 < n4) r[k] = *reinterpret_cast<const float4 *>(w + 4 * e4);
            }
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const int e4 = base + threadIdx.x + 256 * k;
                if (e4 < n4) {
                    const float f[4] = {r[k].x, r[k].y, r[k].z, r[k].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * e4 + j, c = e / 9, t = e - c * 9;
                        wl[(MODE == 2 ? 8 - t : t) * C + c] = f[j];
                    }
                }
            }
        }
        if (MODE != 2)
            for (int c = threadIdx.x; c < C; c += 256) wl[9 * C + c] = bias ? bias[c] : 0.f;
    }
    __syncthreads();
    const int NV = B * HW * HW * G;                                  // (host: < 2^31)
#pragma unroll 1
    for (int it = 0; it < slabs; ++it) {
        const int v = (blockIdx.x * slabs + it) * 256 + threadIdx.x;
        if (v >= NV) break;
        const int g = v % G;
        const int pos = v / G;                                       // b * HW * HW + h * HW + wq
        const int wq = pos % HW, h = (pos / HW) % HW;
        const uint16_t *px = x + (int64_t)pos * C + 8 * g;
        uint4 q[9];
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const bool ok = (unsigned)(h + dh - 1) < (unsigned)HW && (unsigned)(wq + dw - 1) < (unsigned)HW;
                q[dh * 3 + dw] = ok ? *reinterpret_cast<const uint4 *>(px + ((dh - 1) * HW + (dw - 1)) * C) : make_uint4(0, 0, 0, 0);
            }
        uint4 gq = make_uint4(0, 0, 0, 0);
        if constexpr (MODE == 1) gq = *reinterpret_cast<const uint4 *>(dy + (int64_t)pos * C + 8 * g);
        float acc[8];
        if constexpr (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
        } else {
            const float4 b0 = *reinterpret_cast<const float4 *>(wl + 9 * C + 8 * g), b1 = *reinterpret_cast<const float4 *>(wl + 9 * C + 8 * g + 4);
            acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w; acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 w0 = *reinterpret_cast<const float4 *>(wl + t * C + 8 * g), w1 = *reinterpret_cast<const float4 *>(wl + t * C + 8 * g + 4);
            const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
            const uint32_t u[4] = {q[t].x, q[t].y, q[t].z, q[t].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[2 * i] = fmaf(wv[2 * i], __uint_as_float(u[i] << 16), acc[2 * i]);
                acc[2 * i + 1] = fmaf(wv[2 * i + 1], __uint_as_float(u[i] & 0xffff0000u), acc[2 * i + 1]);
            }
        }
        float o[8];
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = acc[j] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-acc[j] * kLog2e));
        } else if constexpr (MODE == 1) {
            float gy[8];
            const uint32_t u[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                gy[2 * i] = __uint_as_float(u[i] << 16);
                gy[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-acc[j] * kLog2e));
                o[j] = gy[j] * sg * fmaf(acc[j], 1.f - sg, 1.f);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = acc[j];
        }
        dwt_store<8>(out + (int64_t)pos * C + 8 * g, o);
    }
}

// ---- LDS-tile kernels: a workgroup owns one sample's map for 64 channels -------------------------------------------------
// [L positions][64 channels] bf16 = rows of 128 bytes (25 KB at 14 x 14): the tile is read from HBM once (all of a thread's
// 16-byte loads in flight), every neighbour access is a 16-byte LDS read whose eight lanes of a position cover one 128-byte row
// (conflict-free), and a thread keeps ONE channel group (tid & 7) for all its positions, so its 9 x 8 weights live in registers.
// (Both memory-only forms above measured 34 us forward / 91 us backward at 64 x 14 x 14 x 384 under rocprofv3, against 9 / 21
//  us of the plane-major kernels: nine L2 reads per output vector, resp. a serial walk on a third of the SIMDs.)
template <int HW>
__device__ __forceinline__ void dwt_tile_load(uint16_t *ts, const uint16_t *src, const int C, const int tid) {
    constexpr int L = HW * HW, NV = L * 8, KV = (NV + 255) / 256;
    uint4 r[KV];                                                     // (unconditional loads at a clamped index: a conditional
#pragma unroll                                                       //  assignment leaves the array in scratch memory)
    for (int k = 0; k < KV; ++k) {
        const int v = min(tid + 256 * k, NV - 1);
        r[k] = *reinterpret_cast<const uint4 *>(src + (int64_t)(v >> 3) * C + 8 * (v & 7));
    }
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        const int v = tid + 256 * k;
        if (v < NV) *reinterpret_cast<uint4 *>(ts + 8 * v) = r[k];
    }
}
// the 3 x 3 neighbourhood of position (h, wq), channel group g (CH channels), as nine unpacked vectors (zeros outside the map)
template <int HW, int CH>
__device__ __forceinline__ void dwt_tile_nb(const uint16_t *ts, const int h, const int wq, const int g, float (&nb)[9][CH]) {
#pragma unroll
    for (int dh = 0; dh < 3; ++dh)
#pragma unroll
        for (int dw = 0; dw < 3; ++dw) {
            const int hh = h + dh - 1, ww = wq + dw - 1;
            const bool ok = (unsigned)hh < (unsigned)HW && (unsigned)ww < (unsigned)HW;
            if (ok) {
                dwt_load<CH>(ts + (hh * HW + ww) * 64 + CH * g, nb[dh * 3 + dw]);
            } else {
#pragma unroll
                for (int j = 0; j < CH; ++j) nb[dh * 3 + dw][j] = 0.f;
            }
        }
}

// the 9 x CH taps of CH consecutive channels: 9 CH contiguous floats of the (C, 9) weight as 16-byte loads
template <int CH>
__device__ __forceinline__ void dwt_tile_weights(const float *w, const float *bias, const int c, float (&wt)[9][CH], float (&bs)[CH]) {
    float f[9 * CH];
#pragma unroll
    for (int q = 0; q < 9 * CH / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4 *>(w + (int64_t)c * 9 + 4 * q);
        f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
#pragma unroll
        for (int t = 0; t < 9; ++t) wt[t][j] = f[j * 9 + t];
        bs[j] = bias ? bias[c + j] : 0.f;
    }
}

// CH channels per thread: a thread keeps channel group tid % (64 / CH) and walks positions tid / (64 / CH) + k * (256 CH / 64)
template <int HW, int CH>
__global__ void __launch_bounds__(256) dwconv_tile_fwd_kernel(const uint16_t *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ bias, uint16_t *__restrict__ y, const int C) {
    constexpr int L = HW * HW, NG = 64 / CH, PS = 256 / NG;
    extern __shared__ float smem[];
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem);
    const int nb64 = C >> 6, b = blockIdx.x / nb64, c0 = (blockIdx.x - b * nb64) * 64;
    const int tid = threadIdx.x, g = tid % NG, pr = tid / NG;
    dwt_tile_load<HW>(xs, x + (int64_t)b * L * C + c0, C, tid);
    float wt[9][CH], bs[CH];
    dwt_tile_weights<CH>(w, bias, c0 + CH * g, wt, bs);
    __syncthreads();
#pragma unroll 1
    for (int p = pr; p < L; p += PS) {
        const int h = p / HW, wq = p - h * HW;
        float nb[9][CH], o[CH];
        dwt_tile_nb<HW, CH>(xs, h, wq, g, nb);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            float acc = bs[j];
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = fmaf(wt[t][j], nb[t][j], acc);
            o[j] = acc * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-acc * kLog2e));
        }
        dwt_store<CH>(y + ((int64_t)b * L + p) * C + c0 + CH * g, o);
    }
}

// backward in ONE launch: dz = dy * silu'(conv(x) + bias) into the dy tile (in place), the weight / bias gradient sums of the
// thread's positions in registers (the conv's unpacked neighbourhood is reused), dx = transposed convolution of the dz tile;
// the sums are folded over the positions of the tile (lanes NG apart, then the four waves through LDS) into ONE partial row per
// (sample, channel block): part (B, 10, C), summed over the samples by dwconv_tok_reduce_kernel
template <int HW, int CH>
__global__ void __launch_bounds__(256) dwconv_tile_bwd_kernel(const uint16_t *__restrict__ x, const float *__restrict__ w,
                                                              const float *__restrict__ bias, const uint16_t *__restrict__ dy,
                                                              uint16_t *__restrict__ dx, float *__restrict__ part, const int C) {
    constexpr int L = HW * HW, NG = 64 / CH, PS = 256 / NG;
    extern __shared__ float smem[];
    uint16_t *xs = reinterpret_cast<uint16_t *>(smem), *gs = xs + L * 64;
    const int nb64 = C >> 6, b = blockIdx.x / nb64, c0 = (blockIdx.x - b * nb64) * 64;
    const int tid = threadIdx.x, g = tid % NG, pr = tid / NG;
    dwt_tile_load<HW>(xs, x + (int64_t)b * L * C + c0, C, tid);
    dwt_tile_load<HW>(gs, dy + (int64_t)b * L * C + c0, C, tid);
    float wt[9][CH], bs[CH];
    dwt_tile_weights<CH>(w, bias, c0 + CH * g, wt, bs);
    float pw[10][CH];                                                // taps 0..8, [9] = bias
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int j = 0; j < CH; ++j) pw[t][j] = 0.f;
    __syncthreads();
#pragma unroll 1
    for (int p = pr; p < L; p += PS) {
        const int h = p / HW, wq = p - h * HW;
        float nb[9][CH], gy[CH], dz[CH];
        dwt_tile_nb<HW, CH>(xs, h, wq, g, nb);
        dwt_load<CH>(gs + p * 64 + CH * g, gy);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            float acc = bs[j];
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = fmaf(wt[t][j], nb[t][j], acc);
            const float sg = __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-acc * kLog2e));
            dz[j] = gy[j] * sg * fmaf(acc, 1.f - sg, 1.f);
            pw[9][j] += dz[j];
#pragma unroll
            for (int t = 0; t < 9; ++t) pw[t][j] = fmaf(dz[j], nb[t][j], pw[t][j]);
        }
        dwt_store<CH>(gs + p * 64 + CH * g, dz);                     // (only this thread reads dy of (p, g))
    }
    __syncthreads();
#pragma unroll 1
    for (int p = pr; p < L; p += PS) {
        const int h = p / HW, wq = p - h * HW;
        float nb[9][CH], o[CH];
        dwt_tile_nb<HW, CH>(gs, h, wq, g, nb);
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = fmaf(wt[8 - t][j], nb[t][j], acc);     // flipped taps: the transposed convolution
            o[j] = acc;
        }
        dwt_store<CH>(dx + ((int64_t)b * L + p) * C + c0 + CH * g, o);
    }
    // ---- fold the sums: lanes with the same channel group are NG apart, then the four waves through LDS
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            float v = pw[t][j];
#pragma unroll
            for (int o = NG; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            pw[t][j] = v;
        }
    __syncthreads();                                                 // (the tiles are free now)
    float *red = smem;                                               // [4 waves][10][64]
    const int wave = tid >> 6, lane = tid & 63;
    if (lane < NG) {
#pragma unroll
        for (int t = 0; t < 10; ++t)
#pragma unroll
            for (int j = 0; j < CH; ++j) red[(wave * 10 + t) * 64 + CH * g + j] = pw[t][j];
    }
    __syncthreads();
    for (int e = tid; e < 640; e += 256) {
        const int t = e >> 6, c = e & 63;
        part[((int64_t)b * 10 + t) * C + c0 + c] = (red[e] + red[640 + e]) + (red[1280 + e] + red[1920 + e]);
    }
}

// dweight (C, 9) / dbias (C) += sums of the partial rows part (rows, 10, C): blockIdx.y owns a slice of 32 rows (one thread per
// (tap, channel) walking all rows ran on 15 workgroups: 100+ us of serial round trips); results ZEROED by the caller
__global__ void __launch_bounds__(256) dwconv_tok_reduce_kernel(const float *__restrict__ part, float *__restrict__ dweight,
                                                                float *__restrict__ dbias, const int rows, const int C) {
    const int e = blockIdx.x * 256 + threadIdx.x;                    // (tap, channel), channel fastest
    if (e >= 10 * C) return;
    const int t = e / C, c = e - t * C;
    const int r0 = blockIdx.y * 32, r1 = min(rows, r0 + 32);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0;
    for (; r + 4 <= r1; r += 4) {
        s0 += part[((int64_t)r * 10) * C + e];
        s1 += part[((int64_t)(r + 1) * 10) * C + e];
        s2 += part[((int64_t)(r + 2) * 10) * C + e];
        s3 += part[((int64_t)(r + 3) * 10) * C + e];
    }
    for (; r < r1; ++r) s0 += part[((int64_t)r * 10) * C + e];
    const float s = (s0 + s1) + (s2 + s3);
    if (t < 9) atomicAdd(dweight + c * 9 + t, s);
    else if (dbias) atomicAdd(dbias + c, s);
}

template <int HW, int MODE>
static void dwt_stencil(const void *x, const float *w, const float *bias, const void *dy, void *out, int B, int C, hipStream_t s) {
    const int64_t nv = (int64_t)B * HW * HW * (C / 8);
    // slabs of 256 vectors per workgroup: ~768 workgroups (three per CU, co-resident), the weight staging amortised over a walk
    int slabs = (int)((nv / 256 + 767) / 768);
    if (slabs < 1) slabs = 1;
    const unsigned grid = (unsigned)((nv + 256 * (int64_t)slabs - 1) / (256 * (int64_t)slabs));
    const size_t lds = (size_t)10 * C * sizeof(float);
    hipLaunchKernelGGL((dwconv_tok_stencil_kernel<HW, MODE>), dim3(grid), dim3(256), lds, s, (const uint16_t *)x, w, bias,
                       (const uint16_t *)dy, (uint16_t *)out, B, C, slabs);
}
template <int HW> static int dwt_fwd(const void *x, const float *w, const float *bias, void *y, int B, int C, hipStream_t s) {
    if (C % 64 == 0) {
        hipLaunchKernelGGL((dwconv_tile_fwd_kernel<HW, DWT_CH>), dim3((unsigned)(B * (C / 64))), dim3(256), (size_t)HW * HW * 128, s,
                           (const uint16_t *)x, w, bias, (uint16_t *)y, C);
        return check_launch();
    }
    dwt_stencil<HW, 0>(x, w, bias, nullptr, y, B, C, s);
    return check_launch();
}
template <int HW>
static int dwt_bwd(const void *x, const float *w, const float *bias, const void *dy, void *dz, void *dx, float *part, float *dweight,
                   float *dbias, int B, int C, hipStream_t s) {
    if (C % 64 == 0) {
        // one launch: part (B, 10, C), one partial row per sample
        hipLaunchKernelGGL((dwconv_tile_bwd_kernel<HW, DWT_CH>), dim3((unsigned)(B * (C / 64))), dim3(256), (size_t)HW * HW * 256, s,
                           (const uint16_t *)x, w, bias, (const uint16_t *)dy, (uint16_t *)dx, part, C);
        hipLaunchKernelGGL(dwconv_tok_reduce_kernel, dim3((10 * C + 255) / 256, (B + 31) / 32), dim3(256), 0, s, part, dweight, dbias, B, C);
        return check_launch();
    }
    dwt_stencil<HW, 1>(x, w, bias, dy, dz, B, C, s);                                  // dz = dy * silu'(conv(x) + bias)
    const int64_t n4 = (int64_t)B * HW * (C / 4);
    hipLaunchKernelGGL((dwconv_tok_kernel<HW, 3, 4>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const uint16_t *)x, w,
                       (const float *)nullptr, (const uint16_t *)dz, (uint16_t *)nullptr, part, B, C);
    hipLaunchKernelGGL(dwconv_tok_reduce_kernel, dim3((10 * C + 255) / 256, (B * HW + 31) / 32), dim3(256), 0, s, part, dweight, dbias,
                       B * HW, C);
    dwt_stencil<HW, 2>(dz, w, nullptr, nullptr, dx, B, C, s);                         // dx = transposed convolution of dz
    return check_launch();
}

}  // namespace xfm

extern "C" {

int xfm_dwconv3x3_tokens_supported(int H, int W, int C) {
    return (H == W && (H == 14 || H == 7) && C > 0 && C % 8 == 0 && C <= 1536) ? 1 : 0;      // (10 C floats of LDS <= 64 KB)
}

int xfm_dwconv3x3_tokens_fwd(const void *x, const float *weight, const float *bias, void *y, int B, int H, int W, int C, void *stream) {
    using namespace xfm;
    if (!x || !weight || !y || B <= 0) return XFM_EINVAL;
    if (!xfm_dwconv3x3_tokens_supported(H, W, C)) return XFM_ELIMIT;
    if ((((uintptr_t)x | (uintptr_t)y) & 15) || (int64_t)B * H * W * (C / 8) >= (1ll << 31)) return XFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    return H == 14 ? dwt_fwd<14>(x, weight, bias, y, B, C, s) : dwt_fwd<7>(x, weight, bias, y, B, C, s);
}

int xfm_dwconv3x3_tokens_bwd(const void *x, const float *weight, const float *bias, const void *dy, void *dz_ws, void *dx,
                             float *part_ws, float *dweight, float *dbias, int B, int H, int W, int C, void *stream) {
    using namespace xfm;
    if (!x || !weight || !dy || !dz_ws || !dx || !part_ws || !dweight || B <= 0) return XFM_EINVAL;
    if (!xfm_dwconv3x3_tokens_supported(H, W, C)) return XFM_ELIMIT;
    if ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dz_ws | (uintptr_t)dx | (uintptr_t)part_ws) & 15) ||
        (int64_t)B * H * W * (C / 8) >= (1ll << 31))
        return XFM_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    return H == 14 ? dwt_bwd<14>(x, weight, bias, dy, dz_ws, dx, part_ws, dweight, dbias, B, C, s)
                   : dwt_bwd<7>(x, weight, bias, dy, dz_ws, dx, part_ws, dweight, dbias, B, C, s);
}
}
