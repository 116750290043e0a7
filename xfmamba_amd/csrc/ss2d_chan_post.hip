// ss2d_chan_post.hip -- the two dense products behind the channel-lane SS2D backward (csrc/ss2d_chan.hip), on MFMA:
//   d x_dbl[b, l, k*C2p + r] = sum_d ddts[b,k,l,d] * W_dt[k,d,r]      (contraction over the channels; the B / C columns of
//                                                                      the row come from the scan kernel's dBC sums)
//   d W_dt[k, d, r]        += sum_{b,l} ddts[b,k,l,d] * x_dbl[b, l, k*C2p + r]   (contraction over batch and positions)
// i.e. the backward of the dt_proj einsum of forward_corev2 (reference models/fusion_vmamba.py:1154-1156 / :492-495) in
// the token-major layout of the channel-lane kernels.  Both read ddts (B,4,L,D) bf16 once; HBM-bound.
#include <algorithm>
#include <cstdlib>

#include "xfm_common.hpp"

namespace xfm {

int wgrad_grouped(const void *a, const void *b, float *dw, int M, int N, int batch, int L, int64_t a_bs, int64_t b_bs, int lda,
                  int ldb, int groups, int64_t a_gs, int64_t b_gs, int64_t dw_gs, hipStream_t s);      // wgrad_gemm.hip

typedef __bf16 pbf16x8_t __attribute__((ext_vector_type(8)));
typedef float pf32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t pu32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t pu32x2_t __attribute__((ext_vector_type(2)));

struct ChanPostArgs {
    const uint16_t *ddts;    // (Bt, 4, L, D) bf16
    const uint16_t *xdbl;    // (Bt, L, XC) bf16
    const uint16_t *wdt;     // (4, D, Rp8) bf16: dt_proj weight, zero columns beyond R (the layout xfm_ss2dc_fwd/_bwd take)
    const float *dBC;        // (Bt, 4, 2, N, L) fp32
    uint16_t *dxdbl;         // (Bt, L, XC) bf16 (every column written)
    float *dwdt;             // (4, D, R) fp32 ZEROED (atomics)
    int Bt, D, L, R, N, Rp8, C2p, XC, ptiles, bchunk;
    int HW;                  // d_state 1: dBC is in WALKING order -- odd routes column-major on the HW x HW map (ss2d_chan1.hip)
};

__device__ __forceinline__ pbf16x8_t post_ld8(const uint16_t *p) {
    const pu32x4_t v = *reinterpret_cast<const pu32x4_t *>(p);
    return *reinterpret_cast<const pbf16x8_t *>(&v);
}
__device__ __forceinline__ pbf16x8_t post_zero8() {
    const pu32x4_t v = {0u, 0u, 0u, 0u};
    return *reinterpret_cast<const pbf16x8_t *>(&v);
}

// ---- d x_dbl: one wave per (sample, route, tile of 32 positions); D^T[r][pos] = sum_d W^T[r][d] * ddts[pos][d] -----------
template <int KT> __global__ void __launch_bounds__(256) chan_dxdbl_kernel(const ChanPostArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int job = blockIdx.x * 4 + wave;                          // (b*4 + k) * ptiles + pt
    if (job >= a.Bt * 4 * a.ptiles) return;
    const int bk = job / a.ptiles, pt = job - bk * a.ptiles;
    const int k = bk & 3, b = bk >> 2;
    const int col = lane & 31, kb = lane >> 5;
    const int p = pt * 32 + col;
    const bool pv = p < a.L;
    const uint16_t *brow = a.ddts + ((int64_t)bk * a.L + (pv ? p : 0)) * a.D + 8 * kb;
    // A operand = W^T (rows r, k = channels): gathered from the (D, Rp8) weight with 2-byte loads (32 lanes = 64
    // contiguous bytes per channel; the weight is a few KB and stays in L1 / L2)
    const uint16_t *arow = a.wdt + ((int64_t)k * a.D + 8 * kb) * a.Rp8 + col;
    pf32x16_t acc[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    const int nks = a.D / 16;
    auto load_a = [&](int s, pbf16x8_t(&af)[KT]) {
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            uint16_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (32 * t + col < a.Rp8) ? arow[(int64_t)(16 * s + j) * a.Rp8 + 32 * t] : (uint16_t)0;
            af[t] = *reinterpret_cast<const pbf16x8_t *>(v);
        }
    };
    pbf16x8_t bf = pv ? post_ld8(brow) : post_zero8();
    pbf16x8_t af[KT];
    load_a(0, af);
    for (int s = 0; s < nks; ++s) {
        const pbf16x8_t bc = bf;
        pbf16x8_t ac[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) ac[t] = af[t];
        if (s + 1 < nks) {
            bf = pv ? post_ld8(brow + 16 * (s + 1)) : post_zero8();
            load_a(s + 1, af);
        }
#pragma unroll
        for (int t = 0; t < KT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[t], bc, acc[t], 0, 0, 0);
    }
    if (!pv) return;
    uint16_t *orow = a.dxdbl + ((int64_t)b * a.L + p) * a.XC + k * a.C2p;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r0 = t * 32 + 8 * g + 4 * kb;                 // rows r0 .. r0+3 of D^T = four consecutive columns
            if (r0 < a.Rp8) {
                pu32x2_t v;
                v[0] = pack_bf16x2(acc[t][4 * g], acc[t][4 * g + 1]);
                v[1] = pack_bf16x2(acc[t][4 * g + 2], acc[t][4 * g + 3]);
                *reinterpret_cast<pu32x2_t *>(orow + r0) = v;
            }
        }
    // B / C columns (and the zero padding after them) of this position
    if (kb == 0) {
        const int pw = (a.N == 1 && (k & 1)) ? (p % a.HW) * a.HW + p / a.HW : p;
        const float *dB = a.dBC + (((int64_t)bk * 2 + 0) * a.N) * a.L + pw;
        const float *dC = a.dBC + (((int64_t)bk * 2 + 1) * a.N) * a.L + pw;
        if (a.N == 1) {
            pu32x4_t v = {pack_bf16x2(dB[0], dC[0]), 0u, 0u, 0u};
            *reinterpret_cast<pu32x4_t *>(orow + a.Rp8) = v;
        } else {
            for (int q = 0; q < a.N / 8; ++q) {
                pu32x4_t vb, vc;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    vb[j] = pack_bf16x2(dB[(int64_t)(8 * q + 2 * j) * a.L], dB[(int64_t)(8 * q + 2 * j + 1) * a.L]);
                    vc[j] = pack_bf16x2(dC[(int64_t)(8 * q + 2 * j) * a.L], dC[(int64_t)(8 * q + 2 * j + 1) * a.L]);
                }
                *reinterpret_cast<pu32x4_t *>(orow + a.Rp8 + 8 * q) = vb;
                *reinterpret_cast<pu32x4_t *>(orow + a.Rp8 + a.N + 8 * q) = vc;
            }
        }
    }
}

// ---- d x_dbl, second form: the dt_proj weight of a route TRANSPOSED in LDS.  The kernel above gathers its A operand (W^T: rows
// r, k = channels) from the (D, Rp8) weight with 2-byte loads, eight per lane and k-step -- more instructions than the product
// itself.  Here a workgroup belongs to ONE route: it stages W[k]^T once ([Rp8][D + 8] bf16, transposed while staging), its four
// waves walk (sample, 32-position tile) jobs, and the A fragments are 16-byte LDS reads.
template <int KT> __global__ void __launch_bounds__(256) chan_dxdbl2_kernel(const ChanPostArgs a, const int wgs_per_route) {
    extern __shared__ __align__(16) uint16_t wl[];                  // [Rp8][P]
    const int P = a.D + 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int k = blockIdx.x / wgs_per_route, g = blockIdx.x - k * wgs_per_route;
    {
        const uint16_t *w = a.wdt + (int64_t)k * a.D * a.Rp8;       // (D, Rp8)
        for (int e = threadIdx.x; e < a.D * a.Rp8; e += 256) {
            const int d = e / a.Rp8, r = e - d * a.Rp8;
            wl[r * P + d] = w[e];
        }
    }
    __syncthreads();
    const int col = lane & 31, kb = lane >> 5;
    const int nks = a.D / 16;
    const int njobs = a.Bt * a.ptiles;
    for (int job = g * 4 + wave; job < njobs; job += wgs_per_route * 4) {
        const int b = job / a.ptiles, pt = job - b * a.ptiles;
        const int bk = b * 4 + k;
        const int p = pt * 32 + col;
        const bool pv = p < a.L;
        const uint16_t *brow = a.ddts + ((int64_t)bk * a.L + (pv ? p : 0)) * a.D + 8 * kb;
        // rows of W^T this lane feeds: r = 32 t + col (rows at or beyond Rp8 are never stored: read a valid row instead)
        const uint16_t *arow[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) arow[t] = wl + min(32 * t + col, a.Rp8 - 1) * P + 8 * kb;
        pf32x16_t acc[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
        pbf16x8_t bf[4];                                             // four k-steps of the ddts row in flight
#pragma unroll
        for (int q = 0; q < 4; ++q) bf[q] = (pv && q < nks) ? post_ld8(brow + 16 * q) : post_zero8();
        for (int s0 = 0; s0 < nks; s0 += 4) {
            pbf16x8_t bc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bc[q] = bf[q];
#pragma unroll
            for (int q = 0; q < 4; ++q) bf[q] = (pv && s0 + 4 + q < nks) ? post_ld8(brow + 16 * (s0 + 4 + q)) : post_zero8();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (s0 + q < nks) {
#pragma unroll
                    for (int t = 0; t < KT; ++t) {
                        const pbf16x8_t af = *reinterpret_cast<const pbf16x8_t *>(arow[t] + 16 * (s0 + q));
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bc[q], acc[t], 0, 0, 0);
                    }
                }
            }
        }
        if (!pv) continue;
        uint16_t *orow = a.dxdbl + ((int64_t)b * a.L + p) * a.XC + k * a.C2p;
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r0 = t * 32 + 8 * q + 4 * kb;                 // rows r0 .. r0+3 of D^T = four consecutive columns
                if (r0 < a.Rp8) {
                    pu32x2_t v;
                    v[0] = pack_bf16x2(acc[t][4 * q], acc[t][4 * q + 1]);
                    v[1] = pack_bf16x2(acc[t][4 * q + 2], acc[t][4 * q + 3]);
                    *reinterpret_cast<pu32x2_t *>(orow + r0) = v;
                }
            }
        if (kb == 0) {                                               // B / C columns (and the zero padding after them)
            const int pw = (a.N == 1 && (k & 1)) ? (p % a.HW) * a.HW + p / a.HW : p;
            const float *dB = a.dBC + (((int64_t)bk * 2 + 0) * a.N) * a.L + pw;
            const float *dC = a.dBC + (((int64_t)bk * 2 + 1) * a.N) * a.L + pw;
            if (a.N == 1) {
                pu32x4_t v = {pack_bf16x2(dB[0], dC[0]), 0u, 0u, 0u};
                *reinterpret_cast<pu32x4_t *>(orow + a.Rp8) = v;
            } else {
                for (int q = 0; q < a.N / 8; ++q) {
                    pu32x4_t vb, vc;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        vb[j] = pack_bf16x2(dB[(int64_t)(8 * q + 2 * j) * a.L], dB[(int64_t)(8 * q + 2 * j + 1) * a.L]);
                        vc[j] = pack_bf16x2(dC[(int64_t)(8 * q + 2 * j) * a.L], dC[(int64_t)(8 * q + 2 * j + 1) * a.L]);
                    }
                    *reinterpret_cast<pu32x4_t *>(orow + a.Rp8 + 8 * q) = vb;
                    *reinterpret_cast<pu32x4_t *>(orow + a.Rp8 + a.N + 8 * q) = vc;
                }
            }
        }
    }
}

// ---- d W_dt: one wave per (route, 32-channel tile, chunk of samples); D[d][r] = sum_pos ddts[pos][d] * x_dbl[pos][r] ------
// Both operands have the contraction index (positions) as their slow index: the fragments are gathered with 2-byte loads
// (32 lanes = 64 contiguous bytes per wave instruction and position).
template <int KT> __global__ void __launch_bounds__(256) chan_dwdt_kernel(const ChanPostArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dtiles = a.D / 32;
    const int nchunks = (a.Bt + a.bchunk - 1) / a.bchunk;
    const int job = blockIdx.x * 4 + wave;                          // (k * dtiles + dt) * nchunks + ch
    if (job >= 4 * dtiles * nchunks) return;
    const int ch = job % nchunks, kd = job / nchunks;
    const int dt = kd % dtiles, k = kd / dtiles;
    const int col = lane & 31, kb = lane >> 5;
    pf32x16_t acc[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    const int b0 = ch * a.bchunk, b1 = min(a.Bt, b0 + a.bchunk);
    const int nps = (a.L + 15) / 16;
    for (int b = b0; b < b1; ++b) {
        const uint16_t *dd = a.ddts + (((int64_t)b * 4 + k) * a.L) * a.D + dt * 32 + col;   // + pos * D
        const uint16_t *xr = a.xdbl + ((int64_t)b * a.L) * a.XC + k * a.C2p + col;         // + pos * XC (+ 32 t)
        for (int s = 0; s < nps; ++s) {
            uint16_t av[8], bv[KT][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int pos = 16 * s + 8 * kb + j;
                const bool ok = pos < a.L;
                av[j] = ok ? dd[(int64_t)pos * a.D] : (uint16_t)0;
#pragma unroll
                for (int t = 0; t < KT; ++t) bv[t][j] = (ok && 32 * t + col < a.R) ? xr[(int64_t)pos * a.XC + 32 * t] : (uint16_t)0;
            }
            const pbf16x8_t af = *reinterpret_cast<const pbf16x8_t *>(av);
#pragma unroll
            for (int t = 0; t < KT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, *reinterpret_cast<const pbf16x8_t *>(bv[t]), acc[t], 0, 0, 0);
        }
    }
    // D[row = channel][col = r]: lane (col, kb) holds rows 8g + 4kb + i
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int r = 32 * t + col;
        if (r < a.R) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int d = dt * 32 + 8 * g + 4 * kb + i;
                    atomicAdd(a.dwdt + ((int64_t)k * a.D + d) * a.R + r, acc[t][4 * g + i]);
                }
        }
    }
}

}  // namespace xfm

extern "C" int xfm_ss2dc_post(const void *ddts, const void *xdbl, const void *wdt, const float *dBC, void *dxdbl, float *dwdt,
                              int batch, int d_inner, int L, int dt_rank, int dstate, void *stream) {
    using namespace xfm;
    if (!ddts || !xdbl || !wdt || !dBC || !dxdbl || !dwdt || batch <= 0 || L <= 0) return XFM_EINVAL;
    if (d_inner % 32 || dt_rank < 1 || dt_rank > 64 || (dstate != 1 && dstate % 8)) return XFM_ELIMIT;
    ChanPostArgs a{};
    a.ddts = (const uint16_t *)ddts; a.xdbl = (const uint16_t *)xdbl; a.wdt = (const uint16_t *)wdt; a.dBC = dBC;
    a.dxdbl = (uint16_t *)dxdbl; a.dwdt = dwdt;
    a.Bt = batch; a.D = d_inner; a.L = L; a.R = dt_rank; a.N = dstate;
    a.Rp8 = (dt_rank + 7) / 8 * 8;
    a.C2p = a.Rp8 + (dstate == 1 ? 8 : 2 * dstate);
    a.XC = 4 * a.C2p;
    a.ptiles = (L + 31) / 32;
    a.HW = 1;
    while (a.HW * a.HW < L) ++a.HW;
    if (dstate == 1 && a.HW * a.HW != L) return XFM_ELIMIT;         // (the d_state-1 scan kernels cover square maps only)
    const int KT = (a.Rp8 + 31) / 32;                                // 32-column tiles covering the dt_proj columns
    hipStream_t s = (hipStream_t)stream;
    static const bool v2 = [] { const char *e = getenv("XFM_CHAN_POST2"); return !(e && e[0] == '0'); }();   // A/B switch
    const size_t wl_bytes = (size_t)a.Rp8 * (d_inner + 8) * 2;
    int rc;
    if (v2 && wl_bytes <= 150 * 1024 && d_inner % 16 == 0) {
        // d x_dbl with the route's weight transposed in LDS: up to 64 workgroups per route walk its (sample, tile) jobs
        // (whole rounds: every wave of a route gets the same number of jobs)
        const int njobs = batch * a.ptiles, rounds = (njobs + 255) / 256;
        const int wgs = std::max(1, (njobs + 4 * rounds - 1) / (4 * rounds));
        const void *fn = KT == 1 ? (const void *)chan_dxdbl2_kernel<1> : (const void *)chan_dxdbl2_kernel<2>;
        if (wl_bytes > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl_bytes);
        int wgs_arg = wgs;
        ChanPostArgs args = a;
        void *kargs[] = {&args, &wgs_arg};
        const hipError_t e = hipLaunchKernel(fn, dim3(4 * wgs), dim3(256), kargs, wl_bytes, s);
        if (e != hipSuccess) {
            set_last_hip_error(e);
            return XFM_ELAUNCH;
        }
        rc = check_launch();
    } else {
        const unsigned g1 = (unsigned)((batch * 4 * a.ptiles + 3) / 4);
        if (KT == 1) hipLaunchKernelGGL(chan_dxdbl_kernel<1>, dim3(g1), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(chan_dxdbl_kernel<2>, dim3(g1), dim3(256), 0, s, a);
        rc = check_launch();
    }
    if (rc != XFM_OK) return rc;
    // d W_dt: the token-contracting weight-gradient kernel (csrc/wgrad_gemm.hip) over the four routes as groups of one launch:
    // A = ddts[:, k] (tokens (b, l), D channels, row pitch D), B = the dt_proj input columns of the x_proj rows (row pitch XC)
    if (v2 && dt_rank % 8 == 0) {
        rc = wgrad_grouped(ddts, (const uint16_t *)xdbl, dwdt, d_inner, dt_rank, batch, L, (int64_t)4 * L * d_inner,
                           (int64_t)L * a.XC, d_inner, a.XC, 4, (int64_t)L * d_inner, a.C2p, (int64_t)d_inner * dt_rank, s);
        if (rc != XFM_ELIMIT) return rc;
    }
    // sample chunks: enough waves to fill the chip, few enough that the fp32 atomics stay small
    const int dtiles = d_inner / 32;
    int bchunk = 1;
    while ((int64_t)4 * dtiles * ((batch + bchunk - 1) / bchunk) > 2048 && bchunk < batch) bchunk *= 2;
    a.bchunk = bchunk;
    const unsigned g2 = (unsigned)((4 * dtiles * ((batch + bchunk - 1) / bchunk) + 3) / 4);
    if (KT == 1) hipLaunchKernelGGL(chan_dwdt_kernel<1>, dim3(g2), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(chan_dwdt_kernel<2>, dim3(g2), dim3(256), 0, s, a);
    return check_launch();
}
