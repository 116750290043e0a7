// ss2d_fused.hip -- fused SS2D core for gfx950: cross-scan + 4-route selective scan + cross-merge.
//
// Replaces the operator chain of SS2Dv2.forward_corev2 (models/fusion_vmamba.py:1145-1174:
// cross_scan_fn -> selective_scan_fn -> cross_merge_fn) and its autograd mirror.  The reference
// materialises xs (B,4,D,L), ys (B,4,D,L fp32) and their gradients in HBM; here a wavefront owns a
// tile of G = 64/LPR feature-map planes (b, d..d+G-1), stages them ONCE in LDS (row pitch odd, so
// the column walks of routes 1/3 are bank-conflict free) and runs the four routes as four sweeps of
// the same register-chunk scan used by selective_scan.hip:
//     route 0: row-major ascending      route 2: the same sequence descending
//     route 1: column-major ascending   route 3: the same sequence descending
// A descending route is not a re-indexing: the lane<->chunk map and the in-chunk element order are
// mirrored at load time, after which the scan code is identical.  Per-route operands (dts, Bs, Cs)
// arrive contiguous in the route's own order (layout contract in include/xfm_hip.h), so every HBM
// access has lanes along the contiguous axis.  The merge is an in-LDS accumulation owned by the same
// wave (deterministic order 0,2,1,3), written back once.
//
// Algorithmic HBM bytes per (b,d,p) element, bf16 I/O: forward 2 (x) + 8 (dts) + 4 (y fp32) = 14;
// backward 2 + 8 + 4 (dy) + 2 (dx) + 8 (ddts) = 24; the unfused chain moves 48 / 84.
#include "scan_core.hpp"

namespace xfm {

struct SS2DArgs {
    xfm_ss2d_params_t p;
    int lg_lpr, n_chunks;
    int PW, PSZ;                 // LDS row pitch (odd) and plane size in floats
    int lds_floats_per_wave;
    int waves_per_block;
    uint32_t magicW;             // ceil(2^32 / W): e / W == __umulhi(e, magicW) for e < 2^16
};

// LDS plane offsets of the C register elements of this lane's chunk (or -1 past the end).
// tp0 = physical (ascending) index of the chunk's first element in the route's own order.
template <int C, bool COL, bool REV>
__device__ __forceinline__ void chunk_offsets(int tp0, int H, int W, int PW, int L, int (&off)[C]) {
    const int inner = COL ? H : W;
    int a = tp0 / inner, b = tp0 - a * inner;
#pragma unroll
    for (int q = 0; q < C; ++q) {
        const int j = REV ? C - 1 - q : q;
        off[j] = (tp0 + q < L) ? (COL ? b * PW + a : a * PW + b) : -1;
        if (++b == inner) {
            b = 0;
            ++a;
        }
    }
}

template <typename T>
__device__ __forceinline__ void planes_load(float *pl, const T *src, int G, int L, int PW, int PSZ, uint32_t magicW,
                                            int W, int lane) {
    for (int g = 0; g < G; ++g)
        for (int e = lane; e < L; e += 64) {
            const int h = magicW ? (int)__umulhi((uint32_t)e, magicW) : e, w = e - h * W;   // magicW == 0 <=> W == 1
            pl[g * PSZ + h * PW + w] = ldf<T>(src + (int64_t)g * L + e);
        }
}

template <typename T>
__device__ __forceinline__ void planes_store(const float *pl, T *dst, int G, int L, int PW, int PSZ, uint32_t magicW,
                                             int W, int lane) {
    for (int g = 0; g < G; ++g)
        for (int e = lane; e < L; e += 64) {
            const int h = magicW ? (int)__umulhi((uint32_t)e, magicW) : e, w = e - h * W;   // magicW == 0 <=> W == 1
            stf<T>(dst + (int64_t)g * L + e, pl[g * PSZ + h * PW + w]);
        }
}

// ---------------------------------------------------------------------------------------------
// one route, forward
// ---------------------------------------------------------------------------------------------
template <typename Tin, int C, bool COL, bool REV>
__device__ __forceinline__ void sweep_fwd(const SS2DArgs &a, const int k, float *buf, float *carry, const float *xg,
                                          float *yg, const bool first, const int b, const int d0, const int g,
                                          const int i, const int lane) {
    const xfm_ss2d_params_t &p = a.p;
    const int lg = a.lg_lpr, LPR = 1 << lg, G = 64 >> lg;
    const int N = p.dstate, H = p.H, W = p.W, L = H * W, D = p.d_inner, SL = C << lg, nseg = a.n_chunks;
    const int d = d0 + g, row = k * D + d;
    const Tin *dts_t = (const Tin *)p.dts + (((int64_t)b * 4 + k) * D + d0) * L;
    const Tin *Bg = (const Tin *)p.Bs + ((int64_t)b * 4 + k) * N * L;
    const Tin *Cg = (const Tin *)p.Cs + ((int64_t)b * 4 + k) * N * L;
    const float *Ar = p.A + (int64_t)row * N;
    const float Dr = p.D[row], bias = p.delta_bias[row];
    const int ci = REV ? LPR - 1 - i : i;          // physical chunk this lane owns
    const int slot = (g << lg) + ci;
    for (int n = i; n < N; n += LPR) carry[g * N + n] = 0.f;
    wave_sync();
    for (int s = 0; s < nseg; ++s) {
        const int s0 = (REV ? nseg - 1 - s : s) * SL;
        const int tp0 = s0 + ci * C;
        float dl[C], u[C], y[C];
        int off[C];
        tile_load<Tin, C, REV>(buf, dts_t, (int64_t)L, G, lg, s0, L, lane, slot, dl);
        chunk_offsets<C, COL, REV>(tp0, H, W, a.PW, L, off);
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const bool ok = off[j] >= 0;
            u[j] = ok ? xg[off[j]] : 0.f;
            float v = dl[j] + bias;
            if (p.delta_softplus) v = softplus20(v);
            dl[j] = ok ? v : 0.f;
            y[j] = 0.f;
        }
        for (int n = 0; n < N; ++n) {
            const float A2 = Ar[n] * kLog2e;
            const Tin *Bn = Bg + (int64_t)n * L + tp0;
            const Tin *Cn = Cg + (int64_t)n * L + tp0;
            float Bv[C], Cv[C], av[C];
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const int q = REV ? C - 1 - j : j;
                const bool ok = off[j] >= 0;
                Bv[j] = ok ? ldf<Tin>(Bn + q) : 0.f;
                Cv[j] = ok ? ldf<Tin>(Cn + q) : 0.f;
            }
            float P = 1.f, S = 0.f;
#pragma unroll
            for (int j = 0; j < C; ++j) {
                av[j] = exp2_fast(dl[j] * A2);
                Bv[j] *= dl[j] * u[j];
                S = fmaf(av[j], S, Bv[j]);
                P *= av[j];
            }
            float h = carry[g * N + n];
            if (LPR > 1) {
                seg_scan_up(P, S, i, LPR);
                const float Pe = __shfl_up(P, 1, LPR), Se = __shfl_up(S, 1, LPR);
                if (i > 0) h = fmaf(Pe, h, Se);
            }
#pragma unroll
            for (int j = 0; j < C; ++j) {
                h = fmaf(av[j], h, Bv[j]);
                y[j] = fmaf(Cv[j], h, y[j]);
            }
            if (i == LPR - 1) {
                carry[g * N + n] = h;
                if (nseg > 1) p.chk[((((int64_t)b * 4 + k) * D + d) * nseg + s) * N + n] = h;
            }
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            if (off[j] >= 0) {
                const float v = fmaf(Dr, u[j], y[j]);
                yg[off[j]] = first ? v : yg[off[j]] + v;
            }
        }
    }
    wave_sync();
}

template <typename Tin, typename Tout, int C>
__global__ void __launch_bounds__(256) ss2d_fwd_kernel(const SS2DArgs a) {
    extern __shared__ float smem[];
    const xfm_ss2d_params_t &p = a.p;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lg = a.lg_lpr, LPR = 1 << lg, G = 64 >> lg;
    const int tiles_pb = p.d_inner >> (6 - lg);
    const int64_t tile = (int64_t)blockIdx.x * a.waves_per_block + wave;
    if (tile >= (int64_t)p.batch * tiles_pb) return;
    const int b = (int)(tile / tiles_pb);
    const int d0 = (int)(tile - (int64_t)b * tiles_pb) * G;
    const int g = lane >> lg, i = lane & (LPR - 1);
    const int L = p.H * p.W;

    float *buf = smem + (size_t)wave * a.lds_floats_per_wave;
    float *carry = buf + 64 * (C | 1);
    float *xpl = carry + G * p.dstate;
    float *ypl = xpl + G * a.PSZ;
    planes_load<Tin>(xpl, (const Tin *)p.x + ((int64_t)b * p.d_inner + d0) * L, G, L, a.PW, a.PSZ, a.magicW, p.W, lane);
    wave_sync();
    const float *xg = xpl + g * a.PSZ;
    float *yg = ypl + g * a.PSZ;
    sweep_fwd<Tin, C, false, false>(a, 0, buf, carry, xg, yg, true, b, d0, g, i, lane);
    sweep_fwd<Tin, C, false, true>(a, 2, buf, carry, xg, yg, false, b, d0, g, i, lane);
    sweep_fwd<Tin, C, true, false>(a, 1, buf, carry, xg, yg, false, b, d0, g, i, lane);
    sweep_fwd<Tin, C, true, true>(a, 3, buf, carry, xg, yg, false, b, d0, g, i, lane);
    planes_store<Tout>(ypl, (Tout *)p.y + ((int64_t)b * p.d_inner + d0) * L, G, L, a.PW, a.PSZ, a.magicW, p.W, lane);
}

// ---------------------------------------------------------------------------------------------
// one route, backward (chunks walked against the route's direction; see selective_scan.hip)
// ---------------------------------------------------------------------------------------------
template <typename Tin, int C, bool COL, bool REV>
__device__ __forceinline__ void sweep_bwd(const SS2DArgs &a, const int k, float *buf, float *carryE, const float *xg,
                                          const float *gg, float *dxg, const bool first, const int b, const int d0,
                                          const int g, const int i, const int lane) {
    const xfm_ss2d_params_t &p = a.p;
    const int lg = a.lg_lpr, LPR = 1 << lg, G = 64 >> lg;
    const int N = p.dstate, H = p.H, W = p.W, L = H * W, D = p.d_inner, SL = C << lg, nseg = a.n_chunks;
    const int d = d0 + g, row = k * D + d;
    const Tin *dts_t = (const Tin *)p.dts + (((int64_t)b * 4 + k) * D + d0) * L;
    Tin *ddts_t = (Tin *)p.ddts + (((int64_t)b * 4 + k) * D + d0) * L;
    const Tin *Bg = (const Tin *)p.Bs + ((int64_t)b * 4 + k) * N * L;
    const Tin *Cg = (const Tin *)p.Cs + ((int64_t)b * 4 + k) * N * L;
    float *dBg = p.dBs + ((int64_t)b * 4 + k) * N * L;
    float *dCg = p.dCs + ((int64_t)b * 4 + k) * N * L;
    const float *Ar = p.A + (int64_t)row * N;
    const float Dr = p.D[row], bias = p.delta_bias[row];
    const int ci = REV ? LPR - 1 - i : i;
    const int slot = (g << lg) + ci;
    for (int n = i; n < N; n += LPR) carryE[g * N + n] = 0.f;
    wave_sync();
    float dD_acc = 0.f, dbias_acc = 0.f;
    for (int s = nseg - 1; s >= 0; --s) {
        const int s0 = (REV ? nseg - 1 - s : s) * SL;
        const int tp0 = s0 + ci * C;
        float dl[C], u[C], go[C], s1[C], s2[C];
        int off[C];
        tile_load<Tin, C, REV>(buf, dts_t, (int64_t)L, G, lg, s0, L, lane, slot, dl);
        chunk_offsets<C, COL, REV>(tp0, H, W, a.PW, L, off);
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const bool ok = off[j] >= 0;
            u[j] = ok ? xg[off[j]] : 0.f;
            go[j] = ok ? gg[off[j]] : 0.f;
            float v = dl[j] + bias;
            if (p.delta_softplus) v = softplus20(v);
            dl[j] = ok ? v : 0.f;
            s1[j] = 0.f;
            s2[j] = 0.f;
        }
        for (int n = 0; n < N; ++n) {
            const float An = Ar[n];
            const float A2 = An * kLog2e;
            const Tin *Bn = Bg + (int64_t)n * L + tp0;
            const Tin *Cn = Cg + (int64_t)n * L + tp0;
            float Bv[C], cg[C], av[C], h[C];
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const int q = REV ? C - 1 - j : j;
                const bool ok = off[j] >= 0;
                Bv[j] = ok ? ldf<Tin>(Bn + q) : 0.f;
                cg[j] = ok ? ldf<Tin>(Cn + q) * go[j] : 0.f;
            }
            float P = 1.f, S = 0.f;
#pragma unroll
            for (int j = 0; j < C; ++j) {
                av[j] = exp2_fast(dl[j] * A2);
                S = fmaf(av[j], S, dl[j] * u[j] * Bv[j]);
                P *= av[j];
            }
            float R = 0.f;
#pragma unroll
            for (int j = C - 1; j >= 0; --j) R = av[j] * (cg[j] + R);
            float hin = (s > 0) ? p.chk[((((int64_t)b * 4 + k) * D + d) * nseg + (s - 1)) * N + n] : 0.f;
            float Ein = carryE[g * N + n];
            if (LPR > 1) {
                float P2 = P;
                seg_scan_up(P, S, i, LPR);
                const float Pe = __shfl_up(P, 1, LPR), Se = __shfl_up(S, 1, LPR);
                if (i > 0) hin = fmaf(Pe, hin, Se);
                seg_scan_down(P2, R, i, LPR);
                const float Pn = __shfl_down(P2, 1, LPR), Rn = __shfl_down(R, 1, LPR);
                if (i < LPR - 1) Ein = fmaf(Pn, Ein, Rn);
            }
            float hh = hin;
#pragma unroll
            for (int j = 0; j < C; ++j) {
                hh = fmaf(av[j], hh, dl[j] * u[j] * Bv[j]);
                h[j] = hh;
            }
            float E = Ein, dA_acc = 0.f;
#pragma unroll
            for (int j = C - 1; j >= 0; --j) {
                const int q = REV ? C - 1 - j : j;
                const float dh = cg[j] + E;
                E = av[j] * dh;
                const float du_ = dl[j] * u[j];
                const float ah = h[j] - du_ * Bv[j];
                s1[j] = fmaf(dh, Bv[j], s1[j]);
                s2[j] = fmaf(dh * An, ah, s2[j]);
                dA_acc = fmaf(dh * dl[j], ah, dA_acc);
                float dBv = dh * du_;
                float dCv = go[j] * h[j];
                for (int o = LPR; o < 64; o <<= 1) {   // sum over the G planes of the tile
                    dBv += __shfl_xor(dBv, o, 64);
                    dCv += __shfl_xor(dCv, o, 64);
                }
                if (g == 0 && tp0 + q < L) {
                    atomicAdd(dBg + (int64_t)n * L + tp0 + q, dBv);
                    atomicAdd(dCg + (int64_t)n * L + tp0 + q, dCv);
                }
            }
            if (i == 0) carryE[g * N + n] = E;
            for (int o = 1; o < LPR; o <<= 1) dA_acc += __shfl_xor(dA_acc, o, 64);
            if (i == 0) atomicAdd(p.dA + (int64_t)row * N + n, dA_acc);
        }
        float dd[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const bool ok = off[j] >= 0;
            const float du = fmaf(dl[j], s1[j], Dr * go[j]);
            float ddl = fmaf(u[j], s1[j], s2[j]);
            if (p.delta_softplus && dl[j] <= 20.f) ddl *= 1.f - __expf(-dl[j]);
            dd[j] = ddl;
            dD_acc = fmaf(go[j], u[j], dD_acc);
            dbias_acc += ok ? ddl : 0.f;
            if (ok) dxg[off[j]] = first ? du : dxg[off[j]] + du;
        }
        tile_store<Tin, C, REV>(buf, ddts_t, (int64_t)L, G, lg, s0, L, lane, slot, dd);
    }
    for (int o = 1; o < LPR; o <<= 1) {
        dD_acc += __shfl_xor(dD_acc, o, 64);
        dbias_acc += __shfl_xor(dbias_acc, o, 64);
    }
    if (i == 0) {
        atomicAdd(p.dD + row, dD_acc);
        atomicAdd(p.ddelta_bias + row, dbias_acc);
    }
    wave_sync();
}

template <typename Tin, typename Tout, int C>
__global__ void __launch_bounds__(256) ss2d_bwd_kernel(const SS2DArgs a) {
    extern __shared__ float smem[];
    const xfm_ss2d_params_t &p = a.p;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lg = a.lg_lpr, LPR = 1 << lg, G = 64 >> lg;
    const int tiles_pb = p.d_inner >> (6 - lg);
    const int64_t tile = (int64_t)blockIdx.x * a.waves_per_block + wave;
    if (tile >= (int64_t)p.batch * tiles_pb) return;
    const int b = (int)(tile / tiles_pb);
    const int d0 = (int)(tile - (int64_t)b * tiles_pb) * G;
    const int g = lane >> lg, i = lane & (LPR - 1);
    const int L = p.H * p.W;

    float *buf = smem + (size_t)wave * a.lds_floats_per_wave;
    float *carryE = buf + 64 * (C | 1);
    float *xpl = carryE + G * p.dstate;
    float *gpl = xpl + G * a.PSZ;
    float *dxpl = gpl + G * a.PSZ;
    const int64_t po = ((int64_t)b * p.d_inner + d0) * L;
    planes_load<Tin>(xpl, (const Tin *)p.x + po, G, L, a.PW, a.PSZ, a.magicW, p.W, lane);
    planes_load<Tout>(gpl, (const Tout *)p.dy + po, G, L, a.PW, a.PSZ, a.magicW, p.W, lane);
    wave_sync();
    const float *xg = xpl + g * a.PSZ, *gg = gpl + g * a.PSZ;
    float *dxg = dxpl + g * a.PSZ;
    sweep_bwd<Tin, C, false, false>(a, 0, buf, carryE, xg, gg, dxg, true, b, d0, g, i, lane);
    sweep_bwd<Tin, C, false, true>(a, 2, buf, carryE, xg, gg, dxg, false, b, d0, g, i, lane);
    sweep_bwd<Tin, C, true, false>(a, 1, buf, carryE, xg, gg, dxg, false, b, d0, g, i, lane);
    sweep_bwd<Tin, C, true, true>(a, 3, buf, carryE, xg, gg, dxg, false, b, d0, g, i, lane);
    planes_store<Tin>(dxpl, (Tin *)p.dx + po, G, L, a.PW, a.PSZ, a.magicW, p.W, lane);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static const int kItems2[] = {4, 7, 9, 13};
static const size_t kLdsPerCU = 160 * 1024;

struct Plan2 {
    int lg, items, n_chunks, waves_per_block;
    size_t lds_wave_floats;
};

static int plan_ss2d(int batch, int D, int H, int W, int N, int n_planes_in_lds, Plan2 *out) {
    if (batch <= 0 || D <= 0 || H <= 0 || W <= 0 || N <= 0) return XFM_EINVAL;
    if (N > 256 || (int64_t)H * W >= 65536) return XFM_ELIMIT;
    const int L = H * W, PSZ = H * (W | 1);
    double best = 1e300;
    bool found = false;
    for (int lg = 0; lg <= 6; ++lg) {
        const int G = 64 >> lg;
        if (D % G) continue;
        if (G * N > 2048) continue;
        for (int c : kItems2) {
            const int SL = c << lg;
            const int nseg = (L + SL - 1) / SL;
            // sized for the backward (3 planes) so forward and backward share the chunking (chk layout)
            const size_t fl = (size_t)64 * (c | 1) + (size_t)G * N + (size_t)3 * G * PSZ;
            if (fl * sizeof(float) > kLdsPerCU) continue;
            const double resident = std::min<double>(16.0, (double)(kLdsPerCU / (fl * sizeof(float))));
            const double row = 4.0 * nseg * (N * (c * 1.0 + lg * 0.8 + 1.5) + c * 0.8 + 2.0) + 2.0 * L / 64.0 * G;
            const double waves = (double)batch * D / G;
            const double rounds = std::max(1.0, waves / (256.0 * resident));
            const double est = row * rounds * (1.0 + 2.0 / resident);   // few resident waves hide latency badly
            if (est < best * 0.999) {
                best = est;
                found = true;
                out->lg = lg;
                out->items = c;
                out->n_chunks = nseg;
            }
        }
    }
    if (!found) return XFM_ELIMIT;
    const int G = 64 >> out->lg;
    out->lds_wave_floats = (size_t)64 * (out->items | 1) + (size_t)G * N + (size_t)n_planes_in_lds * G * PSZ;
    int wpb = (int)std::min<size_t>(4, kLdsPerCU / (out->lds_wave_floats * sizeof(float)));
    out->waves_per_block = wpb < 1 ? 1 : wpb;
    return XFM_OK;
}

template <typename Tin, typename Tout, int C>
static int launch2(const SS2DArgs &a, bool bwd, hipStream_t s) {
    const int G = 64 >> a.lg_lpr;
    const int64_t tiles = (int64_t)a.p.batch * (a.p.d_inner / G);
    const unsigned grid = (unsigned)((tiles + a.waves_per_block - 1) / a.waves_per_block);
    const size_t lds = (size_t)a.waves_per_block * a.lds_floats_per_wave * sizeof(float);
    const dim3 block(64 * a.waves_per_block);
    if (bwd) {
        auto kern = ss2d_bwd_kernel<Tin, Tout, C>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid), block, lds, s, a);
    } else {
        auto kern = ss2d_fwd_kernel<Tin, Tout, C>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid), block, lds, s, a);
    }
    return check_launch();
}

template <typename Tin, typename Tout>
static int dispatch2(const SS2DArgs &a, int items, bool bwd, hipStream_t s) {
    switch (items) {
        case 4: return launch2<Tin, Tout, 4>(a, bwd, s);
        case 7: return launch2<Tin, Tout, 7>(a, bwd, s);
        case 9: return launch2<Tin, Tout, 9>(a, bwd, s);
        case 13: return launch2<Tin, Tout, 13>(a, bwd, s);
    }
    return XFM_EINVAL;
}

static int run2(const xfm_ss2d_params_t *p, bool bwd, void *stream) {
    if (!p || !p->x || !p->dts || !p->Bs || !p->Cs || !p->A || !p->D || !p->delta_bias) return XFM_EINVAL;
    if (!bwd && !p->y) return XFM_EINVAL;
    if (bwd && (!p->dy || !p->dx || !p->ddts || !p->dBs || !p->dCs || !p->dA || !p->dD || !p->ddelta_bias))
        return XFM_EINVAL;
    if (p->in_dtype < 0 || p->in_dtype > 2) return XFM_EDTYPE;
    if (p->out_dtype != XFM_F32 && p->out_dtype != p->in_dtype) return XFM_EDTYPE;
    Plan2 pl;
    int rc = plan_ss2d(p->batch, p->d_inner, p->H, p->W, p->dstate, bwd ? 3 : 2, &pl);
    if (rc) return rc;
    if (pl.n_chunks > 1 && !p->chk) return XFM_EINVAL;
    SS2DArgs a;
    a.p = *p;
    a.lg_lpr = pl.lg;
    a.n_chunks = pl.n_chunks;
    a.PW = p->W | 1;
    a.PSZ = p->H * a.PW;
    a.lds_floats_per_wave = (int)pl.lds_wave_floats;
    a.waves_per_block = pl.waves_per_block;
    a.magicW = (uint32_t)((0x100000000ull + p->W - 1) / p->W);
    hipStream_t s = (hipStream_t)stream;
    const bool of32 = p->out_dtype == XFM_F32;
    switch (p->in_dtype) {
        case XFM_F32: return dispatch2<float, float>(a, pl.items, bwd, s);
        case XFM_F16:
            return of32 ? dispatch2<f16_t, float>(a, pl.items, bwd, s) : dispatch2<f16_t, f16_t>(a, pl.items, bwd, s);
        case XFM_BF16:
            return of32 ? dispatch2<bf16_t, float>(a, pl.items, bwd, s) : dispatch2<bf16_t, bf16_t>(a, pl.items, bwd, s);
    }
    return XFM_EDTYPE;
}

}  // namespace xfm

extern "C" {
int xfm_ss2d_plan(int batch, int d_inner, int H, int W, int dstate, xfm_scan_plan_t *plan) {
    if (!plan) return XFM_EINVAL;
    xfm::Plan2 pl;
    const int rc = xfm::plan_ss2d(batch, d_inner, H, W, dstate, 3, &pl);
    if (rc) return rc;
    plan->lanes_per_row = 1 << pl.lg;
    plan->items = pl.items;
    plan->n_chunks = pl.n_chunks;
    return XFM_OK;
}
int xfm_ss2d_fwd(const xfm_ss2d_params_t *p, void *stream) { return xfm::run2(p, false, stream); }
int xfm_ss2d_bwd(const xfm_ss2d_params_t *p, void *stream) { return xfm::run2(p, true, stream); }
}
