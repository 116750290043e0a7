// ss2d_kernels.hpp -- fused SS2D core for gfx950: cross-scan + 4-route selective scan + cross-merge.
//
// Replaces the operator chain of SS2Dv2.forward_corev2 (models/fusion_vmamba.py:1145-1174:
// cross_scan_fn -> selective_scan_fn -> cross_merge_fn) and its autograd mirror.  The reference
// materialises xs (B,4,D,L), ys (B,4,D,L fp32) and their gradients in HBM; here the four routes
//     route 0: row-major ascending      route 2: the same sequence descending
//     route 1: column-major ascending   route 3: the same sequence descending
// run out of feature-map planes staged ONCE in LDS (odd row pitch: the column walks of routes 1/3
// are bank-conflict free).  A descending route is not a re-indexing: the lane<->chunk map and the
// in-chunk element order are mirrored at load time, after which the scan code is identical.
// Per-route operands (dts, Bs, Cs) arrive contiguous in the route's own order (layout contract in
// include/xfm_hip.h): they are fetched with 16-byte vector loads one chunk AHEAD of the compute
// (register prefetch), bounced through a small per-wave LDS tile to hand every lane its C
// consecutive elements (odd stride, conflict free), and never re-read.
//
// Two workgroup shapes share the sweep code:
//   kind 1 "wave per route": the 4 waves of a workgroup share the staged planes of a tile and each
//           owns one route; a workgroup walks `pli` tiles of one image, so the backward sums each
//           route's dB/dC over all of them in wave-private LDS and flushes with one atomic pass.
//   kind 0 "wave per tile": one wave walks the 4 routes of its own planes (fallback for planes
//           too large to share LDS with the accumulators).
//
// Algorithmic HBM bytes per (b,d,p) element, bf16 I/O: forward 2 (x) + 8 (dts) + 4 (y fp32) = 14;
// backward 2 + 8 + 4 (dy) + 2 (dx) + 8 (ddts) = 24; the unfused chain moves 48 / 84.
#pragma once

#include "scan_core.hpp"

namespace xfm {

struct SS2DArgs {
    xfm_ss2d_params_t p;
    int lg_lpr, n_chunks;
    int PW, PSZ;                 // LDS row pitch (odd) and plane size in floats
    int lds_floats_per_wave;
    int waves_per_block;
    int kind;                    // see above
    int pli;                     // kind 1: consecutive plane tiles handled by one workgroup
    int bc_floats;               // per-wave LDS region holding B,C of the route for all states (0: none)
    int dbg;                     // timing-only switches (XFM_SS2D_DBG): 1 skip sweeps, 2 skip plane loads, 4 skip merge
    uint32_t magicW;             // ceil(2^32 / W): e / W == __umulhi(e, magicW) for e < 2^16
};

// ---------------------------------------------------------------------------------------------
// 16-byte vector helpers
// ---------------------------------------------------------------------------------------------
template <typename T> struct Vec16 { static constexpr int VE = 16 / sizeof(T); };

template <typename T> __device__ __forceinline__ void unpack16(const uint4 &r, float *f);
template <> __device__ __forceinline__ void unpack16<float>(const uint4 &r, float *f) {
    f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const uint4 &r, float *f) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(w[i] << 16);
        f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
template <> __device__ __forceinline__ void unpack16<f16_t>(const uint4 &r, float *f) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __half2 h = *reinterpret_cast<const __half2 *>(&w[i]);
        f[2 * i] = __low2float(h);
        f[2 * i + 1] = __high2float(h);
    }
}

template <typename T> __device__ __forceinline__ uint4 pack16(const float *f);
template <> __device__ __forceinline__ uint4 pack16<float>(const float *f) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}
template <> __device__ __forceinline__ uint4 pack16<bf16_t>(const float *f) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        w[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}
template <> __device__ __forceinline__ uint4 pack16<f16_t>(const float *f) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __half2 h = __floats2half2_rn(f[2 * i], f[2 * i + 1]);
        w[i] = *reinterpret_cast<const uint32_t *>(&h);
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// A "run" is n contiguous elements in HBM.  run_issue() starts the 16-byte loads of a run into
// registers (NV vectors per lane cover 64*NV*VE >= n elements); run_commit() later converts them to
// fp32 and lays the run out flat in the LDS tile.  `room` = elements readable from src without
// leaving the tensor (a trailing vector may over-read inside the tensor; such lanes are masked by
// the consumer).  Runs whose base is not 16-byte aligned take the scalar path at commit time.
template <int NV> struct RunRegs { uint4 v[NV]; };

template <typename T, int NV>
__device__ __forceinline__ void run_issue(RunRegs<NV> &r, const T *src, int n, int64_t room, int lane) {
    constexpr int VE = Vec16<T>::VE;
    const bool vec = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
#pragma unroll
    for (int m = 0; m < NV; ++m) {
        const int e0 = (lane + 64 * m) * VE;
        r.v[m] = make_uint4(0, 0, 0, 0);
        if (vec && e0 < n && e0 + VE <= room) r.v[m] = *reinterpret_cast<const uint4 *>(src + e0);
    }
}

template <typename T, int NV>
__device__ __forceinline__ void run_commit(float *buf, const RunRegs<NV> &r, const T *src, int n, int64_t room, int lane) {
    constexpr int VE = Vec16<T>::VE;
    const bool vec = (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    if (vec) {
#pragma unroll
        for (int m = 0; m < NV; ++m) {
            const int e0 = (lane + 64 * m) * VE;
            if (e0 < n) {
                if (e0 + VE <= room) {
                    float f[VE];
                    unpack16<T>(r.v[m], f);
#pragma unroll
                    for (int e = 0; e < VE; e += 4)
                        *reinterpret_cast<float4 *>(buf + e0 + e) = make_float4(f[e], f[e + 1], f[e + 2], f[e + 3]);
                } else {
                    for (int e = 0; e < VE; ++e)
                        if (e0 + e < n) buf[e0 + e] = ldf<T>(src + e0 + e);
                }
            }
        }
    } else {
        for (int e = lane; e < n; e += 64) buf[e] = ldf<T>(src + e);
    }
}

// flat LDS tile -> HBM run (n elements), vectorised when dst is 16-byte aligned
template <typename T>
__device__ __forceinline__ void run_store(T *dst, const float *buf, int n, int lane) {
    constexpr int VE = Vec16<T>::VE;
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        const int nv = n / VE;
        for (int v = lane; v < nv; v += 64) {
            float f[VE];
#pragma unroll
            for (int e = 0; e < VE; e += 4) {
                const float4 q = *reinterpret_cast<const float4 *>(buf + v * VE + e);
                f[e] = q.x; f[e + 1] = q.y; f[e + 2] = q.z; f[e + 3] = q.w;
            }
            *reinterpret_cast<uint4 *>(dst + v * VE) = pack16<T>(f);
        }
        for (int e = nv * VE + lane; e < n; e += 64) stf<T>(dst + e, buf[e]);
    } else {
        for (int e = lane; e < n; e += 64) stf<T>(dst + e, buf[e]);
    }
}

// LDS plane offsets of the C register elements of this lane's chunk, in TRAVERSAL order (or -1 past
// the end).  tp0 = physical (ascending) index of the chunk's first element in the route's own order;
// a descending route walks the chunk from its last physical element backwards.
template <int C>
__device__ __forceinline__ void chunk_offsets(int tp0, int H, int W, int PW, int L, bool col, bool rev, int (&off)[C]) {
    const int inner = col ? H : W;
    int t = rev ? tp0 + C - 1 : tp0;
    int a = t / inner, b = t - a * inner;
    const int sa = col ? 1 : PW, sb = col ? PW : 1;      // offset = a*sa + b*sb
#pragma unroll
    for (int j = 0; j < C; ++j) {
        off[j] = (t < L) ? a * sa + b * sb : -1;
        if (rev) {
            --t;
            if (--b < 0) { b = inner - 1; --a; }
        } else {
            ++t;
            if (++b == inner) { b = 0; ++a; }
        }
    }
}

// HBM plane(s) <-> padded LDS plane(s); `nthreads` cooperating threads, `tid` this thread.
template <typename T>
__device__ __forceinline__ void planes_load(float *pl, const T *src, int G, int L, int PW, int PSZ, uint32_t magicW,
                                            int W, int tid, int nthreads) {
    constexpr int VE = Vec16<T>::VE;
    const int total = G * L;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0 && (total % VE) == 0) {
        for (int v = tid; v < total / VE; v += nthreads) {
            float f[VE];
            unpack16<T>(*reinterpret_cast<const uint4 *>(src + v * VE), f);
            int e = v * VE;
            int g = e / L, r = e - g * L;
            int h = magicW ? (int)__umulhi((uint32_t)r, magicW) : r, w = r - h * W;
#pragma unroll
            for (int q = 0; q < VE; ++q) {
                pl[g * PSZ + h * PW + w] = f[q];
                if (++w == W) {
                    w = 0;
                    if (++h * W >= L) {
                        h = 0;
                        ++g;
                    }
                }
            }
        }
    } else {
        for (int g = 0; g < G; ++g)
            for (int e = tid; e < L; e += nthreads) {
                const int h = magicW ? (int)__umulhi((uint32_t)e, magicW) : e, w = e - h * W;
                pl[g * PSZ + h * PW + w] = ldf<T>(src + (int64_t)g * L + e);
            }
    }
}

// ---------------------------------------------------------------------------------------------
// per-chunk operand staging shared by the forward and backward sweeps
// ---------------------------------------------------------------------------------------------
template <typename Tin, int C>
struct Stager {
    static constexpr int NV = (C * (int)sizeof(Tin) + 15) / 16;
    RunRegs<NV> rd, rb, rc;
};

// ---------------------------------------------------------------------------------------------
// one route, forward
// ---------------------------------------------------------------------------------------------
template <typename Tin, int C, bool N1>
__device__ __forceinline__ void sweep_fwd(const SS2DArgs &a, const int k, const bool COL, const bool REV, float *buf, float *carry, float *bc,
                                          const float *xg, float *yg, const bool first, const int b, const int d0,
                                          const int g, const int i, const int lane) {
    const xfm_ss2d_params_t &p = a.p;
    const int lg = a.lg_lpr, LPR = 1 << lg, G = 64 >> lg;
    const int N = p.dstate, H = p.H, W = p.W, L = H * W, D = p.d_inner, SL = C << lg, nseg = a.n_chunks;
    const int d = d0 + g, row = k * D + d;
    const int64_t dts_off = (((int64_t)b * 4 + k) * D + d0) * L, bc_off = ((int64_t)b * 4 + k) * N * L;
    const Tin *dts_t = (const Tin *)p.dts + dts_off;
    const Tin *Bg = (const Tin *)p.Bs + bc_off, *Cg = (const Tin *)p.Cs + bc_off;
    const int64_t dts_room = (int64_t)p.batch * 4 * D * L - dts_off, bc_room = (int64_t)p.batch * 4 * N * L - bc_off;
    const float *Ar = p.A + (int64_t)row * N;
    const float Dr = p.D[row], bias = p.delta_softplus == 2 ? 0.f : p.delta_bias[row];   // mode 2: dts = softplus(raw + bias) already
    const int ci = REV ? LPR - 1 - i : i;          // physical chunk this lane owns
    const bool one_run = nseg == 1 || G == 1;      // the tile's dts rows form one contiguous HBM run
    const int pitch = nseg == 1 ? L : SL;
    Stager<Tin, C> st;
    for (int n = i; n < N; n += LPR) carry[g * N + n] = 0.f;
    if (!N1 && bc) {   // B, C of this route for all states, staged once per tile (small: nseg == 1)
        for (int e = lane; e < N * L; e += 64) {
            bc[e] = ldf<Tin>(Bg + e);
            bc[N * L + e] = ldf<Tin>(Cg + e);
        }
    }
    auto seg0 = [&](int s) { return (REV ? nseg - 1 - s : s) * SL; };
    auto issue = [&](int s) {
        const int s0 = seg0(s);
        if (one_run) {
            const int off = nseg == 1 ? 0 : s0;
            run_issue<Tin, Stager<Tin, C>::NV>(st.rd, dts_t + off, nseg == 1 ? G * L : min(SL, L - s0), dts_room - off, lane);
        }
        if (N1) {
            run_issue<Tin, Stager<Tin, C>::NV>(st.rb, Bg + s0, min(SL, L - s0), bc_room - s0, lane);
            run_issue<Tin, Stager<Tin, C>::NV>(st.rc, Cg + s0, min(SL, L - s0), bc_room - s0, lane);
        }
    };
    issue(0);
    wave_sync();
    for (int s = 0; s < nseg; ++s) {
        const int s0 = seg0(s);
        const int tp0 = s0 + ci * C;
        float dl[C], u[C], y[C], Bv[C], Cv[C];
        int off[C];
        // ---- hand each lane its chunk of dts (and B, C when there is a single state) through the tile
        if (one_run) {
            const int o = nseg == 1 ? 0 : s0;
            run_commit<Tin, Stager<Tin, C>::NV>(buf, st.rd, dts_t + o, nseg == 1 ? G * L : min(SL, L - s0), dts_room - o, lane);
        } else {
            for (int gg = 0; gg < G; ++gg)
                for (int e = lane; e < min(SL, L - s0); e += 64) buf[gg * SL + e] = ldf<Tin>(dts_t + (int64_t)gg * L + s0 + e);
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < C; ++j) dl[j] = buf[g * pitch + ci * C + (REV ? C - 1 - j : j)];
        wave_sync();
        if (N1) {
            run_commit<Tin, Stager<Tin, C>::NV>(buf, st.rb, Bg + s0, min(SL, L - s0), bc_room - s0, lane);
            wave_sync();
#pragma unroll
            for (int j = 0; j < C; ++j) Bv[j] = buf[ci * C + (REV ? C - 1 - j : j)];
            wave_sync();
            run_commit<Tin, Stager<Tin, C>::NV>(buf, st.rc, Cg + s0, min(SL, L - s0), bc_room - s0, lane);
            wave_sync();
#pragma unroll
            for (int j = 0; j < C; ++j) Cv[j] = buf[ci * C + (REV ? C - 1 - j : j)];
            wave_sync();
        }
        if (s + 1 < nseg) issue(s + 1);            // next chunk's HBM loads fly during this chunk's maths
        chunk_offsets<C>(tp0, H, W, a.PW, L, COL, REV, off);
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const bool ok = off[j] >= 0;
            u[j] = ok ? xg[off[j]] : 0.f;
            float v = dl[j] + bias;
            if (p.delta_softplus == 1) v = softplus20(v);
            dl[j] = ok ? v : 0.f;
            y[j] = 0.f;
        }
        for (int n = 0; n < N; ++n) {
            const float A2 = Ar[n] * kLog2e;
            float av[C];
            if (!N1) {
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    const int q = REV ? C - 1 - j : j;
                    const bool ok = off[j] >= 0;
                    if (bc) {
                        Bv[j] = ok ? bc[n * L + tp0 + q] : 0.f;
                        Cv[j] = ok ? bc[(N + n) * L + tp0 + q] : 0.f;
                    } else {
                        Bv[j] = ok ? ldf<Tin>(Bg + (int64_t)n * L + tp0 + q) : 0.f;
                        Cv[j] = ok ? ldf<Tin>(Cg + (int64_t)n * L + tp0 + q) : 0.f;
                    }
                }
            }
            float P = 1.f, S = 0.f;
            float bb[C];
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const bool ok = off[j] >= 0;
                av[j] = exp2_fast(dl[j] * A2);
                bb[j] = ok ? dl[j] * u[j] * Bv[j] : 0.f;
                S = fmaf(av[j], S, bb[j]);
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
                h = fmaf(av[j], h, bb[j]);
                y[j] = fmaf(off[j] >= 0 ? Cv[j] : 0.f, h, y[j]);
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

// ---------------------------------------------------------------------------------------------
// one route, backward (chunks walked against the route's direction; maths: selective_scan.hip)
// ---------------------------------------------------------------------------------------------
template <typename Tin, int C, bool N1>
__device__ __forceinline__ void sweep_bwd(const SS2DArgs &a, const int k, const bool COL, const bool REV, float *buf, float *carryE, float *bc,
                                          const float *xg, const float *gg, float *dxg, const bool first, const int b,
                                          const int d0, const int g, const int i, const int lane, float *acc) {
    const xfm_ss2d_params_t &p = a.p;
    const int lg = a.lg_lpr, LPR = 1 << lg, G = 64 >> lg;
    const int N = p.dstate, H = p.H, W = p.W, L = H * W, D = p.d_inner, SL = C << lg, nseg = a.n_chunks;
    const int d = d0 + g, row = k * D + d;
    const int64_t dts_off = (((int64_t)b * 4 + k) * D + d0) * L, bc_off = ((int64_t)b * 4 + k) * N * L;
    const Tin *dts_t = (const Tin *)p.dts + dts_off;
    Tin *ddts_t = (Tin *)p.ddts + dts_off;
    const Tin *Bg = (const Tin *)p.Bs + bc_off, *Cg = (const Tin *)p.Cs + bc_off;
    const int64_t dts_room = (int64_t)p.batch * 4 * D * L - dts_off, bc_room = (int64_t)p.batch * 4 * N * L - bc_off;
    float *dBg = p.dBs + bc_off, *dCg = p.dCs + bc_off;
    const float *Ar = p.A + (int64_t)row * N;
    const float Dr = p.D[row], bias = p.delta_softplus == 2 ? 0.f : p.delta_bias[row];   // mode 2: dts = softplus(raw + bias) already
    const int ci = REV ? LPR - 1 - i : i;
    const bool one_run = nseg == 1 || G == 1;
    const int pitch = nseg == 1 ? L : SL;
    Stager<Tin, C> st;
    for (int n = i; n < N; n += LPR) carryE[g * N + n] = 0.f;
    if (!N1 && bc) {
        for (int e = lane; e < N * L; e += 64) {
            bc[e] = ldf<Tin>(Bg + e);
            bc[N * L + e] = ldf<Tin>(Cg + e);
        }
    }
    auto seg0 = [&](int s) { return (REV ? nseg - 1 - s : s) * SL; };
    auto issue = [&](int s) {
        const int s0 = seg0(s);
        if (one_run) {
            const int off = nseg == 1 ? 0 : s0;
            run_issue<Tin, Stager<Tin, C>::NV>(st.rd, dts_t + off, nseg == 1 ? G * L : min(SL, L - s0), dts_room - off, lane);
        }
        if (N1) {
            run_issue<Tin, Stager<Tin, C>::NV>(st.rb, Bg + s0, min(SL, L - s0), bc_room - s0, lane);
            run_issue<Tin, Stager<Tin, C>::NV>(st.rc, Cg + s0, min(SL, L - s0), bc_room - s0, lane);
        }
    };
    issue(nseg - 1);
    wave_sync();
    float dD_acc = 0.f, dbias_acc = 0.f;
    for (int s = nseg - 1; s >= 0; --s) {
        const int s0 = seg0(s);
        const int tp0 = s0 + ci * C;
        const int run_n = nseg == 1 ? G * L : min(SL, L - s0);
        const int run_o = nseg == 1 ? 0 : s0;
        float dl[C], u[C], go[C], s1[C], s2[C], Bv[C], Cv[C];
        int off[C];
        if (one_run) {
            run_commit<Tin, Stager<Tin, C>::NV>(buf, st.rd, dts_t + run_o, run_n, dts_room - run_o, lane);
        } else {
            for (int g2 = 0; g2 < G; ++g2)
                for (int e = lane; e < min(SL, L - s0); e += 64) buf[g2 * SL + e] = ldf<Tin>(dts_t + (int64_t)g2 * L + s0 + e);
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < C; ++j) dl[j] = buf[g * pitch + ci * C + (REV ? C - 1 - j : j)];
        wave_sync();
        if (N1) {
            run_commit<Tin, Stager<Tin, C>::NV>(buf, st.rb, Bg + s0, min(SL, L - s0), bc_room - s0, lane);
            wave_sync();
#pragma unroll
            for (int j = 0; j < C; ++j) Bv[j] = buf[ci * C + (REV ? C - 1 - j : j)];
            wave_sync();
            run_commit<Tin, Stager<Tin, C>::NV>(buf, st.rc, Cg + s0, min(SL, L - s0), bc_room - s0, lane);
            wave_sync();
#pragma unroll
            for (int j = 0; j < C; ++j) Cv[j] = buf[ci * C + (REV ? C - 1 - j : j)];
            wave_sync();
        }
        if (s > 0) issue(s - 1);
        chunk_offsets<C>(tp0, H, W, a.PW, L, COL, REV, off);
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const bool ok = off[j] >= 0;
            u[j] = ok ? xg[off[j]] : 0.f;
            go[j] = ok ? gg[off[j]] : 0.f;
            float v = dl[j] + bias;
            if (p.delta_softplus == 1) v = softplus20(v);
            dl[j] = ok ? v : 0.f;
            s1[j] = 0.f;
            s2[j] = 0.f;
        }
        for (int n = 0; n < N; ++n) {
            const float An = Ar[n];
            const float A2 = An * kLog2e;
            float cg[C], av[C], h[C], bb[C];
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const int q = REV ? C - 1 - j : j;
                const bool ok = off[j] >= 0;
                if (!N1) {
                    if (bc) {
                        Bv[j] = ok ? bc[n * L + tp0 + q] : 0.f;
                        Cv[j] = ok ? bc[(N + n) * L + tp0 + q] : 0.f;
                    } else {
                        Bv[j] = ok ? ldf<Tin>(Bg + (int64_t)n * L + tp0 + q) : 0.f;
                        Cv[j] = ok ? ldf<Tin>(Cg + (int64_t)n * L + tp0 + q) : 0.f;
                    }
                } else if (!ok) {
                    Bv[j] = 0.f;
                    Cv[j] = 0.f;
                }
                cg[j] = Cv[j] * go[j];
            }
            float P = 1.f, S = 0.f;
#pragma unroll
            for (int j = 0; j < C; ++j) {
                av[j] = exp2_fast(dl[j] * A2);
                bb[j] = dl[j] * u[j] * Bv[j];
                S = fmaf(av[j], S, bb[j]);
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
                hh = fmaf(av[j], hh, bb[j]);
                h[j] = hh;
            }
            float E = Ein, dA_acc = 0.f;
#pragma unroll
            for (int j = C - 1; j >= 0; --j) {
                const int q = REV ? C - 1 - j : j;
                const float dh = cg[j] + E;
                E = av[j] * dh;
                const float du_ = dl[j] * u[j];
                const float ah = h[j] - bb[j];
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
                    if (acc) {   // wave-private LDS accumulators (wave-per-route kernel): plain RMW, flushed once
                        acc[n * L + tp0 + q] += dBv;
                        acc[(N + n) * L + tp0 + q] += dCv;
                    } else {
                        atomicAdd(dBg + (int64_t)n * L + tp0 + q, dBv);
                        atomicAdd(dCg + (int64_t)n * L + tp0 + q, dCv);
                    }
                }
            }
            if (i == 0) carryE[g * N + n] = E;
            for (int o = 1; o < LPR; o <<= 1) dA_acc += __shfl_xor(dA_acc, o, 64);
            if (i == 0) atomicAdd(p.dA + (int64_t)row * N + n, dA_acc);
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const bool ok = off[j] >= 0;
            const float du = fmaf(dl[j], s1[j], Dr * go[j]);
            float ddl = fmaf(u[j], s1[j], s2[j]);
            if (p.delta_softplus && dl[j] <= 20.f) ddl *= 1.f - __expf(-dl[j]);
            dD_acc = fmaf(go[j], u[j], dD_acc);
            dbias_acc += ok ? ddl : 0.f;
            if (ok) {
                if (acc) atomicAdd(dxg + off[j], du);   // plane shared by the 4 route-waves: LDS atomic (ds_add_f32)
                else dxg[off[j]] = first ? du : dxg[off[j]] + du;
                buf[g * pitch + ci * C + (REV ? C - 1 - j : j)] = ddl;   // only in-range slots: rows are packed
            }
        }
        wave_sync();
        if (one_run) {
            run_store<Tin>(ddts_t + run_o, buf, run_n, lane);
        } else {
            for (int g2 = 0; g2 < G; ++g2)
                for (int e = lane; e < min(SL, L - s0); e += 64) stf<Tin>(ddts_t + (int64_t)g2 * L + s0 + e, buf[g2 * SL + e]);
        }
        wave_sync();
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

// ---------------------------------------------------------------------------------------------
// kind 0 kernels: one wave walks the 4 routes of its own tile
// per-wave LDS: tile [64*C+16] | carry [G*N] | bc [bc_floats] | planes
// ---------------------------------------------------------------------------------------------
template <typename Tin, typename Tout, int C, bool N1>
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
    float *carry = buf + 64 * C + 16;
    float *bc = a.bc_floats ? carry + G * p.dstate : nullptr;
    float *xpl = carry + G * p.dstate + a.bc_floats;
    float *ypl = xpl + G * a.PSZ;
    const int64_t po = ((int64_t)b * p.d_inner + d0) * L;
    planes_load<Tin>(xpl, (const Tin *)p.x + po, G, L, a.PW, a.PSZ, a.magicW, p.W, lane, 64);
    wave_sync();
    const float *xg = xpl + g * a.PSZ;
    float *yg = ypl + g * a.PSZ;
    for (int r = 0; r < 4; ++r)      // routes in the fixed order 0, 2, 1, 3 (one inlined copy of the sweep)
        sweep_fwd<Tin, C, N1>(a, (r & 1) * 2 + (r >> 1), (r >> 1) != 0, (r & 1) != 0, buf, carry, bc, xg, yg, r == 0, b, d0, g, i, lane);
    Tout *yo = (Tout *)p.y + po;
    for (int g2 = 0; g2 < G; ++g2)
        for (int e = lane; e < L; e += 64) {
            const int h = a.magicW ? (int)__umulhi((uint32_t)e, a.magicW) : e, w = e - h * p.W;
            stf<Tout>(yo + (int64_t)g2 * L + e, ypl[g2 * a.PSZ + h * a.PW + w]);
        }
}

template <typename Tin, typename Tout, int C, bool N1>
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
    float *carryE = buf + 64 * C + 16;
    float *bc = a.bc_floats ? carryE + G * p.dstate : nullptr;
    float *xpl = carryE + G * p.dstate + a.bc_floats;
    float *gpl = xpl + G * a.PSZ;
    float *dxpl = gpl + G * a.PSZ;
    const int64_t po = ((int64_t)b * p.d_inner + d0) * L;
    planes_load<Tin>(xpl, (const Tin *)p.x + po, G, L, a.PW, a.PSZ, a.magicW, p.W, lane, 64);
    planes_load<Tout>(gpl, (const Tout *)p.dy + po, G, L, a.PW, a.PSZ, a.magicW, p.W, lane, 64);
    wave_sync();
    const float *xg = xpl + g * a.PSZ, *gg = gpl + g * a.PSZ;
    float *dxg = dxpl + g * a.PSZ;
    for (int r = 0; r < 4; ++r)
        sweep_bwd<Tin, C, N1>(a, (r & 1) * 2 + (r >> 1), (r >> 1) != 0, (r & 1) != 0, buf, carryE, bc, xg, gg, dxg, r == 0, b, d0, g, i, lane, nullptr);
    Tin *dxo = (Tin *)p.dx + po;
    for (int g2 = 0; g2 < G; ++g2)
        for (int e = lane; e < L; e += 64) {
            const int h = a.magicW ? (int)__umulhi((uint32_t)e, a.magicW) : e, w = e - h * p.W;
            stf<Tin>(dxo + (int64_t)g2 * L + e, dxpl[g2 * a.PSZ + h * a.PW + w]);
        }
}

// ---------------------------------------------------------------------------------------------
// kind 1 kernels: wave per route (wave 0..3 -> route 0, 2, 1, 3), planes shared by the workgroup
// forward  LDS: xpl [G*PSZ] | per wave: ypl [G*PSZ] | tile [64*C+16] | carry [G*N] | bc
// backward LDS: xpl | gpl | dxpl | per wave: tile | carryE | bc | acc [2*N*L]
// ---------------------------------------------------------------------------------------------
template <typename Tin, typename Tout, int C, bool N1>
__global__ void __launch_bounds__(256) ss2d_fwd_wpr_kernel(const SS2DArgs a) {
    extern __shared__ float smem[];
    const xfm_ss2d_params_t &p = a.p;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lg = a.lg_lpr, LPR = 1 << lg, G = 64 >> lg;
    const int tiles_pb = p.d_inner >> (6 - lg);
    const int groups_pb = (tiles_pb + a.pli - 1) / a.pli;
    const int b = blockIdx.x / groups_pb, tg = blockIdx.x - b * groups_pb;
    const int g = lane >> lg, i = lane & (LPR - 1);
    const int L = p.H * p.W, GP = G * a.PSZ;
    float *xpl = smem;
    float *ypl = smem + GP + (size_t)wave * a.lds_floats_per_wave;   // this route's private output planes
    float *buf = ypl + GP;
    float *carry = buf + 64 * C + 16;
    float *bc = a.bc_floats ? carry + G * p.dstate : nullptr;
    for (int it = 0; it < a.pli; ++it) {
        const int tile = tg * a.pli + it;
        if (tile >= tiles_pb) break;
        const int d0 = tile * G;
        const int64_t po = ((int64_t)b * p.d_inner + d0) * L;
        __syncthreads();
        planes_load<Tin>(xpl, (const Tin *)p.x + po, G, L, a.PW, a.PSZ, a.magicW, p.W, threadIdx.x, 256);
        __syncthreads();
        const float *xg = xpl + g * a.PSZ;
        float *yg = ypl + g * a.PSZ;
        sweep_fwd<Tin, C, N1>(a, (wave & 1) * 2 + (wave >> 1), (wave >> 1) != 0, (wave & 1) != 0, buf, carry, bc, xg, yg, true, b, d0, g, i, lane);
        __syncthreads();
        const float *Y0 = smem + GP, *Y1 = Y0 + a.lds_floats_per_wave, *Y2 = Y1 + a.lds_floats_per_wave,
                    *Y3 = Y2 + a.lds_floats_per_wave;
        Tout *yo = (Tout *)p.y + po;
        for (int g2 = 0; g2 < G; ++g2)
            for (int e = threadIdx.x; e < L; e += 256) {
                const int h = a.magicW ? (int)__umulhi((uint32_t)e, a.magicW) : e, w = e - h * p.W;
                const int idx = g2 * a.PSZ + h * a.PW + w;
                stf<Tout>(yo + (int64_t)g2 * L + e, (Y0[idx] + Y1[idx]) + (Y2[idx] + Y3[idx]));   // fixed order
            }
    }
}

template <typename Tin, typename Tout, int C, bool N1>
__global__ void __launch_bounds__(256) ss2d_bwd_wpr_kernel(const SS2DArgs a) {
    extern __shared__ float smem[];
    const xfm_ss2d_params_t &p = a.p;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lg = a.lg_lpr, LPR = 1 << lg, G = 64 >> lg;
    const int tiles_pb = p.d_inner >> (6 - lg);
    const int groups_pb = (tiles_pb + a.pli - 1) / a.pli;
    const int b = blockIdx.x / groups_pb, tg = blockIdx.x - b * groups_pb;
    const int g = lane >> lg, i = lane & (LPR - 1);
    const int N = p.dstate, L = p.H * p.W, GP = G * a.PSZ;
    float *xpl = smem, *gpl = smem + GP, *dxpl = smem + 2 * GP;
    float *buf = smem + 3 * GP + (size_t)wave * a.lds_floats_per_wave;
    float *carryE = buf + 64 * C + 16;
    float *bc = a.bc_floats ? carryE + G * N : nullptr;
    float *acc = carryE + G * N + a.bc_floats;         // [2][N][L] dB, dC of this wave's route
    for (int e = lane; e < 2 * N * L; e += 64) acc[e] = 0.f;
    const int k = (wave & 1) * 2 + (wave >> 1);        // wave 0..3 -> route 0, 2, 1, 3
    for (int it = 0; it < a.pli; ++it) {
        const int tile = tg * a.pli + it;
        if (tile >= tiles_pb) break;
        const int d0 = tile * G;
        const int64_t po = ((int64_t)b * p.d_inner + d0) * L;
        __syncthreads();
        planes_load<Tin>(xpl, (const Tin *)p.x + po, G, L, a.PW, a.PSZ, a.magicW, p.W, threadIdx.x, 256);
        planes_load<Tout>(gpl, (const Tout *)p.dy + po, G, L, a.PW, a.PSZ, a.magicW, p.W, threadIdx.x, 256);
        for (int e = threadIdx.x; e < GP; e += 256) dxpl[e] = 0.f;
        __syncthreads();
        const float *xg = xpl + g * a.PSZ, *gg = gpl + g * a.PSZ;
        float *dxg = dxpl + g * a.PSZ;
        sweep_bwd<Tin, C, N1>(a, k, (wave >> 1) != 0, (wave & 1) != 0, buf, carryE, bc, xg, gg, dxg, false, b, d0, g, i, lane, acc);
        __syncthreads();
        Tin *dxo = (Tin *)p.dx + po;
        for (int g2 = 0; g2 < G; ++g2)
            for (int e = threadIdx.x; e < L; e += 256) {
                const int h = a.magicW ? (int)__umulhi((uint32_t)e, a.magicW) : e, w = e - h * p.W;
                stf<Tin>(dxo + (int64_t)g2 * L + e, dxpl[g2 * a.PSZ + h * a.PW + w]);
            }
    }
    wave_sync();
    float *dBg = p.dBs + ((int64_t)b * 4 + k) * N * L;
    float *dCg = p.dCs + ((int64_t)b * 4 + k) * N * L;
    for (int e = lane; e < N * L; e += 64) {
        atomicAdd(dBg + e, acc[e]);
        atomicAdd(dCg + e, acc[N * L + e]);
    }
}

}  // namespace xfm

#include "ss2d_direct.hpp"
#include "ss2d_lean.hpp"

namespace xfm {

// launch plan shared by the host code and the per-dtype translation units
struct Plan2 {
    int lg, items, n_chunks;
    int kind, pli, bc_floats;                    // kind 2 = direct (ss2d_direct.hpp), kind 3 = lean d_state==1 (ss2d_lean.hpp)
    int ppt;                                     // kind 3: planes per tile
    int pli_fwd;                                 // kind 3: tiles per workgroup in the forward (no accumulators to amortise)
    int reg_nseg;                                // kind 3: chunk count of the register-accumulator variant (0: LDS)
    int psz;                                     // plane size: floats incl. pitch (kind 0/1) or elements (kind 2)
    size_t lds_fwd_floats, lds_bwd_floats;       // per wave (kind 0) / per wave beyond the shared planes (kind 1)
    size_t lds_fwd_block, lds_bwd_block;         // dynamic LDS bytes per workgroup
    int waves_fwd, waves_bwd;                    // waves per workgroup
};

template <typename Tin, typename Tout>
int ss2d_dispatch(const SS2DArgs &a, const Plan2 &pl, bool bwd, hipStream_t s);

int ss2d_launch_raw(const void *fn, const SS2DArgs &a, const Plan2 &pl, bool bwd, hipStream_t s);
int ss2d_launch_lean(const void *fn, const SS2DArgs &a, const Plan2 &pl, bool bwd, hipStream_t s);

template <typename Tin, typename Tout, int C, bool N1>
static int ss2d_launch(const SS2DArgs &a, const Plan2 &pl, bool bwd, hipStream_t s) {
    const void *fn;
    if (pl.kind == 1)
        fn = bwd ? (const void *)ss2d_bwd_wpr_kernel<Tin, Tout, C, N1> : (const void *)ss2d_fwd_wpr_kernel<Tin, Tout, C, N1>;
    else
        fn = bwd ? (const void *)ss2d_bwd_kernel<Tin, Tout, C, N1> : (const void *)ss2d_fwd_kernel<Tin, Tout, C, N1>;
    return ss2d_launch_raw(fn, a, pl, bwd, s);
}

template <typename Tin, typename Tout>
static int ss2d_dispatch_impl(const SS2DArgs &a, const Plan2 &pl, bool bwd, hipStream_t s) {
    const bool n1 = a.p.dstate == 1;
    if (pl.kind == 3) {
        const void *fn;
        if (!bwd) {
            if constexpr (sizeof(Tin) == 2) {
                if (pl.reg_nseg == 18) return ss2d_launch_lean((const void *)ss2d_fwd_lean_kernel<Tin, Tout, 8, true>, a, pl, bwd, s);
            }
            fn = pl.items == 8 ? (const void *)ss2d_fwd_lean_kernel<Tin, Tout, 8> : (const void *)ss2d_fwd_lean_kernel<Tin, Tout, 4>;
        } else if (pl.items == 8) {
            if constexpr (sizeof(Tin) == 2) {
                if (pl.reg_nseg == 18) return ss2d_launch_lean((const void *)ss2d_bwd_lean_kernel<Tin, Tout, 8, 18>, a, pl, bwd, s);
            }
            switch (pl.reg_nseg) {                      // dB/dC sums in registers when the chunk count is a built variant
                case 1: fn = (const void *)ss2d_bwd_lean_kernel<Tin, Tout, 8, 1>; break;
                case 2: fn = (const void *)ss2d_bwd_lean_kernel<Tin, Tout, 8, 2>; break;
                default: fn = (const void *)ss2d_bwd_lean_kernel<Tin, Tout, 8, 0>;
            }
        } else {
            switch (pl.reg_nseg) {
                case 1: fn = (const void *)ss2d_bwd_lean_kernel<Tin, Tout, 4, 1>; break;
                case 2: fn = (const void *)ss2d_bwd_lean_kernel<Tin, Tout, 4, 2>; break;
                default: fn = (const void *)ss2d_bwd_lean_kernel<Tin, Tout, 4, 0>;
            }
        }
        return ss2d_launch_lean(fn, a, pl, bwd, s);
    }
    if (pl.kind == 2) {
        const void *fn;
        if (n1) fn = bwd ? (const void *)ss2d_bwd_direct_kernel<Tin, Tout, true> : (const void *)ss2d_fwd_direct_kernel<Tin, Tout, true>;
        else fn = bwd ? (const void *)ss2d_bwd_direct_kernel<Tin, Tout, false> : (const void *)ss2d_fwd_direct_kernel<Tin, Tout, false>;
        return ss2d_launch_raw(fn, a, pl, bwd, s);
    }
    switch (pl.items) {
        case 4: return n1 ? ss2d_launch<Tin, Tout, 4, true>(a, pl, bwd, s) : ss2d_launch<Tin, Tout, 4, false>(a, pl, bwd, s);
        case 7: return n1 ? ss2d_launch<Tin, Tout, 7, true>(a, pl, bwd, s) : ss2d_launch<Tin, Tout, 7, false>(a, pl, bwd, s);
        case 9: return n1 ? ss2d_launch<Tin, Tout, 9, true>(a, pl, bwd, s) : ss2d_launch<Tin, Tout, 9, false>(a, pl, bwd, s);
        case 13: return n1 ? ss2d_launch<Tin, Tout, 13, true>(a, pl, bwd, s) : ss2d_launch<Tin, Tout, 13, false>(a, pl, bwd, s);
    }
    return XFM_EINVAL;
}

}  // namespace xfm
