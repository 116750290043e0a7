// ss2d_direct.hpp -- "direct" variant of the fused SS2D sweeps: the chunk a lane owns is exactly one
// 16-byte vector (C = 16 / sizeof(T): 8 bf16/fp16 or 4 fp32 elements).
//
// With that choice every per-route operand (dts, B, C, ddts) is moved by ONE global_load/store_dwordx4
// per lane straight between HBM and registers (coalesced 1 KiB per wave instruction, prefetched one
// chunk ahead), and the feature-map planes are kept in LDS twice -- row-major and transposed -- so the
// row routes (0/2) and the column routes (1/3) both read/write their chunk as one contiguous, aligned
// 16/32-byte LDS access.  No LDS bounce, no per-element index arithmetic, no bank conflicts in the
// sweeps; the only strided LDS traffic is building the transposed copy and the final merge (once per
// plane each).  Requires rows whose byte length is a multiple of 4 (all XFMamba trunk stages >= 14x14).
// Included by ss2d_kernels.hpp.
#pragma once

namespace xfm {

struct __attribute__((packed, aligned(4))) u4a4 { uint32_t x, y, z, w; };   // 16 bytes, dword aligned

// 16-byte vector from HBM at a dword-aligned address; `room` = elements readable from p inside the tensor
template <typename T>
__device__ __forceinline__ uint4 load16_guard(const T *p, int64_t room) {
    constexpr int VE = 16 / (int)sizeof(T);
    if (room >= VE) {
        const u4a4 v = *reinterpret_cast<const u4a4 *>(p);
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    float f[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) f[e] = e < room ? ldf<T>(p + e) : 0.f;
    return pack16<T>(f);
}

template <typename T>
__device__ __forceinline__ void store16_guard(T *p, const float *f, int nvalid) {
    constexpr int VE = 16 / (int)sizeof(T);
    if (nvalid >= VE) {
        const uint4 v = pack16<T>(f);
        u4a4 o;
        o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
        *reinterpret_cast<u4a4 *>(p) = o;
    } else {
        for (int e = 0; e < nvalid; ++e) stf<T>(p + e, f[e]);
    }
}

template <typename T> __device__ __forceinline__ T from_float(float v);
template <> __device__ __forceinline__ float from_float<float>(float v) { return v; }
template <> __device__ __forceinline__ f16_t from_float<f16_t>(float v) { return __float2half(v); }
template <> __device__ __forceinline__ bf16_t from_float<bf16_t>(float v) { return __float2bfloat16(v); }

// HBM planes (type S, natural order) -> LDS planes of type T in BOTH layouts: nat[g][h*W+w], tr[g][w*H+h]
template <typename S, typename T>
__device__ __forceinline__ void planes_load_nt(T *nat, T *tr, const S *src, int G, int L, int PSZ, int H, int W,
                                               uint32_t magicW, int64_t room, int tid, int nthreads) {
    constexpr int VS = 16 / (int)sizeof(S);
    const int nvec = (L + VS - 1) / VS;
    for (int g = 0; g < G; ++g) {
        const S *pg = src + (int64_t)g * L;
        for (int v = tid; v < nvec; v += nthreads) {
            const int e0 = v * VS;
            float f[VS];
            unpack16<S>(load16_guard<S>(pg + e0, room - (int64_t)g * L - e0), f);
            int h = magicW ? (int)__umulhi((uint32_t)e0, magicW) : e0, w = e0 - h * W;
#pragma unroll
            for (int q = 0; q < VS; ++q) {
                if (e0 + q < L) {
                    const T val = from_float<T>(f[q]);
                    nat[g * PSZ + e0 + q] = val;
                    tr[g * PSZ + w * H + h] = val;
                }
                if (++w == W) {
                    w = 0;
                    ++h;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// one route, forward.  xq / yq: this route's layout (natural for routes 0/2, transposed for 1/3).
// ---------------------------------------------------------------------------------------------
// Register ring of prefetched operand vectors: slot j holds flattened work item F+j, where a work item
// is (tile, chunk) in the order this wave consumes them; the ring runs continuously across the tiles of
// the workgroup so ~PD KiB per operand stay in flight per wave regardless of how short a row is.
constexpr int kPD = 3;
struct Pref { uint4 d[kPD], b[kPD], c[kPD]; };

template <typename Tin, bool N1>
__device__ __forceinline__ void pref_issue(const SS2DArgs &a, const int k, const bool REV, const int b, const int g,
                                           const int i, const int tile0, const int n_tiles, const int item,
                                           uint4 &rd, uint4 &rb, uint4 &rc) {
    constexpr int C = 16 / (int)sizeof(Tin);
    const xfm_ss2d_params_t &p = a.p;
    const int lg = a.lg_lpr, LPR = 1 << lg, G = 64 >> lg;
    const int N = p.dstate, L = p.H * p.W, D = p.d_inner, SL = C << lg, nseg = a.n_chunks;
    const int tq = item / nseg, s = item - tq * nseg;      // tile offset inside the workgroup, chunk (traversal order)
    if (tq >= n_tiles) return;
    const int ci = REV ? LPR - 1 - i : i;
    const int tp0 = (REV ? nseg - 1 - s : s) * SL + ci * C;
    if (tp0 >= L) return;
    const int d = (tile0 + tq) * G + g;
    const int64_t dts_off = (((int64_t)b * 4 + k) * D + d) * L + tp0, bc_off = ((int64_t)b * 4 + k) * N * L + tp0;
    rd = load16_guard<Tin>((const Tin *)p.dts + dts_off, (int64_t)p.batch * 4 * D * L - dts_off);
    if (N1) {
        rb = load16_guard<Tin>((const Tin *)p.Bs + bc_off, (int64_t)p.batch * 4 * N * L - bc_off);
        rc = load16_guard<Tin>((const Tin *)p.Cs + bc_off, (int64_t)p.batch * 4 * N * L - bc_off);
    }
}

template <typename Tin, bool N1>
__device__ __forceinline__ void sweep_fwd_d(const SS2DArgs &a, const int k, const bool REV, float *carry, float *bc,
                                            const Tin *xq, float *yq, const int b, const int d0, const int g,
                                            const int i, const int lane, Pref &pf, const int tile0, const int n_tiles,
                                            const int it) {
    constexpr int C = 16 / (int)sizeof(Tin);
    const xfm_ss2d_params_t &p = a.p;
    const int lg = a.lg_lpr, LPR = 1 << lg;
    const int N = p.dstate, L = p.H * p.W, D = p.d_inner, SL = C << lg, nseg = a.n_chunks;
    const int d = d0 + g, row = k * D + d;
    const int64_t dts_off = (((int64_t)b * 4 + k) * D + d) * L, bc_off = ((int64_t)b * 4 + k) * N * L;
    const Tin *dts_r = (const Tin *)p.dts + dts_off;
    const Tin *Bg = (const Tin *)p.Bs + bc_off, *Cg = (const Tin *)p.Cs + bc_off;
    const int64_t dts_room = (int64_t)p.batch * 4 * D * L - dts_off, bc_room = (int64_t)p.batch * 4 * N * L - bc_off;
    const float *Ar = p.A + (int64_t)row * N;
    const float Dr = p.D[row], bias = p.delta_softplus == 2 ? 0.f : p.delta_bias[row];   // mode 2: dts = softplus(raw + bias) already
    const int ci = REV ? LPR - 1 - i : i;
    for (int n = i; n < N; n += LPR) carry[g * N + n] = 0.f;
    if (!N1 && bc) {
        for (int e = lane; e < N * L; e += 64) {
            bc[e] = ldf<Tin>(Bg + e);
            bc[N * L + e] = ldf<Tin>(Cg + e);
        }
    }
    auto seg0 = [&](int s) { return (REV ? nseg - 1 - s : s) * SL; };
    wave_sync();
    for (int s = 0; s < nseg; ++s) {
        const int tp0 = seg0(s) + ci * C;
        const bool live = tp0 < L;
        float dl[C], u[C], y[C], Bv[C], Cv[C], tmp[C];
        bool ok[C];
        const uint4 rd = pf.d[0], rb = pf.b[0], rc = pf.c[0];
#pragma unroll
        for (int j = 0; j + 1 < kPD; ++j) {
            pf.d[j] = pf.d[j + 1];
            pf.b[j] = pf.b[j + 1];
            pf.c[j] = pf.c[j + 1];
        }
        pref_issue<Tin, N1>(a, k, REV, b, g, i, tile0, n_tiles, it * nseg + s + kPD, pf.d[kPD - 1], pf.b[kPD - 1], pf.c[kPD - 1]);
        unpack16<Tin>(rd, tmp);
#pragma unroll
        for (int j = 0; j < C; ++j) dl[j] = REV ? tmp[C - 1 - j] : tmp[j];
        if (N1) {
            unpack16<Tin>(rb, tmp);
#pragma unroll
            for (int j = 0; j < C; ++j) Bv[j] = REV ? tmp[C - 1 - j] : tmp[j];
            unpack16<Tin>(rc, tmp);
#pragma unroll
            for (int j = 0; j < C; ++j) Cv[j] = REV ? tmp[C - 1 - j] : tmp[j];
        }
        uint4 xv = make_uint4(0, 0, 0, 0);
        if (live) xv = *reinterpret_cast<const uint4 *>(xq + tp0);
        unpack16<Tin>(xv, tmp);
#pragma unroll
        for (int j = 0; j < C; ++j) u[j] = REV ? tmp[C - 1 - j] : tmp[j];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            ok[j] = tp0 + (REV ? C - 1 - j : j) < L;
            float v = dl[j] + bias;
            if (p.delta_softplus == 1) v = softplus20(v);
            dl[j] = ok[j] ? v : 0.f;
            u[j] = ok[j] ? u[j] : 0.f;
            y[j] = 0.f;
        }
        for (int n = 0; n < N; ++n) {
            const float A2 = Ar[n] * kLog2e;
            float av[C], bb[C];
            if (!N1) {
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    const int t = tp0 + (REV ? C - 1 - j : j);
                    if (bc) {
                        Bv[j] = ok[j] ? bc[n * L + t] : 0.f;
                        Cv[j] = ok[j] ? bc[(N + n) * L + t] : 0.f;
                    } else {
                        Bv[j] = ok[j] ? ldf<Tin>(Bg + (int64_t)n * L + t) : 0.f;
                        Cv[j] = ok[j] ? ldf<Tin>(Cg + (int64_t)n * L + t) : 0.f;
                    }
                }
            }
            float P = 1.f, S = 0.f;
#pragma unroll
            for (int j = 0; j < C; ++j) {
                av[j] = exp2_fast(dl[j] * A2);
                bb[j] = ok[j] ? dl[j] * u[j] * Bv[j] : 0.f;
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
                y[j] = fmaf(ok[j] ? Cv[j] : 0.f, h, y[j]);
            }
            if (i == LPR - 1) {
                carry[g * N + n] = h;
                if (nseg > 1) p.chk[((((int64_t)b * 4 + k) * D + d) * nseg + s) * N + n] = h;
            }
        }
        if (live) {
#pragma unroll
            for (int j = 0; j < C; ++j) y[j] = fmaf(Dr, u[j], y[j]);
#pragma unroll
            for (int q = 0; q < C; q += 4)
                *reinterpret_cast<float4 *>(yq + tp0 + q) =
                    REV ? make_float4(y[C - 1 - q], y[C - 2 - q], y[C - 3 - q], y[C - 4 - q])
                        : make_float4(y[q], y[q + 1], y[q + 2], y[q + 3]);
        }
    }
    wave_sync();
}

// ---------------------------------------------------------------------------------------------
// one route, backward
// ---------------------------------------------------------------------------------------------
template <typename Tin, bool N1>
__device__ __forceinline__ void pref_issue_bwd(const SS2DArgs &a, const int k, const bool REV, const int b, const int g,
                                               const int i, const int tile0, const int n_tiles, const int item,
                                               uint4 &rd, uint4 &rb, uint4 &rc) {
    const int nseg = a.n_chunks;
    const int tq = item / nseg, m = item - tq * nseg;
    pref_issue<Tin, N1>(a, k, REV, b, g, i, tile0, n_tiles, tq * nseg + (nseg - 1 - m), rd, rb, rc);
}

template <typename Tin, bool N1>
__device__ __forceinline__ void sweep_bwd_d(const SS2DArgs &a, const int k, const bool REV, float *carryE, float *bc,
                                            const Tin *xq, const Tin *gq, float *dxq, const int b, const int d0,
                                            const int g, const int i, const int lane, float *acc, Pref &pf,
                                            const int tile0, const int n_tiles, const int it) {
    constexpr int C = 16 / (int)sizeof(Tin);
    const xfm_ss2d_params_t &p = a.p;
    const int lg = a.lg_lpr, LPR = 1 << lg;
    const int N = p.dstate, L = p.H * p.W, D = p.d_inner, SL = C << lg, nseg = a.n_chunks;
    const int d = d0 + g, row = k * D + d;
    const int64_t dts_off = (((int64_t)b * 4 + k) * D + d) * L, bc_off = ((int64_t)b * 4 + k) * N * L;
    const Tin *dts_r = (const Tin *)p.dts + dts_off;
    Tin *ddts_r = (Tin *)p.ddts + dts_off;
    const Tin *Bg = (const Tin *)p.Bs + bc_off, *Cg = (const Tin *)p.Cs + bc_off;
    const int64_t dts_room = (int64_t)p.batch * 4 * D * L - dts_off, bc_room = (int64_t)p.batch * 4 * N * L - bc_off;
    const float *Ar = p.A + (int64_t)row * N;
    const float Dr = p.D[row], bias = p.delta_softplus == 2 ? 0.f : p.delta_bias[row];   // mode 2: dts = softplus(raw + bias) already
    const int ci = REV ? LPR - 1 - i : i;
    for (int n = i; n < N; n += LPR) carryE[g * N + n] = 0.f;
    if (!N1 && bc) {
        for (int e = lane; e < N * L; e += 64) {
            bc[e] = ldf<Tin>(Bg + e);
            bc[N * L + e] = ldf<Tin>(Cg + e);
        }
    }
    auto seg0 = [&](int s) { return (REV ? nseg - 1 - s : s) * SL; };
    wave_sync();
    float dD_acc = 0.f, dbias_acc = 0.f;
    for (int s = nseg - 1; s >= 0; --s) {
        const int tp0 = seg0(s) + ci * C;
        const bool live = tp0 < L;
        float dl[C], u[C], go[C], s1[C], s2[C], Bv[C], Cv[C], tmp[C];
        bool ok[C];
        const uint4 rd = pf.d[0], rb = pf.b[0], rc = pf.c[0];
#pragma unroll
        for (int j = 0; j + 1 < kPD; ++j) {
            pf.d[j] = pf.d[j + 1];
            pf.b[j] = pf.b[j + 1];
            pf.c[j] = pf.c[j + 1];
        }
        // the backward consumes the chunks of a tile against the route: work item m of tile `it` is chunk nseg-1-m
        pref_issue_bwd<Tin, N1>(a, k, REV, b, g, i, tile0, n_tiles, it * nseg + (nseg - 1 - s) + kPD, pf.d[kPD - 1], pf.b[kPD - 1],
                                pf.c[kPD - 1]);
        unpack16<Tin>(rd, tmp);
#pragma unroll
        for (int j = 0; j < C; ++j) dl[j] = REV ? tmp[C - 1 - j] : tmp[j];
        if (N1) {
            unpack16<Tin>(rb, tmp);
#pragma unroll
            for (int j = 0; j < C; ++j) Bv[j] = REV ? tmp[C - 1 - j] : tmp[j];
            unpack16<Tin>(rc, tmp);
#pragma unroll
            for (int j = 0; j < C; ++j) Cv[j] = REV ? tmp[C - 1 - j] : tmp[j];
        }
        uint4 xv = make_uint4(0, 0, 0, 0), gv = xv;
        if (live) {
            xv = *reinterpret_cast<const uint4 *>(xq + tp0);
            gv = *reinterpret_cast<const uint4 *>(gq + tp0);
        }
        unpack16<Tin>(xv, tmp);
#pragma unroll
        for (int j = 0; j < C; ++j) u[j] = REV ? tmp[C - 1 - j] : tmp[j];
        unpack16<Tin>(gv, tmp);
#pragma unroll
        for (int j = 0; j < C; ++j) go[j] = REV ? tmp[C - 1 - j] : tmp[j];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            ok[j] = tp0 + (REV ? C - 1 - j : j) < L;
            float v = dl[j] + bias;
            if (p.delta_softplus == 1) v = softplus20(v);
            dl[j] = ok[j] ? v : 0.f;
            u[j] = ok[j] ? u[j] : 0.f;
            go[j] = ok[j] ? go[j] : 0.f;
            s1[j] = 0.f;
            s2[j] = 0.f;
        }
        for (int n = 0; n < N; ++n) {
            const float An = Ar[n];
            const float A2 = An * kLog2e;
            float cg[C], av[C], h[C], bb[C];
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const int t = tp0 + (REV ? C - 1 - j : j);
                if (!N1) {
                    if (bc) {
                        Bv[j] = ok[j] ? bc[n * L + t] : 0.f;
                        Cv[j] = ok[j] ? bc[(N + n) * L + t] : 0.f;
                    } else {
                        Bv[j] = ok[j] ? ldf<Tin>(Bg + (int64_t)n * L + t) : 0.f;
                        Cv[j] = ok[j] ? ldf<Tin>(Cg + (int64_t)n * L + t) : 0.f;
                    }
                } else if (!ok[j]) {
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
            float dBv[C], dCv[C];
#pragma unroll
            for (int j = C - 1; j >= 0; --j) {
                const float dh = cg[j] + E;
                E = av[j] * dh;
                const float ah = h[j] - bb[j];
                s1[j] = fmaf(dh, Bv[j], s1[j]);
                s2[j] = fmaf(dh * An, ah, s2[j]);
                dA_acc = fmaf(dh * dl[j], ah, dA_acc);
                dBv[j] = dh * dl[j] * u[j];
                dCv[j] = go[j] * h[j];
            }
            // dB/dC: sum over the G planes of the tile, then into this route's accumulator (or HBM atomics)
            for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    dBv[j] += __shfl_xor(dBv[j], o, 64);
                    dCv[j] += __shfl_xor(dCv[j], o, 64);
                }
            }
            if (g == 0 && live) {
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    const int t = tp0 + (REV ? C - 1 - j : j);
                    if (t < L) {
                        if (acc) {
                            acc[n * L + t] += dBv[j];
                            acc[(N + n) * L + t] += dCv[j];
                        } else {
                            atomicAdd(p.dBs + bc_off + (int64_t)n * L + t, dBv[j]);
                            atomicAdd(p.dCs + bc_off + (int64_t)n * L + t, dCv[j]);
                        }
                    }
                }
            }
            if (i == 0) carryE[g * N + n] = E;
            for (int o = 1; o < LPR; o <<= 1) dA_acc += __shfl_xor(dA_acc, o, 64);
            if (i == 0) atomicAdd(p.dA + (int64_t)row * N + n, dA_acc);
        }
        float dd[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const float du = fmaf(dl[j], s1[j], Dr * go[j]);
            float ddl = fmaf(u[j], s1[j], s2[j]);
            if (p.delta_softplus && dl[j] <= 20.f) ddl *= 1.f - __expf(-dl[j]);
            dD_acc = fmaf(go[j], u[j], dD_acc);
            dbias_acc += ok[j] ? ddl : 0.f;
            const int q = REV ? C - 1 - j : j;
            dd[q] = ddl;                                            // physical order for the vector store
            if (ok[j]) atomicAdd(dxq + tp0 + q, du);                // plane shared by the two waves of this layout
        }
        if (live) store16_guard<Tin>(ddts_r + tp0, dd, min(C, L - tp0));
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
// kernels (wave per route).  a.PSZ = plane size in ELEMENTS (L rounded up to the vector length).
// forward  LDS: xN | xT (Tin)                         per wave: y plane (float) | carry | bc
// backward LDS: xN | xT | gN | gT (Tin) | dxN | dxT   per wave: carryE | bc | acc [2*N*L]
// ---------------------------------------------------------------------------------------------
template <typename Tin, typename Tout, bool N1>
__global__ void __launch_bounds__(256) ss2d_fwd_direct_kernel(const SS2DArgs a) {
    extern __shared__ float smem[];
    const xfm_ss2d_params_t &p = a.p;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lg = a.lg_lpr, LPR = 1 << lg, G = 64 >> lg;
    const int tiles_pb = p.d_inner >> (6 - lg);
    const int groups_pb = (tiles_pb + a.pli - 1) / a.pli;
    const int b = blockIdx.x / groups_pb, tg = blockIdx.x - b * groups_pb;
    const int g = lane >> lg, i = lane & (LPR - 1);
    const int H = p.H, W = p.W, L = H * W, GP = G * a.PSZ;
    Tin *xN = reinterpret_cast<Tin *>(smem), *xT = xN + GP;
    float *wbase = smem + (2 * (size_t)GP * sizeof(Tin) + 15) / 16 * 4 + (size_t)wave * a.lds_floats_per_wave;
    float *ypl = wbase;
    float *carry = ypl + GP;
    float *bc = a.bc_floats ? carry + ((G * p.dstate + 3) & ~3) : nullptr;
    const bool col = wave >> 1, rev = wave & 1;
    const int k = (wave & 1) * 2 + (wave >> 1);
    const int tile0 = tg * a.pli, n_tiles = min(a.pli, tiles_pb - tile0);
    Pref pf;
#pragma unroll
    for (int j = 0; j < kPD; ++j) {
        pf.d[j] = pf.b[j] = pf.c[j] = make_uint4(0, 0, 0, 0);
        pref_issue<Tin, N1>(a, k, rev, b, g, i, tile0, n_tiles, j, pf.d[j], pf.b[j], pf.c[j]);
    }
    for (int it = 0; it < a.pli; ++it) {
        const int tile = tg * a.pli + it;
        if (tile >= tiles_pb) break;
        const int d0 = tile * G;
        const int64_t po = ((int64_t)b * p.d_inner + d0) * L;
        __syncthreads();
        if (!(a.dbg & 2))
            planes_load_nt<Tin, Tin>(xN, xT, (const Tin *)p.x + po, G, L, a.PSZ, H, W, a.magicW,
                                     (int64_t)p.batch * p.d_inner * L - po, threadIdx.x, 256);
        __syncthreads();
        if (!(a.dbg & 1))
            sweep_fwd_d<Tin, N1>(a, k, rev, carry, bc, (col ? xT : xN) + g * a.PSZ, ypl + g * a.PSZ, b, d0, g, i, lane, pf, tile0, n_tiles, it);
        __syncthreads();
        if (a.dbg & 4) continue;
        const size_t ws = a.lds_floats_per_wave;
        const float *Y0 = wbase - (size_t)wave * ws, *Y1 = Y0 + ws, *Y2 = Y1 + ws, *Y3 = Y2 + ws;
        Tout *yo = (Tout *)p.y + po;
        for (int g2 = 0; g2 < G; ++g2)
            for (int e = threadIdx.x; e < L; e += 256) {
                const int h = a.magicW ? (int)__umulhi((uint32_t)e, a.magicW) : e, w = e - h * W;
                const int n_ = g2 * a.PSZ + e, t_ = g2 * a.PSZ + w * H + h;
                stf<Tout>(yo + (int64_t)g2 * L + e, (Y0[n_] + Y1[n_]) + (Y2[t_] + Y3[t_]));   // fixed order
            }
    }
}

template <typename Tin, typename Tout, bool N1>
__global__ void __launch_bounds__(256) ss2d_bwd_direct_kernel(const SS2DArgs a) {
    extern __shared__ float smem[];
    const xfm_ss2d_params_t &p = a.p;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lg = a.lg_lpr, LPR = 1 << lg, G = 64 >> lg;
    const int tiles_pb = p.d_inner >> (6 - lg);
    const int groups_pb = (tiles_pb + a.pli - 1) / a.pli;
    const int b = blockIdx.x / groups_pb, tg = blockIdx.x - b * groups_pb;
    const int g = lane >> lg, i = lane & (LPR - 1);
    const int N = p.dstate, H = p.H, W = p.W, L = H * W, GP = G * a.PSZ;
    Tin *xN = reinterpret_cast<Tin *>(smem), *xT = xN + GP, *gN = xT + GP, *gT = gN + GP;
    float *dxN = smem + (4 * (size_t)GP * sizeof(Tin) + 15) / 16 * 4, *dxT = dxN + GP;
    float *carryE = dxT + GP + (size_t)wave * a.lds_floats_per_wave;
    float *bc = a.bc_floats ? carryE + ((G * N + 3) & ~3) : nullptr;
    float *acc = carryE + ((G * N + 3) & ~3) + a.bc_floats;
    for (int e = lane; e < 2 * N * L; e += 64) acc[e] = 0.f;
    const bool col = wave >> 1, rev = wave & 1;
    const int k = (wave & 1) * 2 + (wave >> 1);
    const int tile0 = tg * a.pli, n_tiles = min(a.pli, tiles_pb - tile0);
    Pref pf;
#pragma unroll
    for (int j = 0; j < kPD; ++j) {
        pf.d[j] = pf.b[j] = pf.c[j] = make_uint4(0, 0, 0, 0);
        pref_issue_bwd<Tin, N1>(a, k, rev, b, g, i, tile0, n_tiles, j, pf.d[j], pf.b[j], pf.c[j]);
    }
    for (int it = 0; it < a.pli; ++it) {
        const int tile = tg * a.pli + it;
        if (tile >= tiles_pb) break;
        const int d0 = tile * G;
        const int64_t po = ((int64_t)b * p.d_inner + d0) * L;
        const int64_t room = (int64_t)p.batch * p.d_inner * L - po;
        __syncthreads();
        if (!(a.dbg & 2)) {
            planes_load_nt<Tin, Tin>(xN, xT, (const Tin *)p.x + po, G, L, a.PSZ, H, W, a.magicW, room, threadIdx.x, 256);
            planes_load_nt<Tout, Tin>(gN, gT, (const Tout *)p.dy + po, G, L, a.PSZ, H, W, a.magicW, room, threadIdx.x, 256);
        }
        for (int e = threadIdx.x; e < 2 * GP; e += 256) dxN[e] = 0.f;
        __syncthreads();
        if (!(a.dbg & 1))
            sweep_bwd_d<Tin, N1>(a, k, rev, carryE, bc, (col ? xT : xN) + g * a.PSZ, (col ? gT : gN) + g * a.PSZ,
                                 (col ? dxT : dxN) + g * a.PSZ, b, d0, g, i, lane, acc, pf, tile0, n_tiles, it);
        __syncthreads();
        if (a.dbg & 4) continue;
        Tin *dxo = (Tin *)p.dx + po;
        for (int g2 = 0; g2 < G; ++g2)
            for (int e = threadIdx.x; e < L; e += 256) {
                const int h = a.magicW ? (int)__umulhi((uint32_t)e, a.magicW) : e, w = e - h * W;
                stf<Tin>(dxo + (int64_t)g2 * L + e, dxN[g2 * a.PSZ + e] + dxT[g2 * a.PSZ + w * H + h]);
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
