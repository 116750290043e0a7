// ss2d_lean.hpp -- instruction-lean fused SS2D kernels for the backbone case d_state == 1
// (92 % of the scan elements of XFMamba).
//
// Same algorithm and layout contract as ss2d_direct.hpp (chunk = one aligned vector per lane, planes in
// LDS in row-major AND transposed order, wave-per-route workgroups, private dB/dC accumulators), but
// written against the instruction count the first profiles exposed (profiles/r01_*): 85 VALU
// lane-instructions per (element, route) of which < 20 % were maths.  Changes:
//   * the route direction is a template parameter, so reversal is register renaming, not selects;
//   * rows are a whole number of vectors (L % C == 0): no per-element validity masks, no tail paths;
//   * the wave scan runs on DPP (row_shr / row_bcast) instead of ds_bpermute shuffles, the chunk carry
//     lives in a register (lane broadcast) instead of LDS, stores are predicated not branched;
//   * a tile is PPT planes (not 64/LPR): small maps amortise the two workgroup barriers per tile;
//   * 32-bit offsets from per-route base pointers computed once.
#pragma once

namespace xfm {

struct LeanArgs {
    // tensors (layout contract of include/xfm_hip.h)
    const void *x, *dts, *Bs, *Cs;
    const float *A, *D, *bias;
    void *y;
    float *chk;
    const void *dy;
    void *dx, *ddts;
    float *dBs, *dCs, *dA, *dD, *dbias;
    int batch, D_, H, W, L;
    int nseg;        // chunks per row = ceil(L / (64*C))
    int ppt;         // planes per tile (staged together in LDS)
    int pli;         // tiles per workgroup
    int softplus;
    int dbg;         // timing-only switches (XFM_SS2D_DBG): 1 skip sweeps, 2 skip plane loads, 4 skip merge/store, 8 skip dB/dC flush
    uint32_t magicW;
    uint32_t magicL;   // ceil(2^32 / L): plane index of a tile element by multiply-high (tile elements < 2^16)
};

// ---- DPP helpers -------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, ROW_MASK, 0xf, false));
}
template <int LANE> __device__ __forceinline__ float bcast_lane(float v) {   // v_readlane_b32: no LDS round trip
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), LANE));
}
constexpr int kRowShr1 = 0x111, kRowShr2 = 0x112, kRowShr4 = 0x114, kRowShr8 = 0x118;
constexpr int kRowShl1 = 0x101, kRowShl2 = 0x102, kRowShl4 = 0x104, kRowShl8 = 0x108;
constexpr int kRowBcast15 = 0x142, kRowBcast31 = 0x143;
constexpr int kWaveShr1 = 0x138, kWaveShl1 = 0x130;

// Inclusive scan of affine maps h -> P*h + S over the 64 lanes, ascending.  Lanes without a source keep
// their value (DPP `old` operand = identity map contribution).
__device__ __forceinline__ void wave_scan_up(float &P, float &S) {
#define XFM_STEP_UP(CTRL, MASK)                                   \
    {                                                             \
        const float Pp = dpp_mov<CTRL, MASK>(1.f, P);             \
        const float Sp = dpp_mov<CTRL, MASK>(0.f, S);             \
        S = fmaf(P, Sp, S);                                       \
        P *= Pp;                                                  \
    }
    XFM_STEP_UP(kRowShr1, 0xf)
    XFM_STEP_UP(kRowShr2, 0xf)
    XFM_STEP_UP(kRowShr4, 0xf)
    XFM_STEP_UP(kRowShr8, 0xf)
    XFM_STEP_UP(kRowBcast15, 0xa)
    XFM_STEP_UP(kRowBcast31, 0xc)
#undef XFM_STEP_UP
}

// Same, descending (lane i composes lanes 63..i).  No row_bcast in this direction: the two cross-row
// steps use readlane broadcasts of the row totals.
__device__ __forceinline__ void wave_scan_down(float &P, float &S, int lane) {
#define XFM_STEP_DN(CTRL)                                         \
    {                                                             \
        const float Pn = dpp_mov<CTRL>(1.f, P);                   \
        const float Sn = dpp_mov<CTRL>(0.f, S);                   \
        S = fmaf(P, Sn, S);                                       \
        P *= Pn;                                                  \
    }
    XFM_STEP_DN(kRowShl1)
    XFM_STEP_DN(kRowShl2)
    XFM_STEP_DN(kRowShl4)
    XFM_STEP_DN(kRowShl8)
#undef XFM_STEP_DN
    // row r (lanes 16r..16r+15) now holds suffixes within the row; fold in the rows above it
    const float P1 = bcast_lane<16>(P), S1 = bcast_lane<16>(S), P2 = bcast_lane<32>(P), S2 = bcast_lane<32>(S),
                P3 = bcast_lane<48>(P), S3 = bcast_lane<48>(S);
    // totals of rows 1..3 = values at their first lanes (suffix over the whole row)
    const float T3S = S3, T3P = P3;
    const float T2S = fmaf(P2, T3S, S2), T2P = P2 * T3P;          // rows 2..3
    const float T1S = fmaf(P1, T2S, S1), T1P = P1 * T2P;          // rows 1..3
    const int r = lane >> 4;
    const float Pa = r == 0 ? T1P : (r == 1 ? T2P : (r == 2 ? T3P : 1.f));
    const float Sa = r == 0 ? T1S : (r == 1 ? T2S : (r == 2 ? T3S : 0.f));
    S = fmaf(P, Sa, S);
    P *= Pa;
}

template <typename T, int C> struct VecIO;
template <typename T> struct VecIO<T, 8> {   // 8 x 16-bit = 16 bytes
    using V = uint4;
    static __device__ __forceinline__ V zero() { return make_uint4(0, 0, 0, 0); }
    static __device__ __forceinline__ void unpack(const V &v, float *f) { unpack16<T>(v, f); }
    static __device__ __forceinline__ V pack(const float *f) { return pack16<T>(f); }
};
template <typename T> struct VecIO<T, 4> {   // 4 x 16-bit = 8 bytes, or 4 x fp32 = 16 bytes
    using V = typename std::conditional<sizeof(T) == 4, uint4, uint2>::type;
    static __device__ __forceinline__ V zero() { return V{}; }
    static __device__ __forceinline__ void unpack(const V &v, float *f) {
        if constexpr (sizeof(T) == 4) {
            unpack16<T>(v, f);
        } else {
            float t[8];
            unpack16<T>(make_uint4(v.x, v.y, 0, 0), t);
            f[0] = t[0]; f[1] = t[1]; f[2] = t[2]; f[3] = t[3];
        }
    }
    static __device__ __forceinline__ V pack(const float *f) {
        if constexpr (sizeof(T) == 4) {
            return pack16<T>(f);
        } else {
            const float t[8] = {f[0], f[1], f[2], f[3], 0.f, 0.f, 0.f, 0.f};
            const uint4 q = pack16<T>(t);
            return make_uint2(q.x, q.y);
        }
    }
};

// register order <-> physical order
template <int C, bool REV> __device__ __forceinline__ void to_traversal(const float *phys, float *reg) {
#pragma unroll
    for (int j = 0; j < C; ++j) reg[j] = phys[REV ? C - 1 - j : j];
}

// ---------------------------------------------------------------------------------------------
// forward, one route over one plane.  xq: this route's LDS plane (type Tin), yq: private fp32 LDS plane.
// ---------------------------------------------------------------------------------------------
// Operand vectors of the NEXT chunk, in flight while the current one is computed; the stream runs on
// across planes (consecutive d => next row = this row + L), so plane starts do not expose HBM latency.
template <typename V> struct LeanPref { V d, b, c; float h; };

template <typename Tin, int C, bool REV>
__device__ __forceinline__ void lean_fwd_plane(const LeanArgs &a, const Tin *__restrict__ dts_row,
                                               const Tin *__restrict__ Brow, const Tin *__restrict__ Crow,
                                               float *__restrict__ chk_row, const float A2, const float Dr,
                                               const float bias, const Tin *xq, Tin *yq, const int lane,
                                               LeanPref<typename VecIO<Tin, C>::V> &pf, const bool has_next) {
    using IO = VecIO<Tin, C>;
    using V = typename IO::V;
    const int L = a.L, nseg = a.nseg;
    const int ci = REV ? 63 - lane : lane;
    float hc = 0.f;                                   // state entering the current chunk row-wide
    const int s0 = REV ? (nseg - 1) * 64 * C : 0;
    const int sstep = REV ? -64 * C : 64 * C;
    int tp0 = s0 + ci * C;
    V rd = pf.d, rb = pf.b, rc = pf.c;
    for (int s = 0; s < nseg; ++s) {
        const bool live = tp0 < L;
        float ph[C], dl[C], Bv[C], Cv[C], u[C];
        IO::unpack(rd, ph); to_traversal<C, REV>(ph, dl);
        IO::unpack(rb, ph); to_traversal<C, REV>(ph, Bv);
        IO::unpack(rc, ph); to_traversal<C, REV>(ph, Cv);
        V xv = IO::zero();
        if (live) xv = *reinterpret_cast<const V *>(xq + tp0);
        IO::unpack(xv, ph); to_traversal<C, REV>(ph, u);
        // prefetch the next chunk (of this plane, or the first chunk of the next plane)
        const int tpn = tp0 + sstep;
        if (s + 1 < nseg) {
            if (tpn < L && tpn >= 0) {
                rd = *reinterpret_cast<const V *>(dts_row + tpn);
                rb = *reinterpret_cast<const V *>(Brow + tpn);
                rc = *reinterpret_cast<const V *>(Crow + tpn);
            }
        } else if (has_next && s0 + ci * C < L) {
            pf.d = *reinterpret_cast<const V *>(dts_row + L + s0 + ci * C);
            pf.b = *reinterpret_cast<const V *>(Brow + s0 + ci * C);
            pf.c = *reinterpret_cast<const V *>(Crow + s0 + ci * C);
        }
        float av[C], bb[C];
        float P = 1.f, S = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            float v = dl[j] + bias;
            if (a.softplus == 1) v = softplus20(v);
            v = live ? v : 0.f;                       // dead lanes: identity map (a = 1, b = 0)
            av[j] = exp2_fast(v * A2);
            bb[j] = v * u[j] * Bv[j];
            S = fmaf(av[j], S, bb[j]);
            P *= av[j];
        }
        wave_scan_up(P, S);
        // exclusive prefix: map of lanes 0..lane-1, applied to the row carry
        const float Pe = dpp_mov<kWaveShr1>(1.f, P), Se = dpp_mov<kWaveShr1>(0.f, S);
        float h = fmaf(Pe, hc, Se);
        hc = fmaf(bcast_lane<63>(P), hc, bcast_lane<63>(S));  // state after this chunk row = inclusive map of lane 63
        float y[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            h = fmaf(av[j], h, bb[j]);
            y[j] = fmaf(Cv[j], h, Dr * u[j]);
        }
        if (nseg > 1 && lane == 63) chk_row[s] = hc;
        if (live) {
            float yo[C];
            to_traversal<C, REV>(y, yo);              // the map is an involution: traversal -> physical
            *reinterpret_cast<V *>(yq + tp0) = IO::pack(yo);   // private plane in the I/O precision (fp32 for fp32 I/O)
        }
        tp0 = tpn;
    }
}

// ---------------------------------------------------------------------------------------------
// backward, one route over one plane
// ---------------------------------------------------------------------------------------------
// NSEG > 0: the chunk loop is unrolled and the route's dB/dC sums over the planes of the workgroup live in
// registers (rB/rC[NSEG][C]); NSEG == 0: runtime chunk count, sums in the wave-private LDS accumulators.
template <typename Tin, int C, bool REV, int NSEG>
__device__ __forceinline__ void lean_bwd_plane(const LeanArgs &a, const Tin *__restrict__ dts_row,
                                               Tin *__restrict__ ddts_row, const Tin *__restrict__ Brow,
                                               const Tin *__restrict__ Crow, const float *__restrict__ chk_row,
                                               const float An, const float Dr, const float bias, const Tin *xq,
                                               const Tin *gq, Tin *dxq, float *accB, float *accC,
                                               float (&rB)[NSEG ? NSEG : 1][C], float (&rC)[NSEG ? NSEG : 1][C],
                                               float &dA_acc, float &dD_acc, float &dbias_acc, const int lane,
                                               LeanPref<typename VecIO<Tin, C>::V> &pf, const bool has_next) {
    using IO = VecIO<Tin, C>;
    using V = typename IO::V;
    const int L = a.L, nseg = NSEG ? NSEG : a.nseg;
    const float A2 = An * kLog2e;
    const int ci = REV ? 63 - lane : lane;
    float Ec = 0.f;                                   // E flowing in from the chunk row processed before (later in the route)
    const int sstep = REV ? 64 * C : -64 * C;         // chunks are walked against the route
    const int tpf = (REV ? 0 : (nseg - 1) * 64 * C) + ci * C;
    int tp0 = tpf;
    V rd = pf.d, rb = pf.b, rc = pf.c;
    float hin_next = pf.h;                            // chunk state entering chunk s (prefetched like the vectors)
#pragma unroll 1
    for (int s = nseg - 1; s >= 0; --s) {
        const bool live = tp0 < L;
        float ph[C], dl[C], Bv[C], Cv[C], u[C], go[C];
        IO::unpack(rd, ph); to_traversal<C, REV>(ph, dl);
        IO::unpack(rb, ph); to_traversal<C, REV>(ph, Bv);
        IO::unpack(rc, ph); to_traversal<C, REV>(ph, Cv);
        V xv = IO::zero(), gv = IO::zero();
        if (live) {
            xv = *reinterpret_cast<const V *>(xq + tp0);
            gv = *reinterpret_cast<const V *>(gq + tp0);
        }
        IO::unpack(xv, ph); to_traversal<C, REV>(ph, u);
        IO::unpack(gv, ph); to_traversal<C, REV>(ph, go);
        const int tpn = tp0 + sstep;
        const float hin0 = hin_next;
        if (s > 0) {
            if (tpn < L && tpn >= 0) {
                rd = *reinterpret_cast<const V *>(dts_row + tpn);
                rb = *reinterpret_cast<const V *>(Brow + tpn);
                rc = *reinterpret_cast<const V *>(Crow + tpn);
            }
            hin_next = (s > 1) ? chk_row[s - 2] : 0.f;
        } else if (has_next) {
            if (tpf < L) {
                pf.d = *reinterpret_cast<const V *>(dts_row + L + tpf);
                pf.b = *reinterpret_cast<const V *>(Brow + tpf);
                pf.c = *reinterpret_cast<const V *>(Crow + tpf);
            }
            pf.h = (nseg > 1) ? chk_row[nseg + nseg - 2] : 0.f;       // next plane's row of chk, entry of its last chunk
        }
        float av[C], bb[C], cg[C], sg[C];
        float P = 1.f, S = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            float v = dl[j] + bias;
            sg[j] = 1.f;
            if (a.softplus == 1) v = softplus20_sig(v, sg[j]);
            else if (a.softplus == 2) sg[j] = v > 20.f ? 1.f : 1.f - exp2_fast(-v * kLog2e);   // dts holds softplus(raw)
            v = live ? v : 0.f;
            dl[j] = v;
            av[j] = exp2_fast(v * A2);
            bb[j] = v * u[j] * Bv[j];
            cg[j] = Cv[j] * go[j];
            S = fmaf(av[j], S, bb[j]);
            P *= av[j];
        }
        float R = 0.f;
#pragma unroll
        for (int j = C - 1; j >= 0; --j) R = av[j] * (cg[j] + R);
        float P2 = P;
        wave_scan_up(P, S);
        const float Pe = dpp_mov<kWaveShr1>(1.f, P), Se = dpp_mov<kWaveShr1>(0.f, S);
        float hh = fmaf(Pe, hin0, Se);
        wave_scan_down(P2, R, lane);
        const float Pn = dpp_mov<kWaveShl1>(1.f, P2), Rn = dpp_mov<kWaveShl1>(0.f, R);
        float E = fmaf(Pn, Ec, Rn);
        Ec = fmaf(bcast_lane<0>(P2), Ec, bcast_lane<0>(R));                 // E leaving this chunk row = inclusive map of lane 0
        float h[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            hh = fmaf(av[j], hh, bb[j]);
            h[j] = hh;
        }
        float du[C], dd[C], dBv[C], dCv[C];
#pragma unroll
        for (int j = C - 1; j >= 0; --j) {
            const float dh = cg[j] + E;
            E = av[j] * dh;
            const float ah = h[j] - bb[j];
            const float s1 = dh * Bv[j];
            const float s2 = dh * An * ah;
            dA_acc = fmaf(dh * dl[j], ah, dA_acc);
            dBv[j] = dh * dl[j] * u[j];
            dCv[j] = go[j] * h[j];
            du[j] = fmaf(dl[j], s1, Dr * go[j]);
            float ddl = fmaf(u[j], s1, s2);
            ddl *= sg[j];                               // d softplus / d raw (1 when softplus is off or linear)
            dd[j] = ddl;
            dD_acc = fmaf(go[j], u[j], dD_acc);
            dbias_acc += live ? ddl : 0.f;              // (a dead lane still carries dh*A*h through s2)
        }
        if (live) {
            float t[C];
            to_traversal<C, REV>(dd, t);
            *reinterpret_cast<V *>(ddts_row + tp0) = IO::pack(t);
            to_traversal<C, REV>(du, t);
            *reinterpret_cast<V *>(dxq + tp0) = IO::pack(t);                  // this route's private dx plane
            if constexpr (NSEG > 0) {
                // one rolled loop body; the chunk index only selects which register set receives the sums
                // (values stay in traversal order and are un-permuted once, at the flush)
#pragma unroll
                for (int ss = 0; ss < NSEG; ++ss)
                    if (ss == s) {
#pragma unroll
                        for (int j = 0; j < C; ++j) {
                            rB[ss][j] += dBv[j];
                            rC[ss][j] += dCv[j];
                        }
                    }
            } else {
                to_traversal<C, REV>(dBv, t);
#pragma unroll
                for (int q = 0; q < C; q += 4) {
                    float4 v = *reinterpret_cast<float4 *>(accB + tp0 + q);
                    v.x += t[q]; v.y += t[q + 1]; v.z += t[q + 2]; v.w += t[q + 3];
                    *reinterpret_cast<float4 *>(accB + tp0 + q) = v;
                }
                to_traversal<C, REV>(dCv, t);
#pragma unroll
                for (int q = 0; q < C; q += 4) {
                    float4 v = *reinterpret_cast<float4 *>(accC + tp0 + q);
                    v.x += t[q]; v.y += t[q + 1]; v.z += t[q + 2]; v.w += t[q + 3];
                    *reinterpret_cast<float4 *>(accC + tp0 + q) = v;
                }
            }
        }
        tp0 = tpn;
    }
}

// natural + transposed LDS copies of PPT planes (source type S in HBM, LDS type T)
template <typename S, typename T, int VS>
__device__ __forceinline__ void lean_planes_load(T *nat, T *tr, const S *src, int nplanes, int L, int H, int W,
                                                 uint32_t magicW) {
    using IO = VecIO<S, VS>;
    const int nvec = L / VS;                           // L % VS == 0 guaranteed by the plan
    if (W % VS == 0) {
        // Rows are whole vectors: consecutive lanes take the SAME column block of consecutive rows, so the scattered
        // 2-byte writes of the transposed copy land on consecutive addresses (bank-conflict free; with the row-major
        // lane order below they are W*VS*2 bytes apart = one bank for W = 56).  The price -- 16-byte global loads that
        // are a row apart -- is paid in L2 hits, not HBM traffic.
        const int vpr = W / VS;                         // vectors per row
        for (int pl = 0; pl < nplanes; ++pl) {
            const S *pg = src + (int64_t)pl * L;
            for (int v = threadIdx.x; v < nvec; v += 256) {
                const int wb = v / H, h = v - wb * H;   // h fastest
                const int e0 = h * W + wb * VS;
                float f[VS];
                IO::unpack(*reinterpret_cast<const typename IO::V *>(pg + e0), f);
                *reinterpret_cast<typename VecIO<T, VS>::V *>(nat + pl * L + e0) = VecIO<T, VS>::pack(f);
#pragma unroll
                for (int q = 0; q < VS; ++q) tr[pl * L + (wb * VS + q) * H + h] = from_float<T>(f[q]);
            }
        }
        (void)vpr;
        return;
    }
    for (int pl = 0; pl < nplanes; ++pl) {
        const S *pg = src + (int64_t)pl * L;
        for (int v = threadIdx.x; v < nvec; v += 256) {
            const int e0 = v * VS;
            float f[VS];
            IO::unpack(*reinterpret_cast<const typename IO::V *>(pg + e0), f);
            int h = (int)__umulhi((uint32_t)e0, magicW), w = e0 - h * W;
            *reinterpret_cast<typename VecIO<T, VS>::V *>(nat + pl * L + e0) = VecIO<T, VS>::pack(f);
#pragma unroll
            for (int q = 0; q < VS; ++q) {
                const T val = from_float<T>(f[q]);
                tr[pl * L + w * H + h] = val;
                if (++w == W) {
                    w = 0;
                    ++h;
                }
            }
        }
    }
}

// The same staging split in two, so that the HBM latency of the NEXT tile's planes hides under the sweeps of the
// current one: `issue` starts the 16-byte loads into registers (NV vectors per thread cover a tile of at most
// 256*NV*VS elements), `commit` converts and writes the natural + transposed LDS copies once the planes are free.
template <typename S, int VS, int NV> struct PlaneRegs { typename VecIO<S, VS>::V v[NV]; };

template <typename S, int VS, int NV>
__device__ __forceinline__ void lean_planes_issue(PlaneRegs<S, VS, NV> &r, const S *src, int nelem) {
    const int nvec = nelem / VS;
#pragma unroll
    for (int m = 0; m < NV; ++m) {
        const int v = threadIdx.x + m * 256;
        if (v < nvec) r.v[m] = *reinterpret_cast<const typename VecIO<S, VS>::V *>(src + (int64_t)v * VS);
    }
}

template <typename S, typename T, int VS, int NV>
__device__ __forceinline__ void lean_planes_commit(const PlaneRegs<S, VS, NV> &r, T *nat, T *tr, int nelem, int L, int H,
                                                   int W, uint32_t magicW, uint32_t magicL) {
    const int nvec = nelem / VS;
#pragma unroll
    for (int m = 0; m < NV; ++m) {
        const int v = threadIdx.x + m * 256;
        if (v >= nvec) continue;
        const int e0 = v * VS;                          // element index inside the tile (planes are contiguous)
        float f[VS];
        VecIO<S, VS>::unpack(r.v[m], f);
        *reinterpret_cast<typename VecIO<T, VS>::V *>(nat + e0) = VecIO<T, VS>::pack(f);
        const int pl = (int)__umulhi((uint32_t)e0, magicL), ep = e0 - pl * L;        // plane, element in plane
        int h = (int)__umulhi((uint32_t)ep, magicW), w = ep - h * W;
#pragma unroll
        for (int q = 0; q < VS; ++q) {
            tr[pl * L + w * H + h] = from_float<T>(f[q]);
            if (++w == W) {
                w = 0;
                ++h;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// kernels: wave w owns route {0,2,1,3}[w]; LDS (forward):  xN | xT (Tin, PPT planes) | 4 x y planes (fp32)
//                                         LDS (backward): xN | xT | gN | gT (Tin) | dxN | dxT (fp32) | 4 x (accB|accC)
// ---------------------------------------------------------------------------------------------
template <typename Tin, typename Tout, int C>
__global__ void __launch_bounds__(256) ss2d_fwd_lean_kernel(const LeanArgs a) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = a.L, H = a.H, W = a.W, D = a.D_, PL = a.ppt * L;
    const int tiles_pb = D / a.ppt;
    const int groups_pb = tiles_pb / a.pli;
    const int b = blockIdx.x / groups_pb, tg = blockIdx.x - b * groups_pb;
    // LDS: xN | xT | 4 private y plane sets, all in the I/O precision: the per-route partial sums are rounded to
    // it once before the fixed-order fp32 merge (for 16-bit I/O that keeps 4 workgroups per CU instead of 2)
    Tin *xN = reinterpret_cast<Tin *>(smem), *xT = xN + PL;
    Tin *Y = xT + PL;
    const bool col = wave >> 1, rev = wave & 1;
    const int k = (wave & 1) * 2 + (wave >> 1);
    const Tin *xq = col ? xT : xN;
    Tin *yq = Y + (size_t)wave * PL;
    const int64_t route = (int64_t)b * 4 + k;
    const Tin *Brow = (const Tin *)a.Bs + route * L, *Crow = (const Tin *)a.Cs + route * L;
    using V = typename VecIO<Tin, C>::V;
    LeanPref<V> pf;
    pf.d = pf.b = pf.c = VecIO<Tin, C>::zero();
    pf.h = 0.f;
    {   // first chunk of the first plane of this workgroup
        const int tpf = (rev ? (a.nseg - 1) * 64 * C : 0) + (rev ? 63 - lane : lane) * C;
        if (tpf < L) {
            pf.d = *reinterpret_cast<const V *>((const Tin *)a.dts + (route * D + (int64_t)tg * a.pli * a.ppt) * L + tpf);
            pf.b = *reinterpret_cast<const V *>(Brow + tpf);
            pf.c = *reinterpret_cast<const V *>(Crow + tpf);
        }
    }
    const int n_planes = a.pli * a.ppt;
    constexpr int NVX = C == 8 ? 2 : 4;                // ppt*L <= 3200 elements (plan)
    constexpr bool kPipe = sizeof(Tin) == 2 && C == 8;  // as in the backward: next tile's planes in flight during the sweeps
    PlaneRegs<Tin, C, NVX> px;
    if (kPipe && !(a.dbg & 2))
        lean_planes_issue<Tin, C, NVX>(px, (const Tin *)a.x + ((int64_t)b * D + (int64_t)tg * a.pli * a.ppt) * L, PL);
    for (int it = 0; it < a.pli; ++it) {
        const int d0 = (tg * a.pli + it) * a.ppt;
        const int64_t po = ((int64_t)b * D + d0) * L;
        __syncthreads();
        if (!(a.dbg & 2)) {
            // (not pipelined: still ONE round trip for the whole tile -- the per-plane loop of lean_planes_load is a
            //  chain of ppt dependent HBM latencies)
            if (!kPipe) lean_planes_issue<Tin, C, NVX>(px, (const Tin *)a.x + po, PL);
            lean_planes_commit<Tin, Tin, C, NVX>(px, xN, xT, PL, L, H, W, a.magicW, a.magicL);
        }
        __syncthreads();
        if (kPipe && !(a.dbg & 2) && it + 1 < a.pli) lean_planes_issue<Tin, C, NVX>(px, (const Tin *)a.x + po + PL, PL);
        for (int pl = 0; pl < ((a.dbg & 1) ? 0 : a.ppt); ++pl) {
            const int d = d0 + pl, row = k * D + d;
            const Tin *dts_row = (const Tin *)a.dts + (route * D + d) * L;
            float *chk_row = a.chk + (route * D + d) * a.nseg;
            const float A2 = a.A[row] * kLog2e, Dr = a.D[row], bias = a.softplus == 2 ? 0.f : a.bias[row];
            const bool has_next = it * a.ppt + pl + 1 < n_planes;
            if (rev) lean_fwd_plane<Tin, C, true>(a, dts_row, Brow, Crow, chk_row, A2, Dr, bias, xq + pl * L, yq + pl * L, lane, pf, has_next);
            else lean_fwd_plane<Tin, C, false>(a, dts_row, Brow, Crow, chk_row, A2, Dr, bias, xq + pl * L, yq + pl * L, lane, pf, has_next);
        }
        __syncthreads();
        if (a.dbg & 4) continue;
        Tout *yo = (Tout *)a.y + po;
        const Tin *Y0 = Y, *Y1 = Y + PL, *Y2 = Y + 2 * PL, *Y3 = Y + 3 * PL;
        for (int pl = 0; pl < a.ppt; ++pl)
            for (int e = threadIdx.x; e < L; e += 256) {
                const int h = (int)__umulhi((uint32_t)e, a.magicW), w = e - h * W;
                const int n_ = pl * L + e, t_ = pl * L + w * H + h;
                stf<Tout>(yo + (int64_t)pl * L + e,
                          (ldf<Tin>(Y0 + n_) + ldf<Tin>(Y1 + n_)) + (ldf<Tin>(Y2 + t_) + ldf<Tin>(Y3 + t_)));   // fixed order
            }
    }
}

template <typename Tin, typename Tout, int C, int NSEG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C == 4 ? 3 : 1)))   // 14x14: keep 3 waves/SIMD
ss2d_bwd_lean_kernel(const LeanArgs a) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = a.L, H = a.H, W = a.W, D = a.D_, PL = a.ppt * L;
    const int tiles_pb = D / a.ppt;
    const int groups_pb = tiles_pb / a.pli;
    const int b = blockIdx.x / groups_pb, tg = blockIdx.x - b * groups_pb;
    // LDS: xN | xT | gN | gT | 4 private dx planes (all Tin, PPT planes each) | [NSEG == 0: 4 x (accB | accC) fp32]
    Tin *xN = reinterpret_cast<Tin *>(smem), *xT = xN + PL, *gN = xT + PL, *gT = gN + PL, *DX = gT + PL;
    float *accB = smem + (8 * (size_t)PL * sizeof(Tin)) / 4 + (size_t)wave * 2 * L, *accC = accB + L;
    if (NSEG == 0)
        for (int e = lane; e < 2 * L; e += 64) accB[e] = 0.f;
    float rB[NSEG ? NSEG : 1][C], rC[NSEG ? NSEG : 1][C];
#pragma unroll
    for (int s = 0; s < (NSEG ? NSEG : 1); ++s)
#pragma unroll
        for (int j = 0; j < C; ++j) rB[s][j] = rC[s][j] = 0.f;
    const bool col = wave >> 1, rev = wave & 1;
    const int k = (wave & 1) * 2 + (wave >> 1);
    const Tin *xq = col ? xT : xN, *gq = col ? gT : gN;
    Tin *dxq = DX + (size_t)wave * PL;
    const int64_t route = (int64_t)b * 4 + k;
    const Tin *Brow = (const Tin *)a.Bs + route * L, *Crow = (const Tin *)a.Cs + route * L;
    using V = typename VecIO<Tin, C>::V;
    const int nseg = NSEG ? NSEG : a.nseg;
    LeanPref<V> pf;
    pf.d = pf.b = pf.c = VecIO<Tin, C>::zero();
    pf.h = 0.f;
    {   // last chunk (in route order) of the first plane of this workgroup
        const int tpf = (rev ? 0 : (nseg - 1) * 64 * C) + (rev ? 63 - lane : lane) * C;
        const int64_t r0 = route * D + (int64_t)tg * a.pli * a.ppt;
        if (tpf < L) {
            pf.d = *reinterpret_cast<const V *>((const Tin *)a.dts + r0 * L + tpf);
            pf.b = *reinterpret_cast<const V *>(Brow + tpf);
            pf.c = *reinterpret_cast<const V *>(Crow + tpf);
        }
        if (nseg > 1) pf.h = a.chk[r0 * nseg + nseg - 2];
    }
    const int n_planes = a.pli * a.ppt;
    // plane staging is software-pipelined: the loads of tile it+1 are in flight while tile it is swept
    constexpr int NVX = C == 8 ? 2 : 4;                // ppt*L <= 3200 elements (plan): 400 / 800 vectors over 256 threads
    // (C == 8 variants only: they are LDS-limited to one or two workgroups per CU, so the extra registers are free and
    //  nothing else hides the staging; the 14x14 variant runs 3 waves per SIMD and loses one to the registers)
    constexpr bool kPipe = sizeof(Tin) == 2 && sizeof(Tout) == 4 && C == 8;
    PlaneRegs<Tin, C, NVX> px;
    PlaneRegs<Tout, 4, 4> pg;
    if (kPipe && !(a.dbg & 2)) {
        const int64_t po0 = ((int64_t)b * D + (int64_t)tg * a.pli * a.ppt) * L;
        lean_planes_issue<Tin, C, NVX>(px, (const Tin *)a.x + po0, PL);
        lean_planes_issue<Tout, 4, 4>(pg, (const Tout *)a.dy + po0, PL);
    }
    for (int it = 0; it < a.pli; ++it) {
        const int d0 = (tg * a.pli + it) * a.ppt;
        const int64_t po = ((int64_t)b * D + d0) * L;
        __syncthreads();
        if (!(a.dbg & 2)) {
            if (!kPipe) {                               // one round trip for the tile (see the forward kernel)
                lean_planes_issue<Tin, C, NVX>(px, (const Tin *)a.x + po, PL);
                lean_planes_issue<Tout, 4, 4>(pg, (const Tout *)a.dy + po, PL);
            }
            lean_planes_commit<Tin, Tin, C, NVX>(px, xN, xT, PL, L, H, W, a.magicW, a.magicL);
            lean_planes_commit<Tout, Tin, 4, 4>(pg, gN, gT, PL, L, H, W, a.magicW, a.magicL);
        }
        __syncthreads();
        if (kPipe && !(a.dbg & 2) && it + 1 < a.pli) {
            lean_planes_issue<Tin, C, NVX>(px, (const Tin *)a.x + po + PL, PL);
            lean_planes_issue<Tout, 4, 4>(pg, (const Tout *)a.dy + po + PL, PL);
        }
        for (int pl = 0; pl < ((a.dbg & 1) ? 0 : a.ppt); ++pl) {
            const int d = d0 + pl, row = k * D + d;
            const int64_t ro = (route * D + d) * L;
            const float *chk_row = a.chk + (route * D + d) * nseg;
            const float An = a.A[row], Dr = a.D[row], bias = a.softplus == 2 ? 0.f : a.bias[row];
            float dA_acc = 0.f, dD_acc = 0.f, dbias_acc = 0.f;
            const bool has_next = it * a.ppt + pl + 1 < n_planes;
            if (rev)
                lean_bwd_plane<Tin, C, true, NSEG>(a, (const Tin *)a.dts + ro, (Tin *)a.ddts + ro, Brow, Crow, chk_row, An, Dr,
                                                   bias, xq + pl * L, gq + pl * L, dxq + pl * L, accB, accC, rB, rC, dA_acc,
                                                   dD_acc, dbias_acc, lane, pf, has_next);
            else
                lean_bwd_plane<Tin, C, false, NSEG>(a, (const Tin *)a.dts + ro, (Tin *)a.ddts + ro, Brow, Crow, chk_row, An, Dr,
                                                    bias, xq + pl * L, gq + pl * L, dxq + pl * L, accB, accC, rB, rC, dA_acc,
                                                    dD_acc, dbias_acc, lane, pf, has_next);
            for (int o = 32; o > 0; o >>= 1) {
                dA_acc += __shfl_xor(dA_acc, o, 64);
                dD_acc += __shfl_xor(dD_acc, o, 64);
                dbias_acc += __shfl_xor(dbias_acc, o, 64);
            }
            if (lane == 0) {
                atomicAdd(a.dA + row, dA_acc);
                atomicAdd(a.dD + row, dD_acc);
                atomicAdd(a.dbias + row, dbias_acc);
            }
        }
        __syncthreads();
        if (a.dbg & 4) continue;
        Tin *dxo = (Tin *)a.dx + po;
        const Tin *X0 = DX, *X1 = DX + PL, *X2 = DX + 2 * PL, *X3 = DX + 3 * PL;
        for (int pl = 0; pl < a.ppt; ++pl)
            for (int e = threadIdx.x; e < L; e += 256) {
                const int h = (int)__umulhi((uint32_t)e, a.magicW), w = e - h * W;
                const int n_ = pl * L + e, t_ = pl * L + w * H + h;
                stf<Tin>(dxo + (int64_t)pl * L + e, (ldf<Tin>(X0 + n_) + ldf<Tin>(X1 + n_)) + (ldf<Tin>(X2 + t_) + ldf<Tin>(X3 + t_)));
            }
    }
    wave_sync();
    if (a.dbg & 8) return;
    float *dBg = a.dBs + route * L, *dCg = a.dCs + route * L;
    if constexpr (NSEG > 0) {
        // registers hold [traversal chunk s][traversal element j] of this lane.  Atomics are only fast when a
        // wave instruction covers contiguous bytes (MI355X_MICROARCH.md, "Global float atomics"), so the sums
        // are first laid out by position in LDS (the plane region is free now: 4 waves x L floats fit in it).
        __syncthreads();
        float *stage = smem + (size_t)wave * L;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int s = 0; s < NSEG; ++s) {
                const int tp0 = (rev ? NSEG - 1 - s : s) * 64 * C + (rev ? 63 - lane : lane) * C;
                if (tp0 < L) {
#pragma unroll
                    for (int j = 0; j < C; ++j) stage[tp0 + (rev ? C - 1 - j : j)] = pass ? rC[s][j] : rB[s][j];
                }
            }
            wave_sync();
            float *dst = pass ? dCg : dBg;
            for (int e = lane; e < L; e += 64) atomicAdd(dst + e, stage[e]);
            wave_sync();
        }
    } else {
        for (int e = lane; e < L; e += 64) {
            atomicAdd(dBg + e, accB[e]);
            atomicAdd(dCg + e, accC[e]);
        }
    }
}

}  // namespace xfm
