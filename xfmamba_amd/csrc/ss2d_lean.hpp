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
    uint32_t magicH;   // ceil(2^32 / H)
    float *parts;      // ss2d_l3.hip: (batch, groups, 4, 2, L) fp32 partial dB / dC sums of the workgroups (null: atomics)
    int xmap;          // ss2d_l3.hip: 1 = XCD-local sample placement (batch % 8 == 0), see l3_block_map
    const void *xrt;   // ss2d_l3.hip, softplus mode 3: dt_proj input rows (batch, 4, L, Rp) bf16, position-major, route order
    const void *dtw;   //                               dt_proj weight (4, D, Rp) bf16
    const float *Bs32, *Cs32;   // ss2d_w.hpp (backward): the B / C rows as fp32 (batch, 4, L)
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

// Backward: the same stream PD chunk rows deep.  The backward kernels run one wave per SIMD (LDS / registers), so
// nothing but the wave's own requests in flight hides HBM latency: one chunk row of maths (~0.5 us) does not cover it.
// Depth by variant: runtime chunk count (NSEG 0) rotates three slots; two chunk rows per plane = two slots (one plane
// ahead); a single chunk row per plane and the 18-row BIG variant keep one (BIG: three copies of its chunk-row body, each
// with its block of pinned accumulators, ran 3.5x slower than one -- registers and instruction cache).
__host__ __device__ constexpr int lean_pd(int nseg) { return nseg == 0 ? 3 : (nseg == 2 ? 2 : 1); }
template <typename Tin, int C, int PD> struct LeanStream {
    using V = typename VecIO<Tin, C>::V;
    V d[PD], b[PD], c[PD];
    float h[PD];
    const Tin *dts_row;                               // cursor: row, chunk row (walked nseg-1 .. 0) and planes left to request
    const float *chk_row;
    int s, planes_left, phase;                        // phase: slot of the next chunk row to consume
    int vzero;                                        // 0 in a VGPR the compiler cannot see through (see lean_stream_issue)
};

// request the cursor's chunk row into slot SLOT and advance the cursor
template <typename Tin, int C, bool REV, int SLOT, int PD>
__device__ __forceinline__ void lean_stream_issue(LeanStream<Tin, C, PD> &st, const Tin *Brow, const Tin *Crow, const int L,
                                                  const int nseg, const int ci) {
    using V = typename VecIO<Tin, C>::V;
    const int tp = (REV ? nseg - 1 - st.s : st.s) * 64 * C + ci * C;
    if (st.planes_left > 0) {
        if (tp < L) {
            st.d[SLOT] = *reinterpret_cast<const V *>(st.dts_row + tp);
            st.b[SLOT] = *reinterpret_cast<const V *>(Brow + tp);
            st.c[SLOT] = *reinterpret_cast<const V *>(Crow + tp);
        }
        // (a VECTOR load on purpose: as a scalar load the checkpoint shares lgkmcnt with the LDS reads and returns out of
        //  order, so every LDS wait of the chunk row would also wait out this HBM round trip)
        st.h[SLOT] = st.s > 0 ? st.chk_row[st.s - 1 + st.vzero] : 0.f;
        if (--st.s < 0) {
            st.s = nseg - 1;
            st.dts_row += L;
            st.chk_row += nseg;
            --st.planes_left;
        }
    }
}

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

// Accumulators of the BIG backward: 18 chunk rows x (dB, dC) x 8 elements = 288 sums per lane.  The first 16 rows are
// pinned in the accumulation half of the register file (256 AGPRs: a v_accvgpr_read / add / v_accvgpr_write per update),
// the last two stay in VGPRs -- left to the register allocator they end up in scratch.
template <bool ACC> __device__ __forceinline__ void lean_acc_zero(float &r) {
    if constexpr (ACC) asm("v_accvgpr_write_b32 %0, 0" : "=a"(r));
    else r = 0.f;
}
template <bool ACC> __device__ __forceinline__ void lean_acc_add(float &r, const float v) {
    if constexpr (ACC) {
        float t;
        asm("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(r));
        t += v;
        asm("v_accvgpr_write_b32 %0, %1" : "=a"(r) : "v"(t));
    } else {
        r += v;
    }
}
template <bool ACC> __device__ __forceinline__ float lean_acc_get(const float &r) {
    if constexpr (ACC) {
        float t;
        asm("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(r));
        return t;
    } else {
        return r;
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
                                               LeanStream<Tin, C, lean_pd(NSEG)> &st) {
    using IO = VecIO<Tin, C>;
    using V = typename IO::V;
    const int L = a.L, nseg = NSEG ? NSEG : a.nseg;
    const float A2 = An * kLog2e;
    const int ci = REV ? 63 - lane : lane;
    float Ec = 0.f;                                   // E flowing in from the chunk row processed before (later in the route)
    const int sstep = REV ? 64 * C : -64 * C;         // chunks are walked against the route
    const int tpf = (REV ? 0 : (nseg - 1) * 64 * C) + ci * C;
    int tp0 = tpf;
    V xv_n = IO::zero(), gv_n = IO::zero();               // x / dy of the NEXT chunk row to process, from LDS
    if (NSEG == 0 && tpf < L) {
        xv_n = *reinterpret_cast<const V *>(xq + tpf);
        gv_n = *reinterpret_cast<const V *>(gq + tpf);
    }
    // One chunk row.  SLOT is static: the row's operands are consumed from stream slot SLOT and the request for the row
    // lean_pd(NSEG) ahead goes into the same registers -- rotating the slots with moves would make every move wait for the
    // load in flight to its source, i.e. collapse the stream to one row of look-ahead.
    auto chunk_row = [&](auto slot_c, const int s) {
        constexpr int SLOT = decltype(slot_c)::value;
        const bool live = tp0 < L;
        float ph[C], dl[C], Bv[C], Cv[C], u[C], go[C];
        IO::unpack(st.d[SLOT], ph); to_traversal<C, REV>(ph, dl);
        IO::unpack(st.b[SLOT], ph); to_traversal<C, REV>(ph, Bv);
        IO::unpack(st.c[SLOT], ph); to_traversal<C, REV>(ph, Cv);
        const float hin0 = st.h[SLOT];
        lean_stream_issue<Tin, C, REV, SLOT, lean_pd(NSEG)>(st, Brow, Crow, L, nseg, ci);
        // LDS operands: x / dy of this chunk row were requested during the previous one (xv_n / gv_n); the running dB/dC
        // sums of the LDS-accumulator variant are requested here, ~500 instructions before they are needed.  The
        // backward kernels are LDS-capacity bound at one wave per SIMD: the registers this costs are free, and nothing
        // else covers an LDS round trip.
        V xv = xv_n, gv = gv_n;
        if constexpr (NSEG != 0) {                       // (register-accumulator variants: no spare registers, plain reads)
            xv = gv = IO::zero();
            if (live) {
                xv = *reinterpret_cast<const V *>(xq + tp0);
                gv = *reinterpret_cast<const V *>(gq + tp0);
            }
        }
        float4 aB[C / 4], aC[C / 4];
        if constexpr (NSEG == 0) {
            if (live) {
#pragma unroll
                for (int q = 0; q < C / 4; ++q) {
                    aB[q] = *reinterpret_cast<const float4 *>(accB + tp0 + 4 * q);
                    aC[q] = *reinterpret_cast<const float4 *>(accC + tp0 + 4 * q);
                }
            }
        }
        const int tpn = tp0 + sstep;
        if constexpr (NSEG == 0) {
            if (s > 0 && tpn < L) {                      // (tpn >= 0 always: the walk ends at s == 0)
                xv_n = *reinterpret_cast<const V *>(xq + tpn);
                gv_n = *reinterpret_cast<const V *>(gq + tpn);
            } else {
                xv_n = gv_n = IO::zero();
            }
        }
        IO::unpack(xv, ph); to_traversal<C, REV>(ph, u);
        IO::unpack(gv, ph); to_traversal<C, REV>(ph, go);
        // The element-wise maths runs on pairs (v_pk_mul/add/fma_f32: two elements per VALU instruction); only the four
        // recurrences (S/P, R, h, E) are scalar chains.  Lanes past the end of the row exist in the tail chunk row only.
        typedef float f2 __attribute__((ext_vector_type(2)));
        constexpr int CP = C / 2;
        const bool tail = REV ? s == 0 : s == nseg - 1;
        f2 v2[CP], sg2[CP], u2[CP], B2[CP], go2[CP], av2[CP], bb2[CP], cg2[CP];
#pragma unroll
        for (int q = 0; q < CP; ++q) {
            u2[q] = f2{u[2 * q], u[2 * q + 1]};
            B2[q] = f2{Bv[2 * q], Bv[2 * q + 1]};
            go2[q] = f2{go[2 * q], go[2 * q + 1]};
            v2[q] = f2{dl[2 * q], dl[2 * q + 1]};
        }
        if (a.softplus == 2) {                          // dts holds softplus(raw): sigmoid(raw) = 1 - exp(-softplus)
#pragma unroll                                          // (beyond 20 the exponential is below half an ulp of 1: no select)
            for (int q = 0; q < CP; ++q) {
                const f2 t = v2[q] * (-kLog2e);
                sg2[q] = 1.f - f2{exp2_fast(t.x), exp2_fast(t.y)};
            }
        } else {
#pragma unroll
            for (int q = 0; q < CP; ++q) {
                v2[q] += bias;
                sg2[q] = f2{1.f, 1.f};
                if (a.softplus == 1) {
                    float s0, s1;
                    const float v0 = softplus20_sig(v2[q].x, s0), v1 = softplus20_sig(v2[q].y, s1);
                    v2[q] = f2{v0, v1};
                    sg2[q] = f2{s0, s1};
                }
            }
        }
        if (tail) {
#pragma unroll
            for (int q = 0; q < CP; ++q) v2[q] = live ? v2[q] : f2{0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < CP; ++q) {
            const f2 t = v2[q] * A2;
            av2[q] = f2{exp2_fast(t.x), exp2_fast(t.y)};
            bb2[q] = v2[q] * u2[q] * B2[q];
            cg2[q] = f2{Cv[2 * q], Cv[2 * q + 1]} * go2[q];
        }
        float P = 1.f, S = 0.f;
#pragma unroll
        for (int q = 0; q < CP; ++q) {
            S = fmaf(av2[q].x, S, bb2[q].x);
            P *= av2[q].x;
            S = fmaf(av2[q].y, S, bb2[q].y);
            P *= av2[q].y;
        }
        float R = 0.f;
#pragma unroll
        for (int q = CP - 1; q >= 0; --q) {
            R = av2[q].y * (cg2[q].y + R);
            R = av2[q].x * (cg2[q].x + R);
        }
        float P2 = P;
        wave_scan_up(P, S);
        const float Pe = dpp_mov<kWaveShr1>(1.f, P), Se = dpp_mov<kWaveShr1>(0.f, S);
        float hh = fmaf(Pe, hin0, Se);
        wave_scan_down(P2, R, lane);
        const float Pn = dpp_mov<kWaveShl1>(1.f, P2), Rn = dpp_mov<kWaveShl1>(0.f, R);
        float E = fmaf(Pn, Ec, Rn);
        Ec = fmaf(bcast_lane<0>(P2), Ec, bcast_lane<0>(R));                 // E leaving this chunk row = inclusive map of lane 0
        f2 h2[CP], dh2[CP];
#pragma unroll
        for (int q = 0; q < CP; ++q) {
            hh = fmaf(av2[q].x, hh, bb2[q].x);
            h2[q].x = hh;
            hh = fmaf(av2[q].y, hh, bb2[q].y);
            h2[q].y = hh;
        }
#pragma unroll
        for (int q = CP - 1; q >= 0; --q) {
            dh2[q].y = cg2[q].y + E;
            E = av2[q].y * dh2[q].y;
            dh2[q].x = cg2[q].x + E;
            E = av2[q].x * dh2[q].x;
        }
        float du[C], dd[C], dBv[C], dCv[C];
        f2 dA2 = f2{0.f, 0.f}, dD2 = dA2, db2 = dA2;
#pragma unroll
        for (int q = 0; q < CP; ++q) {
            const f2 ah = h2[q] - bb2[q];
            const f2 s1 = dh2[q] * B2[q];
            const f2 dhah = dh2[q] * ah;
            dA2 = __builtin_elementwise_fma(v2[q], dhah, dA2);
            const f2 dBq = dh2[q] * v2[q] * u2[q];
            const f2 dCq = go2[q] * h2[q];
            const f2 duq = __builtin_elementwise_fma(v2[q], s1, go2[q] * Dr);
            f2 ddl = __builtin_elementwise_fma(u2[q], s1, dhah * An);
            ddl *= sg2[q];                              // d softplus / d raw (1 when softplus is off or linear)
            dD2 = __builtin_elementwise_fma(go2[q], u2[q], dD2);
            db2 += (tail && !live) ? f2{0.f, 0.f} : ddl;     // (a dead lane still carries dh*A*h through ddl)
            dBv[2 * q] = dBq.x; dBv[2 * q + 1] = dBq.y;
            dCv[2 * q] = dCq.x; dCv[2 * q + 1] = dCq.y;
            du[2 * q] = duq.x; du[2 * q + 1] = duq.y;
            dd[2 * q] = ddl.x; dd[2 * q + 1] = ddl.y;
        }
        dA_acc += dA2.x + dA2.y;
        dD_acc += dD2.x + dD2.y;
        dbias_acc += db2.x + db2.y;
        if (live) {
            float t[C];
            to_traversal<C, REV>(dd, t);
            *reinterpret_cast<V *>(ddts_row + tp0) = IO::pack(t);
            to_traversal<C, REV>(du, t);
            *reinterpret_cast<V *>(dxq + tp0) = IO::pack(t);                  // this route's private dx plane
            if constexpr (NSEG > 0) {
                // one rolled loop body; the chunk index only selects which register set receives the sums
                // (values stay in traversal order and are un-permuted once, at the flush)
                if constexpr (NSEG > 2) {
                    // many chunk rows (the BIG variant): ONE jump per chunk row into a block of static-register adds
#define XFM_ACC_CASE(SS)                                      \
    case SS:                                                  \
        if constexpr (SS < NSEG) {                            \
            _Pragma("unroll") for (int j = 0; j < C; ++j) {   \
                lean_acc_add<(SS < 16)>(rB[SS][j], dBv[j]);   \
                lean_acc_add<(SS < 16)>(rC[SS][j], dCv[j]);   \
            }                                                 \
        }                                                     \
        break;
                    switch (s) {
                        XFM_ACC_CASE(0) XFM_ACC_CASE(1) XFM_ACC_CASE(2) XFM_ACC_CASE(3) XFM_ACC_CASE(4) XFM_ACC_CASE(5)
                        XFM_ACC_CASE(6) XFM_ACC_CASE(7) XFM_ACC_CASE(8) XFM_ACC_CASE(9) XFM_ACC_CASE(10) XFM_ACC_CASE(11)
                        XFM_ACC_CASE(12) XFM_ACC_CASE(13) XFM_ACC_CASE(14) XFM_ACC_CASE(15) XFM_ACC_CASE(16) XFM_ACC_CASE(17)
                    }
#undef XFM_ACC_CASE
                } else {
#pragma unroll
                    for (int ss = 0; ss < NSEG; ++ss)
                        if (ss == s) {
#pragma unroll
                            for (int j = 0; j < C; ++j) {
                                rB[ss][j] += dBv[j];
                                rC[ss][j] += dCv[j];
                            }
                        }
                }
            } else {
                to_traversal<C, REV>(dBv, t);
#pragma unroll
                for (int q = 0; q < C; q += 4) {
                    float4 v = aB[q / 4];
                    v.x += t[q]; v.y += t[q + 1]; v.z += t[q + 2]; v.w += t[q + 3];
                    *reinterpret_cast<float4 *>(accB + tp0 + q) = v;
                }
                to_traversal<C, REV>(dCv, t);
#pragma unroll
                for (int q = 0; q < C; q += 4) {
                    float4 v = aC[q / 4];
                    v.x += t[q]; v.y += t[q + 1]; v.z += t[q + 2]; v.w += t[q + 3];
                    *reinterpret_cast<float4 *>(accC + tp0 + q) = v;
                }
            }
        }
        tp0 = tpn;
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    if constexpr (NSEG == 1) {
        chunk_row(S0{}, 0);
    } else if constexpr (NSEG == 2) {
        chunk_row(S0{}, 1);
        chunk_row(S1{}, 0);
    } else if constexpr (NSEG > 2) {
#pragma unroll 1
        for (int s = NSEG - 1; s >= 0; --s) chunk_row(S0{}, s);
    } else {
        // runtime chunk count: the slot phase carries across planes (the stream does not restart), enter the three-row
        // rotation where it stands
        int s = nseg - 1;
        int ph = st.phase;
        if (ph == 1) goto slot1;
        if (ph == 2) goto slot2;
#pragma unroll 1
        for (;;) {
            chunk_row(S0{}, s);
            ph = 1;
            if (--s < 0) break;
        slot1:
            chunk_row(S1{}, s);
            ph = 2;
            if (--s < 0) break;
        slot2:
            chunk_row(S2{}, s);
            ph = 0;
            if (--s < 0) break;
        }
        st.phase = ph;
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

// Which VS elements of the tile vector v of a workgroup is: when the rows are whole vectors (W % VS == 0) consecutive
// lanes take the SAME column block of consecutive rows, so the 2-byte writes of the transposed copy (and the 2-byte
// reads of the merges) of a wave instruction fall on consecutive addresses instead of W*VS*2 bytes apart (one or two
// banks for W = 56 / 96: a 32- to 64-way conflict per instruction).  The price -- 16-byte global accesses a row
// apart -- is paid in L2 / TA cycles, not HBM traffic.  Otherwise vectors are taken in memory order.
template <int VS>
__device__ __forceinline__ void lean_vec_pos(const LeanArgs &a, const int v, int &e0, int &pl, int &h, int &w) {
    const int idx = v * VS;
    pl = (int)__umulhi((uint32_t)idx, a.magicL);
    const int ep = idx - pl * a.L;
    if (a.W % VS == 0) {
        const int r = ep / VS;
        const int wb = (int)__umulhi((uint32_t)r, a.magicH);
        h = r - wb * a.H;
        w = wb * VS;
    } else {
        h = (int)__umulhi((uint32_t)ep, a.magicW);
        w = ep - h * a.W;
    }
    e0 = pl * a.L + h * a.W + w;
}

template <typename S, int VS, int NV>
__device__ __forceinline__ void lean_planes_issue(const LeanArgs &a, PlaneRegs<S, VS, NV> &r, const S *src, int nelem) {
    const int nvec = nelem / VS;
#pragma unroll
    for (int m = 0; m < NV; ++m) {
        const int v = threadIdx.x + m * 256;
        if (v < nvec) {
            int e0, pl, h, w;
            lean_vec_pos<VS>(a, v, e0, pl, h, w);
            r.v[m] = *reinterpret_cast<const typename VecIO<S, VS>::V *>(src + e0);
        }
    }
}

template <typename S, typename T, int VS, int NV>
__device__ __forceinline__ void lean_planes_commit(const LeanArgs &a, const PlaneRegs<S, VS, NV> &r, T *nat, T *tr, int nelem) {
    const int nvec = nelem / VS;
    const int L = a.L, H = a.H, W = a.W;
#pragma unroll
    for (int m = 0; m < NV; ++m) {
        const int v = threadIdx.x + m * 256;
        if (v >= nvec) continue;
        int e0, pl, h, w;                               // element index inside the tile (planes are contiguous)
        lean_vec_pos<VS>(a, v, e0, pl, h, w);
        float f[VS];
        VecIO<S, VS>::unpack(r.v[m], f);
        *reinterpret_cast<typename VecIO<T, VS>::V *>(nat + e0) = VecIO<T, VS>::pack(f);
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

// Merge of the four private planes of a tile (routes 0/2 in natural, 1/3 in transposed order) into HBM, fixed order.
// Rows of whole 8-element vectors: one 16-byte read per natural plane, conflict-free 2-byte reads of the transposed
// ones (lean_vec_pos), one vector store; otherwise element by element.
template <typename Tin, typename To, int C>
__device__ __forceinline__ void lean_merge_store(const LeanArgs &a, To *out, const Tin *P0, const Tin *P1, const Tin *P2,
                                                 const Tin *P3, const int PL) {
    const int L = a.L, H = a.H, W = a.W;
    if constexpr (C == 8 && sizeof(Tin) == 2) {
        if (W % 8 == 0) {
            for (int v = threadIdx.x; v < PL / 8; v += 256) {
                int e0, pl, h, w;
                lean_vec_pos<8>(a, v, e0, pl, h, w);
                float n0[8], n1[8], o[8];
                VecIO<Tin, 8>::unpack(*reinterpret_cast<const uint4 *>(P0 + e0), n0);
                VecIO<Tin, 8>::unpack(*reinterpret_cast<const uint4 *>(P1 + e0), n1);
                const int t0 = pl * L + w * H + h;
#pragma unroll
                for (int q = 0; q < 8; ++q) o[q] = (n0[q] + n1[q]) + (ldf<Tin>(P2 + t0 + q * H) + ldf<Tin>(P3 + t0 + q * H));
                if constexpr (sizeof(To) == 2) {
                    *reinterpret_cast<uint4 *>(out + e0) = VecIO<To, 8>::pack(o);
                } else {
                    *reinterpret_cast<float4 *>(out + e0) = make_float4(o[0], o[1], o[2], o[3]);
                    *reinterpret_cast<float4 *>(out + e0 + 4) = make_float4(o[4], o[5], o[6], o[7]);
                }
            }
            return;
        }
    }
    for (int e = threadIdx.x; e < PL; e += 256) {
        const int pl = (int)__umulhi((uint32_t)e, a.magicL), ep = e - pl * L;
        const int h = (int)__umulhi((uint32_t)ep, a.magicW), w = ep - h * W;
        const int t_ = pl * L + w * H + h;
        stf<To>(out + e, (ldf<Tin>(P0 + e) + ldf<Tin>(P1 + e)) + (ldf<Tin>(P2 + t_) + ldf<Tin>(P3 + t_)));
    }
}

// ---------------------------------------------------------------------------------------------
// kernels: wave w owns route {0,2,1,3}[w]; LDS (forward):  xN | xT (Tin, PPT planes) | 4 x y planes (fp32)
//                                         LDS (backward): xN | xT | gN | gT (Tin) | dxN | dxT (fp32) | 4 x (accB|accC)
// ---------------------------------------------------------------------------------------------
// BIG: one plane of up to 256 * 5 * 8 = 10240 elements per tile (96 x 96 maps of XFMamba-B at 384^2): the staging
// registers cover 5 (x) / 9 (dy, backward) vectors per thread; the backward keeps the dB/dC sums of all 18 chunk rows
// in registers -- a workgroup of four waves owns the CU there (LDS), so each wave has the whole 512-entry file.
template <typename Tin, typename Tout, int C, bool BIG = false>
__global__ void __launch_bounds__(256) ss2d_fwd_lean_kernel(const LeanArgs a) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = a.L, D = a.D_, PL = a.ppt * L;
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
    constexpr int NVX = BIG ? 5 : (C == 8 ? 2 : 4);    // ppt*L <= 3200 elements (plan), BIG: one plane <= 10240
    constexpr bool kPipe = sizeof(Tin) == 2 && C == 8;  // as in the backward: next tile's planes in flight during the sweeps
    PlaneRegs<Tin, C, NVX> px;
    if (kPipe && !(a.dbg & 2))
        lean_planes_issue<Tin, C, NVX>(a, px, (const Tin *)a.x + ((int64_t)b * D + (int64_t)tg * a.pli * a.ppt) * L, PL);
    for (int it = 0; it < a.pli; ++it) {
        const int d0 = (tg * a.pli + it) * a.ppt;
        const int64_t po = ((int64_t)b * D + d0) * L;
        __syncthreads();
        if (!(a.dbg & 2)) {
            // (not pipelined: still ONE round trip for the whole tile -- the per-plane loop of lean_planes_load is a
            //  chain of ppt dependent HBM latencies)
            if (!kPipe) lean_planes_issue<Tin, C, NVX>(a, px, (const Tin *)a.x + po, PL);
            lean_planes_commit<Tin, Tin, C, NVX>(a, px, xN, xT, PL);
        }
        __syncthreads();
        if (kPipe && !(a.dbg & 2) && it + 1 < a.pli) lean_planes_issue<Tin, C, NVX>(a, px, (const Tin *)a.x + po + PL, PL);
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
        lean_merge_store<Tin, Tout, C>(a, (Tout *)a.y + po, Y, Y + PL, Y + 2 * PL, Y + 3 * PL, PL);   // fixed order
    }
}

template <typename Tin, typename Tout, int C, int NSEG>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((C == 4 && NSEG != 0) ? 2 : 1)))   // 4-element chunks (fp32 I/O, rows of 4k+4): two waves/SIMD
ss2d_bwd_lean_kernel(const LeanArgs a) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = a.L, D = a.D_, PL = a.ppt * L;
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
        for (int j = 0; j < C; ++j) {
            if (NSEG > 2 && s < 16) {
                lean_acc_zero<true>(rB[s][j]);
                lean_acc_zero<true>(rC[s][j]);
            } else {
                rB[s][j] = rC[s][j] = 0.f;
            }
        }
    const bool col = wave >> 1, rev = wave & 1;
    const int k = (wave & 1) * 2 + (wave >> 1);
    const Tin *xq = col ? xT : xN, *gq = col ? gT : gN;
    Tin *dxq = DX + (size_t)wave * PL;
    const int64_t route = (int64_t)b * 4 + k;
    const Tin *Brow = (const Tin *)a.Bs + route * L, *Crow = (const Tin *)a.Cs + route * L;
    using V = typename VecIO<Tin, C>::V;
    const int nseg = NSEG ? NSEG : a.nseg;
    const int n_planes = a.pli * a.ppt;
    constexpr int PD = lean_pd(NSEG);
    LeanStream<Tin, C, PD> st;
#pragma unroll
    for (int q = 0; q < PD; ++q) {
        st.d[q] = st.b[q] = st.c[q] = VecIO<Tin, C>::zero();
        st.h[q] = 0.f;
    }
    {   // the operand stream starts at the last chunk row (in route order) of the first plane of this workgroup
        const int64_t r0 = route * D + (int64_t)tg * a.pli * a.ppt;
        st.dts_row = (const Tin *)a.dts + r0 * L;
        st.chk_row = a.chk + r0 * nseg;
        st.s = nseg - 1;
        st.planes_left = n_planes;
        st.phase = 0;
        asm volatile("v_mov_b32 %0, 0" : "=v"(st.vzero));
        const int ci = rev ? 63 - lane : lane;
        if (rev) {
            lean_stream_issue<Tin, C, true, 0, PD>(st, Brow, Crow, L, nseg, ci);
            if constexpr (PD > 1) lean_stream_issue<Tin, C, true, (PD > 1 ? 1 : 0), PD>(st, Brow, Crow, L, nseg, ci);
            if constexpr (PD > 2) lean_stream_issue<Tin, C, true, (PD > 2 ? 2 : 0), PD>(st, Brow, Crow, L, nseg, ci);
        } else {
            lean_stream_issue<Tin, C, false, 0, PD>(st, Brow, Crow, L, nseg, ci);
            if constexpr (PD > 1) lean_stream_issue<Tin, C, false, (PD > 1 ? 1 : 0), PD>(st, Brow, Crow, L, nseg, ci);
            if constexpr (PD > 2) lean_stream_issue<Tin, C, false, (PD > 2 ? 2 : 0), PD>(st, Brow, Crow, L, nseg, ci);
        }
    }
    // plane staging is software-pipelined: the loads of tile it+1 are in flight while tile it is swept
    constexpr bool BIG = NSEG > 2;                     // 96 x 96: one plane per tile, 18 chunk rows
    constexpr int NVX = BIG ? 5 : (C == 8 ? 2 : 4);    // ppt*L <= 3200 elements (plan): 400 / 800 vectors over 256 threads
    constexpr int NVG = BIG ? 9 : 4;
    // (C == 8 variants only: they are LDS-limited to one or two workgroups per CU, so the extra registers are free and
    //  nothing else hides the staging; the 14x14 variant runs 3 waves per SIMD and loses one to the registers)
    constexpr bool kPipe = sizeof(Tin) == 2 && sizeof(Tout) == 4 && C == 8 && !BIG;   // (BIG: the registers hold the sums)
    PlaneRegs<Tin, C, NVX> px;
    PlaneRegs<Tout, 4, NVG> pg;
    if (kPipe && !(a.dbg & 2)) {
        const int64_t po0 = ((int64_t)b * D + (int64_t)tg * a.pli * a.ppt) * L;
        lean_planes_issue<Tin, C, NVX>(a, px, (const Tin *)a.x + po0, PL);
        lean_planes_issue<Tout, 4, NVG>(a, pg, (const Tout *)a.dy + po0, PL);
    }
    for (int it = 0; it < a.pli; ++it) {
        const int d0 = (tg * a.pli + it) * a.ppt;
        const int64_t po = ((int64_t)b * D + d0) * L;
        __syncthreads();
        if (!(a.dbg & 2)) {
            if (!kPipe) {                               // one round trip for the tile (see the forward kernel)
                lean_planes_issue<Tin, C, NVX>(a, px, (const Tin *)a.x + po, PL);
                lean_planes_issue<Tout, 4, NVG>(a, pg, (const Tout *)a.dy + po, PL);
            }
            lean_planes_commit<Tin, Tin, C, NVX>(a, px, xN, xT, PL);
            lean_planes_commit<Tout, Tin, 4, NVG>(a, pg, gN, gT, PL);
        }
        __syncthreads();
        if (kPipe && !(a.dbg & 2) && it + 1 < a.pli) {
            lean_planes_issue<Tin, C, NVX>(a, px, (const Tin *)a.x + po + PL, PL);
            lean_planes_issue<Tout, 4, NVG>(a, pg, (const Tout *)a.dy + po + PL, PL);
        }
        for (int pl = 0; pl < ((a.dbg & 1) ? 0 : a.ppt); ++pl) {
            const int d = d0 + pl, row = k * D + d;
            const int64_t ro = (route * D + d) * L;
            const float *chk_row = a.chk + (route * D + d) * nseg;
            const float An = a.A[row], Dr = a.D[row], bias = a.softplus == 2 ? 0.f : a.bias[row];
            float dA_acc = 0.f, dD_acc = 0.f, dbias_acc = 0.f;
            if (rev)
                lean_bwd_plane<Tin, C, true, NSEG>(a, (const Tin *)a.dts + ro, (Tin *)a.ddts + ro, Brow, Crow, chk_row, An, Dr,
                                                   bias, xq + pl * L, gq + pl * L, dxq + pl * L, accB, accC, rB, rC, dA_acc,
                                                   dD_acc, dbias_acc, lane, st);
            else
                lean_bwd_plane<Tin, C, false, NSEG>(a, (const Tin *)a.dts + ro, (Tin *)a.ddts + ro, Brow, Crow, chk_row, An, Dr,
                                                    bias, xq + pl * L, gq + pl * L, dxq + pl * L, accB, accC, rB, rC, dA_acc,
                                                    dD_acc, dbias_acc, lane, st);
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
        lean_merge_store<Tin, Tin, C>(a, (Tin *)a.dx + po, DX, DX + PL, DX + 2 * PL, DX + 3 * PL, PL);
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
                    for (int j = 0; j < C; ++j) {
                        float v;
                        if (NSEG > 2 && s < 16) v = pass ? lean_acc_get<true>(rC[s][j]) : lean_acc_get<true>(rB[s][j]);
                        else v = pass ? rC[s][j] : rB[s][j];
                        stage[tp0 + (rev ? C - 1 - j : j)] = v;
                    }
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
