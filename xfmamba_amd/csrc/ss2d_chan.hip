// ss2d_chan.hip -- "channel-lane" fused SS2D core for SHORT maps (5x5 ... 14x14): x_proj output -> dt_proj (MFMA,
// in-kernel) -> softplus -> 4-route selective scan -> cross-merge in ONE kernel, forward and backward.
//
// Replaces, for maps of at most 14 x 14, the chain  dt_proj kernel -> xfm_ss2d_fwd/_bwd  (trunk stages 2/3) and the
// operator chain cross_scan -> einsum x2 -> selective_scan -> cross_merge of the 7x7 fusion blocks (reference
// models/fusion_vmamba.py:1145-1174, :483-576, :808-843).  The (B,4,D,L) step-size tensor `dts` never exists.
//
// Work decomposition (the opposite of ss2d_lean.hpp, which puts lanes along the sequence):
//   * a LANE owns one CHANNEL; the recurrence h_t = a_t h_{t-1} + b_t runs sequentially in the lane -- one FMA per
//     element, no cross-lane scan, no chunk fold/replay, every lane busy whatever the map size;
//   * a WAVEFRONT owns the 32 channel planes of one sample (x as bf16 and the merged output as fp32 in its own LDS
//     region) and walks all four routes in two passes: rows (routes 0 and 2), then columns (routes 1 and 3).  The two
//     lane halves are a route and its reverse, so every plane has ONE owner: the merge is a plain wave-private
//     read-modify-write -- no atomics (ds_add_f32 measured ~190 cycles per wave instruction on gfx950, 18x an integer
//     add), no workgroup barrier;
//   * the step sizes of a 32-channel x 16-position tile come out of v_mfma_f32_32x32x16_bf16 (rows = positions of the
//     token-major x_proj output, columns = channels of the dt_proj weight, bias rides in as the C operand) directly in
//     that layout: D[row][col] has the column on the lane and 16 rows in its registers, rows 4h..4h+3 of every group
//     of 8 on lane half h.  Rows of half 0 carry the forward route's inputs in k-slots [0, Kp), rows of half 1 the
//     reverse route's in [Kp, 2Kp), against the two routes' weights stacked along k;
//   * B_t and C_t of a position are the same for every channel: an MFMA with an indicator matrix as its B operand
//     broadcasts column R+n of the same x_proj rows to every channel lane (exact: 1.0 x bf16, fp32 accumulate).
// Roofline: HBM (forward 2 + 4 B per (b,d,p) element plus the small x_proj rows; backward 2 + 4 + 2 + 8).
#include <cstdlib>

#include "xfm_common.hpp"

namespace xfm {

typedef __bf16 cbf16x8_t __attribute__((ext_vector_type(8)));
typedef float cf32x16_t __attribute__((ext_vector_type(16)));
typedef uint32_t cu32x4_t __attribute__((ext_vector_type(4)));

struct ChanArgs {
    const uint16_t *x;       // (Bt, D, L) bf16, natural row-major planes
    const uint16_t *xdbl;    // (Bt, L, XC) bf16 token-major x_proj rows; route k owns columns [k*C2p, (k+1)*C2p):
                             //   [0,R) dt_proj input | [Rp8, Rp8+N) B | [Rp8+NB, Rp8+NB+N) C   (NB = 1 if N == 1 else N)
    const uint16_t *wdt;     // (4, D, Rp8) bf16 dt_proj weight (zero columns beyond R when R % 8 != 0)
    const float *A;          // (4*D, N)
    const float *Dp, *bias;  // (4*D)
    float *y;                // (Bt, D, L) fp32
    float *chk;              // (Bt, 4, NSTEP, N, D) fp32 states at step ends (written by fwd, read by bwd)
    const float *dy;         // (Bt, D, L) fp32
    uint16_t *dx;            // (Bt, D, L) bf16
    uint16_t *ddts;          // (Bt, 4, L, D) bf16, NATURAL position order, channel fastest: d loss / d raw step size
    float *dBC;              // (Bt, 4, 2, N, L) fp32 ZEROED: dB (index 0) / dC (index 1), natural position order
    float *dA, *dD, *dbias;  // (4*D, N), (4*D), (4*D) fp32 ZEROED
    int Bt, D, R, C2p, XC, Kp, Rp8;
    int c_mod, c_off;        // c_mod > 0: the C operand of sample sb is read from sample c_off + sb % c_mod
    const uint16_t *zeros;   // >= 2 * Kp zero bf16 (16-byte aligned): k-slots of the other route of a pair
    int ct;                  // consecutive 32-channel tiles walked by one workgroup (amortises the dB / dC flush)
};

// N (d_state) only sets the step length: a step of d_state > 1 works 16x longer per position, and its per-position
// operands (not its state) fill the registers, so it takes ONE row / column where d_state 1 takes two on 5x5 / 7x7 maps
template <int HW, int N = 1> struct ChanGeom {
    static constexpr int L = HW * HW;
    static constexpr int Lp = L + 2 - (L & 1);               // bf16 plane pitch: Lp / 2 odd -> conflict-free channel lanes
    static constexpr int Lq = L | 1;                          // fp32 plane pitch (odd)
    // positions per step (one or two rows / columns), <= 16.  (d_state 16 with ONE row per step and two waves per SIMD
    // measured slower on the deep block -- backward 1159 vs 975 us, forward 264 vs 219 -- so N does not enter here.)
    static constexpr int P = HW <= 8 ? 2 * HW : HW;
    static constexpr int Q = P / HW;
    static constexpr int NSTEP = (L + P - 1) / P;
    static constexpr int TAIL = L - (NSTEP - 1) * P;          // valid positions of the last step
    // step whose forward and reverse halves touch the same rows (odd maps only): merged half by half
    static constexpr int MIDSTEP = (L & 1) ? ((L - 1) / 2) / P : -1;
    // natural position, on the FORWARD route of a pass, of index i of step st: base(st) + off(i)
    template <bool COL> static __host__ __device__ constexpr int off(int i) { return COL ? (i % HW) * HW + i / HW : i; }
    template <bool COL> static __device__ __forceinline__ int base(int st) { return COL ? st * Q : st * P; }
};

__device__ __forceinline__ float bf16_bits_to_float(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }

// softplus (threshold 20, reference models/csms6s.py:49-50) for step sizes that come from bf16 operands: log2(1 + z)
// straight from v_log_f32 (relative error ~6e-8 / z: below 1e-4 for every step size above 1e-3, far inside the bf16
// bound), no series branch.  `sig` = d softplus / d raw = z / (1 + z).
__device__ __forceinline__ float chan_softplus(float x) {
    const float zp1 = 1.0f + __builtin_amdgcn_exp2f(x * kLog2e);
    const float lg = __builtin_amdgcn_logf(zp1) * 0.6931471805599453f;
    return x > 20.f ? x : lg;
}
__device__ __forceinline__ float chan_softplus_sig(float x, float &sig) {
    const float z = __builtin_amdgcn_exp2f(x * kLog2e);
    const float zp1 = 1.0f + z;
    const float lg = __builtin_amdgcn_logf(zp1) * 0.6931471805599453f;
    const bool lin = x > 20.f;
    sig = lin ? 1.0f : z * __builtin_amdgcn_rcpf(zp1);
    return lin ? x : lg;
}

// indicator B operand: k-slot `slot` (0..15) of every column is 1.0, the rest 0 -> the MFMA copies column `slot` of the A
// rows into every channel lane
__device__ __forceinline__ cbf16x8_t chan_indicator(int kb, int slot) {
    cu32x4_t v = {0u, 0u, 0u, 0u};
    const uint32_t one = (slot & 1) ? 0x3F800000u : 0x00003F80u;
    const uint32_t val = (kb == (slot >> 3)) ? one : 0u;
    const int dw = (slot & 7) >> 1;
    v[0] = dw == 0 ? val : 0u;
    v[1] = dw == 1 ? val : 0u;
    v[2] = dw == 2 ? val : 0u;
    v[3] = dw == 3 ? val : 0u;
    return *reinterpret_cast<cbf16x8_t *>(&v);
}

__device__ __forceinline__ cbf16x8_t chan_ld8(const uint16_t *p) {
    const cu32x4_t v = *reinterpret_cast<const cu32x4_t *>(p);
    return *reinterpret_cast<const cbf16x8_t *>(&v);
}
// dt_proj weight fragment of route rm: 8 consecutive k of channel row `ch` from the (4, D, Rp8) weight; k-slots at or beyond
// Rp8 (the contraction is walked in steps of 16) come from the block of zeros -- an address select, not a masked load
template <typename Args>
__device__ __forceinline__ const uint16_t *chan_w_ptr(const Args &a, const int rm, const int ch, const int k0) {
    return k0 < a.Rp8 ? a.wdt + ((int64_t)rm * a.D + ch) * a.Rp8 + k0 : a.zeros;
}

__device__ __forceinline__ cbf16x8_t chan_zero8() {
    const cu32x4_t v = {0u, 0u, 0u, 0u};
    return *reinterpret_cast<const cbf16x8_t *>(&v);
}

// stage the 32 x L bf16 planes of one sample into this wave's LDS region ([c][Lp]); 32*L elements contiguous in HBM
template <int HW, int NT> __device__ __forceinline__ void chan_load_planes(uint16_t *dst, const uint16_t *src, int lane) {
    constexpr int L = HW * HW, Lp = ChanGeom<HW>::Lp;
    constexpr int NV = 32 * L / 8;
    for (int v = lane; v < NV; v += NT) {
        const cu32x4_t r = *reinterpret_cast<const cu32x4_t *>(src + 8 * v);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = 8 * v + 2 * q;
            const int c = e / L, l = e - c * L;
            if constexpr ((L & 1) == 0) {
                *reinterpret_cast<uint32_t *>(dst + c * Lp + l) = r[q];
            } else {
                dst[c * Lp + l] = (uint16_t)(r[q] & 0xffffu);
                const int e1 = e + 1, c1 = e1 / L, l1 = e1 - c1 * L;
                dst[c1 * Lp + l1] = (uint16_t)(r[q] >> 16);
            }
        }
    }
}

// fp32 planes (dy) -> bf16 LDS planes, same layout
template <int HW, int NT> __device__ __forceinline__ void chan_load_planes_f32(uint16_t *dst, const float *src, int lane) {
    constexpr int L = HW * HW, Lp = ChanGeom<HW>::Lp;
    constexpr int NV = 32 * L / 4;
    for (int v = lane; v < NV; v += NT) {
        const float4 r = *reinterpret_cast<const float4 *>(src + 4 * v);
        const float f[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = 4 * v + q;
            const int c = e / L, l = e - c * L;
            dst[c * Lp + l] = (uint16_t)(pack_bf16x2(f[q], 0.f) & 0xffffu);
        }
    }
}

// Per-lane roles shared by the forward and backward passes of one route pair (pass COL: routes COL and COL + 2)
template <int HW, int N, int KS, bool COL> struct ChanLane {
    using G = ChanGeom<HW, N>;
    static constexpr int L = G::L, P = G::P;
    static constexpr int NB = N == 1 ? 1 : N;
    int c, h, kb, ha, offA;
    int wrow;                     // row of this lane's (route, channel) in A / D / bias
    const uint16_t *rowA, *rowC;  // x_proj rows of the A-operand role (own route's columns), sample / C-source sample
    const uint16_t *zeros;
    int jB;
    __device__ __forceinline__ ChanLane(const ChanArgs &a, int sb, int c0, int lane) {
        c = lane & 31;
        h = lane >> 5;
        const int rho = lane & 31;
        kb = lane >> 5;
        ha = (rho >> 2) & 1;
        const int ia = min(4 * (rho >> 3) + (rho & 3), P - 1);
        offA = COL ? (ia % HW) * HW + ia / HW : ia;
        const int ra = (COL ? 1 : 0) + 2 * ha;
        const int sbC = a.c_mod > 0 ? a.c_off + sb % a.c_mod : sb;
        rowA = a.xdbl + (int64_t)sb * L * a.XC + ra * a.C2p;
        rowC = a.xdbl + (int64_t)sbC * L * a.XC + ra * a.C2p;
        wrow = ((COL ? 1 : 0) + 2 * h) * a.D + c0 + c;
        jB = a.Rp8 >> 3;
        zeros = a.zeros;
    }
    // natural position of this lane's A-operand row at step st (clamped into the map for the padding rows)
    __device__ __forceinline__ int natA(int st) const {
        int nf = G::template base<COL>(st) + offA;
        nf = nf > L - 1 ? L - 1 : nf;
        return ha ? L - 1 - nf : nf;
    }
};

// the lane's operand fragments of one step: dt_proj input (own route's k-slots only), B block(s), C block(s)
template <int N, int KS> struct ChanFrags { cbf16x8_t f0[KS], f1[KS], fB, fC; };

template <int HW, int N, int KS, bool COL>
__device__ __forceinline__ void chan_load_frags(const ChanArgs &a, const ChanLane<HW, N, KS, COL> &ln, int st,
                                                ChanFrags<N, KS> &f) {
    constexpr int NB = N == 1 ? 1 : N;
    const int nat = ln.natA(st);
    const uint16_t *ra = ln.rowA + (int64_t)nat * a.XC;
    const uint16_t *rc = ln.rowC + (int64_t)nat * a.XC;
    // rows of half 0 feed k-slots [0, Kp) (forward route), rows of half 1 feed [Kp, 2 Kp) (reverse route); the other
    // k-slots of a row are read from a block of zeros (address select, not data select: the loads go straight into the
    // MFMA operands and stay in flight under the previous step's work)
    const uint16_t *p0 = ln.ha == 0 ? ra : ln.zeros;
    const uint16_t *p1 = ln.ha == 0 ? ln.zeros : ra;
#pragma unroll
    for (int m = 0; m < KS; ++m) f.f0[m] = chan_ld8(p0 + 16 * m + 8 * ln.kb);
#pragma unroll
    for (int m = 0; m < KS; ++m) f.f1[m] = chan_ld8(p1 + 16 * m + 8 * ln.kb);
    if constexpr (N == 1) {
        f.fB = chan_ld8(ra + 8 * ln.jB);               // B at element 0, C at element 1 of this block
        f.fC = chan_ld8(rc + 8 * ln.jB);
    } else {
        f.fB = chan_ld8(ra + 8 * ln.jB + 8 * ln.kb);   // 16 states: two blocks
        f.fC = chan_ld8(rc + 8 * ln.jB + NB + 8 * ln.kb);
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------
// YT: element type of the pass-private output planes (float, or bf16 bits where LDS capacity decides: 14 x 14)
// Sum per-lane values over the 32 channel lanes of each half WITHOUT an LDS round trip: the values (bf16) are the A
// operand of an MFMA against a selector matrix, which hands lane `col` the 32 channels' copies of value `col` as its 16
// accumulator registers (rows 8q + 4g + i on lane half g); an in-lane sum, one cross-half add, and lanes 0..15 hold the
// totals: lanes 0..7 value j of half 0 (the forward route), lanes 8..15 value j of half 1 (the reverse route).
__device__ __forceinline__ cbf16x8_t chan_selector(int lane) {
    const int col = lane & 31, kb = lane >> 5;
    return chan_indicator((col < 16 && (col >> 3) == kb) ? 0 : 1, (col < 16 && (col >> 3) == kb) ? (col & 7) : 0);
}
__device__ __forceinline__ float chan_colsum8(const float (&v)[8], const cbf16x8_t sel) {
    cu32x4_t pk;
#pragma unroll
    for (int j = 0; j < 4; ++j) pk[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
    const cf32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const cf32x16_t t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const cbf16x8_t *>(&pk), sel, zero16, 0, 0, 0);
    float s = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    s += ((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15]));
    // add the other lane half's partial (channels 4..7, 12..15, ...): lane L <- s[L] + s[L + 32]
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const uint32_t sb = __float_as_uint(s);
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(sb, sb, false, false);
    return s + __uint_as_float(r[1]);                  // r[1] lanes 0..31 = s of lanes 32..63
}

// dt_proj of one step on MFMA: rows = the step's positions (both directions), columns = this tile's channels.
// d_state 1 keeps the weight fragments and the bias vector in registers for the whole pass and prefetches the next
// step's x_proj rows; d_state > 1 (a step is 16x longer, registers are the scarce resource) loads them per step.
template <int HW, int N, int KS, bool COL>
__device__ __forceinline__ cf32x16_t chan_dt_step(const ChanArgs &a, const ChanLane<HW, N, KS, COL> &ln, const int c0,
                                                  const int st, const float bv, cbf16x8_t &fB, cbf16x8_t &fC) {
    ChanFrags<N, KS> fr;
    chan_load_frags<HW, N, KS, COL>(a, ln, st, fr);
    cf32x16_t acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = bv;
#pragma unroll
    for (int m = 0; m < 2 * KS; ++m) {
        const int rm = (COL ? 1 : 0) + 2 * (m / KS);
        const cbf16x8_t w = chan_ld8(chan_w_ptr(a, rm, c0 + ln.c, 16 * (m % KS) + 8 * ln.kb));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(m < KS ? fr.f0[m] : fr.f1[m - KS], w, acc, 0, 0, 0);
    }
    fB = fr.fB;
    fC = fr.fC;
    return acc;
}

template <typename YT> struct ChanTile;
template <> struct ChanTile<float> {
    template <int HW> static constexpr int pitch() { return ChanGeom<HW>::Lq; }
    static __device__ __forceinline__ float ld(const char *p) { return *reinterpret_cast<const float *>(p); }
    static __device__ __forceinline__ void st(char *p, float v) { *reinterpret_cast<float *>(p) = v; }
};
template <> struct ChanTile<uint16_t> {
    template <int HW> static constexpr int pitch() { return ChanGeom<HW>::Lp; }
    static __device__ __forceinline__ float ld(const char *p) { return bf16_bits_to_float(*reinterpret_cast<const uint16_t *>(p)); }
    static __device__ __forceinline__ void st(char *p, float v) {
        *reinterpret_cast<uint16_t *>(p) = (uint16_t)(pack_bf16x2(v, 0.f) & 0xffffu);
    }
};

// Merge one step's per-position values of both directions into the pass-private planes.  Every position is visited twice
// in a pass, once by each direction.  Walking the steps in order (ASC: 0 .. NSTEP-1, else NSTEP-1 .. 0), both directions
// are the FIRST visitor of their rows in the first half of the walk (plain store: the planes are never zero-filled) and
// the second in the other half (read-modify-write); in the middle step of an odd map, where the two directions meet,
// the visitor with the earlier sequence index (in walk order) stores first and the other adds after a wave-level sync.
template <typename YT, int HW, int N, bool COL, bool ASC, int NV>
__device__ __forceinline__ void chan_merge(char *lds, const int yb, const int sgy, const int h, const int st,
                                           const float (&v)[NV]) {
    using G = ChanGeom<HW, N>;
    using TL = ChanTile<YT>;
    constexpr int L = G::L, P = G::P, NSTEP = G::NSTEP;
    if (G::MIDSTEP >= 0 && st == G::MIDSTEP) {
        constexpr int SM = (L - 1) / 2 - (G::MIDSTEP < 0 ? 0 : G::MIDSTEP) * P;   // index of the centre inside the step
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bool first = ASC ? (h ? i < SM : i <= SM) : (h ? i > SM : i >= SM);
            if (first) TL::st(lds + yb + sgy * G::template off<COL>(i), v[i]);
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bool first = ASC ? (h ? i < SM : i <= SM) : (h ? i > SM : i >= SM);
            char *q = lds + yb + sgy * G::template off<COL>(i);
            if (!first) TL::st(q, TL::ld(q) + v[i]);
        }
    } else if (ASC ? (2 * st + 1 < NSTEP) : (2 * st + 1 > NSTEP)) {
#pragma unroll
        for (int i = 0; i < NV; ++i) TL::st(lds + yb + sgy * G::template off<COL>(i), v[i]);
    } else {
        float o[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) o[i] = TL::ld(lds + yb + sgy * G::template off<COL>(i));
#pragma unroll
        for (int i = 0; i < NV; ++i) TL::st(lds + yb + sgy * G::template off<COL>(i), o[i] + v[i]);
    }
}

template <int HW, int N, int KS, bool COL, typename YT>
__device__ __forceinline__ void chan_fwd_pass(const ChanArgs &a, const int sb, const int c0, const uint16_t *xs, YT *ys,
                                              float *scr) {
    using G = ChanGeom<HW, N>;
    using TL = ChanTile<YT>;
    constexpr int L = G::L, P = G::P, NSTEP = G::NSTEP, Lp = G::Lp, Lq = TL::template pitch<HW>();
    constexpr int YS = (int)sizeof(YT);                                 // bytes per output position
    const int lane = threadIdx.x & 63;
    const ChanLane<HW, N, KS, COL> ln(a, sb, c0, lane);
    const int c = ln.c, h = ln.h, kb = ln.kb;
    cbf16x8_t wf[N == 1 ? 2 * KS : 1];
    if constexpr (N == 1) {
#pragma unroll
        for (int m = 0; m < 2 * KS; ++m) {
            const int rm = (COL ? 1 : 0) + 2 * (m / KS);
            wf[m] = chan_ld8(chan_w_ptr(a, rm, c0 + c, 16 * (m % KS) + 8 * kb));
        }
    }
    const float bv = a.bias[ln.wrow];
    // d_state 1: decay rate and state in registers; d_state > 1: in the wave's LDS scratch ([n][lane]), the loop over
    // the states stays rolled (one state's B / C broadcast in registers at a time)
    float A2[N == 1 ? 1 : 1], hst[N == 1 ? 1 : 1];
    float *hs = scr + lane, *A2s = scr + N * 64 + lane;
    if constexpr (N == 1) {
        A2[0] = a.A[ln.wrow] * kLog2e;
        hst[0] = 0.f;
    } else {
        for (int n = 0; n < N; ++n) {
            hs[n * 64] = 0.f;
            A2s[n * 64] = a.A[(int64_t)ln.wrow * N + n] * kLog2e;
        }
    }
    cf32x16_t biasv;
    if constexpr (N == 1) {
#pragma unroll
        for (int j = 0; j < 16; ++j) biasv[j] = bv;
    }
    // LDS byte addressing relative to the wave's region start (ys is its first array)
    char *lds = reinterpret_cast<char *>(ys);
    const int sg2 = h ? -2 : 2;                                         // bytes per bf16 position step, signed by direction
    const int sgy = h ? -YS : YS;
    const int xbase = (int)((const char *)(xs + c * Lp) - (const char *)ys) + (h ? 2 * (L - 1) : 0);
    const int ybase = c * Lq * YS + (h ? YS * (L - 1) : 0);
    const int route = (COL ? 1 : 0) + 2 * h;
    float *chk = a.chk + (((int64_t)sb * 4 + route) * NSTEP) * N * a.D + c0 + c;
    ChanFrags<N, KS> fr;
    if constexpr (N == 1) chan_load_frags<HW, N, KS, COL>(a, ln, 0, fr);
    const cf32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int st = 0; st < NSTEP; ++st) {
        cf32x16_t acc, bB1, bC1;
        cbf16x8_t fBc, fCc;
        if constexpr (N == 1) {
            acc = biasv;
#pragma unroll
            for (int m = 0; m < KS; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.f0[m], wf[m], acc, 0, 0, 0);
#pragma unroll
            for (int m = 0; m < KS; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.f1[m], wf[KS + m], acc, 0, 0, 0);
            fBc = fr.fB;
            fCc = fr.fC;
            bB1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fBc, chan_indicator(kb, 0), zero16, 0, 0, 0);
            bC1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fCc, chan_indicator(kb, 1), zero16, 0, 0, 0);
            if (st + 1 < NSTEP) chan_load_frags<HW, N, KS, COL>(a, ln, st + 1, fr);   // in flight under this step's work
        } else {
            acc = chan_dt_step<HW, N, KS, COL>(a, ln, c0, st, bv, fBc, fCc);
        }
        const int nb = G::template base<COL>(st);
        auto body = [&](auto nv_tag) {
            constexpr int NV = decltype(nv_tag)::value;
            // element i of this step sits at natural position sg * (nb + off(i)) from the lane's end of the plane:
            // byte offsets = per-step base + loop-invariant signed constants
            float u[NV], yv[NV];
            const int xb = xbase + sg2 * nb, yb = ybase + sgy * nb;
#pragma unroll
            for (int i = 0; i < NV; ++i)
                u[i] = bf16_bits_to_float(*reinterpret_cast<const uint16_t *>(lds + xb + sg2 * G::template off<COL>(i)));
            if constexpr (N == 1) {
                float hh = hst[0];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const float dl = chan_softplus(acc[i]);
                    const float av = exp2_fast(dl * A2[0]);
                    hh = fmaf(av, hh, dl * u[i] * bB1[i]);
                    yv[i] = bC1[i] * hh;
                }
                hst[0] = hh;
            } else {
                float dl[NV], du[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    dl[i] = chan_softplus(acc[i]);
                    du[i] = dl[i] * u[i];
                    yv[i] = 0.f;
                }
                cbf16x8_t ind = chan_indicator(kb, 0);
                cf32x16_t bBn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fBc, ind, zero16, 0, 0, 0);
                cf32x16_t bCn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fCc, ind, zero16, 0, 0, 0);
#pragma unroll 1
                for (int n = 0; n < N; ++n) {
                    const cf32x16_t bB = bBn, bC = bCn;
                    if (n + 1 < N) {                                 // next state's broadcast under this state's work
                        ind = chan_indicator(kb, n + 1);
                        bBn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fBc, ind, zero16, 0, 0, 0);
                        bCn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fCc, ind, zero16, 0, 0, 0);
                    }
                    const float A2n = A2s[n * 64];
                    float hh = hs[n * 64];
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        const float av = exp2_fast(dl[i] * A2n);
                        hh = fmaf(av, hh, du[i] * bB[i]);
                        yv[i] = fmaf(bC[i], hh, yv[i]);
                    }
                    hs[n * 64] = hh;
                    chk[((int64_t)st * N + n) * a.D] = hh;
                }
            }
            chan_merge<YT, HW, N, COL, true, NV>(lds, yb, sgy, h, st, yv);
        };
        if (G::TAIL == P || st + 1 < NSTEP) body(std::integral_constant<int, P>{});
        else body(std::integral_constant<int, G::TAIL>{});
        if constexpr (N == 1) chk[(int64_t)st * a.D] = hst[0];
    }
}

template <int HW, int N, int KS, typename YT>
__global__ void __launch_bounds__(128) ss2dc_fwd_kernel(const ChanArgs a) {
    using G = ChanGeom<HW, N>;
    using TL = ChanTile<YT>;
    constexpr int L = G::L, Lp = G::Lp, Lq = TL::template pitch<HW>();
    extern __shared__ float smem[];
    // wave 0 walks the rows (routes 0, 2), wave 1 the columns (routes 1, 3): one private output plane set each,
    // the x planes shared.  [2][32][Lq] YT | [32][Lp] bf16 | dsum [32] | d_state > 1: per wave [2][N][64] fp32 scratch
    YT *ys = reinterpret_cast<YT *>(smem);
    uint16_t *xs = reinterpret_cast<uint16_t *>(ys + 2 * 32 * Lq);
    float *dsum = reinterpret_cast<float *>(xs + 32 * Lp);
    const int wave = threadIdx.x >> 6;
    float *scr = dsum + 32 + wave * 2 * N * 64;
    const int tiles = a.D / 32, groups = (tiles + a.ct - 1) / a.ct;
    const int sb = blockIdx.x / groups, t0 = (blockIdx.x - sb * groups) * a.ct;
#pragma unroll 1
    for (int t = t0; t < min(tiles, t0 + a.ct); ++t) {
        const int c0 = 32 * t;
        chan_load_planes<HW, 128>(xs, a.x + ((int64_t)sb * a.D + c0) * L, threadIdx.x);
        if (threadIdx.x < 32) {
            const int q = threadIdx.x;
            dsum[q] = (a.Dp[c0 + q] + a.Dp[a.D + c0 + q]) + (a.Dp[2 * a.D + c0 + q] + a.Dp[3 * a.D + c0 + q]);
        }
        __syncthreads();
        // (the pass-private planes are addressed relative to their own start, the x planes from there as well)
        if (wave == 0) chan_fwd_pass<HW, N, KS, false, YT>(a, sb, c0, xs, ys, scr);
        else chan_fwd_pass<HW, N, KS, true, YT>(a, sb, c0, xs, ys + 32 * Lq, scr);
        __syncthreads();
        // y = rows + columns + (sum_k D_k) * x: the contiguous run of 32*L floats of this (sample, channel tile)
        float *dst = a.y + ((int64_t)sb * a.D + c0) * L;
        const char *y0 = reinterpret_cast<const char *>(ys), *y1 = reinterpret_cast<const char *>(ys + 32 * Lq);
        for (int v = threadIdx.x; v < 32 * L / 4; v += 128) {
            float o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = 4 * v + q;
                const int c = e / L, l = e - c * L;
                const int yo = (c * Lq + l) * (int)sizeof(YT);
                o[q] = fmaf(dsum[c], bf16_bits_to_float(xs[c * Lp + l]), TL::ld(y0 + yo) + TL::ld(y1 + yo));
            }
            *reinterpret_cast<float4 *>(dst + 4 * v) = make_float4(o[0], o[1], o[2], o[3]);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------------
// LDS of a workgroup (wave 0: rows, wave 1: columns):
//   dxs [2][32][pitch] YT (pass-private dx planes) | xs [32][Lp] bf16 | gs [32][Lp] bf16 | dsum [32] |
//   per wave: bcacc [2 halves][2][N][L] fp32, ddts staging rows [2][P][32] bf16,
//             d_state > 1: E / dA accumulators [2][N][64] fp32
template <int HW, int N, typename YT> struct ChanBwdLds {
    using G = ChanGeom<HW, N>;
    static constexpr int P = G::P;
    static constexpr int Lq = ChanTile<YT>::template pitch<HW>();
    static constexpr size_t dxs = 0;
    static constexpr size_t xs = (dxs + (size_t)2 * 32 * Lq * sizeof(YT) + 15) / 16 * 16;
    static constexpr size_t gs = xs + (size_t)32 * G::Lp * 2;
    static constexpr size_t dsum = gs + (size_t)32 * G::Lp * 2;
    static constexpr size_t wave0 = (dsum + 32 * 4 + 15) / 16 * 16;
    static constexpr size_t bcacc_sz = (size_t)2 * 2 * N * G::L * 4;
    static constexpr size_t red_off = (bcacc_sz + 15) / 16 * 16;                 // inside a wave's block
    static constexpr size_t scr_off = red_off + ((size_t)2 * P * 32 * 2 + 15) / 16 * 16;   // [2][N][64] fp32 (d_state > 1)
    static constexpr size_t wave_sz = scr_off + (N == 1 ? 0 : (size_t)2 * N * 64 * 4);
    static constexpr size_t total = wave0 + 2 * wave_sz;
};

template <int HW, int N, int KS, bool COL, typename YT>
__device__ __forceinline__ void chan_bwd_pass(const ChanArgs &a, const int sb, const int c0, const uint16_t *xs,
                                              const uint16_t *gs, YT *dxs, float *bcacc, float *red, float *scr) {
    using G = ChanGeom<HW, N>;
    constexpr int L = G::L, P = G::P, NSTEP = G::NSTEP, Lp = G::Lp, Lq = ChanTile<YT>::template pitch<HW>();
    constexpr int YS = (int)sizeof(YT);
    const int lane = threadIdx.x & 63;
    const ChanLane<HW, N, KS, COL> ln(a, sb, c0, lane);
    const int c = ln.c, h = ln.h, kb = ln.kb;
    cbf16x8_t wf[N == 1 ? 2 * KS : 1];
    if constexpr (N == 1) {
#pragma unroll
        for (int m = 0; m < 2 * KS; ++m) {
            const int rm = (COL ? 1 : 0) + 2 * (m / KS);
            wf[m] = chan_ld8(chan_w_ptr(a, rm, c0 + c, 16 * (m % KS) + 8 * kb));
        }
    }
    const float bv = a.bias[ln.wrow];
    // d_state 1: decay rate, adjoint carry E and dA sum in registers; d_state > 1: E / dA in the wave's LDS scratch
    // ([n][lane]) and the decay rates re-read per state, the loop over the states stays rolled
    float An[1], A2[1], E[1], dAacc[1];
    float *Es = scr + lane, *dAs = scr + N * 64 + lane;
    const float *Arow = a.A + (int64_t)ln.wrow * N;
    if constexpr (N == 1) {
        An[0] = Arow[0];
        A2[0] = An[0] * kLog2e;
        E[0] = dAacc[0] = 0.f;
    } else {
        for (int n = 0; n < N; ++n) Es[n * 64] = dAs[n * 64] = 0.f;
    }
    cf32x16_t biasv;
    if constexpr (N == 1) {
#pragma unroll
        for (int j = 0; j < 16; ++j) biasv[j] = bv;
    }
    float dbacc = 0.f;
    char *lds = reinterpret_cast<char *>(dxs);                         // addressing relative to this pass's dx planes
    const int sg2 = h ? -2 : 2;
    const int sgy = h ? -YS : YS;
    const int xbase = (int)((const char *)(xs + c * Lp) - (const char *)dxs) + (h ? 2 * (L - 1) : 0);
    const int gbase = (int)((const char *)(gs + c * Lp) - (const char *)dxs) + (h ? 2 * (L - 1) : 0);
    const int dbase = c * Lq * YS + (h ? YS * (L - 1) : 0);
    const int route = (COL ? 1 : 0) + 2 * h;
    const float *chk = a.chk + (((int64_t)sb * 4 + route) * NSTEP) * N * a.D + c0 + c;
    uint16_t *stg = reinterpret_cast<uint16_t *>(red);    // [2][P][32] bf16 rows of ddts
    const cbf16x8_t sel = chan_selector(lane);
    ChanFrags<N, KS> fr;
    if constexpr (N == 1) chan_load_frags<HW, N, KS, COL>(a, ln, NSTEP - 1, fr);
    const cf32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int st = NSTEP - 1; st >= 0; --st) {
        cf32x16_t acc, bB1, bC1;
        cbf16x8_t fBc, fCc;
        if constexpr (N == 1) {
            acc = biasv;
#pragma unroll
            for (int m = 0; m < KS; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.f0[m], wf[m], acc, 0, 0, 0);
#pragma unroll
            for (int m = 0; m < KS; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.f1[m], wf[KS + m], acc, 0, 0, 0);
            fBc = fr.fB;
            fCc = fr.fC;
            bB1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fBc, chan_indicator(kb, 0), zero16, 0, 0, 0);
            bC1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fCc, chan_indicator(kb, 1), zero16, 0, 0, 0);
            if (st > 0) chan_load_frags<HW, N, KS, COL>(a, ln, st - 1, fr);
        } else {
            acc = chan_dt_step<HW, N, KS, COL>(a, ln, c0, st, bv, fBc, fCc);
        }
        float hin1 = 0.f;
        if constexpr (N == 1) hin1 = st > 0 ? chk[(int64_t)(st - 1) * a.D] : 0.f;
        const int nb = G::template base<COL>(st);
        auto body = [&](auto nv_tag) {
            constexpr int NV = decltype(nv_tag)::value;
            float dl[NV], sg[NV], u[NV], g[NV], sB[NV], sA[NV];
            const int xb = xbase + sg2 * nb, gb = gbase + sg2 * nb, db = dbase + sgy * nb;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                u[i] = bf16_bits_to_float(*reinterpret_cast<const uint16_t *>(lds + xb + sg2 * G::template off<COL>(i)));
                g[i] = bf16_bits_to_float(*reinterpret_cast<const uint16_t *>(lds + gb + sg2 * G::template off<COL>(i)));
            }
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                dl[i] = chan_softplus_sig(acc[i], sg[i]);
                sB[i] = sA[i] = 0.f;
            }
            // one state: forward recompute from the state entering the step, reverse sweep; dB / dC of the state summed
            // over the channel lanes by transposing MFMAs and added to the route's LDS accumulators (natural order)
            auto one_state = [&](const cf32x16_t &bB, const cf32x16_t &bC, const float A2n, const float Ann, const float hin,
                                 float &Ev, float &dAn, const int n) {
                float av[NV], hv_[NV];
                float hh = hin;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    av[i] = exp2_fast(dl[i] * A2n);
                    hh = fmaf(av[i], hh, dl[i] * u[i] * bB[i]);
                    hv_[i] = hh;
                }
                float dBv[16], dCv[16];
#pragma unroll
                for (int i = NV; i < 16; ++i) dBv[i] = dCv[i] = 0.f;
#pragma unroll
                for (int i = NV - 1; i >= 0; --i) {
                    const float dh = fmaf(bC[i], g[i], Ev);
                    Ev = av[i] * dh;
                    const float dlu = dl[i] * u[i];
                    const float dha = dh * (hv_[i] - dlu * bB[i]);           // dh * a_t h_{t-1}
                    sB[i] = fmaf(dh, bB[i], sB[i]);
                    sA[i] = fmaf(dha, Ann, sA[i]);
                    dAn = fmaf(dha, dl[i], dAn);
                    dBv[i] = dh * dlu;                                        // dB_t of this channel
                    dCv[i] = g[i] * hv_[i];                                   // dC_t of this channel
                }
                // lanes 0..15 receive the sums: lane (hh2, j) = value 8 m + j of half hh2
                const int hh2 = (lane >> 3) & 1, jj = lane & 7;
#pragma unroll
                for (int m = 0; m < (NV > 8 ? 2 : 1); ++m)
#pragma unroll
                    for (int op = 0; op < 2; ++op) {
                        float v8[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) v8[j] = op ? dCv[8 * m + j] : dBv[8 * m + j];
                        const float tot = chan_colsum8(v8, sel);
                        const int i = 8 * m + jj;
                        if (lane < 16 && i < NV) {
                            const int nf = nb + (COL ? (i % HW) * HW + i / HW : i);
                            atomicAdd(bcacc + ((hh2 * 2 + op) * N + n) * L + (hh2 ? L - 1 - nf : nf), tot);
                        }
                    }
            };
            if constexpr (N == 1) {
                one_state(bB1, bC1, A2[0], An[0], hin1, E[0], dAacc[0], 0);
            } else {
                cbf16x8_t ind = chan_indicator(kb, 0);
                cf32x16_t bBn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fBc, ind, zero16, 0, 0, 0);
                cf32x16_t bCn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fCc, ind, zero16, 0, 0, 0);
                float hin_n = st > 0 ? chk[(int64_t)(st - 1) * N * a.D] : 0.f;
                float An_n = Arow[0], E_n = Es[0], dA_n = dAs[0];
#pragma unroll 1
                for (int n = 0; n < N; ++n) {
                    const cf32x16_t bB = bBn, bC = bCn;
                    const float hin = hin_n, Ann = An_n;
                    float Ev = E_n, dAn = dA_n;
                    if (n + 1 < N) {                                 // next state's operands under this state's work
                        ind = chan_indicator(kb, n + 1);
                        bBn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fBc, ind, zero16, 0, 0, 0);
                        bCn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fCc, ind, zero16, 0, 0, 0);
                        hin_n = st > 0 ? chk[((int64_t)(st - 1) * N + n + 1) * a.D] : 0.f;
                        An_n = Arow[n + 1];
                        E_n = Es[(n + 1) * 64];
                        dA_n = dAs[(n + 1) * 64];
                    }
                    one_state(bB, bC, Ann * kLog2e, Ann, hin, Ev, dAn, n);
                    Es[n * 64] = Ev;
                    dAs[n * 64] = dAn;
                }
            }
            // ---- per-position results: du of this route into the wave's dx planes, d raw step size to the staging rows
            float duv[NV];
            wave_sync();
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                duv[i] = dl[i] * sB[i];                             // (D g is added once, at the merge)
                const float ddl = fmaf(u[i], sB[i], sA[i]) * sg[i];
                dbacc += ddl;
                stg[(h * P + i) * 32 + c] = (uint16_t)(pack_bf16x2(ddl, 0.f) & 0xffffu);
            }
            chan_merge<YT, HW, N, COL, false, NV>(lds, db, sgy, h, st, duv);
            // ---- ddts rows of this step: [half][position][32 channels] bf16 -> 16-byte stores (4 lanes per position)
            wave_sync();
            {
                constexpr int NCH = 2 * NV * 4;
                for (int q = lane; q < NCH; q += 64) {
                    const int hh2 = q / (NV * 4), r = q - hh2 * NV * 4, i = r >> 2, part = r & 3;
                    const int nf = nb + (COL ? (i % HW) * HW + i / HW : i);
                    const int np = hh2 ? L - 1 - nf : nf;
                    const int rt = (COL ? 1 : 0) + 2 * hh2;
                    const cu32x4_t v = *reinterpret_cast<const cu32x4_t *>(stg + (hh2 * P + i) * 32 + 8 * part);
                    uint16_t *dst = a.ddts + ((((int64_t)sb * 4 + rt) * L + np) * a.D + c0 + 8 * part);
                    *reinterpret_cast<cu32x4_t *>(dst) = v;
                }
            }
            wave_sync();
        };
        if (G::TAIL == P || st + 1 < NSTEP) body(std::integral_constant<int, P>{});
        else body(std::integral_constant<int, G::TAIL>{});
    }
    if constexpr (N == 1) atomicAdd(a.dA + ln.wrow, dAacc[0]);
    else
        for (int n = 0; n < N; ++n) atomicAdd(a.dA + (int64_t)ln.wrow * N + n, dAs[n * 64]);
    atomicAdd(a.dbias + ln.wrow, dbacc);
}

template <int HW, int N, int KS, typename YT>
__global__ void __launch_bounds__(128) ss2dc_bwd_kernel(const ChanArgs a) {
    using G = ChanGeom<HW, N>;
    using LD = ChanBwdLds<HW, N, YT>;
    using TL = ChanTile<YT>;
    constexpr int L = G::L, Lp = G::Lp, Lq = LD::Lq;
    extern __shared__ float smem[];
    char *sm = reinterpret_cast<char *>(smem);
    YT *dxs = reinterpret_cast<YT *>(sm + LD::dxs);
    uint16_t *xs = reinterpret_cast<uint16_t *>(sm + LD::xs);
    uint16_t *gs = reinterpret_cast<uint16_t *>(sm + LD::gs);
    float *dsum = reinterpret_cast<float *>(sm + LD::dsum);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *bcacc = reinterpret_cast<float *>(sm + LD::wave0 + wave * LD::wave_sz);
    float *red = reinterpret_cast<float *>(sm + LD::wave0 + wave * LD::wave_sz + LD::red_off);
    float *scr = reinterpret_cast<float *>(sm + LD::wave0 + wave * LD::wave_sz + LD::scr_off);
    const int tiles = a.D / 32, groups = (tiles + a.ct - 1) / a.ct;
    const int sb = blockIdx.x / groups, t0 = (blockIdx.x - sb * groups) * a.ct;
    for (int e = lane; e < 2 * 2 * N * L; e += 64) bcacc[e] = 0.f;     // dB / dC of this wave's two routes, all tiles
#pragma unroll 1
    for (int t = t0; t < min(tiles, t0 + a.ct); ++t) {
        const int c0 = 32 * t;
        chan_load_planes<HW, 128>(xs, a.x + ((int64_t)sb * a.D + c0) * L, threadIdx.x);
        chan_load_planes_f32<HW, 128>(gs, a.dy + ((int64_t)sb * a.D + c0) * L, threadIdx.x);
        if (threadIdx.x < 32) {
            const int q = threadIdx.x;
            dsum[q] = (a.Dp[c0 + q] + a.Dp[a.D + c0 + q]) + (a.Dp[2 * a.D + c0 + q] + a.Dp[3 * a.D + c0 + q]);
        }
        __syncthreads();
        if (wave == 0) chan_bwd_pass<HW, N, KS, false, YT>(a, sb, c0, xs, gs, dxs, bcacc, red, scr);
        else chan_bwd_pass<HW, N, KS, true, YT>(a, sb, c0, xs, gs, dxs + 32 * Lq, bcacc, red, scr);
        __syncthreads();
        // ---- dx = rows + columns + (sum_k D_k) g ; dD_k[c] += sum_l g u (the same for every route k)
        uint16_t *dst = a.dx + ((int64_t)sb * a.D + c0) * L;
        const char *d0 = reinterpret_cast<const char *>(dxs), *d1 = reinterpret_cast<const char *>(dxs + 32 * Lq);
        for (int v = threadIdx.x; v < 32 * L / 2; v += 128) {
            float o[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = 2 * v + q;
                const int c = e / L, l = e - c * L;
                const int off = (c * Lq + l) * (int)sizeof(YT);
                o[q] = fmaf(dsum[c], bf16_bits_to_float(gs[c * Lp + l]), TL::ld(d0 + off) + TL::ld(d1 + off));
            }
            *reinterpret_cast<uint32_t *>(dst + 2 * v) = pack_bf16x2(o[0], o[1]);
        }
        {
            const int c = threadIdx.x >> 2, part = threadIdx.x & 3;   // four lanes per channel split the plane
            float s = 0.f;
            for (int l = part; l < L; l += 4)
                s = fmaf(bf16_bits_to_float(gs[c * Lp + l]), bf16_bits_to_float(xs[c * Lp + l]), s);
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            if (part == 0)
                for (int k = 0; k < 4; ++k) atomicAdd(a.dD + k * a.D + c0 + c, s);
        }
        __syncthreads();
    }
    {   // dB / dC of this wave's two routes over all its tiles: contiguous fp32 atomics (natural position order)
        const int sbC = a.c_mod > 0 ? a.c_off + sb % a.c_mod : sb;      // dC of a borrowed C goes to its owner
        for (int hs = 0; hs < 2; ++hs)
            for (int op = 0; op < 2; ++op) {
                const int rt = wave + 2 * hs;
                float *dst = a.dBC + ((((int64_t)(op ? sbC : sb) * 4 + rt) * 2 + op) * N) * L;
                const float *src = bcacc + ((size_t)hs * 2 + op) * N * L;
                for (int e = lane; e < N * L; e += 64) atomicAdd(dst + e, src[e]);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
template <int HW, int N, int KS> static int chan_launch(const ChanArgs &a, bool bwd, hipStream_t s) {
    using G = ChanGeom<HW, N>;
    // forward: fp32 pass-private planes while four workgroups (8 waves) fit a CU, bf16 planes beyond (14 x 14)
    using YT = typename std::conditional<(HW > 12), uint16_t, float>::type;
    const size_t lds_f = (size_t)2 * 32 * ChanTile<YT>::template pitch<HW>() * sizeof(YT) + (size_t)32 * G::Lp * 2 + 32 * 4 +
                         (N == 1 ? 0 : (size_t)2 * 2 * N * 64 * 4);
    const size_t lds = bwd ? ChanBwdLds<HW, N, YT>::total : lds_f;
    const void *fn = bwd ? (const void *)ss2dc_bwd_kernel<HW, N, KS, YT> : (const void *)ss2dc_fwd_kernel<HW, N, KS, YT>;
    if (lds > 160 * 1024) return XFM_ELIMIT;
    static bool opted[2] = {false, false};                        // (per template instantiation: once per kernel)
    if (lds > 64 * 1024 && !opted[bwd]) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return XFM_ELAUNCH;
        opted[bwd] = true;
    }
    ChanArgs args = a;
    // one 32-channel tile per workgroup measured fastest also for d_state 16 (deep block, 96 x 48 tiles: 999 us against
    // 1256 / 1648 / 2111 us with 2 / 4 / 8 tiles per workgroup: parallelism beats the smaller dB / dC flush)
    args.ct = 1;
    if (const char *e = getenv("XFM_CHAN_CT")) args.ct = atoi(e) > 0 ? atoi(e) : 1;   // tuning hook
    const unsigned grid = (unsigned)(a.Bt * ((a.D / 32 + args.ct - 1) / args.ct));
    void *kargs[] = {&args};
    const hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(128), kargs, lds, s);
    if (e != hipSuccess) {
        set_last_hip_error(e);
        return XFM_ELAUNCH;
    }
    return check_launch();
}

template <int HW, int N> static int chan_dispatch_ks(const ChanArgs &a, bool bwd, hipStream_t s) {
    switch (a.Kp / 16) {
        case 1: return chan_launch<HW, N, 1>(a, bwd, s);
        case 2: return chan_launch<HW, N, 2>(a, bwd, s);
        case 3: return chan_launch<HW, N, 3>(a, bwd, s);
        case 4: return chan_launch<HW, N, 4>(a, bwd, s);
    }
    return XFM_ELIMIT;
}

static int chan_supported(int HW_h, int HW_w, int N, int NR, int D, int R) {
    if (HW_h != HW_w || D % 32 || R < 1 || R > 64) return 0;
    if (N == 1 && NR == 4 && (HW_h == 7 || HW_h == 12 || HW_h == 14)) return 1;
#ifdef XFM_CHAN_N16
    if (N == 16 && NR == 4 && (HW_h == 5 || HW_h == 7 || HW_h == 12)) return 1;
#endif
    return 0;
}

static int chan_run(const xfm_ss2dc_params_t *p, bool bwd, void *stream) {
    if (!p || !p->x || !p->xdbl || !p->wdt || !p->A || !p->D || !p->delta_bias || !p->chk || !p->zeros) return XFM_EINVAL;
    if (!bwd && !p->y) return XFM_EINVAL;
    if (bwd && (!p->dy || !p->dx || !p->ddts || !p->dBC || !p->dA || !p->dD || !p->ddelta_bias)) return XFM_EINVAL;
    if (!chan_supported(p->H, p->W, p->dstate, p->n_routes, p->d_inner, p->dt_rank)) return XFM_ELIMIT;
    ChanArgs a{};
    a.x = (const uint16_t *)p->x; a.xdbl = (const uint16_t *)p->xdbl; a.wdt = (const uint16_t *)p->wdt;
    a.A = p->A; a.Dp = p->D; a.bias = p->delta_bias; a.y = (float *)p->y; a.chk = p->chk;
    a.dy = (const float *)p->dy; a.dx = (uint16_t *)p->dx; a.ddts = (uint16_t *)p->ddts; a.dBC = p->dBC;
    a.dA = p->dA; a.dD = p->dD; a.dbias = p->ddelta_bias;
    a.Bt = p->batch; a.D = p->d_inner; a.R = p->dt_rank;
    a.Rp8 = (p->dt_rank + 7) / 8 * 8;
    const int N = p->dstate;
    a.C2p = a.Rp8 + (N == 1 ? 8 : 2 * N);
    a.XC = p->n_routes * a.C2p;
    a.Kp = (p->dt_rank + 15) / 16 * 16;
    a.c_mod = p->c_mod; a.c_off = p->c_off;
    a.zeros = (const uint16_t *)p->zeros;
    hipStream_t s = (hipStream_t)stream;
    const int HW = p->H;
    if (N == 1) {
        if (HW == 7) return chan_dispatch_ks<7, 1>(a, bwd, s);
        if (HW == 12) return chan_dispatch_ks<12, 1>(a, bwd, s);
        if (HW == 14) return chan_dispatch_ks<14, 1>(a, bwd, s);
    }
#ifdef XFM_CHAN_N16
    else if (p->n_routes == 4) {
        if (HW == 5) return chan_dispatch_ks<5, 16>(a, bwd, s);
        if (HW == 7) return chan_dispatch_ks<7, 16>(a, bwd, s);
        if (HW == 12) return chan_dispatch_ks<12, 16>(a, bwd, s);
    }
#endif
    return XFM_ELIMIT;
}

}  // namespace xfm

extern "C" {
int xfm_ss2dc_supported(int H, int W, int dstate, int n_routes, int d_inner, int dt_rank) {
    return xfm::chan_supported(H, W, dstate, n_routes, d_inner, dt_rank);
}
int xfm_ss2dc_nsteps(int H, int W, int dstate) {
    (void)dstate;
    const int P = H <= 8 ? 2 * H : H;
    return (H * W + P - 1) / P;
}
int xfm_ss2dc_fwd(const xfm_ss2dc_params_t *p, void *stream) { return xfm::chan_run(p, false, stream); }
int xfm_ss2dc_bwd(const xfm_ss2dc_params_t *p, void *stream) { return xfm::chan_run(p, true, stream); }
}
