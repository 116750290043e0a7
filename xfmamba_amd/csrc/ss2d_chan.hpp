// ss2d_chan.hpp -- shared pieces of the channel-lane SS2D kernels (ss2d_chan.hip: generic d_state, deep block;
// ss2d_chan1.hip: the d_state-1 kernels of the trunk): argument block, map geometry, workgroup -> (sample, tile) map,
// operand fragments of the in-kernel dt_proj, channel sums on transposing MFMAs, the first-visitor merge.
#pragma once
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
    uint16_t *chk16;         // d_state 16 on 7 x 7 maps, when the backward is namespace deep's: the states after every ROW /
                             // column as bf16, two states per dword: (Bt, 4, 7, N / 2, D, 2) -- written instead of chk
    const float *dy;         // (Bt, D, L) fp32
    uint16_t *dx;            // (Bt, D, L) bf16
    uint16_t *ddts;          // (Bt, 4, L, D) bf16, NATURAL position order, channel fastest: d loss / d raw step size
    float *dBC;              // (Bt, 4, 2, N, L) fp32 ZEROED: dB (index 0) / dC (index 1), natural position order
    float *dA, *dD, *dbias;  // (4*D, N), (4*D), (4*D) fp32 ZEROED
    int Bt, D, R, C2p, XC, Kp, Rp8;
    int c_mod, c_off;        // c_mod > 0: the C operand of sample sb is read from sample c_off + sb % c_mod
    const uint16_t *zeros;   // >= 2 * Kp zero bf16 (16-byte aligned): k-slots of the other route of a pair
    int ct;                  // consecutive 32-channel tiles walked by one workgroup (amortises the dB / dC flush)
    int xmap;                // 1: samples are dealt to the XCDs (Bt % 8 == 0), see chan_block_map
    int wdiv;                // n_routes == 1: sample sb uses weight set sb / wdiv (include/xfm_hip.h)
    int xtok;                // 1: x and dx are TOKEN-MAJOR (Bt, L, D) bf16 (same kernels as ytok)
    int ytok;                // 1: y and dy are TOKEN-MAJOR (Bt, L, D) fp32 (second-generation d_state-1 kernels at 14 x 14 / 7 x 7)
};

// Workgroup -> (sample, first index inside the sample).  Every workgroup of a sample re-reads that sample's x_proj rows
// (L x 4 C2p bf16: 37 KB at 14 x 14); workgroups go to the 8 XCDs round-robin, so with consecutive ids on one sample all
// XCDs fetch all samples' rows -- and re-fetch them after the streaming planes evicted them: PMC traffic 1.9x (forward) /
// 1.5x (backward) the algorithmic bytes at 14 x 14.  With xmap sample sb lives on XCD sb % 8 (as ss2d_l3.hip does).
__device__ __forceinline__ void chan_block_map(const int xmap, const int per_sample, int &sb, int &idx) {
    const int bid = blockIdx.x;
    if (xmap) {
        const int j = bid >> 3, q = j / per_sample;
        sb = (bid & 7) + 8 * q;
        idx = j - q * per_sample;
    } else {
        sb = bid / per_sample;
        idx = bid - sb * per_sample;
    }
}

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

}  // namespace xfm
