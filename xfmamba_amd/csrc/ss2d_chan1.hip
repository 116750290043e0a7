// ss2d_chan1.hip -- channel-lane SS2D core for d_state 1 (trunk stages 2 / 3: 14 x 14, 12 x 12, 7 x 7 maps), second
// generation of the BACKWARD kernel of ss2d_chan.hip (reference models/fusion_vmamba.py:1145-1174; adjoint
// selective_scan_bwd_kernel.cuh:141-273).
//
// Same decomposition and layout contract as ss2d_chan.hip (a lane owns a channel, the two lane halves are a route and its
// reverse, wave 0 walks the rows, wave 1 the columns of the 32 channel planes of one sample; dt_proj on MFMA inside the
// kernel).  What changed, and why: the first-generation kernel needed 320 registers and 59 KB of LDS per workgroup, i.e. two
// workgroups = ONE wave per SIMD on a CU, and the 768 workgroups of a (64, 384, 14 x 14) launch ran as 512 + 256.  A lane's
// work is a sequential recurrence over the L positions of its route, so the launch time is (waves in sequence on a SIMD) x
// (instructions per wave) x (4 cycles per instruction of a wave that is alone on its SIMD):
//   * LDS is two planes of 32-bit words and nothing else -- x | dy of a position in one word, the row pass's | the column
//     pass's du in the other (50.6 KB at 14 x 14) --: THREE workgroups per CU, every workgroup of the bench launch resident
//     at once, one per-position offset addressing everything.  The ddts rows leave through 2-byte global
//     stores straight from the registers (64 contiguous bytes per lane half, what the 16-byte staged rows also wrote), the
//     dB / dC sums of a step leave as ONE fp32 atomic instruction per 8 positions -- dB in columns 0..15 and dC in columns
//     16..31 of one transposing-MFMA accumulator -- into a dBC laid out in WALKING order (odd routes column-major), so that
//     a step's sums are contiguous for the column pass too; no LDS accumulators, no staging rows, no wave-level syncs;
//   * at most 256 registers (two waves per SIMD where a CU's six waves pair up): the bias vector is folded into the
//     softplus argument (zero C operand instead of a 16-register bias accumulator), per-position values are produced and
//     consumed in one reverse sweep (no sB / sA arrays; with d_state 1 they are one product each);
//   * fewer instructions per position: a.h_{t-1} is E'.h_{t-1} / dh, i.e. the adjoint's dha = E' h_{t-1} needs no a_t h_{t-1}
//     product of its own; softplus without selects (log2(1 + 2^t) clamped from below by t is x for x > 20 in fp32).
#include "ss2d_chan.hpp"

namespace xfm {
namespace chan1 {

// Geometry of this file: ONE row / column per step at every map size (the first generation takes two at 7 x 7), i.e. P = HW,
// NSTEP = HW, no ragged last step; the forward kernel of this file writes one state checkpoint per step.
template <int HW> struct Geom1 {
    static constexpr int L = HW * HW, P = HW, NSTEP = HW;
    static constexpr int Lp = L + 2 - (L & 1);               // bf16 plane pitch: Lp / 2 odd -> conflict-free channel lanes
    static constexpr int Lq = L | 1;                          // 32-bit plane pitch (odd)
    static constexpr int MIDSTEP = (L & 1) ? ((L - 1) / 2) / P : -1;   // step whose two directions touch the same row (odd maps)
    template <bool COL> static __host__ __device__ constexpr int off(int i) { return COL ? i * HW : i; }
    template <bool COL> static __device__ __forceinline__ int base(int st) { return COL ? st : st * P; }
};

// per-lane roles of a pass (COL: routes COL and COL + 2) and the operand fragments of a step, as ChanLane / ChanFrags of
// ss2d_chan.hpp for this geometry and d_state 1 (B and C of a position are k-slots 0 and 1 of ONE fragment)
template <int HW, int KS, bool COL> struct Lane1 {
    using G = Geom1<HW>;
    int c, h, kb, ha, offA, wrow, jB;
    const uint16_t *rowA, *zeros;
    __device__ __forceinline__ Lane1(const ChanArgs &a, int sb, int c0, int lane) {
        c = lane & 31;
        h = kb = lane >> 5;
        const int rho = lane & 31;
        ha = (rho >> 2) & 1;
        const int ia = min(4 * (rho >> 3) + (rho & 3), G::P - 1);
        offA = G::template off<COL>(ia);
        rowA = a.xdbl + (int64_t)sb * G::L * a.XC + ((COL ? 1 : 0) + 2 * ha) * a.C2p;
        wrow = ((COL ? 1 : 0) + 2 * h) * a.D + c0 + c;
        jB = a.Rp8 >> 3;
        zeros = a.zeros;
    }
    __device__ __forceinline__ int natA(int st) const {
        const int nf = G::template base<COL>(st) + offA;
        return ha ? G::L - 1 - nf : nf;
    }
};
template <int KS> struct Frags1 { cbf16x8_t f0[KS], f1[KS], fB; };
template <int HW, int KS, bool COL>
__device__ __forceinline__ void load_frags(const ChanArgs &a, const Lane1<HW, KS, COL> &ln, const int st, Frags1<KS> &f) {
    const uint16_t *ra = ln.rowA + (int64_t)ln.natA(st) * a.XC;
    // rows of half 0 feed k-slots [0, Kp) (forward route), rows of half 1 feed [Kp, 2 Kp); the other k-slots of a row come
    // from a block of zeros (an address select: the loads go straight into the MFMA operands)
    const uint16_t *p0 = ln.ha == 0 ? ra : ln.zeros, *p1 = ln.ha == 0 ? ln.zeros : ra;
#pragma unroll
    for (int m = 0; m < KS; ++m) f.f0[m] = chan_ld8(p0 + 16 * m + 8 * ln.kb);
#pragma unroll
    for (int m = 0; m < KS; ++m) f.f1[m] = chan_ld8(p1 + 16 * m + 8 * ln.kb);
    f.fB = chan_ld8(ra + 8 * ln.jB);
}

// LDS of a workgroup: two planes of 32-bit words, [32 channels][Lq] each (Lq odd: the 32 channel lanes of a half hit 32 banks):
//   XG  word (c, p) = x[c][p] (bf16, low half) | dy[c][p] (bf16, high half): ONE read hands a position's u and g
//   DD  word (c, p) = du of the row pass (low half) | du of the column pass (high half): each wave read-modify-writes its
//       own 16 bits, the epilogue reads both partial sums at once
// -> one per-position byte offset addresses everything (x, dy, both dx planes)
template <int HW> struct Lds {
    using G = Geom1<HW>;
    static constexpr int Lq = G::Lq;
    static constexpr int PW = 32 * Lq * 4;
    static constexpr int XG = 0, DD = PW, DSUM = 2 * PW;
    static constexpr int total = 2 * PW + 32 * 4;
};

// timing-only switches (build with -DXFM_CHAN1_TIMING, then XFM_CHAN1_DBG=<bits>: 2 skip the sweeps, 4 the epilogue, 8 the
// dB / dC atomics, 16 the ddts stores, 32 the merge, 64 the channel sums); the production build compiles them away
__device__ __forceinline__ bool dbg_on(const ChanArgs &a, const int bit) {
#ifdef XFM_CHAN1_TIMING
    return (a.ct & bit) != 0;
#else
    return false;
#endif
}

__device__ __forceinline__ float lds_bf16(const char *p) { return bf16_bits_to_float(*reinterpret_cast<const uint16_t *>(p)); }

// selectors of the transposing channel sum: column j of the accumulator receives value (j & 7) of lane half ((j >> 3) & 1)
// from the FIRST operand (columns 0..15, `hi` = 0) or from the SECOND one (columns 16..31, `hi` = 1)
__device__ __forceinline__ cbf16x8_t selector(const int lane, const int hi) {
    const int col = lane & 31, kb = lane >> 5;
    const bool on = (col >> 4) == hi && ((col >> 3) & 1) == kb;
    return chan_indicator(on ? 0 : 1, on ? (col & 7) : 0);
}

// sum over the 32 channel lanes of each half of 8 + 8 per-lane values (packed bf16 pairs): lane j < 32 returns the total of
//   value (j & 7) of pb of half (j >> 3) & 1 for j < 16,   value (j & 7) of pc of half (j >> 3) & 1 for j >= 16
__device__ __forceinline__ float colsum2(const cu32x4_t pb, const cu32x4_t pc, const cbf16x8_t sel_lo, const cbf16x8_t sel_hi) {
    const cf32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    cf32x16_t t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const cbf16x8_t *>(&pb), sel_lo, zero16, 0, 0, 0);
    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const cbf16x8_t *>(&pc), sel_hi, t, 0, 0, 0);
    float s = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    s += ((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15]));
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const uint32_t sbits = __float_as_uint(s);
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(sbits, sbits, false, false);
    return s + __uint_as_float(r[1]);
}

// merge a step's du of both directions into the pass-private bf16 planes: the first visitor of a position stores, the
// second adds (ss2d_chan.hpp chan_merge, walking the steps downwards), addressed by the step's per-position byte offsets
template <int HW, int NV>
__device__ __forceinline__ void merge16(char *pl, const int (&ad)[NV], const int h, const int st, const float (&v)[NV]) {
    using G = Geom1<HW>;
    constexpr int L = G::L, P = G::P, NSTEP = G::NSTEP;
    auto put = [&](const int i, const float x) {
        *reinterpret_cast<uint16_t *>(pl + ad[i]) = (uint16_t)(pack_bf16x2(x, 0.f) & 0xffffu);
    };
    if (G::MIDSTEP >= 0 && st == G::MIDSTEP) {
        constexpr int SM = (L - 1) / 2 - (G::MIDSTEP < 0 ? 0 : G::MIDSTEP) * P;   // index of the centre inside the step
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (h ? i > SM : i >= SM) put(i, v[i]);
        wave_sync();
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (!(h ? i > SM : i >= SM)) put(i, lds_bf16(pl + ad[i]) + v[i]);
    } else if (2 * st + 1 > NSTEP) {
#pragma unroll
        for (int i = 0; i < NV; ++i) put(i, v[i]);
    } else {
        float o[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) o[i] = lds_bf16(pl + ad[i]);
#pragma unroll
        for (int i = 0; i < NV; ++i) put(i, o[i] + v[i]);
    }
}

// stage the tile's x (bf16) and dy (fp32 -> bf16) into the XG plane: ALL loads of a thread are requested before the first is
// used -- a loop of load / scatter pairs exposed one HBM round trip per 16 bytes (18 of them: 9 of the 54 us of a launch whose
// workgroups all run this phase at the same time)
template <int HW>
__device__ __forceinline__ void stage_tile(uint32_t *xg, const uint16_t *xsrc, const float *gsrc, const int tid) {
    constexpr int L = HW * HW, Lq = Lds<HW>::Lq;
    constexpr int NX = 32 * L / 8;                                  // groups of 8 positions: 16 bytes of x, 32 bytes of dy
    constexpr int KX = (NX + 127) / 128;
    cu32x4_t xr[KX];
    float4 g0[KX], g1[KX];
#pragma unroll
    for (int k = 0; k < KX; ++k)
        if (tid + 128 * k < NX) {
            xr[k] = *reinterpret_cast<const cu32x4_t *>(xsrc + 8 * (tid + 128 * k));
            g0[k] = *reinterpret_cast<const float4 *>(gsrc + 8 * (tid + 128 * k));
            g1[k] = *reinterpret_cast<const float4 *>(gsrc + 8 * (tid + 128 * k) + 4);
        }
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        const int v = tid + 128 * k;
        if (v < NX) {
            const float f[8] = {g0[k].x, g0[k].y, g0[k].z, g0[k].w, g1[k].x, g1[k].y, g1[k].z, g1[k].w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = 8 * v + 2 * q, c = e / L, l = e - c * L;
                const uint32_t gp = pack_bf16x2(f[2 * q], f[2 * q + 1]);
                const uint32_t w0 = (xr[k][q] & 0xffffu) | (gp << 16), w1 = (xr[k][q] >> 16) | (gp & 0xffff0000u);
                xg[c * Lq + l] = w0;
                if constexpr ((L & 1) == 0) {
                    xg[c * Lq + l + 1] = w1;
                } else {
                    const int e1 = e + 1, c1 = e1 / L, l1 = e1 - c1 * L;
                    xg[c1 * Lq + l1] = w1;
                }
            }
        }
    }
}

// The same staging with dy TOKEN-MAJOR (a.ytok: dy (Bt, L, D) fp32, the layout a row LayerNorm / token GEMM hands back): a
// position's 32 channels of this tile are 128 contiguous bytes.  x (plane-major run) fills the low halves of the words, dy the
// high halves, as 2-byte LDS writes; all of a thread's loads are in flight before the first write.
template <int HW>
__device__ __forceinline__ void stage_tile_tok(uint32_t *xg, const uint16_t *xsrc, const float *gsrc, const int D, const bool xtok,
                                               const bool gtok, const int tid) {
    constexpr int L = HW * HW, Lq = Lds<HW>::Lq;
    constexpr int NX = 32 * L / 8, KX = (NX + 127) / 128;          // x: 16-byte pieces (8 positions of a channel, or 8 channels of a position)
    constexpr int NG = L * 8, KG = (NG + 127) / 128;               // dy: 16-byte pieces (4 positions of a channel, or 4 channels of a position)
    cu32x4_t xr[KX];
    float4 g[KG];
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        const int v = tid + 128 * k;
        if (v < NX) xr[k] = *reinterpret_cast<const cu32x4_t *>(xtok ? xsrc + (int64_t)(v >> 2) * D + 8 * (v & 3) : xsrc + 8 * v);
    }
#pragma unroll
    for (int k = 0; k < KG; ++k) {
        const int v = tid + 128 * k;
        if (v < NG) g[k] = *reinterpret_cast<const float4 *>(gtok ? gsrc + (int64_t)(v >> 3) * D + 4 * (v & 7) : gsrc + 4 * v);
    }
    uint16_t *x16 = reinterpret_cast<uint16_t *>(xg);
#pragma unroll
    for (int k = 0; k < KX; ++k) {
        const int v = tid + 128 * k;
        if (v < NX) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int i0, i1;                                          // word index (c * Lq + l) of the two elements of dword q
                if (xtok) {
                    const int l = v >> 2, c = 8 * (v & 3) + 2 * q;
                    i0 = c * Lq + l;
                    i1 = i0 + Lq;
                } else {
                    const int e = 8 * v + 2 * q, c = e / L, l = e - c * L;
                    const int e1 = e + 1, c1 = e1 / L, l1 = e1 - c1 * L;
                    i0 = c * Lq + l;
                    i1 = c1 * Lq + l1;
                }
                x16[2 * i0] = (uint16_t)(xr[k][q] & 0xffffu);
                x16[2 * i1] = (uint16_t)(xr[k][q] >> 16);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < KG; ++k) {
        const int v = tid + 128 * k;
        if (v < NG) {
            const uint32_t p0 = pack_bf16x2(g[k].x, g[k].y), p1 = pack_bf16x2(g[k].z, g[k].w);
            const uint16_t h[4] = {(uint16_t)(p0 & 0xffffu), (uint16_t)(p0 >> 16), (uint16_t)(p1 & 0xffffu), (uint16_t)(p1 >> 16)};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int i;
                if (gtok) {
                    i = (4 * (v & 7) + j) * Lq + (v >> 3);
                } else {
                    const int e = 4 * v + j, c = e / L;
                    i = c * Lq + (e - c * L);
                }
                x16[2 * i + 1] = h[j];
            }
        }
    }
}

template <int HW, int KS, bool COL>
__device__ __forceinline__ void bwd_pass(const ChanArgs &a, const int sb, const int c0, char *sm) {
    using G = Geom1<HW>;
    using LD = Lds<HW>;
    constexpr int L = G::L, P = G::P, NSTEP = G::NSTEP;
    const int lane = threadIdx.x & 63;
    const Lane1<HW, KS, COL> ln(a, sb, c0, lane);
    const int c = ln.c, h = ln.h, kb = ln.kb;
    cbf16x8_t wf[2 * KS];
#pragma unroll
    for (int m = 0; m < 2 * KS; ++m) {
        const int rm = (COL ? 1 : 0) + 2 * (m / KS);
        wf[m] = chan_ld8(chan_w_ptr(a, rm, c0 + c, 16 * (m % KS) + 8 * kb));
    }
    const float A1 = a.A[ln.wrow], A2 = A1 * kLog2e, bvl = a.bias[ln.wrow] * kLog2e;
    float E = 0.f, dAacc = 0.f, dbacc = 0.f;
    // byte offset, inside a [32][Lq] plane of 32-bit words, of sequence element (step nb, index i): pb + s2 * (nb + off(i))
    constexpr int Lq = LD::Lq;
    const int s2 = h ? -4 : 4;
    const int pb = c * Lq * 4 + (h ? 4 * (L - 1) : 0);
    const int route = (COL ? 1 : 0) + 2 * h;
    const float *chk = a.chk + (((int64_t)sb * 4 + route) * NSTEP) * a.D + c0 + c;
    // ddts (Bt, 4, L, D) bf16: byte offset of (route, natural position, channel) = plane offset * D / 2 + g1 (mod 2^32: the
    // host checks that the tensor is smaller than 4 GB)
    const uint32_t g1 = (uint32_t)((((int64_t)sb * 4 + route) * L * a.D + c0 + c) * 2 - (int64_t)c * Lq * 2 * a.D);
    const uint32_t Dh = (uint32_t)a.D >> 1;                         // (plane offsets count 4 bytes per position, ddts rows 2 D)
    char *const ddts = reinterpret_cast<char *>(a.ddts);
    // dB / dC sums of a step: lane j < 32 owns value (j & 7) [+ 8] of direction (j >> 3) & 1, dB for j < 16, dC above;
    // dBC (Bt, 4, 2, L) fp32 in walking order: index of sequence element s of a route = s (forward) or L - 1 - s (reverse)
    const int jt = lane & 7, jdir = (lane >> 3) & 1, jop = (lane >> 4) & 1;
    float *bcp = a.dBC + ((((int64_t)sb * 4 + (COL ? 1 : 0) + 2 * jdir) * 2 + jop) * L) + (jdir ? L - 1 - jt : jt);
    const cbf16x8_t sel_lo = selector(lane, 0), sel_hi = selector(lane, 1);
    // Software pipeline of the vector-memory operations.  They retire in ISSUE order, and the compiler's wait for a load that
    // is still in flight across the loop's back edge is a full drain (s_waitcnt vmcnt(0)): a step that consumed fragments
    // requested by the previous iteration waited for that iteration's ddts stores and dB / dC atomics as well (~3000 cycles
    // with every CU issuing; SQ counters: 43 % of a wave's cycles parked).  So the next step's operands are requested and
    // consumed inside ONE iteration: requested after the adjoint sweep (when its registers are free), in flight under the
    // merge / channel sums, consumed by the dt_proj MFMAs of the NEXT step -- their accumulator is what crosses the back
    // edge, in registers -- and only then come the iteration's stores and atomics.
    Frags1<KS> fr;
    load_frags<HW, KS, COL>(a, ln, NSTEP - 1, fr);
    float hin = NSTEP > 1 ? chk[(int64_t)(NSTEP - 2) * a.D] : 0.f;
    const cf32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto dt_mfma = [&]() {
        cf32x16_t r = zero16;
#pragma unroll
        for (int m = 0; m < KS; ++m) r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.f0[m], wf[m], r, 0, 0, 0);
#pragma unroll
        for (int m = 0; m < KS; ++m) r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.f1[m], wf[KS + m], r, 0, 0, 0);
        return r;
    };
    cf32x16_t acc = dt_mfma();
    cbf16x8_t fbc = fr.fB;                   // B and C of a position: k-slots 0 and 1 of the SAME fragment (no borrowed C here)
    asm volatile("" : "+v"(fbc), "+v"(hin));  // (complete before the loop: a load pending at its head makes the first use of
                                              //  `hin` inside it a full drain of the queue -- behind the previous step's atomics)
#pragma unroll 1
    for (int st = NSTEP - 1; st >= 0; --st) {
        const cf32x16_t bB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbc, chan_indicator(kb, 0), zero16, 0, 0, 0);
        const cf32x16_t bC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbc, chan_indicator(kb, 1), zero16, 0, 0, 0);
        float hin_next = 0.f;
        const int nb = G::template base<COL>(st);
        float *const bc0 = bcp + (jdir ? -st * P : st * P);
        auto body = [&](auto nv_tag) {
            constexpr int NV = decltype(nv_tag)::value;
            int ad[NV];
            uint32_t ug[NV];                                           // x (low half) | dy (high half) of the step's positions
            float dl[NV], sg[NV], av[NV], hv[NV];
            const int bs = pb + s2 * nb;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                ad[i] = bs + s2 * G::template off<COL>(i);
                ug[i] = *reinterpret_cast<const uint32_t *>(sm + LD::XG + ad[i]);
            }
            auto u = [&](const int i) { return __uint_as_float(ug[i] << 16); };
            // raw step size -> softplus and its derivative: t = x log2(e); log2(1 + 2^t) >= t, and equals it in fp32 from
            // t ~ 25 on, so clamping from below by t IS the reference's "x > 20 ? x" (csms6s.py:49-50) without a select;
            // the exponent is clamped so that 2^t stays finite (sig = z / (1 + z) = 1 there)
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const float t = fmaf(acc[i], kLog2e, bvl);
                const float z = __builtin_amdgcn_exp2f(fminf(t, 126.f));
                const float zp1 = 1.0f + z;
                dl[i] = 0.6931471805599453f * fmaxf(__builtin_amdgcn_logf(zp1), t);
                sg[i] = z * __builtin_amdgcn_rcpf(zp1);
            }
            // states of the step from the state entering it
            float hh = hin;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                av[i] = exp2_fast(dl[i] * A2);
                hh = fmaf(av[i], hh, dl[i] * u(i) * bB[i]);
                hv[i] = hh;
            }
            // adjoint sweep; everything a position yields is formed and PACKED here (bf16 pairs: the ddts values, the per-lane
            // dB / dC terms = operands of the channel-sum MFMAs).  No LDS access inside the sweep.
            uint32_t ddp[(NV + 1) / 2];
            cu32x4_t pB[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}}, pC[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
            float duv[NV];
            float dd_hi = 0.f, dB_hi = 0.f, dC_hi = 0.f;
#pragma unroll
            for (int i = NV - 1; i >= 0; --i) {
                const float g = __uint_as_float(ug[i] & 0xffff0000u), ui = u(i);
                const float dh = fmaf(bC[i], g, E);
                E = av[i] * dh;
                const float dha = E * (i > 0 ? hv[i - 1] : hin);          // dh * a_t h_{t-1}
                const float q = dh * bB[i];
                duv[i] = dl[i] * q;                                        // (D g is added once, at the merge)
                const float dd = fmaf(ui, q, dha * A1) * sg[i];          // d loss / d raw step size
                dbacc += dd;
                dAacc = fmaf(dha, dl[i], dAacc);
                const float dBi = dh * (dl[i] * ui), dCi = g * hv[i];
                if (i & 1) {
                    dd_hi = dd; dB_hi = dBi; dC_hi = dCi;
                } else {
                    const bool pair = i + 1 < NV;
                    ddp[i / 2] = pack_bf16x2(dd, pair ? dd_hi : 0.f);
                    pB[i / 8][(i % 8) / 2] = pack_bf16x2(dBi, pair ? dB_hi : 0.f);
                    pC[i / 8][(i % 8) / 2] = pack_bf16x2(dCi, pair ? dC_hi : 0.f);
                }
            }
            // ---- the order of what follows is the point (see the note above the loop):
            //   request the next step's operands (their step index DEPENDS on the sweep's last value: left alone, the
            //   scheduler hoists the loads to the top of the iteration where they hold 24 registers through the sweeps)
            //   (and the sweep's results are pinned HERE: the compiler otherwise sinks the half of the sweep that only the
            //   stores need -- the sigmoid factors, dd, the packs -- below the MFMAs at the bottom, 100 live registers long)
#pragma unroll
            for (int k2 = 0; k2 < (NV + 1) / 2; ++k2) asm volatile("" : "+v"(ddp[k2]));
            asm volatile("" : "+v"(pB[0]), "+v"(pB[1]), "+v"(pC[0]), "+v"(pC[1]), "+v"(dAacc), "+v"(dbacc));
            int stn = st - 1;
            asm volatile("" : "+s"(stn) : "v"(E));
            if (st > 0) load_frags<HW, KS, COL>(a, ln, stn, fr);
            if (st > 1) hin_next = chk[(int64_t)(stn - 1) * a.D];
            //   LDS / ALU work under the loads: merge du into the pass-private plane, channel sums of dB / dC
            if (!dbg_on(a, 32)) merge16<HW, NV>(sm + LD::DD + (COL ? 2 : 0), ad, h, st, duv);
            float t0 = 0.f, t1 = 0.f;
            if (!dbg_on(a, 64)) {
                t0 = colsum2(pB[0], pC[0], sel_lo, sel_hi);
                if constexpr (NV > 8) t1 = colsum2(pB[1], pC[1], sel_lo, sel_hi);
            }
            //   next step's raw step sizes; the accumulator crosses the back edge in registers
            if (st > 0) {
                acc = dt_mfma();
                fbc = fr.fB;
            }
            //   only now the stores and atomics of this step: their data is made to DEPEND on the loaded operands, so no load
            //   is ever queued behind them
            asm volatile("" : "+v"(ddp[0]) : "v"(fbc), "v"(hin_next));
            // ddts rows: (route, natural position) x 32 channels = 64 contiguous bytes per lane half
            // (the offsets are formed here from the plane offsets, one multiply-add each: as loop-carried values of their
            //  own they cost 14 more registers)
            auto goff = [&](int v) {
                asm volatile("" : "+v"(v));
                return (uint32_t)v * Dh + g1;
            };
            if (!dbg_on(a, 16))
#pragma unroll
            for (int i = 0; i < NV; i += 2) {
                *reinterpret_cast<uint16_t *>(ddts + goff(ad[i])) = (uint16_t)(ddp[i / 2] & 0xffffu);
                if (i + 1 < NV) *reinterpret_cast<uint16_t *>(ddts + goff(ad[i + 1])) = (uint16_t)(ddp[i / 2] >> 16);
            }
            // dB / dC: one atomic instruction per 8 positions
            if (!dbg_on(a, 8)) {
                if (lane < 32 && jt < NV) atomicAdd(bc0, t0);
                if constexpr (NV > 8)
                    if (lane < 32 && 8 + jt < NV) atomicAdd(bc0 + (jdir ? -8 : 8), t1);
            }
        };
        body(std::integral_constant<int, P>{});
        hin = hin_next;
    }
    atomicAdd(a.dA + ln.wrow, dAacc);
    atomicAdd(a.dbias + ln.wrow, dbacc);
}

// one workgroup = one sample x 32 channels: wave 0 the rows (routes 0, 2), wave 1 the columns (routes 1, 3).
// (128, 2): at most 256 registers -- three workgroups per CU put two waves on two of its SIMDs
template <int HW, int KS>
__global__ void __launch_bounds__(128, 2) bwd_kernel(const ChanArgs a) {
    using G = Geom1<HW>;
    using LD = Lds<HW>;
    constexpr int L = G::L, Lq = LD::Lq;
    extern __shared__ float smem[];
    char *sm = reinterpret_cast<char *>(smem);
    uint32_t *xg = reinterpret_cast<uint32_t *>(sm + LD::XG);
    const uint32_t *dd = reinterpret_cast<const uint32_t *>(sm + LD::DD);
    float *dsum = reinterpret_cast<float *>(sm + LD::DSUM);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int sb, t;
    chan_block_map(a.xmap, a.D / 32, sb, t);
    const int c0 = 32 * t;
    if (a.ytok || a.xtok)
        stage_tile_tok<HW>(xg, a.xtok ? a.x + (int64_t)sb * L * a.D + c0 : a.x + ((int64_t)sb * a.D + c0) * L,
                           a.ytok ? a.dy + (int64_t)sb * L * a.D + c0 : a.dy + ((int64_t)sb * a.D + c0) * L, a.D, a.xtok != 0,
                           a.ytok != 0, threadIdx.x);
    else stage_tile<HW>(xg, a.x + ((int64_t)sb * a.D + c0) * L, a.dy + ((int64_t)sb * a.D + c0) * L, threadIdx.x);
    if (threadIdx.x < 32) {
        const int q = threadIdx.x;
        dsum[q] = (a.Dp[c0 + q] + a.Dp[a.D + c0 + q]) + (a.Dp[2 * a.D + c0 + q] + a.Dp[3 * a.D + c0 + q]);
    }
    __syncthreads();
    if (!dbg_on(a, 2)) {
        if (wave == 0) bwd_pass<HW, KS, false>(a, sb, c0, sm);
        else bwd_pass<HW, KS, true>(a, sb, c0, sm);
    }
    __syncthreads();
    if (dbg_on(a, 4)) return;
    // ---- dx = rows + columns + (sum_k D_k) g ; dD_k[c] += sum_l g u (the same for every route k)
    uint16_t *dst = a.dx + ((int64_t)sb * a.D + c0) * L;
    auto lo = [](const uint32_t w) { return __uint_as_float(w << 16); };
    auto hi = [](const uint32_t w) { return __uint_as_float(w & 0xffff0000u); };
    if (a.xtok) {
        // dx TOKEN-MAJOR (Bt, L, D) bf16: four channels of a position per thread and trip, one 8-byte store (64 contiguous bytes
        // per position and tile)
        uint16_t *tok = a.dx + (int64_t)sb * L * a.D + c0;
        for (int v = threadIdx.x; v < L * 8; v += 128) {
            const int l = v >> 3, c4 = 4 * (v & 7);
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int off = (c4 + j) * Lq + l;
                o[j] = fmaf(dsum[c4 + j], hi(xg[off]), lo(dd[off]) + hi(dd[off]));
            }
            typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
            const u32x2_t pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            *reinterpret_cast<u32x2_t *>(tok + (int64_t)l * a.D + c4) = pk;
        }
    } else if constexpr (L % 4 == 0) {
        // four positions of one channel per thread and trip, one 8-byte store
        for (int v = threadIdx.x; v < 32 * L / 4; v += 128) {
            const int e = 4 * v, c = e / L, l = e - c * L, off = c * Lq + l;
            const float ds = dsum[c];
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaf(ds, hi(xg[off + j]), lo(dd[off + j]) + hi(dd[off + j]));
            typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
            const u32x2_t pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            *reinterpret_cast<u32x2_t *>(dst + 4 * v) = pk;
        }
    } else {
        for (int v = threadIdx.x; v < 32 * L / 2; v += 128) {
            float o[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = 2 * v + q;
                const int c = e / L, l = e - c * L, off = c * Lq + l;
                o[q] = fmaf(dsum[c], hi(xg[off]), lo(dd[off]) + hi(dd[off]));
            }
            *reinterpret_cast<uint32_t *>(dst + 2 * v) = pack_bf16x2(o[0], o[1]);
        }
    }
    {
        const int c = threadIdx.x >> 2, part = threadIdx.x & 3;      // four lanes per channel split the plane
        // x g of a word = half the dot product of (x, g) with (g, x): one rotate, one v_dot2c_f32_bf16 per position
        typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
        float s = 0.f;
        for (int l = part; l < L; l += 4) {
            const uint32_t w = xg[c * Lq + l], r = (w >> 16) | (w << 16);
            s = __builtin_amdgcn_fdot2_f32_bf16(*reinterpret_cast<const b2_t *>(&w), *reinterpret_cast<const b2_t *>(&r), s, false);
        }
        s *= 0.5f;
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        if (part == 0)
            for (int k = 0; k < 4; ++k) atomicAdd(a.dD + k * a.D + c0 + c, s);
    }
}

template <int HW, int KS> static int launch_bwd(const ChanArgs &a, hipStream_t s) {
    static const int pad = [] { const char *e = getenv("XFM_CHAN1_LDSPAD"); return e ? atoi(e) : 0; }();   // occupancy experiments
    const size_t lds = Lds<HW>::total + pad;
    const void *fn = (const void *)bwd_kernel<HW, KS>;
    static LdsOptIn opted;                                          // (per template instantiation: once per kernel and device)
    if (lds > 64 * 1024 && !lds_opt_in(opted, fn, lds)) return XFM_ELAUNCH;
    ChanArgs args = a;
    static const int dbg = [] { const char *e = getenv("XFM_CHAN1_DBG"); return e ? atoi(e) : 0; }();
    args.ct = dbg & ~1;
    void *kargs[] = {&args};
    const hipError_t e = hipLaunchKernel(fn, dim3((unsigned)(a.Bt * (a.D / 32))), dim3(128), kargs, lds, s);
    if (e != hipSuccess) {
        set_last_hip_error(e);
        return XFM_ELAUNCH;
    }
    return check_launch();
}

template <int HW> static int dispatch_bwd(const ChanArgs &a, hipStream_t s) {
    switch (a.Kp / 16) {
        case 1: return launch_bwd<HW, 1>(a, s);
        case 2: return launch_bwd<HW, 2>(a, s);
        case 3: return launch_bwd<HW, 3>(a, s);
        case 4: return launch_bwd<HW, 4>(a, s);
    }
    return XFM_ELIMIT;
}

// =====================================================================================================================
// forward (same decomposition; one state checkpoint per step = per row / column: what bwd_pass above reads)
// =====================================================================================================================
// LDS: x [32][Lp] bf16 | pass-private y planes with 4-byte position stride: YT = bf16 (14 x 14: LDS capacity decides) ONE
// plane of words, rows' partial sums in the low half, columns' in the high half; YT = float: two fp32 planes
template <int HW, typename YT> struct FwdLds {
    using G = Geom1<HW>;
    static constexpr bool F32 = sizeof(YT) == 4;
    static constexpr int XS = 0;
    static constexpr int YR = (32 * G::Lp * 2 + 15) / 16 * 16;
    static constexpr int YC = F32 ? YR + 32 * G::Lq * 4 : YR + 2;
    static constexpr int DSUM = YR + (F32 ? 2 : 1) * 32 * G::Lq * 4;
    static constexpr int total = DSUM + 32 * 4;
};

template <typename YT> __device__ __forceinline__ float y_ld(const char *p);
template <> __device__ __forceinline__ float y_ld<float>(const char *p) { return *reinterpret_cast<const float *>(p); }
template <> __device__ __forceinline__ float y_ld<uint16_t>(const char *p) { return lds_bf16(p); }
template <typename YT> __device__ __forceinline__ void y_st(char *p, float v);
template <> __device__ __forceinline__ void y_st<float>(char *p, float v) { *reinterpret_cast<float *>(p) = v; }
template <> __device__ __forceinline__ void y_st<uint16_t>(char *p, float v) {
    *reinterpret_cast<uint16_t *>(p) = (uint16_t)(pack_bf16x2(v, 0.f) & 0xffffu);
}

template <int HW, int KS, bool COL, typename YT>
__device__ __forceinline__ void fwd_pass(const ChanArgs &a, const int sb, const int c0, char *sm) {
    using G = Geom1<HW>;
    using LD = FwdLds<HW, YT>;
    constexpr int L = G::L, P = G::P, NSTEP = G::NSTEP, Lp = G::Lp, Lq = G::Lq;
    const int lane = threadIdx.x & 63;
    const Lane1<HW, KS, COL> ln(a, sb, c0, lane);
    const int c = ln.c, h = ln.h, kb = ln.kb;
    cbf16x8_t wf[2 * KS];
#pragma unroll
    for (int m = 0; m < 2 * KS; ++m) {
        const int rm = (COL ? 1 : 0) + 2 * (m / KS);
        wf[m] = chan_ld8(chan_w_ptr(a, rm, c0 + c, 16 * (m % KS) + 8 * kb));
    }
    const float A2 = a.A[ln.wrow] * kLog2e, bvl = a.bias[ln.wrow] * kLog2e;
    // byte offset of sequence element (step nb, index i) in the bf16 x plane: pb + s2 (nb + off(i)); the same element of a y
    // plane (4-byte stride, pitch Lq) sits at twice that plus a per-lane constant
    const int s2 = h ? -2 : 2;
    const int pb = c * Lp * 2 + (h ? 2 * (L - 1) : 0);
    const int kfix = 4 * c * (Lq - Lp);
    char *const yplane = sm + (COL ? LD::YC : LD::YR);
    float *chk = a.chk + (((int64_t)sb * 4 + (COL ? 1 : 0) + 2 * h) * NSTEP) * a.D + c0 + c;
    Frags1<KS> fr;
    load_frags<HW, KS, COL>(a, ln, 0, fr);
    const cf32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto dt_mfma = [&]() {
        cf32x16_t r = zero16;
#pragma unroll
        for (int m = 0; m < KS; ++m) r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.f0[m], wf[m], r, 0, 0, 0);
#pragma unroll
        for (int m = 0; m < KS; ++m) r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr.f1[m], wf[KS + m], r, 0, 0, 0);
        return r;
    };
    cf32x16_t acc = dt_mfma();
    cbf16x8_t fbc = fr.fB;
    asm volatile("" : "+v"(fbc));                  // (nothing loaded is in flight at the loop head: see bwd_pass)
    float hh = 0.f;
#pragma unroll 1
    for (int st = 0; st < NSTEP; ++st) {
        const cf32x16_t bB = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbc, chan_indicator(kb, 0), zero16, 0, 0, 0);
        const cf32x16_t bC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbc, chan_indicator(kb, 1), zero16, 0, 0, 0);
        const int bs = pb + s2 * G::template base<COL>(st);
        int ad[P];
        float u[P], yv[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            ad[i] = bs + s2 * G::template off<COL>(i);
            u[i] = lds_bf16(sm + LD::XS + ad[i]);
        }
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const float t = fmaf(acc[i], kLog2e, bvl);
            const float z = __builtin_amdgcn_exp2f(fminf(t, 126.f));
            const float dl = 0.6931471805599453f * fmaxf(__builtin_amdgcn_logf(1.0f + z), t);
            const float av = exp2_fast(dl * A2);
            hh = fmaf(av, hh, dl * u[i] * bB[i]);
            yv[i] = bC[i] * hh;
        }
        // next step's operands: requested after the sweep, consumed by the MFMAs that close the iteration (bwd_pass)
        int stn = st + 1;
        asm volatile("" : "+s"(stn) : "v"(hh));
        if (st + 1 < NSTEP) load_frags<HW, KS, COL>(a, ln, stn, fr);
        // merge into the pass-private plane: the first visitor of a position stores, the second adds
        {
            auto yad = [&](const int i) { return yplane + 2 * ad[i] + kfix; };
            if (G::MIDSTEP >= 0 && st == G::MIDSTEP) {
                constexpr int SM = (L - 1) / 2 - (G::MIDSTEP < 0 ? 0 : G::MIDSTEP) * P;
#pragma unroll
                for (int i = 0; i < P; ++i)
                    if (h ? i < SM : i <= SM) y_st<YT>(yad(i), yv[i]);
                wave_sync();
#pragma unroll
                for (int i = 0; i < P; ++i)
                    if (!(h ? i < SM : i <= SM)) y_st<YT>(yad(i), y_ld<YT>(yad(i)) + yv[i]);
            } else if (2 * st + 1 < NSTEP) {
#pragma unroll
                for (int i = 0; i < P; ++i) y_st<YT>(yad(i), yv[i]);
            } else {
                float o[P];
#pragma unroll
                for (int i = 0; i < P; ++i) o[i] = y_ld<YT>(yad(i));
#pragma unroll
                for (int i = 0; i < P; ++i) y_st<YT>(yad(i), o[i] + yv[i]);
            }
        }
        if (st + 1 < NSTEP) {
            acc = dt_mfma();
            fbc = fr.fB;
        }
        float hs = hh;
        asm volatile("" : "+v"(hs) : "v"(fbc));        // the checkpoint store is issued behind the loads' completion
        chk[(int64_t)st * a.D] = hs;
    }
}

template <int HW, int KS, typename YT>
__global__ void __launch_bounds__(128, 2) fwd_kernel(const ChanArgs a) {
    using G = Geom1<HW>;
    using LD = FwdLds<HW, YT>;
    constexpr int L = G::L, Lp = G::Lp, Lq = G::Lq;
    extern __shared__ float smem[];
    char *sm = reinterpret_cast<char *>(smem);
    uint16_t *xs = reinterpret_cast<uint16_t *>(sm + LD::XS);
    float *dsum = reinterpret_cast<float *>(sm + LD::DSUM);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int sb, t;
    chan_block_map(a.xmap, a.D / 32, sb, t);
    const int c0 = 32 * t;
    if (a.xtok) {
        // x TOKEN-MAJOR (Bt, L, D) bf16: a position's 32 channels of this tile are 64 contiguous bytes = four 16-byte pieces
        constexpr int NV = L * 4, KV = (NV + 127) / 128;
        const uint16_t *xsrc = a.x + (int64_t)sb * L * a.D + c0;
        cu32x4_t xr[KV];
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const int v = threadIdx.x + 128 * k;
            if (v < NV) xr[k] = *reinterpret_cast<const cu32x4_t *>(xsrc + (int64_t)(v >> 2) * a.D + 8 * (v & 3));
        }
#pragma unroll
        for (int k = 0; k < KV; ++k) {
            const int v = threadIdx.x + 128 * k;
            if (v < NV) {
                const int l = v >> 2, c8 = 8 * (v & 3);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    xs[(c8 + 2 * q) * Lp + l] = (uint16_t)(xr[k][q] & 0xffffu);
                    xs[(c8 + 2 * q + 1) * Lp + l] = (uint16_t)(xr[k][q] >> 16);
                }
            }
        }
    } else {   // stage x: all loads of a thread first
        constexpr int NX = 32 * L / 8, KX = (NX + 127) / 128;
        const uint16_t *xsrc = a.x + ((int64_t)sb * a.D + c0) * L;
        cu32x4_t xr[KX];
#pragma unroll
        for (int k = 0; k < KX; ++k)
            if ((int)threadIdx.x + 128 * k < NX) xr[k] = *reinterpret_cast<const cu32x4_t *>(xsrc + 8 * (threadIdx.x + 128 * k));
#pragma unroll
        for (int k = 0; k < KX; ++k) {
            const int v = threadIdx.x + 128 * k;
            if (v < NX) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = 8 * v + 2 * q, c = e / L, l = e - c * L;
                    if constexpr ((L & 1) == 0) {
                        *reinterpret_cast<uint32_t *>(xs + c * Lp + l) = xr[k][q];
                    } else {
                        xs[c * Lp + l] = (uint16_t)(xr[k][q] & 0xffffu);
                        const int e1 = e + 1, c1 = e1 / L, l1 = e1 - c1 * L;
                        xs[c1 * Lp + l1] = (uint16_t)(xr[k][q] >> 16);
                    }
                }
            }
        }
    }
    if (threadIdx.x < 32) {
        const int q = threadIdx.x;
        dsum[q] = (a.Dp[c0 + q] + a.Dp[a.D + c0 + q]) + (a.Dp[2 * a.D + c0 + q] + a.Dp[3 * a.D + c0 + q]);
    }
    __syncthreads();
    if (wave == 0) fwd_pass<HW, KS, false, YT>(a, sb, c0, sm);
    else fwd_pass<HW, KS, true, YT>(a, sb, c0, sm);
    __syncthreads();
    // y = rows + columns + (sum_k D_k) x: the contiguous run of 32 L floats of this (sample, channel tile)
    float *dst = a.y + ((int64_t)sb * a.D + c0) * L;
    const char *yr = sm + LD::YR, *yc = sm + LD::YC;
    auto yval = [&](const int c, const int l) {
        const int o = (c * Lq + l) * 4;
        return fmaf(dsum[c], bf16_bits_to_float(xs[c * Lp + l]), y_ld<YT>(yr + o) + y_ld<YT>(yc + o));
    };
    if (a.ytok) {
        // y TOKEN-MAJOR (Bt, L, D): a position's 32 channels of this tile as eight 16-byte stores (128 contiguous bytes)
        float *tok = a.y + (int64_t)sb * L * a.D + c0;
        for (int v = threadIdx.x; v < L * 8; v += 128) {
            const int l = v >> 3, c4 = 4 * (v & 7);
            *reinterpret_cast<float4 *>(tok + (int64_t)l * a.D + c4) =
                make_float4(yval(c4, l), yval(c4 + 1, l), yval(c4 + 2, l), yval(c4 + 3, l));
        }
    } else if constexpr (L % 4 == 0) {
        for (int v = threadIdx.x; v < 32 * L / 4; v += 128) {
            const int e = 4 * v, c = e / L, l = e - c * L;
            *reinterpret_cast<float4 *>(dst + e) = make_float4(yval(c, l), yval(c, l + 1), yval(c, l + 2), yval(c, l + 3));
        }
    } else {
        for (int e = threadIdx.x; e < 32 * L; e += 128) {
            const int c = e / L, l = e - c * L;
            dst[e] = yval(c, l);
        }
    }
}

template <int HW, int KS> static int launch_fwd(const ChanArgs &a, hipStream_t s) {
    // fp32 pass-private planes while the LDS allows four workgroups per CU, bf16 beyond (14 x 14): as the first generation
    using YT = typename std::conditional<(HW > 12), uint16_t, float>::type;
    const size_t lds = FwdLds<HW, YT>::total;
    const void *fn = (const void *)fwd_kernel<HW, KS, YT>;
    static LdsOptIn opted;
    if (lds > 64 * 1024 && !lds_opt_in(opted, fn, lds)) return XFM_ELAUNCH;
    ChanArgs args = a;
    args.ct = 0;
    void *kargs[] = {&args};
    const hipError_t e = hipLaunchKernel(fn, dim3((unsigned)(a.Bt * (a.D / 32))), dim3(128), kargs, lds, s);
    if (e != hipSuccess) {
        set_last_hip_error(e);
        return XFM_ELAUNCH;
    }
    return check_launch();
}

template <int HW> static int dispatch_fwd(const ChanArgs &a, hipStream_t s) {
    switch (a.Kp / 16) {
        case 1: return launch_fwd<HW, 1>(a, s);
        case 2: return launch_fwd<HW, 2>(a, s);
        case 3: return launch_fwd<HW, 3>(a, s);
        case 4: return launch_fwd<HW, 4>(a, s);
    }
    return XFM_ELIMIT;
}

}  // namespace chan1

// d_state 1, four routes, no borrowed C operand: the kernels of this file (forward AND backward: they share the checkpoint
// layout -- one state per row / column -- so a shape is served by both or by neither).  XFM_ELIMIT: not covered (the caller
// falls back to ss2d_chan.hip).  XFM_CHAN1=0 switches the file off (A/B runs).
static bool chan1_on() {
    static const bool on = [] {
        const char *e = getenv("XFM_CHAN1");
        return !(e && e[0] == '0');
    }();
    return on;
}
int chan1_covers(int H, int W, int N, int n_routes) {
    return chan1_on() && N == 1 && n_routes == 4 && H == W && (H == 14 || H == 12 || H == 7);
}
// maps whose second-generation kernels run BOTH ways (so y / dy can be token-major: ChanArgs::ytok)
int chan1_ytok(int H, int W, int N, int n_routes) {
    return chan1_covers(H, W, N, n_routes) && (H == 14 || H == 7);
}
int chan1_run(const ChanArgs &a, int HW, bool bwd, hipStream_t s) {
    if (!chan1_on() || a.c_mod > 0) return XFM_ELIMIT;
    if ((a.ytok || a.xtok) && HW != 14 && HW != 7) return XFM_ELIMIT;
    if ((int64_t)a.Bt * 4 * HW * HW * a.D * 2 >= ((int64_t)1 << 32)) return XFM_ELIMIT;     // 32-bit ddts offsets
    if (HW == 14) return bwd ? chan1::dispatch_bwd<14>(a, s) : chan1::dispatch_fwd<14>(a, s);
    // (12 x 12: the first-generation forward is faster -- 38.8 vs 42.9 us at XFMamba-B's stage 3 -- and writes the same
    //  checkpoints: one row / column per step at that size in both generations)
    if (HW == 12) return bwd ? chan1::dispatch_bwd<12>(a, s) : XFM_ELIMIT;
    if (HW == 7) return bwd ? chan1::dispatch_bwd<7>(a, s) : chan1::dispatch_fwd<7>(a, s);
    return XFM_ELIMIT;
}

}  // namespace xfm
