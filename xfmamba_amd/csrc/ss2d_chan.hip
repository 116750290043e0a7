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
#include "ss2d_chan.hpp"

namespace xfm {

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
    uint16_t *chk16 = a.chk16 + ((((int64_t)sb * 4 + route) * 7) * (N / 2) * a.D + c0 + c) * 2;
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
                // (packed fp32 over position pairs for the two products outside the recurrence -- step size x decay rate,
                //  B x step size x input -- was built and LOST: 240 -> 320 us; v_pk_mul_f32 beside the broadcast MFMAs of the next
                //  state costs more than the scalar pair, MI355X_MICROARCH.md "price of one filler beside MFMAs")
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
                    float hh = hs[n * 64], hmid = 0.f;
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        const float av = exp2_fast(dl[i] * A2n);
                        hh = fmaf(av, hh, du[i] * bB[i]);
                        yv[i] = fmaf(bC[i], hh, yv[i]);
                        if (i == HW - 1) hmid = hh;
                    }
                    hs[n * 64] = hh;
                    if (HW == 7 && a.chk16) {
                        // one checkpoint per row / column of the 7 x 7 map (this step covers two, the last one one)
                        uint16_t *q = chk16 + ((int64_t)(2 * st) * (N / 2) + (n >> 1)) * a.D * 2 + (n & 1);
                        *q = (uint16_t)(pack_bf16x2(NV > HW ? hmid : hh, 0.f) & 0xffffu);
                        if (NV > HW) q[(int64_t)(N / 2) * a.D * 2] = (uint16_t)(pack_bf16x2(hh, 0.f) & 0xffffu);
                    } else {
                        chk[((int64_t)st * N + n) * a.D] = hh;
                    }
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
    int sb, t0;
    chan_block_map(a.xmap, groups, sb, t0);
    t0 *= a.ct;
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
                            // one tile per workgroup (the default): every slot is written exactly once -- a plain store;
                            // ds_add_f32 costs ~190 cycles per wave instruction (tools/ubench/lds_atomics.hip)
                            float *slot = bcacc + ((hh2 * 2 + op) * N + n) * L + (hh2 ? L - 1 - nf : nf);
                            if (a.ct == 1) *slot = tot;
                            else atomicAdd(slot, tot);
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
    int sb, t0;
    chan_block_map(a.xmap, groups, sb, t0);
    t0 *= a.ct;
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
                // d_state 1: dBC is kept in WALKING order (odd routes column-major), the contract of ss2d_chan1.hip
                for (int e = lane; e < N * L; e += 64) atomicAdd(dst + ((N == 1 && wave == 1) ? (e % HW) * HW + e / HW : e), src[e]);
            }
    }
}

// =====================================================================================================================
// d_state 16 on 7 x 7 maps -- the deep cross-fusion block (reference models/fusion_vmamba.py:483-576), second design.
//
// The kernels above give a lane one channel and its two halves a route and its reverse; with 16 states that needs the
// B_n / C_n broadcasts of a step in 64 registers (MFMA results, one state ahead), 408 VGPRs in all, and ran the backward at
// 1.5 % of the HBM peak (948 us).  Here a wave owns 64 channels of ONE route at a time:
//   * B_n[t], C_n[t] are the same for all 64 lanes: the wave copies its route's rows once per pass into an fp32 LDS
//     table in route order and a state takes its step with two 16-byte broadcast reads (no MFMA in the dependency
//     chain of a state; as scalar operands they missed the scalar cache twice per state pair);
//   * the states are walked in PAIRS (two independent recurrences in flight), the state / adjoint carries in LDS;
//   * dt_proj of a step: one accumulator, the two 32-channel blocks of the tile stacked along k (rows of lane half h
//     carry their x_proj slots in k-block h, the weights of block h sit in the same slots);
//   * the step checkpoints are bf16 pairs (two states per dword): 132 MB instead of 264 MB each way;
//   * wave 0 walks routes 0 then 2, wave 1 routes 1 then 3: every du plane has one owner, the second route adds.
// =====================================================================================================================
namespace deep {

constexpr int HW = 7, L = 49, P = 7, NSTEP = 7, N = 16, LP = 50;   // LP: bf16 plane pitch (LP / 2 odd: conflict-free lanes)
constexpr int TBL = 2 * N * NSTEP * 8;                           // floats of a wave's B / C table: [state pair][step][8][B0 B1 C0 C1]
typedef float df2 __attribute__((ext_vector_type(2)));           // a state PAIR: the pair's arithmetic is packed fp32 (v_pk_*_f32)
// slot of (B: w = 0 / C: w = 1, state n, sequence position s) in the table
__device__ __forceinline__ int deep_slot(const int w, const int n, const int s) {
    return ((((n >> 1) * NSTEP + s / P) * 8 + s % P) << 2) + 2 * w + (n & 1);
}

struct DeepArgs {
    ChanArgs a;
    uint32_t *chkp;          // (Bt, 4, NSTEP, N / 2, D) packed bf16 pairs: states (2 np, 2 np + 1) after each step
    float *dAt;              // (4, N, D) fp32 scratch, zeroed per backward call: dA with the CHANNEL fastest, so that a wave's
                             // atomic instruction covers 256 contiguous bytes (in the (4 D, N) layout of dA its 64 lanes hit
                             // 64 different lines: ~17x slower per instruction, MI355X_MICROARCH.md "Global float atomics")
    int dbg;                 // timing-only switches (builds with -DXFM_DEEP_TIMING, XFM_DEEP_DBG=<bits>): 1 no pair arithmetic,
                             // 2 no final atomics, 4 no dt_proj step, 8 no B / C table fill, 16 no ddts / du stores, 32 no column sums
};
#ifdef XFM_DEEP_TIMING
#define DEEP_DBG(da, bit) (((da).dbg & (bit)) != 0)
#else
#define DEEP_DBG(da, bit) false
#endif

// natural position of sequence index s (0..48) of route k
__host__ __device__ __forceinline__ int deep_nat(const int k, const int s) {
    const int t = (k & 2) ? L - 1 - s : s;
    return (k & 1) ? (t % HW) * HW + t / HW : t;
}

__device__ __forceinline__ float deep_lo(const uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float deep_hi(const uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// B_n[t] / C_n[t] of route k are the same for every channel: the wave copies them once per pass from the token-major x_proj
// rows into its LDS table, fp32, in ROUTE order with eight slots per step (two aligned 16-byte broadcast reads hand a state
// its step).  (As scalar operands -- s_load from a global table -- they shared lgkmcnt with the LDS reads of the state loop,
// returned out of order and missed the scalar cache: two exposed L2 round trips per state pair, 3 k cycles of 4.)
__device__ __forceinline__ void deep_fill_bc(const ChanArgs &a, const int sb, const int sbC, const int k, float *T,
                                             const int lane) {
    // a position's B (16 states) and C columns are 2 x 32 contiguous bytes of its x_proj row: 49 x 4 pieces of 16 bytes,
    // all of a lane's loads in flight together (a rolled element-wise gather exposed 28 dependent L2 round trips per pass)
    const uint16_t *src = a.xdbl + k * a.C2p + a.Rp8;
    cu32x4_t r[4];
    int slot[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int p = q * 64 + lane, s = p >> 2, part = p & 3, w = part >> 1, nb = 8 * (part & 1);
        slot[q] = -1;
        if (s < L) {
            r[q] = *reinterpret_cast<const cu32x4_t *>(src + ((int64_t)(w ? sbC : sb) * L + deep_nat(k, s)) * a.XC + w * N + nb);
            slot[q] = deep_slot(w, nb, s);                                  // states nb, nb + 1: adjacent floats; + 2 j: pair j on
        }
    }
    for (int e = lane; e < (N / 2) * NSTEP; e += 64)                        // the pad slot of every (pair, step)
        *reinterpret_cast<float4 *>(T + (e * 8 + 7) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (slot[q] >= 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float2 *>(T + slot[q] + j * (NSTEP * 8 * 4)) = make_float2(deep_lo(r[q][j]), deep_hi(r[q][j]));
        }
    }
}

// dt_proj of one step for the 64 channels of the tile: acc[i] = raw step size of position i of the step (i < 7).
// One accumulator for both 32-channel blocks: the x_proj slots of lane half h sit in k-block h (the other block reads
// zeros), the weights of channel block h in the same slots.  The x_proj fragments are requested one step ahead.
template <int KS> struct DeepX { cbf16x8_t f[2 * KS]; };

template <int KS>
__device__ __forceinline__ void deep_load_x(const ChanArgs &a, const int sb, const int k, const int st, const int lane,
                                            DeepX<KS> &x) {
    const int row32 = lane & 31, kb = lane >> 5;
    const int i = 4 * (row32 >> 3) + (row32 & 3), typ = (row32 >> 2) & 1;
    const int nat = deep_nat(k, st * P + (i < P ? i : P - 1));
    const uint16_t *xrow = a.xdbl + ((int64_t)sb * L + nat) * a.XC + k * a.C2p + 8 * kb;
#pragma unroll
    for (int m = 0; m < 2 * KS; ++m) {
        const int ks = m % KS, blk = m / KS;
        x.f[m] = chan_ld8((typ == blk && 16 * ks + 8 * kb < a.Rp8) ? xrow + 16 * ks : a.zeros);
    }
}

// the dt_proj weight fragments of the tile's two 32-channel blocks for route k: the same for every step of a pass, so the
// backward holds them in registers for the whole pass (requested per step they exposed an L2 round trip in front of the MFMAs)
template <int KS> struct DeepW { cbf16x8_t f[2 * KS]; };
template <int KS>
__device__ __forceinline__ void deep_load_w(const ChanArgs &a, const int c0, const int k, const int lane, DeepW<KS> &w) {
    const int row32 = lane & 31, kb = lane >> 5;
    const uint16_t *w0 = a.wdt + ((int64_t)k * a.D + c0 + row32) * a.Rp8 + 8 * kb;
    const uint16_t *w1 = w0 + (int64_t)32 * a.Rp8;
#pragma unroll
    for (int m = 0; m < 2 * KS; ++m) {
        const int ks = m % KS, blk = m / KS;
        w.f[m] = chan_ld8(16 * ks + 8 * kb < a.Rp8 ? (blk ? w1 : w0) + 16 * ks : a.zeros);
    }
}
template <int KS>
__device__ __forceinline__ cf32x16_t deep_dt_step(const float bv, const DeepX<KS> &x, const DeepW<KS> &w) {
    const cbf16x8_t(&fb)[2 * KS] = w.f;
    cf32x16_t acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = bv;
#pragma unroll
    for (int m = 0; m < 2 * KS; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.f[m], fb[m], acc, 0, 0, 0);
    return acc;
}

// sum of eight per-lane values over the 64 channel lanes: the values (bf16) are the A operand of an MFMA whose selector
// hands lane `col` value (col & 7) of 16 channel rows from each lane half; an in-lane sum and one cross-half add.
// Lanes 0..7 (and their copies) return the totals.
__device__ __forceinline__ float deep_colsum8(const float (&v)[8], const cbf16x8_t sel) {
    cu32x4_t pk;
#pragma unroll
    for (int j = 0; j < 4; ++j) pk[j] = pack_bf16x2(v[2 * j], v[2 * j + 1]);
    const cf32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const cf32x16_t t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const cbf16x8_t *>(&pk), sel, zero16, 0, 0, 0);
    float s = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    s += ((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15]));
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const uint32_t sb = __float_as_uint(s);
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(sb, sb, false, false);
    return s + __uint_as_float(r[1]);
}

// the four value sets of a state pair (dB, dC of both states: 4 x 7 per-lane values) in ONE accumulator: set s lands in
// columns 8 s .. 8 s + 7 through its own selector, so the in-lane sum of the 16 accumulator rows and the cross-half add are
// paid once instead of four times.  Lane j < 32 returns the total of value (j & 7) of set (j >> 3).
__device__ __forceinline__ cf32x16_t deep_colsum4_issue(const float (&v0)[8], const float (&v1)[8], const float (&v2)[8],
                                                        const float (&v3)[8], const int lane) {
    const float *vs[4] = {v0, v1, v2, v3};
    const cf32x16_t zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    cf32x16_t t = zero16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        cu32x4_t pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) pk[j] = pack_bf16x2(vs[q][2 * j], vs[q][2 * j + 1]);
        // B operand: k-slot (lane & 7) of BOTH k-blocks (channels c and c + 32 of a row add up) for the columns of set q
        const cbf16x8_t sel = chan_indicator(((lane >> 3) & 3) == q ? 0 : 1, lane & 7);
        t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const cbf16x8_t *>(&pk), sel, t, 0, 0, 0);
    }
    return t;
}
// ... and the fold of its accumulator (the caller puts a state pair's worth of other work between the two)
__device__ __forceinline__ float deep_colsum4_fold(const cf32x16_t &t) {
    float s = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    s += ((t[8] + t[9]) + (t[10] + t[11])) + ((t[12] + t[13]) + (t[14] + t[15]));
    typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
    const uint32_t sb = __float_as_uint(s);
    const u32x2_t r = __builtin_amdgcn_permlane32_swap(sb, sb, false, false);
    return s + __uint_as_float(r[1]);
}

__device__ __forceinline__ uint16_t deep_bf16(const float v) { return (uint16_t)(pack_bf16x2(v, 0.f) & 0xffffu); }

// ---- backward --------------------------------------------------------------------------------------------------------
// LDS of a workgroup: xs | gs [64][LP] bf16 | dsum [64] fp32 | per wave: dup [64][LP] bf16 (du planes), T [2][N][NSTEP][8]
// fp32 (the B / C table; a state's dB / dC sums of a step overwrite the slots it has just read), stg [P][64] bf16,
// Es | dAs | As [N][64] fp32 (adjoint carries, dA sums, decay rates), hb [N / 2][64] packed entering states
struct DeepBwdLds {
    static constexpr size_t xs = 0, gs = xs + 64 * LP * 2, dsum = gs + 64 * LP * 2, wave0 = dsum + 64 * 4;
    static constexpr size_t dup = 0, T = dup + 64 * LP * 2, stg = T + (size_t)TBL * 4;
    static constexpr size_t Es = (stg + P * 64 * 2 + 15) / 16 * 16, dAs = Es + N * 64 * 4, As = dAs + N * 64 * 4;
    static constexpr size_t hb = As + N * 64 * 4, wave_sz = hb + (N / 2) * 64 * 4;
    static constexpr size_t total = wave0 + 2 * wave_sz;
};

// k: the route (its geometry, its columns of the x_proj rows, its slice of ddts / dBC / the checkpoints); nr: routes per
// sample; ws: the weight set (rows of wdt / A / bias / dA) -- route k itself with four routes, sample sb / wdiv with one
template <int KS, bool FIRST>
__device__ __forceinline__ void deep_bwd_pass(const DeepArgs &da, const int sb, const int c0, const int k, const int nr,
                                              const int ws, const uint16_t *xs, const uint16_t *gs, char *wl) {
    const ChanArgs &a = da.a;
    const int lane = threadIdx.x & 63;
    uint16_t *dup = reinterpret_cast<uint16_t *>(wl + DeepBwdLds::dup);
    float *T = reinterpret_cast<float *>(wl + DeepBwdLds::T);
    uint16_t *stg = reinterpret_cast<uint16_t *>(wl + DeepBwdLds::stg);
    float *Es = reinterpret_cast<float *>(wl + DeepBwdLds::Es), *dAs = reinterpret_cast<float *>(wl + DeepBwdLds::dAs);
    float *As = reinterpret_cast<float *>(wl + DeepBwdLds::As);
    uint32_t *hb = reinterpret_cast<uint32_t *>(wl + DeepBwdLds::hb);
    const int64_t wrow = (int64_t)ws * a.D + c0 + lane;
    const float bv = a.bias[wrow];
    const float *Arow = a.A + wrow * N;
    const int sbC = a.c_mod > 0 ? a.c_off + sb % a.c_mod : sb;
    const uint32_t *chk = da.chkp + (((int64_t)sb * nr + k) * NSTEP) * (N / 2) * a.D + c0 + lane;
    DeepX<KS> xf;
    DeepW<KS> wf;
    deep_load_x<KS>(a, sb, k, NSTEP - 1, lane, xf);
    deep_load_w<KS>(a, c0, ws, lane, wf);
    // states entering a step: requested one step ahead (eight loads in flight under a whole step of work: inside the
    // state loop each would expose an HBM round trip), handed over through LDS
    // (TWO steps ahead: these 132 MB are the largest stream of the kernel and a wave is alone on its SIMD -- with one step
    //  of look-ahead the bytes in flight per CU capped the launch at ~1.4 TB/s, 95 us of it with everything else switched off)
    uint32_t hn[N / 2], hn2[N / 2];
#pragma unroll
    for (int np = 0; np < N / 2; ++np) {
        hn[np] = chk[((int64_t)(NSTEP - 2) * (N / 2) + np) * a.D];
        hn2[np] = chk[((int64_t)(NSTEP - 3) * (N / 2) + np) * a.D];
    }
    if (!DEEP_DBG(da, 8)) deep_fill_bc(a, sb, sbC, k, T, lane);
    {
        float av[N];
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4 *>(Arow + 4 * q);
            av[4 * q] = t.x; av[4 * q + 1] = t.y; av[4 * q + 2] = t.z; av[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int n = 0; n < N; ++n) {
            Es[n * 64 + lane] = dAs[n * 64 + lane] = 0.f;
            As[n * 64 + lane] = av[n];
        }
    }
    wave_sync();
    float dbacc = 0.f;
#pragma unroll 1
    for (int st = NSTEP - 1; st >= 0; --st) {
#pragma unroll
        for (int np = 0; np < N / 2; ++np) {
            hb[np * 64 + lane] = st > 0 ? hn[np] : 0u;
            hn[np] = hn2[np];
            if (st > 2) hn2[np] = chk[((int64_t)(st - 3) * (N / 2) + np) * a.D];
        }
        cf32x16_t acc;
        if (!DEEP_DBG(da, 4)) {
            acc = deep_dt_step<KS>(bv, xf, wf);
            if (st > 0) deep_load_x<KS>(a, sb, k, st - 1, lane, xf);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = bv;
        }
        float dl[P], sg[P], u[P], g[P], dlu[P];
        df2 sB2[P], sA2[P];                                          // per-position sums over the states, even | odd states apart
        int nat[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            nat[i] = deep_nat(k, st * P + i);
            dl[i] = chan_softplus_sig(acc[i], sg[i]);
            u[i] = bf16_bits_to_float(xs[lane * LP + nat[i]]);
            g[i] = bf16_bits_to_float(gs[lane * LP + nat[i]]);
            dlu[i] = dl[i] * u[i];
            sB2[i] = sA2[i] = df2{0.f, 0.f};
        }
        // A wave is alone on its SIMD here: every LDS round trip and every MFMA result it waits for is dead time (measured:
        // ~2400 cycles per state pair and step for ~215 instructions).  So the pair loop is software-pipelined by hand: the
        // operands of pair np + 1 (decay rates, entering states, carries, the 7 table rows: 14 LDS reads) are requested
        // before pair np computes, and the column sums of pair np (four dependent MFMAs) are folded and stored only after
        // the arithmetic of pair np + 1.  Two register sets, the loop written out twice (no moves).
        struct PairOps { df2 An, E, dA; uint32_t hp; float4 q[P]; };
        auto load_pair = [&](const int np, PairOps &o) {
            o.An = df2{As[(2 * np) * 64 + lane], As[(2 * np + 1) * 64 + lane]};
            o.hp = hb[np * 64 + lane];
            const float4 *Tq = reinterpret_cast<const float4 *>(T) + (np * NSTEP + st) * 8;
#pragma unroll
            for (int i = 0; i < P; ++i) o.q[i] = Tq[i];              // (B, C) of both states at position i: one broadcast read
            o.E = df2{Es[(2 * np) * 64 + lane], Es[(2 * np + 1) * 64 + lane]};
            o.dA = df2{dAs[(2 * np) * 64 + lane], dAs[(2 * np + 1) * 64 + lane]};
        };
        cf32x16_t pend;                                              // column-sum accumulator of the previous pair
        int pend_np = -1;
        auto fold_pending = [&]() {
            if (pend_np < 0) return;
            const float tot = deep_colsum4_fold(pend);
            if (lane < 32 && (lane & 7) < P)
                T[(((pend_np * NSTEP + st) * 8 + (lane & 7)) << 2) + 2 * ((lane >> 3) & 1) + (lane >> 4)] = tot;
        };
        auto run_pair = [&](const int np, const PairOps &o) {
            // The two states of a pair run as the two halves of packed fp32 instructions (left to the compiler the pair's
            // scalars were packed in 17 of 166 places): 11 packed + 2 transcendental instructions per position and pair.
            const df2 An = o.An, A2 = An * kLog2e;
            df2 E = o.E, dA = o.dA;
            const df2 hin = {deep_lo(o.hp), deep_hi(o.hp)};
            df2 h = hin, av[P], hv[P];
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const df2 t = A2 * dl[i];
                av[i] = df2{exp2_fast(t.x), exp2_fast(t.y)};
                h = __builtin_elementwise_fma(av[i], h, df2{o.q[i].x, o.q[i].y} * dlu[i]);
                hv[i] = h;
            }
            df2 dBp[P], dCp[P];
#pragma unroll
            for (int i = P - 1; i >= 0; --i) {
                const df2 bq = {o.q[i].x, o.q[i].y}, cq = {o.q[i].z, o.q[i].w};
                const df2 dh = __builtin_elementwise_fma(cq, df2{g[i], g[i]}, E);
                E = av[i] * dh;
                const df2 dha = E * (i > 0 ? hv[i - 1] : hin);      // dh * a_t h_{t-1} = (a_t dh) h_{t-1}
                sB2[i] = __builtin_elementwise_fma(dh, bq, sB2[i]);
                sA2[i] = __builtin_elementwise_fma(dha, An, sA2[i]);
                dA = __builtin_elementwise_fma(dha, df2{dl[i], dl[i]}, dA);
                dBp[i] = dh * dlu[i];
                dCp[i] = hv[i] * g[i];
            }
            Es[(2 * np) * 64 + lane] = E.x;
            Es[(2 * np + 1) * 64 + lane] = E.y;
            dAs[(2 * np) * 64 + lane] = dA.x;
            dAs[(2 * np + 1) * 64 + lane] = dA.y;
            float dB0[8], dC0[8], dB1[8], dC1[8];
            dB0[7] = dC0[7] = dB1[7] = dC1[7] = 0.f;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                dB0[i] = dBp[i].x; dB1[i] = dBp[i].y;
                dC0[i] = dCp[i].x; dC1[i] = dCp[i].y;
            }
            // dB / dC of the two states: sums over the 64 channel lanes, into the table slots just read
            // sets: 0 = dB of state 2 np, 1 = dC of it, 2 = dB of state 2 np + 1, 3 = dC of it
            if (DEEP_DBG(da, 32)) {
                sB2[0] += df2{dB0[0] + dC0[1], dB1[2] + dC1[3]};
                return;
            }
            fold_pending();                                          // (the previous pair's MFMAs finished long ago)
            pend = deep_colsum4_issue(dB0, dC0, dB1, dC1, lane);
            pend_np = np;
        };
        PairOps oa, ob;
        load_pair(0, oa);
#pragma unroll 1
        for (int np = 0; np < (DEEP_DBG(da, 1) ? 0 : N / 2); np += 2) {
            load_pair(np + 1, ob);
            run_pair(np, oa);
            if (np + 2 < N / 2) load_pair(np + 2, oa);
            run_pair(np + 1, ob);
        }
        fold_pending();
        float sB[P], sA[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            sB[i] = sB2[i].x + sB2[i].y;
            sA[i] = sA2[i].x + sA2[i].y;
        }
        // ---- per-position results: du of this route into the wave's planes, d raw step size to the staging rows
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const float du = dl[i] * sB[i];                          // (D g is added once, at the merge)
            const float ddl = fmaf(u[i], sB[i], sA[i]) * sg[i];
            dbacc += ddl;
            stg[i * 64 + lane] = deep_bf16(ddl);
            uint16_t *q = dup + lane * LP + nat[i];
            *q = deep_bf16(FIRST ? du : bf16_bits_to_float(*q) + du);
        }
        wave_sync();
        if (lane < P * 8 && !DEEP_DBG(da, 16)) {                                          // ddts rows: [position][64 channels] bf16, 16-byte stores
            const int i = lane >> 3, part = lane & 7;
            const int nf = deep_nat(k, st * P + i);
            const cu32x4_t v = *reinterpret_cast<const cu32x4_t *>(stg + i * 64 + 8 * part);
            *reinterpret_cast<cu32x4_t *>(a.ddts + ((((int64_t)sb * nr + k) * L + nf) * a.D + c0 + 8 * part)) = v;
        }
        wave_sync();
    }
    if (DEEP_DBG(da, 2)) {
        if (dbacc == 12345.678f) atomicAdd(a.dbias + wrow, dbacc + dAs[lane] + T[lane]);
        wave_sync();
        return;
    }
    for (int n = 0; n < N; ++n) atomicAdd(da.dAt + ((int64_t)ws * N + n) * a.D + c0 + lane, dAs[n * 64 + lane]);
    atomicAdd(a.dbias + wrow, dbacc);
    // dB / dC of this route over the tile: contiguous fp32 atomics in natural position order (the table is in route order)
    for (int op = 0; op < 2; ++op) {
        float *dst = a.dBC + ((((int64_t)(op ? sbC : sb) * nr + k) * 2 + op) * N) * L;
        for (int e = lane; e < N * L; e += 64) {
            const int n = e / L, pnat = e - n * L;
            int t = (k & 1) ? (pnat % HW) * HW + pnat / HW : pnat;           // sequence index of the position on route k
            t = (k & 2) ? L - 1 - t : t;
            atomicAdd(dst + e, T[deep_slot(op, n, t)]);
        }
    }
    wave_sync();
}

// NR = 4: the deep block (two waves: rows, columns; routes k and k + 2 one after the other).  NR = 1: ONE forward row-major
// route per sample and no merge across routes -- the shallow swap block (reference models/fusion_vmamba.py:808-845) with its
// two channel-swapped views as 2B samples, weight set sb / wdiv; a workgroup is one wave.
template <int KS, int NR>
__global__ void __launch_bounds__(NR == 4 ? 128 : 64) deep_bwd_kernel(const DeepArgs da) {
    constexpr int NT = NR == 4 ? 128 : 64;
    const ChanArgs &a = da.a;
    using LD = DeepBwdLds;
    extern __shared__ float smem[];
    char *sm = reinterpret_cast<char *>(smem);
    uint16_t *xs = reinterpret_cast<uint16_t *>(sm + LD::xs), *gs = reinterpret_cast<uint16_t *>(sm + LD::gs);
    float *dsum = reinterpret_cast<float *>(sm + LD::dsum);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *wl = sm + LD::wave0 + wave * LD::wave_sz;
    const int tiles = a.D / 64;
    int sb, c0;
    chan_block_map(a.xmap, tiles, sb, c0);
    c0 *= 64;
    const int ws1 = NR == 1 ? sb / a.wdiv : 0;                       // weight set of a single-route sample
    {
        // the tile's x (bf16) and dy (fp32) planes are one contiguous run each: ALL of a thread's vectors are requested before
        // the first LDS write (a load / convert / scatter loop exposed one HBM round trip per iteration, 13 per workgroup:
        // most of the ~140 us this launch took with every other phase switched off)
        const uint16_t *src = a.x + ((int64_t)sb * a.D + c0) * L;
        const float *gsrc = a.dy + ((int64_t)sb * a.D + c0) * L;
        constexpr int NVX = 64 * L / 8, NVG = 64 * L / 4, PX = (NVX + NT - 1) / NT, PG = (NVG + NT - 1) / NT;
        cu32x4_t rx[PX];
        float4 rg[PG];
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int v = threadIdx.x + NT * j;
            rx[j] = *reinterpret_cast<const cu32x4_t *>(src + 8 * (v < NVX ? v : 0));
        }
#pragma unroll
        for (int j = 0; j < PG; ++j) {
            const int v = threadIdx.x + NT * j;
            rg[j] = *reinterpret_cast<const float4 *>(gsrc + 4 * (v < NVG ? v : 0));
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int v = threadIdx.x + NT * j;
            if (v < NVX) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = 8 * v + 2 * q, c = e / L, l = e - c * L;
                    const int e1 = e + 1, c1 = e1 / L, l1 = e1 - c1 * L;
                    xs[c * LP + l] = (uint16_t)(rx[j][q] & 0xffffu);
                    xs[c1 * LP + l1] = (uint16_t)(rx[j][q] >> 16);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < PG; ++j) {
            const int v = threadIdx.x + NT * j;
            if (v < NVG) {
                const float f[4] = {rg[j].x, rg[j].y, rg[j].z, rg[j].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = 4 * v + q, c = e / L, l = e - c * L;
                    gs[c * LP + l] = deep_bf16(f[q]);
                }
            }
        }
        if (threadIdx.x < 64) {
            const int q = threadIdx.x;
            if constexpr (NR == 4) dsum[q] = (a.Dp[c0 + q] + a.Dp[a.D + c0 + q]) + (a.Dp[2 * a.D + c0 + q] + a.Dp[3 * a.D + c0 + q]);
            else dsum[q] = a.Dp[(int64_t)ws1 * a.D + c0 + q];
        }
    }
    __syncthreads();
    if constexpr (NR == 1) {
        deep_bwd_pass<KS, true>(da, sb, c0, 0, 1, ws1, xs, gs, wl);
    } else if (wave == 0) {
        deep_bwd_pass<KS, true>(da, sb, c0, 0, 4, 0, xs, gs, wl);
        deep_bwd_pass<KS, false>(da, sb, c0, 2, 4, 2, xs, gs, wl);
    } else {
        deep_bwd_pass<KS, true>(da, sb, c0, 1, 4, 1, xs, gs, wl);
        deep_bwd_pass<KS, false>(da, sb, c0, 3, 4, 3, xs, gs, wl);
    }
    __syncthreads();
    // ---- dx = rows + columns + (sum_k D_k) g ; dD_k[c] += sum_l g u (the same for every route k)
    const uint16_t *d0 = reinterpret_cast<const uint16_t *>(sm + LD::wave0 + LD::dup);
    const uint16_t *d1 = NR == 4 ? reinterpret_cast<const uint16_t *>(sm + LD::wave0 + LD::wave_sz + LD::dup) : d0;
    uint16_t *dst = a.dx + ((int64_t)sb * a.D + c0) * L;
    for (int v = threadIdx.x; v < 64 * L / 2; v += NT) {
        float o[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = 2 * v + q, c = e / L, l = e - c * L, off = c * LP + l;
            const float du = NR == 4 ? bf16_bits_to_float(d0[off]) + bf16_bits_to_float(d1[off]) : bf16_bits_to_float(d0[off]);
            o[q] = fmaf(dsum[c], bf16_bits_to_float(gs[off]), du);
        }
        *reinterpret_cast<uint32_t *>(dst + 2 * v) = pack_bf16x2(o[0], o[1]);
    }
    if constexpr (NR == 4) {
        const int c = threadIdx.x >> 1, part = threadIdx.x & 1;      // two lanes per channel split the plane
        float s = 0.f;
        for (int l = part; l < L; l += 2) s = fmaf(bf16_bits_to_float(gs[c * LP + l]), bf16_bits_to_float(xs[c * LP + l]), s);
        s += __shfl_xor(s, 1, 64);
        if (part == 0)
            for (int kk = 0; kk < 4; ++kk) atomicAdd(a.dD + kk * a.D + c0 + c, s);
    } else {
        const int c = threadIdx.x;
        float s = 0.f;
        for (int l = 0; l < L; ++l) s = fmaf(bf16_bits_to_float(gs[c * LP + l]), bf16_bits_to_float(xs[c * LP + l]), s);
        atomicAdd(a.dD + (int64_t)ws1 * a.D + c0 + c, s);
    }
}

__global__ void __launch_bounds__(256) deep_zero_kernel(float *p, const int n) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n) p[e] = 0.f;
}

// dA (sets D, N) += dAt (sets, N, D)
__global__ void __launch_bounds__(256) deep_dA_finish_kernel(const float *dAt, float *dA, const int D, const int sets) {
    const int e = blockIdx.x * 256 + threadIdx.x;                   // index into dA: (k D + c) N + n
    if (e >= sets * D * N) return;
    const int n = e % N, r = e / N, k = r / D, c = r - k * D;
    dA[e] += dAt[((int64_t)k * N + n) * D + c];
}

static int deep_supported(int H, int W, int N_, int NR, int D, int R) {
    return H == 7 && W == 7 && N_ == 16 && (NR == 4 || NR == 1) && D % 64 == 0 && R >= 1 && R <= 48;
}

template <int KS, int NR> static int deep_launch(const DeepArgs &da, hipStream_t s) {
    const size_t lds = NR == 4 ? DeepBwdLds::total : DeepBwdLds::wave0 + DeepBwdLds::wave_sz;
    const void *fn = (const void *)deep_bwd_kernel<KS, NR>;
    static LdsOptIn opted;
    if (lds > 64 * 1024 && !lds_opt_in(opted, fn, lds)) return XFM_ELAUNCH;
    DeepArgs args = da;
    args.dbg = 0;
#ifdef XFM_DEEP_TIMING
    static const int env_dbg = [] { const char *e = getenv("XFM_DEEP_DBG"); return e ? atoi(e) : 0; }();
    args.dbg = env_dbg;
#endif
    void *kargs[] = {&args};
    const int sets = NR == 4 ? 4 : (da.a.Bt + da.a.wdiv - 1) / da.a.wdiv;
    // (a kernel, not a memset node: under stream capture the memset of this workspace slice replayed with stale contents)
    hipLaunchKernelGGL(deep_zero_kernel, dim3((sets * N * da.a.D + 255) / 256), dim3(256), 0, s, da.dAt, sets * N * da.a.D);
    const hipError_t e = hipLaunchKernel(fn, dim3((unsigned)(da.a.Bt * (da.a.D / 64))), dim3(NR == 4 ? 128 : 64), kargs, lds, s);
    if (e != hipSuccess) {
        set_last_hip_error(e);
        return XFM_ELAUNCH;
    }
    hipLaunchKernelGGL(deep_dA_finish_kernel, dim3((sets * da.a.D * N + 255) / 256), dim3(256), 0, s, da.dAt, da.a.dA, da.a.D, sets);
    return check_launch();
}

// ---- forward with ONE route per sample (n_routes == 1: the shallow swap block) ------------------------------------------
// The first-design forward above puts a route and its reverse into the two halves of a wave; a sample with a single forward
// route would idle half of it.  Here a workgroup is one wave = 64 channels of the sample's route, the second design's B / C
// table in LDS (pair-interleaved: one 16-byte broadcast read per position and state pair), and -- nothing but the dt_proj
// MFMAs of a step's head runs beside them -- the two states of a pair as packed fp32, all eight pairs unrolled with their
// decay rates and states in registers.  Writes the packed row checkpoints deep_bwd_kernel<KS, 1> reads.
struct DeepFwd1Lds {
    static constexpr size_t xs = 0, ys = xs + 64 * LP * 2, T = ys + 64 * LP * 4, total = T + (size_t)TBL * 4;
};

template <int KS>
__global__ void __launch_bounds__(64) deep_fwd1_kernel(const DeepArgs da) {
    const ChanArgs &a = da.a;
    extern __shared__ float smem[];
    char *sm = reinterpret_cast<char *>(smem);
    uint16_t *xs = reinterpret_cast<uint16_t *>(sm + DeepFwd1Lds::xs);
    float *ys = reinterpret_cast<float *>(sm + DeepFwd1Lds::ys);
    float *T = reinterpret_cast<float *>(sm + DeepFwd1Lds::T);
    const int lane = threadIdx.x;
    const int tiles = a.D / 64;
    int sb, c0;
    chan_block_map(a.xmap, tiles, sb, c0);
    c0 *= 64;
    const int ws = sb / a.wdiv;
    {
        const uint16_t *src = a.x + ((int64_t)sb * a.D + c0) * L;
        constexpr int NVX = 64 * L / 8, PX = (NVX + 63) / 64;
        cu32x4_t rx[PX];
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int v = lane + 64 * j;
            rx[j] = *reinterpret_cast<const cu32x4_t *>(src + 8 * (v < NVX ? v : 0));
        }
#pragma unroll
        for (int j = 0; j < PX; ++j) {
            const int v = lane + 64 * j;
            if (v < NVX) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = 8 * v + 2 * q, c = e / L, l = e - c * L;
                    const int e1 = e + 1, c1 = e1 / L, l1 = e1 - c1 * L;
                    xs[c * LP + l] = (uint16_t)(rx[j][q] & 0xffffu);
                    xs[c1 * LP + l1] = (uint16_t)(rx[j][q] >> 16);
                }
            }
        }
    }
    deep_fill_bc(a, sb, sb, 0, T, lane);
    const int64_t wrow = (int64_t)ws * a.D + c0 + lane;
    const float bv = a.bias[wrow], Dv = a.Dp[wrow];
    df2 A2p[N / 2], h2[N / 2];
    {
        const float *Arow = a.A + wrow * N;
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            const float4 t = *reinterpret_cast<const float4 *>(Arow + 4 * q);
            A2p[2 * q] = df2{t.x, t.y} * kLog2e;
            A2p[2 * q + 1] = df2{t.z, t.w} * kLog2e;
        }
#pragma unroll
        for (int np = 0; np < N / 2; ++np) h2[np] = df2{0.f, 0.f};
    }
    DeepX<KS> xf;
    DeepW<KS> wf;
    deep_load_x<KS>(a, sb, 0, 0, lane, xf);
    deep_load_w<KS>(a, c0, ws, lane, wf);
    uint32_t *chk = da.chkp + ((int64_t)sb * NSTEP) * (N / 2) * a.D + c0 + lane;
    wave_sync();
#pragma unroll 1
    for (int st = 0; st < NSTEP; ++st) {
        const cf32x16_t acc = deep_dt_step<KS>(bv, xf, wf);
        if (st + 1 < NSTEP) deep_load_x<KS>(a, sb, 0, st + 1, lane, xf);
        float dl[P], u[P], dlu[P];
        df2 y2[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            dl[i] = chan_softplus(acc[i]);
            u[i] = bf16_bits_to_float(xs[lane * LP + st * P + i]);
            dlu[i] = dl[i] * u[i];
            y2[i] = df2{0.f, 0.f};
        }
        const float4 *Tst = reinterpret_cast<const float4 *>(T) + st * 8;
#pragma unroll
        for (int np = 0; np < N / 2; ++np) {
            df2 h = h2[np];
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const float4 q = Tst[np * NSTEP * 8 + i];            // (B, C) of both states at position i: one broadcast read
                const df2 t = A2p[np] * dl[i];
                const df2 av = {exp2_fast(t.x), exp2_fast(t.y)};
                h = __builtin_elementwise_fma(av, h, df2{q.x, q.y} * dlu[i]);
                y2[i] = __builtin_elementwise_fma(df2{q.z, q.w}, h, y2[i]);
            }
            h2[np] = h;
        }
#pragma unroll
        for (int i = 0; i < P; ++i) ys[lane * LP + st * P + i] = fmaf(Dv, u[i], y2[i].x + y2[i].y);
        if (st + 1 < NSTEP) {
#pragma unroll
            for (int np = 0; np < N / 2; ++np) chk[((int64_t)st * (N / 2) + np) * a.D] = pack_bf16x2(h2[np].x, h2[np].y);
        }
    }
    wave_sync();
    float *dst = a.y + ((int64_t)sb * a.D + c0) * L;
    for (int v = lane; v < 64 * L / 4; v += 64) {
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = 4 * v + q, c = e / L, l = e - c * L;
            o[q] = ys[c * LP + l];
        }
        *reinterpret_cast<float4 *>(dst + 4 * v) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

template <int KS> static int deep_fwd1_launch(const DeepArgs &da, hipStream_t s) {
    DeepArgs args = da;
    args.dbg = 0;
    hipLaunchKernelGGL(deep_fwd1_kernel<KS>, dim3((unsigned)(da.a.Bt * (da.a.D / 64))), dim3(64), DeepFwd1Lds::total, s, args);
    return check_launch();
}

}  // namespace deep

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
    static LdsOptIn opted[2];                                     // (per template instantiation: once per kernel and device)
    if (lds > 64 * 1024 && !lds_opt_in(opted[bwd], fn, lds)) return XFM_ELAUNCH;
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

int chan1_run(const ChanArgs &a, int HW, bool bwd, hipStream_t s);      // ss2d_chan1.hip: d_state 1, second generation
int chan1_covers(int H, int W, int N, int n_routes);
int chan1_ytok(int H, int W, int N, int n_routes);

static int chan_supported(int HW_h, int HW_w, int N, int NR, int D, int R) {
    if (HW_h != HW_w || D % 32 || R < 1 || R > 64) return 0;
    if (N == 1 && NR == 4 && (HW_h == 7 || HW_h == 12 || HW_h == 14)) return 1;
#ifdef XFM_CHAN_N16
    if (N == 16 && NR == 4 && (HW_h == 5 || HW_h == 7 || HW_h == 12)) return 1;
    if (N == 16 && NR == 1 && HW_h == 7 && D % 64 == 0 && R <= 48) return 1;      // the shallow swap block: namespace deep
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
    a.wdiv = p->wdiv > 0 ? p->wdiv : 1;
    a.ytok = p->y_tokens ? 1 : 0;
    a.xtok = p->x_tokens ? 1 : 0;
    if ((a.ytok || a.xtok) && (p->c_mod != 0 || !chan1_ytok(p->H, p->W, p->dstate, p->n_routes))) return XFM_ELIMIT;
    a.zeros = (const uint16_t *)p->zeros;
    a.xmap = (p->batch % 8 == 0 && !getenv("XFM_CHAN_NO_XMAP")) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    const int HW = p->H;
    static const bool deep_on = [] {
        const char *e = getenv("XFM_SS2D_DEEP");
        return !(e && e[0] == '0');
    }();
    if (deep_on && deep::deep_supported(p->H, p->W, N, p->n_routes, p->d_inner, p->dt_rank)) {
        // workspace (xfm_ss2dc_nsteps = 7 steps): packed bf16 checkpoints in the first half, the dA scratch behind.
        // Forward: the first-design kernel (two directions per wave; 222 us against 307 us for a 64-lane forward of the
        // second design) writing one packed checkpoint per row / column; backward: namespace deep.
        deep::DeepArgs da;
        da.a = a;
        da.chkp = reinterpret_cast<uint32_t *>(p->chk);
        da.dAt = p->chk + (size_t)p->batch * p->n_routes * deep::NSTEP * (deep::N / 2) * p->d_inner;   // (sets * 16 D floats of the second half)
        if (p->n_routes == 1) {
            // one forward route per sample (the shallow swap block): the single-route kernels of namespace deep, both ways
            if (p->wdiv < 1 || p->c_mod != 0 || (p->batch + p->wdiv - 1) / p->wdiv > p->batch * deep::NSTEP / 4) return XFM_EINVAL;
            switch (a.Kp / 16) {
                case 1: return bwd ? deep::deep_launch<1, 1>(da, s) : deep::deep_fwd1_launch<1>(da, s);
                case 2: return bwd ? deep::deep_launch<2, 1>(da, s) : deep::deep_fwd1_launch<2>(da, s);
                case 3: return bwd ? deep::deep_launch<3, 1>(da, s) : deep::deep_fwd1_launch<3>(da, s);
            }
            return XFM_ELIMIT;
        }
        if (!bwd) {
            a.chk16 = reinterpret_cast<uint16_t *>(p->chk);
        } else {
            switch (a.Kp / 16) {
                case 1: return deep::deep_launch<1, 4>(da, s);
                case 2: return deep::deep_launch<2, 4>(da, s);
                case 3: return deep::deep_launch<3, 4>(da, s);
            }
        }
    }
    if (chan1_covers(p->H, p->W, N, p->n_routes) && p->c_mod == 0) {      // second generation (ss2d_chan1.hip), both directions
        // (XFM_ELIMIT -- dt_rank beyond four k-steps, a ddts tensor of 4 GB -- is the same answer for both directions of a
        //  shape, so the generations never mix: their checkpoints differ at 7 x 7)
        const int rc1 = chan1_run(a, HW, bwd, s);
        // (token-major y / dy / x / dx exist in the second generation only: such a request never reaches the kernels below)
        if (rc1 != XFM_ELIMIT || a.ytok || a.xtok) return rc1;
    }
    if (a.ytok || a.xtok) return XFM_ELIMIT;
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
int xfm_ss2dc_ytokens_supported(int H, int W, int dstate, int n_routes) { return xfm::chan1_ytok(H, W, dstate, n_routes); }
int xfm_ss2dc_nsteps(int H, int W, int dstate) {
    if (H == 7 && W == 7 && dstate == 16) return 7;      // one row / column per step (second design, namespace deep)
    if (xfm::chan1_covers(H, W, dstate, 4)) return H;    // ss2d_chan1.hip: one row / column per step at every size
    const int P = H <= 8 ? 2 * H : H;
    return (H * W + P - 1) / P;
}
int xfm_ss2dc_fwd(const xfm_ss2dc_params_t *p, void *stream) { return xfm::chan_run(p, false, stream); }
int xfm_ss2dc_bwd(const xfm_ss2dc_params_t *p, void *stream) { return xfm::chan_run(p, true, stream); }
}
