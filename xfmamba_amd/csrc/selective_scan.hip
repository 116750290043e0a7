// selective_scan.hip -- selective scan (S6) forward / backward for gfx950.
//
// Replaces the reference's CUDA extension at its FFI boundary
// (models/selective_scan/csrc/selective_scan/selective_scan.cpp:165-362); the maths is the one
// of models/csms6s.py:25-68 (forward) and SURVEY.md appendix B (backward).  Not a translation of
// the cub::BlockScan design: the unit of work is a 64-lane WAVEFRONT that owns a tile of
// G = 64/LPR consecutive (batch, dim) rows; LPR lanes split a row's sequence into chunks of C
// elements held in registers.  Per state n the lane reduces its chunk to the affine map
// h -> P*h + S, the LPR maps are combined by a segmented wave scan (ds_bpermute shuffles, no LDS
// traffic, no workgroup barrier), and the lane replays its chunk from the exact incoming state.
// Rows longer than LPR*C are walked in chunks with the carried state kept in LDS (forward) and
// check-pointed to `x` for the backward, which walks the chunks in reverse.
//
// Global <-> register traffic goes through a per-wave LDS transposition tile: HBM is always
// touched with lanes along the contiguous sequence axis, and each lane then picks its C
// consecutive elements with an odd LDS stride (bank-conflict free).
#include "scan_core.hpp"
#include "rowscan.hpp"

namespace xfm {

struct ScanArgs {
    xfm_scan_params_t p;
    int lg_lpr;          // log2(lanes per row)
    int n_chunks;
    int dim_per_group;
    int lds_floats_per_wave;
};

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <typename Tin, typename Tout, int C>
__global__ void __launch_bounds__(256) scan_fwd_kernel(const ScanArgs a) {
    extern __shared__ float smem[];
    const xfm_scan_params_t &p = a.p;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lg = a.lg_lpr, LPR = 1 << lg, G = 64 >> lg;
    const int tiles_pb = p.dim >> (6 - lg);
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile >= (int64_t)p.batch * tiles_pb) return;
    const int b = (int)(tile / tiles_pb);
    const int r0 = (int)(tile - (int64_t)b * tiles_pb) * G;
    const int g = lane >> lg, i = lane & (LPR - 1);
    const int r = r0 + g;
    const int k = r0 / a.dim_per_group;
    const int N = p.dstate, L = p.seqlen, SL = C << lg;

    float *buf = smem + (size_t)wave * a.lds_floats_per_wave;
    float *carry = buf + 64 * (C | 1);  // [G][N] running state between chunks
    for (int n = i; n < N; n += LPR) carry[g * N + n] = 0.f;

    const Tin *u_t = (const Tin *)p.u + (int64_t)b * p.u_batch_stride + (int64_t)r0 * p.u_d_stride;
    const Tin *d_t = (const Tin *)p.delta + (int64_t)b * p.delta_batch_stride + (int64_t)r0 * p.delta_d_stride;
    Tout *o_t = (Tout *)p.out + (int64_t)b * p.out_batch_stride + (int64_t)r0 * p.out_d_stride;
    const Tin *Bg = (const Tin *)p.B + (int64_t)b * p.B_batch_stride + (int64_t)k * p.B_group_stride;
    const Tin *Cg = (const Tin *)p.C + (int64_t)b * p.C_batch_stride + (int64_t)k * p.C_group_stride;
    const float *Ar = p.A + (int64_t)r * p.A_d_stride;
    const float Dr = p.D ? p.D[r] : 0.f;
    const float bias = p.delta_bias ? p.delta_bias[r] : 0.f;

    for (int seg = 0; seg < a.n_chunks; ++seg) {
        const int s0 = seg * SL;
        const int t0 = s0 + i * C;
        float u[C], dl[C], y[C];
        tile_load<Tin, C>(buf, u_t, p.u_d_stride, G, lg, s0, L, lane, lane, u);
        tile_load<Tin, C>(buf, d_t, p.delta_d_stride, G, lg, s0, L, lane, lane, dl);
#pragma unroll
        for (int jj = 0; jj < C; ++jj) {
            const bool valid = t0 + jj < L;
            float v = dl[jj] + bias;
            if (p.delta_softplus) v = softplus20(v);
            dl[jj] = valid ? v : 0.f;  // identity element for t >= L: a = 1, b = 0
            y[jj] = 0.f;
        }
        for (int n = 0; n < N; ++n) {
            const float A2 = Ar[n] * kLog2e;
            const Tin *Bn = Bg + (int64_t)n * p.B_dstate_stride + t0;
            const Tin *Cn = Cg + (int64_t)n * p.C_dstate_stride + t0;
            float Bv[C], Cv[C], av[C];
#pragma unroll
            for (int jj = 0; jj < C; ++jj) {
                const bool valid = t0 + jj < L;
                Bv[jj] = valid ? ldf<Tin>(Bn + jj) : 0.f;
                Cv[jj] = valid ? ldf<Tin>(Cn + jj) : 0.f;
            }
            float P = 1.f, S = 0.f;
#pragma unroll
            for (int jj = 0; jj < C; ++jj) {
                av[jj] = exp2_fast(dl[jj] * A2);
                Bv[jj] *= dl[jj] * u[jj];  // b_t = delta * B * u
                S = fmaf(av[jj], S, Bv[jj]);
                P *= av[jj];
            }
            float h = carry[g * N + n];
            if (LPR > 1) {
                seg_scan_up(P, S, i, LPR);
                const float Pe = __shfl_up(P, 1, LPR), Se = __shfl_up(S, 1, LPR);
                if (i > 0) h = fmaf(Pe, h, Se);
            }
#pragma unroll
            for (int jj = 0; jj < C; ++jj) {
                h = fmaf(av[jj], h, Bv[jj]);
                y[jj] = fmaf(Cv[jj], h, y[jj]);
            }
            if (i == LPR - 1) {
                carry[g * N + n] = h;
                if (p.x) p.x[(((int64_t)b * p.dim + r) * a.n_chunks + seg) * N + n] = h;
            }
        }
#pragma unroll
        for (int jj = 0; jj < C; ++jj) y[jj] = fmaf(Dr, u[jj], y[jj]);
        tile_store<Tout, C>(buf, o_t, p.out_d_stride, G, lg, s0, L, lane, lane, y);
    }
}

// ---------------------------------------------------------------------------------------------
// backward.  With E_t := a_t * dh_t the reverse recurrence is  dh_t = C_t*g_t + E_{t+1},
// E_t = a_t*dh_t -- an affine map in E with the same P = prod(a) as the forward chunk.
// ---------------------------------------------------------------------------------------------
template <typename Tin, typename Tout, int C>
__global__ void __launch_bounds__(256) scan_bwd_kernel(const ScanArgs a) {
    extern __shared__ float smem[];
    const xfm_scan_params_t &p = a.p;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lg = a.lg_lpr, LPR = 1 << lg, G = 64 >> lg;
    const int tiles_pb = p.dim >> (6 - lg);
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile >= (int64_t)p.batch * tiles_pb) return;
    const int b = (int)(tile / tiles_pb);
    const int r0 = (int)(tile - (int64_t)b * tiles_pb) * G;
    const int g = lane >> lg, i = lane & (LPR - 1);
    const int r = r0 + g;
    const int k = r0 / a.dim_per_group;
    const int N = p.dstate, L = p.seqlen, SL = C << lg;

    float *buf = smem + (size_t)wave * a.lds_floats_per_wave;
    float *carryE = buf + 64 * (C | 1);  // [G][N]: E flowing from chunk seg+1 into chunk seg
    for (int n = i; n < N; n += LPR) carryE[g * N + n] = 0.f;

    const int64_t row_off = ((int64_t)b * p.dim + r0) * L;  // du / ddelta are contiguous
    const Tin *u_t = (const Tin *)p.u + (int64_t)b * p.u_batch_stride + (int64_t)r0 * p.u_d_stride;
    const Tin *d_t = (const Tin *)p.delta + (int64_t)b * p.delta_batch_stride + (int64_t)r0 * p.delta_d_stride;
    const Tout *g_t = (const Tout *)p.dout + (int64_t)b * p.dout_batch_stride + (int64_t)r0 * p.dout_d_stride;
    Tin *du_t = (Tin *)p.du + row_off;
    Tin *dd_t = (Tin *)p.ddelta + row_off;
    const Tin *Bg = (const Tin *)p.B + (int64_t)b * p.B_batch_stride + (int64_t)k * p.B_group_stride;
    const Tin *Cg = (const Tin *)p.C + (int64_t)b * p.C_batch_stride + (int64_t)k * p.C_group_stride;
    float *dBg = p.dB + ((int64_t)b * p.n_groups + k) * N * L;
    float *dCg = p.dC + ((int64_t)b * p.n_groups + k) * N * L;
    const float *Ar = p.A + (int64_t)r * p.A_d_stride;
    const float Dr = p.D ? p.D[r] : 0.f;
    const float bias = p.delta_bias ? p.delta_bias[r] : 0.f;
    float dD_acc = 0.f, dbias_acc = 0.f;

    for (int seg = a.n_chunks - 1; seg >= 0; --seg) {
        const int s0 = seg * SL;
        const int t0 = s0 + i * C;
        float u[C], dl[C], go[C], s1[C], s2[C];
        tile_load<Tin, C>(buf, u_t, p.u_d_stride, G, lg, s0, L, lane, lane, u);
        tile_load<Tin, C>(buf, d_t, p.delta_d_stride, G, lg, s0, L, lane, lane, dl);
        tile_load<Tout, C>(buf, g_t, p.dout_d_stride, G, lg, s0, L, lane, lane, go);
#pragma unroll
        for (int jj = 0; jj < C; ++jj) {
            const bool valid = t0 + jj < L;
            float v = dl[jj] + bias;
            if (p.delta_softplus) v = softplus20(v);
            dl[jj] = valid ? v : 0.f;
            s1[jj] = 0.f;  // sum_n dh*B
            s2[jj] = 0.f;  // sum_n dh*A*(a*h_prev)
        }
        for (int n = 0; n < N; ++n) {
            const float An = Ar[n];
            const float A2 = An * kLog2e;
            const Tin *Bn = Bg + (int64_t)n * p.B_dstate_stride + t0;
            const Tin *Cn = Cg + (int64_t)n * p.C_dstate_stride + t0;
            float Bv[C], cg[C], av[C], h[C];
#pragma unroll
            for (int jj = 0; jj < C; ++jj) {
                const bool valid = t0 + jj < L;
                Bv[jj] = valid ? ldf<Tin>(Bn + jj) : 0.f;
                cg[jj] = valid ? ldf<Tin>(Cn + jj) * go[jj] : 0.f;  // C_t * g_t
            }
            // chunk summaries: forward (P,S) and reverse (P,R)
            float P = 1.f, S = 0.f;
#pragma unroll
            for (int jj = 0; jj < C; ++jj) {
                av[jj] = exp2_fast(dl[jj] * A2);
                S = fmaf(av[jj], S, dl[jj] * u[jj] * Bv[jj]);
                P *= av[jj];
            }
            float R = 0.f;
#pragma unroll
            for (int jj = C - 1; jj >= 0; --jj) R = av[jj] * (cg[jj] + R);
            float hin = (seg > 0) ? p.x[(((int64_t)b * p.dim + r) * a.n_chunks + (seg - 1)) * N + n] : 0.f;
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
            // replay forward: h_t
            float hh = hin;
#pragma unroll
            for (int jj = 0; jj < C; ++jj) {
                hh = fmaf(av[jj], hh, dl[jj] * u[jj] * Bv[jj]);
                h[jj] = hh;
            }
            // replay reverse: dh_t and every per-(t,n) product
            float E = Ein, dA_acc = 0.f;
#pragma unroll
            for (int jj = C - 1; jj >= 0; --jj) {
                const float dh = cg[jj] + E;
                E = av[jj] * dh;
                const float du_ = dl[jj] * u[jj];
                const float ah = h[jj] - du_ * Bv[jj];  // a_t * h_{t-1}
                s1[jj] = fmaf(dh, Bv[jj], s1[jj]);
                s2[jj] = fmaf(dh * An, ah, s2[jj]);
                dA_acc = fmaf(dh * dl[jj], ah, dA_acc);
                float dBv = dh * du_;       // dB contribution of this row
                float dCv = go[jj] * h[jj];  // dC contribution of this row
                for (int off = LPR; off < 64; off <<= 1) {  // sum over the G rows of the tile
                    dBv += __shfl_xor(dBv, off, 64);
                    dCv += __shfl_xor(dCv, off, 64);
                }
                if (g == 0 && t0 + jj < L) {
                    atomicAdd(dBg + (int64_t)n * L + t0 + jj, dBv);
                    atomicAdd(dCg + (int64_t)n * L + t0 + jj, dCv);
                }
            }
            if (i == 0) carryE[g * N + n] = E;
            for (int off = 1; off < LPR; off <<= 1) dA_acc += __shfl_xor(dA_acc, off, 64);
            if (i == 0) atomicAdd(p.dA + (int64_t)r * N + n, dA_acc);
        }
        float du[C], dd[C];
#pragma unroll
        for (int jj = 0; jj < C; ++jj) {
            du[jj] = fmaf(dl[jj], s1[jj], Dr * go[jj]);
            float ddl = fmaf(u[jj], s1[jj], s2[jj]);
            // d softplus(raw)/d raw = sigmoid(raw) = 1 - exp(-softplus(raw)); linear branch above 20
            if (p.delta_softplus && dl[jj] <= 20.f) ddl *= 1.f - __expf(-dl[jj]);
            dd[jj] = ddl;
            dD_acc = fmaf(go[jj], u[jj], dD_acc);
            dbias_acc += (t0 + jj < L) ? ddl : 0.f;
        }
        tile_store<Tin, C>(buf, du_t, (int64_t)L, G, lg, s0, L, lane, lane, du);
        tile_store<Tin, C>(buf, dd_t, (int64_t)L, G, lg, s0, L, lane, lane, dd);
    }
    for (int off = 1; off < LPR; off <<= 1) {
        dD_acc += __shfl_xor(dD_acc, off, 64);
        dbias_acc += __shfl_xor(dbias_acc, off, 64);
    }
    if (i == 0) {
        if (p.dD) atomicAdd(p.dD + r, dD_acc);
        if (p.ddelta_bias) atomicAdd(p.ddelta_bias + r, dbias_acc);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static const int kItems[] = {4, 7, 9, 13};

static int make_plan(int batch, int dim, int L, int N, int K, xfm_scan_plan_t *plan) {
    if (batch <= 0 || dim <= 0 || L <= 0 || N <= 0 || K <= 0 || dim % K) return XFM_EINVAL;
    if (N > 256) return XFM_ELIMIT;  // same bound as the reference FFI (selective_scan.cpp:204)
    const int Dg = dim / K;
    double best = 1e300;
    int bl = -1, bc = -1;
    for (int lg = 0; lg <= 6; ++lg) {
        const int LPR = 1 << lg, G = 64 >> lg;
        if (Dg % G) continue;            // a tile must not straddle two B/C groups
        if (G * N > 2048) continue;      // carried state in LDS: <= 8 KiB per wave
        for (int c : kItems) {
            const int SL = LPR * c;
            const int nseg = (L + SL - 1) / SL;
            // issue slots per row: per-element work + scan steps + fixed cost, all per state
            const double row = nseg * (N * (c * 1.0 + lg * 0.8 + 1.5) + c * 0.6 + 2.0);
            const double waves = (double)batch * dim / G;
            const double rounds = waves < 3072.0 ? 1.0 : waves / 3072.0;  // ~3 waves/SIMD resident
            const double est = row * rounds;
            if (est < best * 0.999) {
                best = est;
                bl = lg;
                bc = c;
            }
        }
    }
    if (bl < 0) return XFM_ELIMIT;
    plan->lanes_per_row = 1 << bl;
    plan->items = bc;
    plan->n_chunks = (L + (bc << bl) - 1) / (bc << bl);
    return XFM_OK;
}

static int validate(const xfm_scan_params_t *p, bool bwd) {
    if (!p || !p->u || !p->delta || !p->A || !p->B || !p->C) return XFM_EINVAL;
    if (p->in_dtype < 0 || p->in_dtype > 2) return XFM_EDTYPE;
    if (p->out_dtype != XFM_F32 && p->out_dtype != p->in_dtype) return XFM_EDTYPE;
    if (!bwd && !p->out) return XFM_EINVAL;
    if (bwd && (!p->dout || !p->du || !p->ddelta || !p->dA || !p->dB || !p->dC)) return XFM_EINVAL;
    if (bwd && ((p->D && !p->dD) || (p->delta_bias && !p->ddelta_bias))) return XFM_EINVAL;
    return XFM_OK;
}

template <typename Tin, typename Tout, int C>
static int launch(const ScanArgs &a, bool bwd, hipStream_t s) {
    const int G = 64 >> a.lg_lpr;
    const int64_t tiles = (int64_t)a.p.batch * (a.p.dim / G);
    const unsigned grid = (unsigned)((tiles + 3) / 4);
    const size_t lds = (size_t)4 * a.lds_floats_per_wave * sizeof(float);
    if (bwd)
        hipLaunchKernelGGL((scan_bwd_kernel<Tin, Tout, C>), dim3(grid), dim3(256), lds, s, a);
    else
        hipLaunchKernelGGL((scan_fwd_kernel<Tin, Tout, C>), dim3(grid), dim3(256), lds, s, a);
    return check_launch();
}

template <typename Tin, typename Tout>
static int dispatch_items(const ScanArgs &a, int items, bool bwd, hipStream_t s) {
    switch (items) {
        case 4: return launch<Tin, Tout, 4>(a, bwd, s);
        case 7: return launch<Tin, Tout, 7>(a, bwd, s);
        case 9: return launch<Tin, Tout, 9>(a, bwd, s);
        case 13: return launch<Tin, Tout, 13>(a, bwd, s);
    }
    return XFM_EINVAL;
}

static int run(const xfm_scan_params_t *p, bool bwd, void *stream) {
    int rc = validate(p, bwd);
    if (rc) return rc;
    xfm_scan_plan_t plan;
    rc = make_plan(p->batch, p->dim, p->seqlen, p->dstate, p->n_groups, &plan);
    if (rc) return rc;
    if (plan.n_chunks > 1 && !p->x) return XFM_EINVAL;
    ScanArgs a;
    a.p = *p;
    a.lg_lpr = __builtin_ctz(plan.lanes_per_row);
    a.n_chunks = plan.n_chunks;
    a.dim_per_group = p->dim / p->n_groups;
    a.lds_floats_per_wave = 64 * (plan.items | 1) + (64 / plan.lanes_per_row) * p->dstate;
    hipStream_t s = (hipStream_t)stream;
    const bool of32 = p->out_dtype == XFM_F32;
    {   // short rows (7x7 maps): one lane per row, no scan (rowscan.hpp)
        int rrc = 0;
        bool hit = false;
        if (p->in_dtype == XFM_F32) hit = rowscan_try<float, float>(*p, bwd, s, &rrc);
        else if (p->in_dtype == XFM_BF16) hit = of32 ? rowscan_try<bf16_t, float>(*p, bwd, s, &rrc) : rowscan_try<bf16_t, bf16_t>(*p, bwd, s, &rrc);
        else if (p->in_dtype == XFM_F16) hit = of32 ? rowscan_try<f16_t, float>(*p, bwd, s, &rrc) : rowscan_try<f16_t, f16_t>(*p, bwd, s, &rrc);
        if (hit) return rrc;
    }
    switch (p->in_dtype) {
        case XFM_F32: return dispatch_items<float, float>(a, plan.items, bwd, s);
        case XFM_F16:
            return of32 ? dispatch_items<f16_t, float>(a, plan.items, bwd, s)
                        : dispatch_items<f16_t, f16_t>(a, plan.items, bwd, s);
        case XFM_BF16:
            return of32 ? dispatch_items<bf16_t, float>(a, plan.items, bwd, s)
                        : dispatch_items<bf16_t, bf16_t>(a, plan.items, bwd, s);
    }
    return XFM_EDTYPE;
}

}  // namespace xfm

extern "C" {
int xfm_scan_plan(int batch, int dim, int seqlen, int dstate, int n_groups, xfm_scan_plan_t *plan) {
    if (!plan) return XFM_EINVAL;
    return xfm::make_plan(batch, dim, seqlen, dstate, n_groups, plan);
}
int xfm_selective_scan_fwd(const xfm_scan_params_t *p, void *stream) { return xfm::run(p, false, stream); }
int xfm_selective_scan_bwd(const xfm_scan_params_t *p, void *stream) { return xfm::run(p, true, stream); }
}
