// ss2d_fused.hip -- host side of the fused SS2D core (planning, launch, C ABI).  Kernels: ss2d_kernels.hpp.
#include <cstdio>
#include <cstdlib>

#include "ss2d_kernels.hpp"

namespace xfm {

static const int kItems2[] = {4, 7, 9, 13};
static const size_t kLdsPerCU = 160 * 1024;
static inline size_t up4(size_t v) { return (v + 3) & ~(size_t)3; }

// One decomposition serves forward and backward (they share the chunk-state layout `chk`).
static int plan_ss2d(int batch, int D, int H, int W, int N, int in_dtype, Plan2 *out) {
    if (batch <= 0 || D <= 0 || H <= 0 || W <= 0 || N <= 0) return XFM_EINVAL;
    if (N > 256 || (int64_t)H * W >= 65536) return XFM_ELIMIT;
    const int L = H * W;
    const size_t PSZ = up4((size_t)H * (W | 1));
    // tuning hook (tools/kbench.py): XFM_SS2D_FORCE="kind,lg,items,pli" restricts the search
    int f_kind = -1, f_lg = -1, f_items = -1, f_pli = -1;
    if (const char *env = getenv("XFM_SS2D_FORCE")) sscanf(env, "%d,%d,%d,%d", &f_kind, &f_lg, &f_items, &f_pli);
    out->psz = (int)PSZ;
    const int sz = in_dtype == XFM_F32 ? 4 : 2;
    // ---- lean variant for d_state == 1 (ss2d_lean.hpp): rows are whole vectors, PPT planes per tile
    if ((f_kind < 0 || f_kind == 3) && N == 1 && L % 4 == 0 && W > 1) {
        const int c = (sz == 2 && L % 8 == 0) ? 8 : 4;
        const int nseg = (L + 64 * c - 1) / (64 * c);
        int ppt = 1;
        for (int q = 1; q <= D; ++q)
            if (D % q == 0 && (int64_t)q * L <= 3200) ppt = q;
        if (f_lg > 0 && D % f_lg == 0) ppt = f_lg;                     // (tuning hook: "lg" field = planes per tile)
        const size_t PL = (size_t)ppt * L;
        const size_t fwd_blk = 6 * PL * sz;
        // registers win for short rows; from ~4 chunks on the LDS accumulators are faster (measured, stage 0: 473 vs 563 us)
        const bool has_reg = nseg <= 2;
        // one-plane tiles beyond the staging span of the regular variants (96 x 96, XFMamba-B at 384^2): the BIG
        // variants -- 16-bit I/O only, accumulators of all 18 chunk rows in registers (ss2d_lean.hpp)
        const bool big = PL > 4096 && ppt == 1 && c == 8 && nseg == 18 && L <= 9216 && f_lg <= 0;
        const int reg_nseg = big ? 18 : ((has_reg && !getenv("XFM_SS2D_LDSACC")) ? nseg : 0);
        const size_t bwd_blk = 8 * PL * sz + (reg_nseg ? 0 : (size_t)8 * L * sizeof(float));
        // (PL <= 4096: the staging registers of the lean kernels cover a tile of 256 threads x 16 elements)
        if (fwd_blk <= kLdsPerCU && bwd_blk <= kLdsPerCU && (PL <= 4096 || big)) {
            out->lg = 6;
            out->items = c;
            out->n_chunks = nseg;
            out->kind = 3;
            out->ppt = ppt;
            out->reg_nseg = reg_nseg;
            out->bc_floats = 0;
            out->psz = L;
            out->lds_fwd_block = fwd_blk;
            out->lds_bwd_block = bwd_blk;
            out->lds_fwd_floats = out->lds_bwd_floats = 0;
            out->waves_fwd = out->waves_bwd = 4;
            const int tiles_pb = D / ppt;
            auto pick_pli = [&](size_t blk) {      // ~2 rounds of resident workgroups
                const int resident = (int)std::min<size_t>(8, std::max<size_t>(1, kLdsPerCU / blk));
                int pli = (int)((int64_t)batch * tiles_pb / ((int64_t)256 * resident * 2));
                if (pli < 1) pli = 1;
                if (pli > tiles_pb) pli = tiles_pb;
                if (f_pli > 0) pli = std::min(f_pli, tiles_pb);
                while (tiles_pb % pli) --pli;
                return pli;
            };
            out->pli = pick_pli(bwd_blk);          // backward: also the span of the dB/dC accumulation
            out->pli_fwd = pick_pli(fwd_blk);
            return XFM_OK;
        }
    }
    if (f_kind == 3) return XFM_ELIMIT;
    // ---- direct variant: chunk = one 16-byte vector, planes in both layouts (ss2d_direct.hpp)
    if ((f_kind < 0 || f_kind == 2) && ((int64_t)L * sz) % 4 == 0) {
        const int c = 16 / sz;
        const int chunks = (L + c - 1) / c;
        int lg = 0;
        while (lg < 6 && (1 << lg) < chunks) ++lg;
        if (f_lg >= 0) lg = f_lg;
        while (lg < 6 && (D % (64 >> lg) || (64 >> lg) * N > 2048)) ++lg;
        const int G = 64 >> lg;
        if (D % G == 0 && G * N <= 2048) {
            const int nseg = (chunks + (1 << lg) - 1) >> lg;
            const size_t pe = (size_t)(L + c - 1) / c * c, GP = (size_t)G * pe;
            const size_t bcf = (N > 1 && nseg == 1 && (size_t)2 * N * L <= 4096) ? up4((size_t)2 * N * L) : 0;
            const size_t fw = up4(GP + up4((size_t)G * N) + bcf);
            const size_t bw = up4(up4((size_t)G * N) + bcf + (size_t)2 * N * L);
            const size_t fwd_blk = (2 * GP * sz + 15) / 16 * 16 + 4 * fw * sizeof(float);
            const size_t bwd_blk = (4 * GP * sz + 15) / 16 * 16 + 2 * GP * sizeof(float) + 4 * bw * sizeof(float);
            if (fwd_blk <= kLdsPerCU && bwd_blk <= kLdsPerCU) {
                out->lg = lg;
                out->items = c;
                out->n_chunks = nseg;
                out->kind = 2;
                out->bc_floats = (int)bcf;
                out->psz = (int)pe;
                out->lds_fwd_floats = fw;
                out->lds_bwd_floats = bw;
                out->lds_fwd_block = fwd_blk;
                out->lds_bwd_block = bwd_blk;
                out->waves_fwd = out->waves_bwd = 4;
                const int tiles_pb = D / G;
                const int resident = (int)std::max<size_t>(1, kLdsPerCU / bwd_blk);
                int pli = (int)((int64_t)batch * tiles_pb / ((int64_t)256 * resident * 2));
                if (pli < 1) pli = 1;
                if (pli > tiles_pb) pli = tiles_pb;
                if (f_pli > 0) pli = std::min(f_pli, tiles_pb);
                while (tiles_pb % pli) --pli;
                out->pli = pli;
                return XFM_OK;
            }
        }
    }
    if (f_kind == 2) return XFM_ELIMIT;
    for (int kind = 1; kind >= 0; --kind) {
        if (f_kind >= 0 && kind != f_kind) continue;
        double best = 1e300;
        bool found = false;
        for (int lg = 0; lg <= 6; ++lg) {
            const int G = 64 >> lg;
            if (D % G) continue;
            if (G * N > 2048) continue;
            if (f_lg >= 0 && lg != f_lg) continue;
            for (int c : kItems2) {
                if (f_items >= 0 && c != f_items) continue;
                const int SL = c << lg;
                const int nseg = (L + SL - 1) / SL;
                const size_t bcf = (N > 1 && nseg == 1 && (size_t)2 * N * L <= 4096) ? up4((size_t)2 * N * L) : 0;
                const size_t tile = (size_t)64 * c + 16 + up4((size_t)G * N) + bcf;
                size_t fwd_blk, bwd_blk, fw, bw;
                int wf, wb;
                if (kind == 1) {
                    fw = up4((size_t)G * PSZ + tile);                  // private y planes + tile + carry (+ bc)
                    bw = up4(tile + (size_t)2 * N * L);                // tile + carry (+ bc) + dB/dC accumulators
                    fwd_blk = ((size_t)G * PSZ + 4 * fw) * sizeof(float);
                    bwd_blk = ((size_t)3 * G * PSZ + 4 * bw) * sizeof(float);
                    wf = wb = 4;
                } else {
                    fw = up4(tile + (size_t)2 * G * PSZ);
                    bw = up4(tile + (size_t)3 * G * PSZ);
                    wf = (int)std::min<size_t>(4, kLdsPerCU / (fw * sizeof(float)));
                    wb = (int)std::min<size_t>(4, kLdsPerCU / (bw * sizeof(float)));
                    if (wf < 1 || wb < 1) continue;
                    fwd_blk = wf * fw * sizeof(float);
                    bwd_blk = wb * bw * sizeof(float);
                }
                if (fwd_blk > kLdsPerCU || bwd_blk > kLdsPerCU) continue;
                const double res_waves = std::min<double>(16.0, (double)(kLdsPerCU / bwd_blk) * wb);
                const double per_route = nseg * (N * (c * 1.0 + lg * 0.8 + 1.5) + c * 0.8 + 4.0);
                const double tiles = (double)batch * D / G;
                const double wave_jobs = kind == 1 ? tiles * 4.0 : tiles;       // route-sweeps vs whole tiles
                const double job = kind == 1 ? per_route : 4.0 * per_route;
                const double rounds = std::max(1.0, wave_jobs / (256.0 * res_waves));
                double est = job * rounds * (1.0 + 2.0 / res_waves);
                if (!(nseg == 1 || G == 1)) est *= 3.0;        // rows of a tile not one HBM run: scalar staging
                if (est < best * 0.999) {
                    best = est;
                    found = true;
                    out->lg = lg;
                    out->items = c;
                    out->n_chunks = nseg;
                    out->kind = kind;
                    out->bc_floats = (int)bcf;
                    out->lds_fwd_floats = fw;
                    out->lds_bwd_floats = bw;
                    out->lds_fwd_block = fwd_blk;
                    out->lds_bwd_block = bwd_blk;
                    out->waves_fwd = wf;
                    out->waves_bwd = wb;
                }
            }
        }
        if (found) {
            out->pli = 1;
            if (kind == 1) {
                // tiles per workgroup: aim at ~2 rounds of resident workgroups so the dB/dC flush is amortised
                const int G = 64 >> out->lg;
                const int tiles_pb = D / G;
                const int resident = (int)std::max<size_t>(1, kLdsPerCU / out->lds_bwd_block);
                int pli = (int)((int64_t)batch * tiles_pb / ((int64_t)256 * resident * 2));
                if (pli < 1) pli = 1;
                if (pli > tiles_pb) pli = tiles_pb;
                if (f_pli > 0) pli = std::min(f_pli, tiles_pb);
                while (tiles_pb % pli) --pli;                              // equal work per workgroup
                out->pli = pli;
            }
            return XFM_OK;
        }
    }
    return XFM_ELIMIT;
}

int ss2d_launch_raw(const void *fn, const SS2DArgs &a, const Plan2 &pl, bool bwd, hipStream_t s) {
    const int G = 64 >> a.lg_lpr;
    const int tiles_pb = a.p.d_inner / G;
    const size_t lds = bwd ? pl.lds_bwd_block : pl.lds_fwd_block;
    unsigned grid;
    dim3 block;
    if (pl.kind >= 1) {
        grid = (unsigned)((int64_t)a.p.batch * ((tiles_pb + pl.pli - 1) / pl.pli));
        block = dim3(256);
    } else {
        const int wpb = bwd ? pl.waves_bwd : pl.waves_fwd;
        grid = (unsigned)(((int64_t)a.p.batch * tiles_pb + wpb - 1) / wpb);
        block = dim3(64 * wpb);
    }
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    SS2DArgs args = a;
    void *kargs[] = {&args};
    const hipError_t e = hipLaunchKernel(fn, dim3(grid), block, kargs, lds, s);
    if (e != hipSuccess) {
        set_last_hip_error(e);
        return XFM_ELAUNCH;
    }
    return check_launch();
}

int ss2d_launch_lean(const void *fn, const SS2DArgs &a, const Plan2 &pl, bool bwd, hipStream_t s) {
    const xfm_ss2d_params_t &p = a.p;
    LeanArgs la;
    la.x = p.x; la.dts = p.dts; la.Bs = p.Bs; la.Cs = p.Cs;
    la.A = p.A; la.D = p.D; la.bias = p.delta_bias;
    la.y = p.y; la.chk = p.chk; la.dy = p.dy; la.dx = p.dx; la.ddts = p.ddts;
    la.dBs = p.dBs; la.dCs = p.dCs; la.dA = p.dA; la.dD = p.dD; la.dbias = p.ddelta_bias;
    la.batch = p.batch; la.D_ = p.d_inner; la.H = p.H; la.W = p.W; la.L = p.H * p.W;
    la.nseg = pl.n_chunks; la.ppt = pl.ppt; la.pli = bwd ? pl.pli : pl.pli_fwd; la.softplus = p.delta_softplus;
    la.magicW = a.magicW;
    la.magicL = (uint32_t)((0x100000000ull + la.L - 1) / la.L);
    la.magicH = (uint32_t)((0x100000000ull + la.H - 1) / la.H);
    la.dbg = a.dbg;
    const size_t lds = bwd ? pl.lds_bwd_block : pl.lds_fwd_block;
    const unsigned grid = (unsigned)((int64_t)p.batch * (p.d_inner / pl.ppt / la.pli));
    if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    void *kargs[] = {&la};
    const hipError_t e = hipLaunchKernel(fn, dim3(grid), dim3(256), kargs, lds, s);
    if (e != hipSuccess) {
        set_last_hip_error(e);
        return XFM_ELAUNCH;
    }
    return check_launch();
}

int ss2d_l3_run(const xfm_ss2d_params_t *p, bool bwd, hipStream_t s, float *ws, size_t ws_bytes);   // ss2d_l3.hip: wide maps
size_t ss2d_l3_ws_bytes(const xfm_ss2d_params_t *p);
int ss2d_l3_nseg(int batch, int D, int H, int W, int N, int in_dtype);
int ss2d_l3_dtfused_rank(int batch, int D, int H, int W, int N, int R, int in_dtype);
int ss2d_w_covers(int batch, int D, int H, int W, int N, int in_dtype);                                // ss2d_l3.hip / ss2d_w.hpp
static bool ss2d_forced() {
    static const bool f = getenv("XFM_SS2D_FORCE") != nullptr;      // tuning hook, read once per process
    return f;
}

static int run2(const xfm_ss2d_params_t *p, bool bwd, void *stream, float *ws = nullptr, size_t ws_bytes = 0) {
    if (!p || !p->x || !p->Bs || !p->Cs || !p->A || !p->D || !p->delta_bias) return XFM_EINVAL;
    if (p->delta_softplus == 3) {
        // dt_proj inside the kernel: only the wide-map kernels of ss2d_l3.hip have it (xfm_ss2d_dtfused_rank)
        if (!p->xrt || !p->dt_w || p->dt_rank_p <= 0) return XFM_EINVAL;
        if (!bwd && !p->y) return XFM_EINVAL;
        if (bwd && (!p->dy || !p->dx || !p->ddts || !p->dBs || !p->dCs || !p->dA || !p->dD || !p->ddelta_bias)) return XFM_EINVAL;
        if (p->out_dtype != XFM_F32) return XFM_EDTYPE;
        return ss2d_l3_run(p, bwd, (hipStream_t)stream, ws, ws_bytes);
    }
    if (!p->dts) return XFM_EINVAL;
    if (!bwd && !p->y) return XFM_EINVAL;
    if (bwd && (!p->dy || !p->dx || !p->ddts || !p->dBs || !p->dCs || !p->dA || !p->dD || !p->ddelta_bias))
        return XFM_EINVAL;
    if (p->in_dtype < 0 || p->in_dtype > 2) return XFM_EDTYPE;
    if (p->out_dtype != XFM_F32) return XFM_EDTYPE;          // the fused core always emits fp32 ("oflex")
    if (!ss2d_forced()) {
        const int rc3 = ss2d_l3_run(p, bwd, (hipStream_t)stream, ws, ws_bytes);
        if (rc3 != XFM_ELIMIT) return rc3;
    }
    if (p->bc_f32) return XFM_EINVAL;                        // fp32 B / C rows beside 16-bit planes: ss2d_w.hpp only (xfm_ss2d_bc_f32)
    Plan2 pl;
    int rc = plan_ss2d(p->batch, p->d_inner, p->H, p->W, p->dstate, p->in_dtype, &pl);
    if (rc) return rc;
    if (pl.n_chunks > 1 && !p->chk) return XFM_EINVAL;
    SS2DArgs a;
    a.p = *p;
    a.lg_lpr = pl.lg;
    a.n_chunks = pl.n_chunks;
    a.PW = p->W | 1;
    a.PSZ = pl.psz;
    a.lds_floats_per_wave = (int)(bwd ? pl.lds_bwd_floats : pl.lds_fwd_floats);
    a.waves_per_block = bwd ? pl.waves_bwd : pl.waves_fwd;
    a.kind = pl.kind;
    a.pli = pl.pli;
    a.bc_floats = pl.bc_floats;
    static const int env_dbg = [] { const char *e = getenv("XFM_SS2D_DBG"); return e ? atoi(e) : 0; }();
    a.dbg = env_dbg;
    if (env_dbg) {
        static int once = 0;
        if (!once++) fprintf(stderr, "[xfm] ss2d timing switches dbg=%d kind=%d items=%d chunks=%d ppt=%d pli=%d lds=%zu/%zu\n", a.dbg, pl.kind, pl.items, pl.n_chunks, pl.ppt, pl.pli, pl.lds_fwd_block, pl.lds_bwd_block);
    }
    a.magicW = (uint32_t)((0x100000000ull + p->W - 1) / p->W);
    hipStream_t s = (hipStream_t)stream;
    switch (p->in_dtype) {
        case XFM_F32: return ss2d_dispatch<float, float>(a, pl, bwd, s);
        case XFM_F16: return ss2d_dispatch<f16_t, float>(a, pl, bwd, s);
        case XFM_BF16: return ss2d_dispatch<bf16_t, float>(a, pl, bwd, s);
    }
    return XFM_EDTYPE;
}

}  // namespace xfm

extern "C" {
int xfm_ss2d_plan(int batch, int d_inner, int H, int W, int dstate, int in_dtype, xfm_scan_plan_t *plan) {
    if (!plan) return XFM_EINVAL;
    if (in_dtype < 0 || in_dtype > 2) return XFM_EDTYPE;
    xfm::Plan2 pl;
    const int rc = xfm::plan_ss2d(batch, d_inner, H, W, dstate, in_dtype, &pl);
    if (rc) return rc;
    plan->lanes_per_row = 1 << pl.lg;
    plan->items = pl.items;
    plan->n_chunks = pl.n_chunks;
    if (!xfm::ss2d_forced())                       // the wide-map kernels (ss2d_l3.hip) index chk by their own geometry
        plan->n_chunks = std::max(plan->n_chunks, xfm::ss2d_l3_nseg(batch, d_inner, H, W, dstate, in_dtype));
    return XFM_OK;
}
int xfm_ss2d_dtfused_rank(int batch, int d_inner, int H, int W, int dstate, int dt_rank, int in_dtype) {
    if (xfm::ss2d_forced()) return 0;
    return xfm::ss2d_l3_dtfused_rank(batch, d_inner, H, W, dstate, dt_rank, in_dtype);
}
int xfm_ss2d_bc_f32(int batch, int d_inner, int H, int W, int dstate, int in_dtype) {
    if (xfm::ss2d_forced()) return 0;
    return xfm::ss2d_w_covers(batch, d_inner, H, W, dstate, in_dtype);
}
int xfm_ss2d_fwd(const xfm_ss2d_params_t *p, void *stream) { return xfm::run2(p, false, stream); }
int xfm_ss2d_bwd(const xfm_ss2d_params_t *p, void *stream) { return xfm::run2(p, true, stream); }
size_t xfm_ss2d_bwd_ws_bytes(const xfm_ss2d_params_t *p) {
    if (!p || xfm::ss2d_forced() || getenv("XFM_L3_ATOMICS")) return 0;      // (A/B switch of the tests: read per call)
    const char *e = getenv("XFM_SS2D_L3");
    return (e && e[0] == '0') ? 0 : xfm::ss2d_l3_ws_bytes(p);
}
int xfm_ss2d_bwd_ws(const xfm_ss2d_params_t *p, void *workspace, size_t workspace_bytes, void *stream) {
    return xfm::run2(p, true, stream, (float *)workspace, workspace_bytes);
}
}
