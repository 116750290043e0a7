// ss2d_fused.hip -- host side of the fused SS2D core (planning, launch, C ABI).  Kernels: ss2d_kernels.hpp.
#include <cstdio>
#include <cstdlib>

#include "ss2d_kernels.hpp"

namespace xfm {

static const int kItems2[] = {4, 7, 9, 13};
static const size_t kLdsPerCU = 160 * 1024;
static inline size_t up4(size_t v) { return (v + 3) & ~(size_t)3; }

// One decomposition serves forward and backward (they share the chunk-state layout `chk`).
static int plan_ss2d(int batch, int D, int H, int W, int N, Plan2 *out) {
    if (batch <= 0 || D <= 0 || H <= 0 || W <= 0 || N <= 0) return XFM_EINVAL;
    if (N > 256 || (int64_t)H * W >= 65536) return XFM_ELIMIT;
    const int L = H * W;
    const size_t PSZ = up4((size_t)H * (W | 1));
    // tuning hook (tools/kbench.py): XFM_SS2D_FORCE="kind,lg,items,pli" restricts the search
    int f_kind = -1, f_lg = -1, f_items = -1, f_pli = -1;
    if (const char *env = getenv("XFM_SS2D_FORCE")) sscanf(env, "%d,%d,%d,%d", &f_kind, &f_lg, &f_items, &f_pli);
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
    if (pl.kind == 1) {
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

static int run2(const xfm_ss2d_params_t *p, bool bwd, void *stream) {
    if (!p || !p->x || !p->dts || !p->Bs || !p->Cs || !p->A || !p->D || !p->delta_bias) return XFM_EINVAL;
    if (!bwd && !p->y) return XFM_EINVAL;
    if (bwd && (!p->dy || !p->dx || !p->ddts || !p->dBs || !p->dCs || !p->dA || !p->dD || !p->ddelta_bias))
        return XFM_EINVAL;
    if (p->in_dtype < 0 || p->in_dtype > 2) return XFM_EDTYPE;
    if (p->out_dtype != XFM_F32) return XFM_EDTYPE;          // the fused core always emits fp32 ("oflex")
    Plan2 pl;
    int rc = plan_ss2d(p->batch, p->d_inner, p->H, p->W, p->dstate, &pl);
    if (rc) return rc;
    if (pl.n_chunks > 1 && !p->chk) return XFM_EINVAL;
    SS2DArgs a;
    a.p = *p;
    a.lg_lpr = pl.lg;
    a.n_chunks = pl.n_chunks;
    a.PW = p->W | 1;
    a.PSZ = (int)up4((size_t)p->H * a.PW);
    a.lds_floats_per_wave = (int)(bwd ? pl.lds_bwd_floats : pl.lds_fwd_floats);
    a.waves_per_block = bwd ? pl.waves_bwd : pl.waves_fwd;
    a.kind = pl.kind;
    a.pli = pl.pli;
    a.bc_floats = pl.bc_floats;
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
int xfm_ss2d_plan(int batch, int d_inner, int H, int W, int dstate, xfm_scan_plan_t *plan) {
    if (!plan) return XFM_EINVAL;
    xfm::Plan2 pl;
    const int rc = xfm::plan_ss2d(batch, d_inner, H, W, dstate, &pl);
    if (rc) return rc;
    plan->lanes_per_row = 1 << pl.lg;
    plan->items = pl.items;
    plan->n_chunks = pl.n_chunks;
    return XFM_OK;
}
int xfm_ss2d_fwd(const xfm_ss2d_params_t *p, void *stream) { return xfm::run2(p, false, stream); }
int xfm_ss2d_bwd(const xfm_ss2d_params_t *p, void *stream) { return xfm::run2(p, true, stream); }
}
