#!/usr/bin/env python
"""Micro-benchmark of the hand-written kernels at the BASELINE shapes (XFMamba-T, batch 32 x 2 views).

    python tools/kbench.py [--only ss2d] [--force kind,lg,items,pli]   (XFM_SS2D_FORCE plan override)

Prints per-kernel time (HIP events on the launch stream, median of repeats), algorithmic GB/s
(SURVEY.md 8(d) byte counts) and the fraction of the 8 TB/s HBM peak.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def time_raw(cfn, params, reps=30, warm=3):
    """Back-to-back launches of one C-ABI entry point between two events (no Python work in between)."""
    import ctypes
    from xfmamba_amd import _lib
    st = _lib.stream_ptr()
    for _ in range(warm):
        _lib.check(cfn(ctypes.byref(params), st), "kbench")
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        cfn(ctypes.byref(params), st)
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--force", default=None)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    if a.force:
        os.environ["XFM_SS2D_FORCE"] = a.force
    from xfmamba_amd import _lib
    from xfmamba_amd.ss2d import SS2DCoreHip
    from xfmamba_amd.dwconv import DWConv3x3SiLUHip
    import ctypes
    dt = dict(bf16=torch.bfloat16, fp32=torch.float32)[a.dtype]
    dev = "cuda"
    B = a.batch
    shapes = [("stage0", B, 96, 56, 1), ("stage1", B, 192, 28, 1), ("stage2", B, 384, 14, 1), ("stage3", B, 768, 7, 1),
              ("deep", B // 2, 1536, 7, 16)]
    print(f"{'kernel':28s} {'us':>9s} {'GB/s':>8s} {'%HBM':>6s}  plan")
    if not a.only or a.only == "ln2d":
        lib = _lib.lib()
        for name, Bt, C, H in (("stage0", B, 96, 56), ("stage1", B, 192, 28), ("stage2", B, 384, 14), ("stage3", B, 768, 7)):
            L = H * H
            x = torch.randn(Bt, C, L, device=dev)
            y = torch.empty(Bt, C, L, device=dev, dtype=dt)
            dy = torch.randn(Bt, C, L, device=dev).to(dt)
            dx = torch.empty_like(x)
            w, bb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            mean, rstd = torch.empty(Bt * L, device=dev), torch.empty(Bt * L, device=dev)
            dw, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
            st, cy = _lib.stream_ptr(), _lib.dtype_code(dt)

            def raw(fn, reps=30):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / reps

            t = raw(lambda: lib.xfm_layernorm2d_fwd(x.data_ptr(), w.data_ptr(), bb.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                                    rstd.data_ptr(), Bt, C, L, 1e-5, 0, cy, st))
            nb = x.numel() * (4 + y.element_size())
            print(f"{'ln2d_fwd ' + name:28s} {t:9.1f} {nb / t / 1e3:8.1f} {nb / t / 1e3 / 80:6.2f}")
            t = raw(lambda: lib.xfm_layernorm2d_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                    dx.data_ptr(), dw.data_ptr(), db.data_ptr(), Bt, C, L, 0, cy, st))
            nb = x.numel() * (8 + y.element_size())
            print(f"{'ln2d_bwd(dx+wb) ' + name:28s} {t:9.1f} {nb / t / 1e3:8.1f} {nb / t / 1e3 / 80:6.2f}")
    if not a.only or a.only == "dtproj":
        lib = _lib.lib()
        for name, Bt, D, H in (("stage0", B, 96, 56), ("stage1", B, 192, 28), ("stage2", B, 384, 14)):
            R, L = D // 16, H * H
            xr = torch.randn(Bt, 4, R, L, device=dev).to(dt)
            w = torch.randn(4, D, R, device=dev)
            out = torch.empty(Bt, 4, D, L, device=dev, dtype=dt)
            st, code = _lib.stream_ptr(), _lib.dtype_code(dt)
            fn = lambda: lib.xfm_ss2d_dt_proj_fwd(xr.data_ptr(), w.data_ptr(), None, out.data_ptr(), Bt, D, R, L, code, st)
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e3 / 30
            nb = out.numel() * out.element_size()
            print(f"{'dt_proj_fwd ' + name:28s} {t:9.1f} {nb / t / 1e3:8.1f} {nb / t / 1e3 / 80:6.2f}")
            rp = lib.xfm_ss2d_dt_proj_mfma_rp(D, R, L)
            if rp and dt == torch.bfloat16:
                bias = 0.1 * torch.rand(4 * D, device=dev)
                wp = w.to(dt).contiguous()
                out2 = torch.empty_like(out)
                lib.xfm_ss2d_dt_proj_fwd(xr.data_ptr(), w.to(dt).float().data_ptr(), bias.data_ptr(), out.data_ptr(), Bt, D, R, L, code, st)
                fn2 = lambda: lib.xfm_ss2d_dt_proj_fwd_mfma(xr.data_ptr(), wp.data_ptr(), bias.data_ptr(), out2.data_ptr(), Bt, D, R, L, st)
                fn2()
                torch.cuda.synchronize()
                err = float((out2.float() - out.float()).abs().max() / out.float().abs().max())
                for _ in range(3):
                    fn2()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(30):
                    fn2()
                e1.record()
                torch.cuda.synchronize()
                t = e0.elapsed_time(e1) * 1e3 / 30
                print(f"{'dt_proj_mfma ' + name:28s} {t:9.1f} {nb / t / 1e3:8.1f} {nb / t / 1e3 / 80:6.2f}  max rel diff vs VALU kernel {err:.2e}")
                ddts = torch.randn_like(out)
                dxr = torch.empty_like(xr)
                dwa = torch.zeros(4, D, R, device=dev)
                fb = lambda: lib.xfm_ss2d_dt_proj_bwd_mfma(ddts.data_ptr(), xr.data_ptr(), wp.data_ptr(), dxr.data_ptr(), dwa.data_ptr(), Bt, D, R, L, st)
                wT = wp.transpose(1, 2).contiguous()
                ft = lambda: (torch.matmul(wT, ddts), torch.bmm(ddts.view(Bt * 4, D, L), xr.view(Bt * 4, R, L).transpose(1, 2)).view(Bt, 4, D, R).sum(0))
                for nm, f in (("dt_proj_bwd_mfma", fb), ("dt_proj_bwd_torch", ft)):
                    for _ in range(3):
                        f()
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(30):
                        f()
                    e1.record()
                    torch.cuda.synchronize()
                    t = e0.elapsed_time(e1) * 1e3 / 30
                    print(f"{nm + ' ' + name:28s} {t:9.1f} {2 * nb / t / 1e3:8.1f}")
    if not a.only or a.only in "rowscan":
        from xfmamba_amd import csms6s
        for name, Bt, KD, N in (("rowscan fusion", B // 2, 6144, 16), ("rowscan stage3", B, 3072, 1)):
            L, K = 49, 4
            u = torch.randn(Bt, KD, L, device=dev).to(dt)
            delta = (0.5 * torch.rand(Bt, KD, L, device=dev)).to(dt)
            Bm = torch.randn(Bt, K, N, L, device=dev).to(dt)
            Cm = torch.randn(Bt, K, N, L, device=dev).to(dt)
            A = -torch.rand(KD, N, device=dev) - 0.1
            Dp = torch.randn(KD, device=dev)
            bias = 0.1 * torch.rand(KD, device=dev)
            p = _lib.ScanParams()
            csms6s._fill_common(p, u, delta, A, Bm, Cm, Dp, bias, True, torch.float32)
            out = torch.empty(Bt, KD, L, device=dev)
            p.out, p.out_batch_stride, p.out_d_stride = out.data_ptr(), out.stride(0), out.stride(1)
            plan = _lib.scan_plan(Bt, KD, L, N, K)
            xs = torch.empty(Bt, KD, max(plan.n_chunks, 1), N, device=dev)
            p.x = xs.data_ptr()
            dout = torch.randn_like(out)
            du, dd = torch.empty_like(u), torch.empty_like(delta)
            dA, dBm, dCm = torch.zeros_like(A), torch.zeros(Bt, K, N, L, device=dev), torch.zeros(Bt, K, N, L, device=dev)
            dD, db = torch.zeros_like(Dp), torch.zeros_like(bias)
            p.dout, p.dout_batch_stride, p.dout_d_stride = dout.data_ptr(), dout.stride(0), dout.stride(1)
            p.du, p.ddelta = du.data_ptr(), dd.data_ptr()
            p.dA, p.dB, p.dC, p.dD, p.ddelta_bias = dA.data_ptr(), dBm.data_ptr(), dCm.data_ptr(), dD.data_ptr(), db.data_ptr()
            isz = u.element_size()
            fb, bb = Bt * KD * L * (2 * isz + 4), Bt * KD * L * (4 * isz + 4)
            t = time_raw(_lib.lib().xfm_selective_scan_fwd, p)
            print(f"{'scan_fwd ' + name:28s} {t:9.1f} {fb / t / 1e3:8.1f} {fb / t / 1e3 / 80:6.2f}")
            t = time_raw(_lib.lib().xfm_selective_scan_bwd, p)
            print(f"{'scan_bwd ' + name:28s} {t:9.1f} {bb / t / 1e3:8.1f} {bb / t / 1e3 / 80:6.2f}")
    for name, Bt, D, H, N in shapes:
        if a.only and a.only not in ("ss2d", "dwconv") and a.only not in name:
            continue
        do_ss2d = a.only != "dwconv"
        L = H * H
        x = torch.randn(Bt, D, L, device=dev).to(dt)
        dts = (0.5 * torch.rand(Bt, 4, D, L, device=dev)).to(dt)
        Bs = torch.randn(Bt, 4, N, L, device=dev).to(dt)
        Cs = torch.randn(Bt, 4, N, L, device=dev).to(dt)
        A = -torch.rand(4 * D, N, device=dev) - 0.1
        Dp = torch.randn(4 * D, device=dev)
        bias = 0.1 * torch.rand(4 * D, device=dev)
        plan = _lib.ScanPlan()
        _lib.lib().xfm_ss2d_plan(Bt, D, H, H, N, _lib.dtype_code(dt), ctypes.byref(plan))
        ptxt = f"lpr={plan.lanes_per_row} items={plan.items} chunks={plan.n_chunks}"
        isz = x.element_size()
        fb = Bt * D * L * (5 * isz + 4) + 2 * Bt * 4 * N * L * isz
        bb = Bt * D * L * (10 * isz + 4) + 2 * Bt * 4 * N * L * (isz + 4)
        from xfmamba_amd import ss2d as S2
        chk = torch.empty((Bt, 4, D, max(plan.n_chunks, 1), N), dtype=torch.float32, device=dev)
        y = torch.empty((Bt, D, L), dtype=torch.float32, device=dev)
        gy = torch.randn_like(y)
        dx, ddts = torch.empty_like(x), torch.empty_like(dts)
        dBs = torch.zeros(Bs.shape, dtype=torch.float32, device=dev)
        dCs = torch.zeros_like(dBs)
        dA, dD, dbias = torch.zeros_like(A), torch.zeros_like(Dp), torch.zeros_like(bias)
        p = _lib.SS2DParams()
        S2._fill(p, x, dts, A, Bs, Cs, Dp, bias, H, H, torch.float32, chk)
        p.y, p.dy, p.dx, p.ddts = y.data_ptr(), gy.data_ptr(), dx.data_ptr(), ddts.data_ptr()
        p.dBs, p.dCs, p.dA, p.dD, p.ddelta_bias = dBs.data_ptr(), dCs.data_ptr(), dA.data_ptr(), dD.data_ptr(), dbias.data_ptr()
        if do_ss2d:
            t = time_raw(_lib.lib().xfm_ss2d_fwd, p)
            print(f"{'ss2d_fwd ' + name:28s} {t:9.1f} {fb / t / 1e3:8.1f} {fb / t / 1e3 / 80:6.2f}  {ptxt}")
            t = time_raw(_lib.lib().xfm_ss2d_bwd, p)
            print(f"{'ss2d_bwd ' + name:28s} {t:9.1f} {bb / t / 1e3:8.1f} {bb / t / 1e3 / 80:6.2f}")
        if (N == 1 or a.only == "dwconv") and a.only != "ss2d":
            lib = _lib.lib()
            w = torch.randn(D, 1, 3, 3, device=dev)
            y4, g4, dx4 = torch.empty_like(x), torch.randn_like(x), torch.empty_like(x)
            dw, code = torch.zeros(D * 9, device=dev), _lib.dtype_code(dt)
            st = _lib.stream_ptr()

            def raw(fn, reps=30):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) * 1e3 / reps

            t = raw(lambda: lib.xfm_dwconv3x3_fwd(x.data_ptr(), w.data_ptr(), None, y4.data_ptr(), Bt, D, H, H, code, 1, st))
            nb = 2 * x.numel() * isz
            print(f"{'dwconv_fwd ' + name:28s} {t:9.1f} {nb / t / 1e3:8.1f} {nb / t / 1e3 / 80:6.2f}")
            t = raw(lambda: lib.xfm_dwconv3x3_bwd(x.data_ptr(), w.data_ptr(), None, g4.data_ptr(), dx4.data_ptr(),
                                                  dw.data_ptr(), None, Bt, D, H, H, code, 1, st))
            nb = 3 * x.numel() * isz
            print(f"{'dwconv_bwd ' + name:28s} {t:9.1f} {nb / t / 1e3:8.1f} {nb / t / 1e3 / 80:6.2f}")


if __name__ == "__main__":
    main()
