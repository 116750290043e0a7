#!/usr/bin/env python
"""Run one kernel many times while a second process does the same on the same GPU, and compare every result with the first:
memory contention and time slicing between two processes expose waits that are too short (a tile read before it landed) and
other timing assumptions that a single process on an idle GPU never violates.  tools/stress2.py <op> [iterations]; launch two."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    from xfmamba_amd import _lib
    lib = _lib.lib()
    op = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    g = torch.Generator().manual_seed(1)
    if op.startswith("dtbwd"):
        B, D, R, L = {"dtbwd0": (64, 96, 6, 3136), "dtbwd1": (64, 192, 12, 784), "dtbwd2": (64, 192, 6, 3136)}[op]
        ddts = torch.randn(B, 4, D, L, generator=g).bfloat16().cuda()
        xr = torch.randn(B, 4, R, L, generator=g).bfloat16().cuda()
        w = (torch.randn(4, D, R, generator=g) * R ** -0.5).bfloat16().cuda()
        dxr = torch.empty(B, 4, R, L, dtype=torch.bfloat16, device="cuda")
        dw = torch.zeros(4, D, R, device="cuda")

        def run():
            dxr.fill_(float("nan"))
            dw.zero_()
            _lib.check(lib.xfm_ss2d_dt_proj_bwd_mfma(ddts.data_ptr(), xr.data_ptr(), w.data_ptr(), dxr.data_ptr(), dw.data_ptr(), B, D,
                                                     R, L, _lib.stream_ptr()), "dt_proj_bwd_mfma")
            return dxr.clone(), dw.clone()
    elif op in ("l3_56", "l3_28", "w_48", "w_24", "chan14", "chan7"):
        # the fused SS2D cores at the trunk's shapes, forward + backward through the autograd nodes: y / dx are exact (fixed-order
        # sums), the parameter gradients come from atomics (tolerance)
        from xfmamba_amd.ss2d import ss2d_xproj_core_fn
        from xfmamba_amd.ss2d_chan import ss2d_chan_fn
        # (l3_* / w_*: the wide-map kernels -- csrc/ss2d_w.hpp since round 6; w_48 / w_24: XFMamba-B's widths at two samples, the
        #  launch in which a 16-byte buffer store's data registers were overwritten behind it with a second workgroup on the CU)
        B, D, HW, R = {"l3_56": (64, 96, 56, 6), "l3_28": (64, 192, 28, 12), "w_48": (2, 512, 48, 32), "w_24": (2, 1024, 24, 64),
                       "chan14": (64, 384, 14, 24), "chan7": (64, 768, 7, 48)}[op]
        L = HW * HW
        x = torch.randn(B, D, L, generator=g).bfloat16().cuda().requires_grad_()
        xw = (torch.randn(4, R + 2, D, generator=g) * D ** -0.5).cuda().requires_grad_()
        dtw = (torch.randn(4, D, R, generator=g) * R ** -0.5).cuda().requires_grad_()
        A = (-torch.rand(4 * D, 1, generator=g) - 0.2).cuda().requires_grad_()
        Dp = torch.randn(4 * D, generator=g).cuda().requires_grad_()
        bias = (torch.randn(4 * D, generator=g) * 0.5).cuda().requires_grad_()
        gy = torch.randn(B, D, L, generator=g).cuda()
        fn = ss2d_xproj_core_fn if op.startswith(("l3", "w_")) else ss2d_chan_fn

        def run():
            for t in (x, xw, dtw, A, Dp, bias):
                t.grad = None
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = fn(x, xw, dtw, A, Dp, bias, HW, HW)
            y.backward(gy)
            run.segs = [("dxw", xw.numel()), ("ddtw", dtw.numel()), ("dA", A.numel()), ("dD", Dp.numel()), ("dbias", bias.numel())]
            extra = []
            if os.environ.get("XFM_DBG_KEEP"):
                from xfmamba_amd import ss2d as _ss
                extra = [_ss._DBG["dBs"].float().flatten(), _ss._DBG["dCs"].float().flatten()]
                run.segs += [("dBs", extra[0].numel()), ("dCs", extra[1].numel())]
            return y.detach().clone(), x.grad.clone(), torch.cat([t.grad.float().flatten() for t in (xw, dtw, A, Dp, bias)] + extra)
    elif op in ("wgradpp", "wgradpp0"):
        # the plane x plane weight-gradient product of the wide-map SS2D node: dW (56, 192) = sum d x_dbl (x) x at 28 x 28
        # (wgradpp0: (32, 96) at 56 x 56)
        from xfmamba_amd.proj import wgrad_mfma
        B, M, N, L = (64, 56, 192, 784) if op == "wgradpp" else (64, 32, 96, 3136)
        ap = torch.randn(B, M, L, generator=g).bfloat16().cuda()
        bp = torch.randn(B, N, L, generator=g).bfloat16().cuda()

        def run():
            dw = wgrad_mfma(ap, True, bp, True)
            assert dw is not None
            return ap[:1, :1, :8].clone(), ap[:1, :1, :8].clone(), dw.clone().flatten()
    elif op in ("gemm3", "wgrad", "wgradx"):
        T, C, O = 12544, 1536, 384
        xt = torch.randn(T, C, generator=g).bfloat16().cuda()
        wt = (torch.randn(O, C, generator=g) * C ** -0.5).bfloat16().cuda()
        yt = torch.empty(T, O, dtype=torch.bfloat16, device="cuda")
        dyt = torch.randn(T, O, generator=g).bfloat16().cuda()
        dwt = torch.zeros(O, C, device="cuda")
        xm = torch.randn(64, 56, 56, 96, generator=g).bfloat16().cuda()
        dym = torch.randn(64, 28, 28, 192, generator=g).bfloat16().cuda()
        dwm = torch.zeros(192, 9 * 96, device="cuda")

        def run():
            if op == "gemm3":
                _lib.check(lib.xfm_tokens_gemm2(xt.data_ptr(), wt.data_ptr(), None, yt.data_ptr(), None, None, T, C, O, 0, 0,
                                                _lib.stream_ptr()), "gemm2")
                return yt.clone(), yt.clone()[:1], torch.zeros(1, device="cuda")
            if op == "wgrad":
                dwt.zero_()
                _lib.check(lib.xfm_wgrad(dyt.data_ptr(), xt.data_ptr(), dwt.data_ptr(), O, C, 1, T, 0, 0, 0, 0, _lib.stream_ptr()), "wgrad")
                return yt[:1].clone(), yt[:1].clone(), dwt.clone().flatten()
            dwm.zero_()
            _lib.check(lib.xfm_conv3x3s2_tokens_bwd_weight_x(dym.data_ptr(), xm.data_ptr(), dwm.data_ptr(), 64, 56, 56, 96, 192,
                                                             _lib.stream_ptr()), "wgradx")
            return yt[:1].clone(), yt[:1].clone(), dwm.clone().flatten()
    else:
        raise SystemExit("unknown op")
    if op.startswith("dtbwd"):
        run0 = run

        def run():
            a, b = run0()
            return a, a[:1], b.flatten()
    ref = run()
    torch.cuda.synchronize()
    bad = 0
    t0 = time.time()
    for i in range(n):
        out = run()
        torch.cuda.synchronize()
        if op.startswith("chan"):                      # (their dx sums four routes with LDS float atomics: order-dependent rounding)
            d1 = float((out[1].float() - ref[1].float()).abs().max())
            e0 = not bool(torch.equal(out[0], ref[0])) or not (d1 <= 1.6e-2 * float(ref[1].float().abs().max()))
        else:
            e0 = not bool(torch.equal(out[0], ref[0])) or not bool(torch.equal(out[1], ref[1]))
        e1 = float((out[2] - ref[2]).abs().max()) > 2e-3 * float(ref[2].abs().max()) + 1e-6 or not bool(torch.isfinite(out[2]).all())
        bad1 = locals().get("bad1", 0) + (1 if e1 else 0)
        if e0 or e1:
            bad += 1
            if bad <= 3:
                d = (out[0].float() - ref[0].float())
                print(f"[{os.getpid()}] iteration {i}: exact outputs differ at {int((d != 0).sum() + torch.isnan(d).sum())} elements "
                      f"(nan {int(torch.isnan(out[0].float()).sum())}), summed outputs off {e1}", flush=True)
                o0 = 0
                for name, cnt in getattr(run, "segs", []):          # which of the summed outputs, where, by how much
                    dd = (out[2][o0:o0 + cnt] - ref[2][o0:o0 + cnt]).abs()
                    if name in ("dBs", "dCs"):
                        own = float(ref[2][o0:o0 + cnt].abs().max())
                        idx = (dd > 1e-3 * own).nonzero().flatten()
                        if idx.numel():
                            L_ = 784 if op == "l3_28" else 3136
                            bk = sorted(set((idx // L_).tolist()))
                            pos = (idx % L_)
                            print(f"[{os.getpid()}]   {name}: {idx.numel()} elements off by > 1e-3 of {own:.4g}; (b*4+k) rows {bk[:8]}; positions "
                                  f"{int(pos.min())}..{int(pos.max())}; max diff {float(dd.max()):.4g}; sample diffs {dd[idx[:6]].tolist()}", flush=True)
                        o0 += cnt
                        continue
                    if float(dd.max()) > 2e-3 * float(ref[2].abs().max()) + 1e-6:
                        idx = (dd > 2e-3 * float(ref[2].abs().max()) + 1e-6).nonzero().flatten()
                        print(f"[{os.getpid()}]   {name}: {idx.numel()} of {cnt} off, first at {idx[:8].tolist()}, max diff {float(dd.max()):.4g} "
                              f"(scale {float(ref[2][o0:o0 + cnt].abs().max()):.4g})", flush=True)
                    o0 += cnt
    print(f"[{os.getpid()}] {op}: {bad} of {n} runs differ from the first ({time.time() - t0:.1f} s); summed outputs off in {locals().get('bad1', 0)}", flush=True)


if __name__ == "__main__":
    main()
