#!/usr/bin/env python
"""Micro-benchmark: channel-lane fused SS2D core (xfm_ss2dc_fwd/_bwd) next to the lean chunk-scan chain
(dt_proj kernels + xfm_ss2d_fwd/_bwd) on the short-map shapes of the trunk.  HIP-event timing of whole autograd
nodes (forward, backward) on the current stream, median of repeats.

    python tools/chanbench.py [--batch 64]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def med(fn, reps=20, warm=3):
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
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    from xfmamba_amd import _lib
    from xfmamba_amd.ss2d import ss2d_xproj_core_fn
    from xfmamba_amd.ss2d_chan import ss2d_chan_fn
    dev = "cuda"
    shapes = [("stage2 T", a.batch, 384, 14, 24, 1), ("stage3 T", a.batch, 768, 7, 48, 1), ("stage2 S", a.batch, 768, 14, 24, 1),
              ("stage3 B384", a.batch // 4, 2048, 12, 64, 1), ("deep T", a.batch // 2 * 3, 1536, 7, 48, 16)]
    if a.only:
        shapes = [s for s in shapes if a.only in s[0]]
    for name, B, D, HW, R, N in shapes:
        L, K = HW * HW, 4
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, D, L, generator=g).to(dev).bfloat16().requires_grad_()
        xw = (torch.randn(K, R + 2 * N, D, generator=g) * D ** -0.5).to(dev).requires_grad_()
        dtw = (torch.randn(K, D, R, generator=g) * R ** -0.5).to(dev).requires_grad_()
        A = (-torch.rand(K * D, N, generator=g) - 0.1).to(dev).requires_grad_()
        Dp = torch.randn(K * D, generator=g).to(dev).requires_grad_()
        bias = (0.1 * torch.rand(K * D, generator=g) - 4.0).to(dev).requires_grad_()
        gy = torch.randn(B, D, L, device=dev)
        cm = (B // 3, 2 * (B // 3)) if N > 1 else (0, 0)
        chan = (lambda *t: ss2d_chan_fn(*t, c_mod=cm[0], c_off=cm[1]))
        for label, fn in ((("lean chain", ss2d_xproj_core_fn),) if N == 1 else ()) + (("chan", chan),):
            timer = _lib.KernelTimer()
            _lib.set_timer(timer)
            ys = []

            def fwd():
                ys.clear()
                ys.append(fn(x, xw, dtw, A, Dp, bias, HW, HW))

            def bwd():
                ys[0].backward(gy, retain_graph=True)

            tf = med(fwd)
            tb = med(bwd)
            _lib.set_timer(None)
            ks = timer.summary()
            kstr = "  ".join(f"{k}={v['avg_us']:.1f}us" for k, v in ks.items())
            elems = B * D * L
            print(f"{name:12s} {label:10s} node fwd {tf:7.1f} us  bwd {tb:7.1f} us   [{kstr}]   "
                  f"({elems * 6 / tf / 1e3:.0f} / {elems * 16 / tb / 1e3:.0f} GB/s at 6 / 16 B per element)")


if __name__ == "__main__":
    main()
