#!/usr/bin/env python
"""Per-shape timing of the lean / generic fused SS2D chain (x_proj + dt_proj + xfm_ss2d_fwd/_bwd) at the trunk shapes of
XFMamba-T / S / B@384 that the channel-lane kernels do not cover.  Run under rocprofv3 --stats for kernel durations."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    from xfmamba_amd import _lib
    from xfmamba_amd.ss2d import ss2d_xproj_core_fn
    dev = "cuda"
    shapes = [("T s0 56x56", 64, 96, 56, 6), ("T s1 28x28", 64, 192, 28, 12), ("B384 s0 96x96", 32, 256, 96, 8),
              ("B384 s1 48x48", 32, 512, 48, 16), ("B384 s2 24x24", 32, 1024, 24, 32), ("S s0 56x56", 64, 192, 56, 6)]
    for name, B, D, HW, R in shapes:
        if a.only and a.only not in name:
            continue
        L, K, N = HW * HW, 4, 1
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, D, L, generator=g).to(dev).bfloat16().requires_grad_()
        xw = (torch.randn(K, R + 2 * N, D, generator=g) * D ** -0.5).to(dev).requires_grad_()
        dtw = (torch.randn(K, D, R, generator=g) * R ** -0.5).to(dev).requires_grad_()
        A = (-torch.rand(K * D, N, generator=g) - 0.1).to(dev).requires_grad_()
        Dp = torch.randn(K * D, generator=g).to(dev).requires_grad_()
        bias = (0.1 * torch.rand(K * D, generator=g) - 4.0).to(dev).requires_grad_()
        gy = torch.randn(B, D, L, device=dev)
        for _ in range(2):                                  # warm-up (first launches pay module load / attribute calls)
            ss2d_xproj_core_fn(x, xw, dtw, A, Dp, bias, HW, HW).backward(gy)
        timer = _lib.KernelTimer()
        _lib.set_timer(timer)
        for _ in range(5):
            y = ss2d_xproj_core_fn(x, xw, dtw, A, Dp, bias, HW, HW)
            y.backward(gy)
        _lib.set_timer(None)
        ks = timer.summary()
        print(name, "  ".join(f"{k}={v['avg_us']:.0f}us" for k, v in ks.items()))


if __name__ == "__main__":
    main()
