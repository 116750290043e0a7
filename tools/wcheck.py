#!/usr/bin/env python
"""A/B of the wide-map SS2D node: the ss2d_w.hpp kernels against the ss2d_l3.hip kernels (XFM_SS2D_W=0 in a child process) on the
same seeded inputs -- y and every gradient; shapes on the command line as B,D,HW,R."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def run(shape):
    from xfmamba_amd.ss2d import ss2d_xproj_core_fn
    B, D, HW, R = shape
    L, K, N = HW * HW, 4, 1
    g = torch.Generator().manual_seed(B * D + HW)
    dev = "cuda"
    x = torch.randn(B, D, L, generator=g).to(dev).bfloat16().requires_grad_()
    xw = (torch.randn(K, R + 2 * N, D, generator=g) * D ** -0.5).to(dev).requires_grad_()
    dtw = (torch.randn(K, D, R, generator=g) * R ** -0.5).to(dev).requires_grad_()
    A = (-torch.rand(K * D, N, generator=g) - 0.1).to(dev).requires_grad_()
    Dp = torch.randn(K * D, generator=g).to(dev).requires_grad_()
    bias = (0.1 * torch.rand(K * D, generator=g) - 4.0).to(dev).requires_grad_()
    gy = torch.randn(B, D, L, generator=g).to(dev)
    y = ss2d_xproj_core_fn(x, xw, dtw, A, Dp, bias, HW, HW)
    y.backward(gy)
    return [t.detach().float().cpu() for t in (y, x.grad, xw.grad, dtw.grad, A.grad, Dp.grad, bias.grad)]


def main():
    shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:] if "," in s] or [(2, 512, 48, 16), (2, 1024, 24, 32)]
    if "--child" in sys.argv:
        torch.save([run(s) for s in shapes], "/tmp/wcheck_l3.pt")
        return
    env = dict(os.environ, XFM_SS2D_W="0")
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + [",".join(map(str, s)) for s in shapes], env=env, check=True)
    ref = torch.load("/tmp/wcheck_l3.pt")
    for s, r in zip(shapes, ref):
        got = run(s)
        for name, a, b in zip(("y", "dx", "dxw", "ddtw", "dA", "dD", "dbias"), got, r):
            err = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)
            print(s, name, f"rel err {err:.3e}", "finite" if bool(torch.isfinite(a).all()) else "NON-FINITE", flush=True)


if __name__ == "__main__":
    main()
