#!/usr/bin/env python
"""Timing switches of the weight-gradient kernels at one shape: XFM_WGRAD_DBG bits 1 no atomics, 2 no MFMA / fragment reads,
4 no global loads, 8 fragment reads without MFMA.   python tools/wgraddbg.py M N batch L a_planes b_planes"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one():
    import torch
    from xfmamba_amd.proj import wgrad_mfma
    M, N, Bt, L, ap, bp = [int(v) for v in sys.argv[1:7]]
    g = torch.Generator().manual_seed(0)
    a = torch.randn((Bt, M, L) if ap else (Bt, L, M), generator=g).bfloat16().cuda()
    b = torch.randn((Bt, N, L) if bp else (Bt, L, N), generator=g).bfloat16().cuda()
    out = torch.zeros(M, N, device="cuda")
    for _ in range(5):
        wgrad_mfma(a, bool(ap), b, bool(bp), out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    nrep = 40
    for _ in range(5):
        torch.cuda._sleep(20_000_000)                 # the host queues the batch while the GPU spins: no launch gaps inside
        e0.record()
        for _ in range(nrep):
            wgrad_mfma(a, bool(ap), b, bool(bp), out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / nrep)
    ts.sort()
    print(f"dbg={os.environ.get('XFM_WGRAD_DBG', '0'):>2s} {os.environ.get('XFM_WGRAD_EXTRA', ''):12s} median {ts[2]:7.1f} us  min {ts[0]:7.1f} us  (per launch, 40 back to back)")


if __name__ == "__main__":
    if os.environ.get("_WGD_CHILD"):
        one()
    else:
        for dbg in os.environ.get("WGD_LIST", "0 1 2 3 4 5 8 9 12 13").split():
            env = dict(os.environ, XFM_WGRAD_DBG=dbg, _WGD_CHILD="1")
            subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env)
