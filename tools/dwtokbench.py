#!/usr/bin/env python
"""Token-major vs plane-major depthwise 3x3 + SiLU at the trunk's short-map shapes (run under rocprofv3 --kernel-trace --stats)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from xfmamba_amd.dwconv import dwconv3x3_silu_fn, dwconv3x3_silu_tokens_fn  # noqa: E402

for B, HW, C in ((64, 14, 384), (64, 7, 768)):
    g = torch.Generator().manual_seed(0)
    xt = torch.randn(B, HW, HW, C, generator=g).bfloat16().cuda().requires_grad_()
    xp = xt.detach().permute(0, 3, 1, 2).contiguous().requires_grad_()
    w = (torch.randn(C, 1, 3, 3, generator=g) * 0.3).cuda().requires_grad_()
    b = torch.randn(C, generator=g).cuda().requires_grad_()
    for name, fn, x in (("tokens", lambda t: dwconv3x3_silu_tokens_fn(t, w, b), xt), ("planes", lambda t: dwconv3x3_silu_fn(t, w, b, True), xp)):
        gy = torch.randn_like(x)
        for _ in range(3):
            fn(x).backward(gy)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            y = fn(x)
        e.record(); e.synchronize()
        tf = s.elapsed_time(e) / 20 * 1e3
        s.record()
        for _ in range(20):
            y = fn(x); y.backward(gy)
        e.record(); e.synchronize()
        tb = s.elapsed_time(e) / 20 * 1e3 - tf
        print(f"B{B} {HW}x{HW}x{C} {name}: fwd {tf:.1f} us, bwd {tb:.1f} us")
