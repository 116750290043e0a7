#!/usr/bin/env python
"""Mlp of a VSS block (fc1 -> GELU -> fc2, forward + backward) at the trunk shapes: fused GELU products (xfm_tokens_gemm2)
against the three-node chain.  Run under rocprofv3 --stats for kernel durations (tools/prof_mlp.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    from xfmamba_amd import mlp_tokens as M
    dev = "cuda"
    fused = os.environ.get("XFM_MLP_FUSED", "1") == "1"
    for name, B, HW, C in (("stage0", 64, 56, 96), ("stage1", 64, 28, 192), ("stage2", 64, 14, 384), ("stage3", 64, 7, 768)):
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, HW, HW, C, generator=g).to(dev).bfloat16().requires_grad_()
        w1 = (C ** -0.5 * torch.randn(4 * C, C, generator=g)).to(dev).requires_grad_()
        b1 = torch.zeros(4 * C, device=dev, requires_grad=True)
        w2 = ((4 * C) ** -0.5 * torch.randn(C, 4 * C, generator=g)).to(dev).requires_grad_()
        b2 = torch.zeros(C, device=dev, requires_grad=True)
        gy = torch.randn(B, HW, HW, C, device=dev).bfloat16()
        for _ in range(8):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = M.mlp_tokens_fn(x, w1, b1, w2, b2)
            y.backward(gy)
        torch.cuda.synchronize()
        print(name, "fused" if fused else "chain")


if __name__ == "__main__":
    main()
