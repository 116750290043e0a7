#!/usr/bin/env python
"""xfm_wgrad against the library formulation (partial products + sum) at the weight-gradient shapes of the XFMamba-T
step; run under tools/prof_wgrad.sh for kernel durations."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    from xfmamba_amd.proj import wgrad_mfma, _bmm_f32
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    # (batch, L, M, N, a_planes, b_planes)
    shapes = [(64, 3136, 384, 96, 0, 0), (64, 784, 768, 192, 0, 0), (64, 196, 1536, 384, 0, 0), (64, 49, 3072, 768, 0, 0),
              (64, 196, 384, 1536, 0, 0), (64, 196, 384, 384, 1, 0), (64, 196, 384, 384, 0, 1), (64, 196, 128, 384, 0, 1),
              (64, 784, 192, 192, 1, 0), (64, 49, 768, 768, 0, 1)]
    for Bt, L, M, N, ap, bp in shapes:
        a = torch.randn((Bt, M, L) if ap else (Bt, L, M), generator=g).bfloat16().to(dev)
        b = torch.randn((Bt, N, L) if bp else (Bt, L, N), generator=g).bfloat16().to(dev)
        for _ in range(5):
            wgrad_mfma(a, bool(ap), b, bool(bp))
        torch.cuda.synchronize()
        at = a if ap else a.transpose(1, 2)
        bt = b.transpose(1, 2) if bp else b
        for _ in range(5):
            _bmm_f32(at, bt).sum(0)
        torch.cuda.synchronize()
        print("shape", Bt, L, M, N, ap, bp)


if __name__ == "__main__":
    main()
