#!/usr/bin/env python
"""Tile GEMM / implicit-GEMM convolution against the library at the shapes of the XFMamba-T step (run under
rocprofv3 --kernel-trace --stats for kernel durations: tools/prof_tile.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def main():
    from xfmamba_amd import _lib
    lib = _lib.lib()
    dev = "cuda"
    torch.backends.cudnn.benchmark = os.environ.get("TILEBENCH_FIND", "0") == "1"
    g = torch.Generator().manual_seed(0)
    reps = 10
    for T, K, N in [(50176, 192, 768), (50176, 768, 192), (12544, 384, 1536), (12544, 1536, 384), (3136, 768, 3072),
                    (3136, 3072, 768), (12544, 384, 384), (3136, 768, 768)]:
        x = torch.randn(T, K, generator=g).bfloat16().to(dev)
        w = torch.randn(N, K, generator=g).bfloat16().to(dev)
        b = torch.randn(N, generator=g).to(dev)
        y = torch.empty(T, N, dtype=torch.bfloat16, device=dev)
        for _ in range(reps):
            _lib.check(lib.xfm_tile_gemm(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), T, K, N, _lib.stream_ptr()), "t")
        for _ in range(reps):
            torch.mm(x, w.t())
        torch.cuda.synchronize()
        print("gemm", T, K, N)
    for B, H, C, N in [(64, 112, 48, 96), (64, 56, 96, 192), (64, 28, 192, 384), (64, 14, 384, 768)]:
        x = torch.randn(B, H, H, C, generator=g).bfloat16().to(dev)
        w = torch.randn(N, C, 3, 3, generator=g).bfloat16().to(dev)
        w9 = w.permute(0, 2, 3, 1).contiguous()
        wt = w.permute(2, 3, 1, 0).contiguous()
        Ho = H // 2
        y = torch.empty(B, Ho, Ho, N, dtype=torch.bfloat16, device=dev)
        dy = torch.randn(B, Ho, Ho, N, generator=g).bfloat16().to(dev)
        dx = torch.empty_like(x)
        for _ in range(reps):
            _lib.check(lib.xfm_conv3x3s2_fwd(x.data_ptr(), w9.data_ptr(), None, y.data_ptr(), B, H, H, C, N, _lib.stream_ptr()), "f")
            _lib.check(lib.xfm_conv3x3s2_dgrad(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), B, H, H, C, N, _lib.stream_ptr()), "d")
        xc = x.permute(0, 3, 1, 2).requires_grad_()                 # channels_last view, as the model hands it to MIOpen
        wc = w.contiguous(memory_format=torch.channels_last).requires_grad_()
        for _ in range(reps):
            yy = F.conv2d(xc, wc, None, 2, 1)
            yy.backward(dy.permute(0, 3, 1, 2))
        torch.cuda.synchronize()
        print("conv", B, H, C, N)


if __name__ == "__main__":
    main()
