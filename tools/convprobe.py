#!/usr/bin/env python
"""The trunk's 3 x 3 stride-2 convolutions: xfm_conv3x3s2_tokens_* (csrc/conv_tok.hip) against the convolution library on the
channels_last view, per pass (forward, data gradient, weight gradient), launches queued back to back."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def timed(fn, nrep=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    ts = []
    for _ in range(5):
        torch.cuda._sleep(20_000_000)
        e0.record()
        for _ in range(nrep):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / nrep)
    return sorted(ts)[2]


def main():
    from xfmamba_amd import _lib
    lib = _lib.lib()
    torch.backends.cudnn.benchmark = True
    g = torch.Generator().manual_seed(0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    for H, C, O in [(112, 48, 96), (56, 96, 192), (28, 192, 384), (14, 384, 768)]:
        x = torch.randn(B, H, H, C, generator=g).bfloat16().cuda()
        w = ((9 * C) ** -0.5 * torch.randn(O, 3, 3, C, generator=g)).bfloat16().cuda()
        OH = H // 2
        T = B * OH * OH
        col = torch.empty(T, 9 * C, dtype=torch.bfloat16, device="cuda")
        dcol = torch.empty_like(col)
        y = torch.empty(B, OH, OH, O, dtype=torch.bfloat16, device="cuda")
        dy = torch.randn(B, OH, OH, O, generator=g).bfloat16().cuda()
        dx = torch.empty_like(x)
        dw = torch.zeros(O, 9 * C, device="cuda")
        s = _lib.stream_ptr

        def fwd():
            _lib.check(lib.xfm_conv3x3s2_tokens_fwd(x.data_ptr(), w.data_ptr(), col.data_ptr(), y.data_ptr(), B, H, H, C, O, s()), "f")

        def bd():
            _lib.check(lib.xfm_conv3x3s2_tokens_bwd_data(dy.data_ptr(), w.data_ptr(), dcol.data_ptr(), dx.data_ptr(), B, H, H, C, O, s()), "d")

        def bw():
            _lib.check(lib.xfm_conv3x3s2_tokens_bwd_weight(dy.data_ptr(), col.data_ptr(), dw.data_ptr(), B, H, H, C, O, s()), "w")

        def bwx():
            _lib.check(lib.xfm_conv3x3s2_tokens_bwd_weight_x(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), B, H, H, C, O, s()), "wx")

        wx = lib.xfm_conv3x3s2_tokens_bwd_weight_x_supported(B, H, H, C, O)
        xn = x.permute(0, 3, 1, 2)                         # channels_last views
        wn = w.permute(0, 3, 1, 2)
        dyn = dy.permute(0, 3, 1, 2)

        def lf():
            return torch.ops.aten.convolution(xn, wn, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1)

        def lbd():
            return torch.ops.aten.convolution_backward(dyn, xn, wn, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])

        def lbw():
            return torch.ops.aten.convolution_backward(dyn, xn, wn, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])

        fwd()
        r = lf()
        err = float((y.float() - r.permute(0, 2, 3, 1).float()).abs().max()) / float(r.float().abs().max())
        print(f"B {B} {H}x{H}x{C} -> {O}:  fwd own {timed(fwd):6.1f} lib {timed(lf):6.1f}   dgrad own {timed(bd):6.1f} lib {timed(lbd):6.1f}"
              f"   wgrad own {timed(bw):6.1f} from the map {(timed(bwx) if wx else 0.0):6.1f} lib {timed(lbw):6.1f} us   max diff / max {err:.1e}")


def gray():
    """the first convolution of the patch embedding on one replicated channel: own kernels vs the library on the 3-channel image"""
    from xfmamba_amd import _lib
    lib = _lib.lib()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    g = torch.Generator().manual_seed(0)
    H, CI, O = 224, 3, 48
    x1 = torch.randn(B, H, H, generator=g).bfloat16().cuda()
    w = (27 ** -0.5 * torch.randn(O, CI, 3, 3, generator=g)).bfloat16().cuda()
    y = torch.empty(B, H // 2, H // 2, O, dtype=torch.bfloat16, device="cuda")
    dy = torch.randn(B, H // 2, H // 2, O, generator=g).bfloat16().cuda()
    dw9 = torch.zeros(O, 9, device="cuda")
    ws = torch.zeros(lib.xfm_conv3x3s2_gray_ws_floats(O), device="cuda")
    s = _lib.stream_ptr
    x3 = x1.unsqueeze(-1).expand(-1, -1, -1, CI).contiguous().permute(0, 3, 1, 2)
    wn = w.contiguous(memory_format=torch.channels_last)
    dyn = dy.permute(0, 3, 1, 2)

    def fwd():
        _lib.check(lib.xfm_conv3x3s2_gray_fwd(x1.data_ptr(), w.data_ptr(), y.data_ptr(), B, H, H, CI, O, s()), "f")

    def bw():
        _lib.check(lib.xfm_conv3x3s2_gray_bwd_weight(dy.data_ptr(), x1.data_ptr(), dw9.data_ptr(), ws.data_ptr(), B, H, H, O, s()), "w")

    def lf():
        return torch.ops.aten.convolution(x3, wn, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1)

    def lbw():
        return torch.ops.aten.convolution_backward(dyn, x3, wn, None, [2, 2], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])

    print(f"B {B} gray 224x224 -> 48:  fwd own {timed(fwd):6.1f} lib {timed(lf):6.1f}   wgrad own {timed(bw):6.1f} lib {timed(lbw):6.1f} us")


if __name__ == "__main__":
    gray()
    main()
