"""Probe of csrc/tokens_gemm.hip (xfm_tokens_gemm): correctness against torch and time against torch.mm."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def t(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


def main():
    from xfmamba_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    f = lib.xfm_tokens_gemm
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int,
                  ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    dev, bf = "cuda", torch.bfloat16
    st = _lib.stream_ptr()
    for (T, K, N) in [(200704, 96, 384), (200704, 384, 96), (200704, 96, 96), (200704, 96, 192), (1000, 96, 384), (77, 384, 96)]:
        x = torch.randn(T, K, device=dev, dtype=bf)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(bf)
        wt = w.t().contiguous()
        b = torch.randn(N, device=dev)
        y = torch.empty(T, N, device=dev, dtype=bf)
        ref = (x.float() @ w.float().t() + b)
        for name, wp, flag in (("w", w, 0), ("wt", wt, 1)):
            y.zero_()
            rc = f(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), T, K, N, flag, st)
            torch.cuda.synchronize()
            err = float((y.float() - ref).abs().max() / ref.abs().max())
            tm = t(lambda: f(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), T, K, N, flag, st))
            gb = (T * K + T * N) * 2 / 1e3
            print(f"T={T:6d} K={K:3d} N={N:3d} [{name:2s}] rc={rc} rel.err={err:.2e}  {tm:7.1f} us  {gb / tm:7.1f} GB/s", end="")
            if name == "w":
                tl = t(lambda: torch.addmm(b.to(bf), x, w.t()))
                print(f"   torch.addmm {tl:7.1f} us   torch.mm {t(lambda: torch.mm(x, w.t())):7.1f} us")
            else:
                print()


if __name__ == "__main__" and "--proj" not in sys.argv:
    main()


def main_proj():
    from xfmamba_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    f = lib.xfm_proj_gemm
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6 + [ctypes.c_void_p]
    dev, bf = "cuda", torch.bfloat16
    st = _lib.stream_ptr()
    for (B, L, K, M) in [(64, 3136, 96, 96), (64, 784, 192, 192), (12, 40, 96, 96), (12, 40, 192, 192)]:
        w = (torch.randn(M, K, device=dev) / K ** 0.5).to(bf)
        wt = w.t().contiguous()
        b = torch.randn(M, device=dev)
        xt = torch.randn(B, L, K, device=dev, dtype=bf)
        xp = xt.transpose(1, 2).contiguous()
        ref = torch.einsum("mk,blk->bml", w.float(), xt.float()) + b[None, :, None]        # planes (B, M, L)
        for in_pl in (0, 1):
            x = xp if in_pl else xt
            y = torch.zeros((B, L, M) if in_pl else (B, M, L), device=dev, dtype=bf)
            want = ref.transpose(1, 2) if in_pl else ref
            for name, wp, flag in (("w", w, 0), ("wt", wt, 1)):
                y.zero_()
                rc = f(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), B, L, K, M, in_pl, flag, st)
                torch.cuda.synchronize()
                err = float((y.float() - want).abs().max() / want.abs().max())
                tm = t(lambda: f(x.data_ptr(), wp.data_ptr(), b.data_ptr(), y.data_ptr(), B, L, K, M, in_pl, flag, st))
                gb = B * L * (K + M) * 2 / 1e3
                print(f"B={B} L={L} {'planes->tokens' if in_pl else 'tokens->planes'} [{name:2s}] rc={rc} rel.err={err:.2e} {tm:7.1f} us {gb / tm:7.1f} GB/s")
        if B == 64:  # library reference
            print("   torch bmm tok->pl", t(lambda: torch.bmm(w.unsqueeze(0).expand(B, M, K), xt.transpose(1, 2))),
                  " pl->tok", t(lambda: torch.bmm(xp.transpose(1, 2), w.t().unsqueeze(0).expand(B, K, M))))


if __name__ == "__main__" and "--proj" in sys.argv:
    main_proj()
