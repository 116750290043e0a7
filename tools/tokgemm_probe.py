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


if __name__ == "__main__":
    main()
