#!/usr/bin/env python
"""xfm_tokens_gemm2 epilogue 0 (tiled form) against the library GEMM at the projection shapes of the trunk: per-launch time
with launches queued back to back, and the largest difference of the two results."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def timed(fn, nrep=30):
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
    g = torch.Generator().manual_seed(0)
    # (T, con, out, weight given transposed)
    shapes = [(12544, 1536, 384, 0), (12544, 1536, 384, 1), (12544, 384, 384, 0), (12544, 384, 384, 1), (12544, 384, 1536, 0),
              (50176, 768, 192 + 64, 0), (50176, 192, 256, 0), (3136, 3072, 768, 0), (3136, 3072, 768, 1), (3136, 768, 768, 0),
              (200704, 384, 128, 0), (200704, 128, 384, 0)]
    for T, con, out, wt in shapes:
        if not lib.xfm_tokens_gemm2_supported(con, out):
            print("unsupported", T, con, out)
            continue
        x = torch.randn(T, con, generator=g).bfloat16().cuda()
        w = (con ** -0.5 * torch.randn((con, out) if wt else (out, con), generator=g)).bfloat16().cuda()
        b = torch.randn(out, generator=g).cuda()
        y = torch.empty(T, out, dtype=torch.bfloat16, device="cuda")

        def own():
            _lib.check(lib.xfm_tokens_gemm2(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, None, T, con, out, wt, 0,
                                            _lib.stream_ptr()), "gemm2")

        wl = w if not wt else w.t()

        def ref():
            return torch.nn.functional.linear(x, wl, b.bfloat16())

        own()
        r = ref()
        err = float((y.float() - r.float()).abs().max()) / float(r.float().abs().max())
        print(f"T {T:6d} con {con:4d} out {out:4d} wt {wt}:  own {timed(own):7.1f} us   library {timed(ref):7.1f} us   max diff / max {err:.1e}")


if __name__ == "__main__":
    main()
