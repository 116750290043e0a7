#!/usr/bin/env python
"""Run one kernel many times while a second process does the same on the same GPU, and compare every result with the first:
memory contention and time slicing between two processes expose waits that are too short (a tile read before it landed) and
other timing assumptions that a single process on an idle GPU never violates.  tools/stress2.py <op> [iterations]; launch two."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    from xfmamba_amd import _lib
    lib = _lib.lib()
    op = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    g = torch.Generator().manual_seed(1)
    if op.startswith("dtbwd"):
        B, D, R, L = {"dtbwd0": (64, 96, 6, 3136), "dtbwd1": (64, 192, 12, 784), "dtbwd2": (64, 192, 6, 3136)}[op]
        ddts = torch.randn(B, 4, D, L, generator=g).bfloat16().cuda()
        xr = torch.randn(B, 4, R, L, generator=g).bfloat16().cuda()
        w = (torch.randn(4, D, R, generator=g) * R ** -0.5).bfloat16().cuda()
        dxr = torch.empty(B, 4, R, L, dtype=torch.bfloat16, device="cuda")
        dw = torch.zeros(4, D, R, device="cuda")

        def run():
            dxr.fill_(float("nan"))
            dw.zero_()
            _lib.check(lib.xfm_ss2d_dt_proj_bwd_mfma(ddts.data_ptr(), xr.data_ptr(), w.data_ptr(), dxr.data_ptr(), dw.data_ptr(), B, D,
                                                     R, L, _lib.stream_ptr()), "dt_proj_bwd_mfma")
            return dxr.clone(), dw.clone()
    else:
        raise SystemExit("unknown op")
    ref = run()
    torch.cuda.synchronize()
    bad = 0
    t0 = time.time()
    for i in range(n):
        out = run()
        torch.cuda.synchronize()
        e0 = not bool(torch.equal(out[0], ref[0]))
        e1 = float((out[1] - ref[1]).abs().max()) > 1e-3 * float(ref[1].abs().max()) or not bool(torch.isfinite(out[1]).all())
        if e0 or e1:
            bad += 1
            if bad <= 3:
                d = (out[0].float() - ref[0].float())
                print(f"[{os.getpid()}] iteration {i}: dxr differs at {int((d != 0).sum() + torch.isnan(d).sum())} elements "
                      f"(nan {int(torch.isnan(out[0].float()).sum())}), dw off {e1}", flush=True)
    print(f"[{os.getpid()}] {op}: {bad} of {n} runs differ from the first ({time.time() - t0:.1f} s)", flush=True)


if __name__ == "__main__":
    main()
