#!/usr/bin/env python
"""Per-workgroup timeline of the LDS-direct weight-gradient kernel (xfm_dbg_wgrad_prof): when do workgroups start, how long
are prologue / stage loop / adds?   python tools/wgradprof.py M N L   (token-major x token-major, one token run)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def stages(st):
    base = st[0, :, 0].min()
    print("stage: per wave [at barrier, after barrier, after issue, after MFMAs] in cycles since the first stamp")
    for i in range(40):
        if st[i, 0, 0] == 0:
            break
        print(f"  {i:2d} " + "  ".join("[" + " ".join(f"{int(st[i, w, k] - base):6d}" for k in range(5) if st[i, w, k]) + "]" for w in range(4)))


def main():
    from xfmamba_amd import _lib
    from xfmamba_amd.proj import wgrad_mfma
    M, N, L = [int(v) for v in sys.argv[1:4]]
    Bt, ap, bp = ([int(v) for v in sys.argv[4:7]] + [1, 0, 0])[:3] if len(sys.argv) > 4 else (1, 0, 0)
    lib = _lib.lib()
    lib.xfm_dbg_wgrad_prof.argtypes = [ctypes.c_void_p]
    lib.xfm_dbg_wgrad_prof.restype = None
    g = torch.Generator().manual_seed(0)
    a = torch.randn((Bt, M, L) if ap else (Bt, L, M), generator=g).bfloat16().cuda()
    b = torch.randn((Bt, N, L) if bp else (Bt, L, N), generator=g).bfloat16().cuda()
    out = torch.zeros(M, N, device="cuda")
    for _ in range(3):
        wgrad_mfma(a, bool(ap), b, bool(bp), out=out)
    prof = torch.zeros(4 * 512 + 40 * 20, dtype=torch.int64, device="cuda")
    lib.xfm_dbg_wgrad_prof(prof.data_ptr())
    wgrad_mfma(a, bool(ap), b, bool(bp), out=out)
    torch.cuda.synchronize()
    lib.xfm_dbg_wgrad_prof(None)
    st = prof[2048:].view(40, 4, 5).cpu()
    p = prof[:2048].view(512, 4).cpu()
    p = p[p[:, 0] > 0]
    t0 = p[:, 0].min()
    p = (p - t0).double() / 100.0                       # us (100 MHz)
    print(f"{p.shape[0]} workgroups; kernel span {p[:, 3].max():.1f} us")
    for name, col in (("start", p[:, 0]), ("prologue issued", p[:, 1] - p[:, 0]), ("stage loop", p[:, 2] - p[:, 1]),
                      ("adds drained", p[:, 3] - p[:, 2]), ("end", p[:, 3])):
        print(f"  {name:16s} min {col.min():7.2f}  median {col.median():7.2f}  max {col.max():7.2f} us")
    stages(st)


if __name__ == "__main__":
    main()
