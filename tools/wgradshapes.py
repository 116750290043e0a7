#!/usr/bin/env python
"""Which weight-gradient products does one XFMamba-T step launch, and how long does each take?  (eager step, HIP events
around every xfm_wgrad call; shapes (M, N, batch, L, a_planes, b_planes))"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from xfmamba_amd import _lib
    from xfmamba_amd.amp import WeightCache
    from xfmamba_amd.deferred import defer_partial_sums
    from xfmamba_amd.dp import GradBuckets
    from xfmamba_amd.net_fusionmamba import TwoViewXFMambaTop
    from xfmamba_amd.optim import FusedAdam
    from xfmamba_amd.proj import WgradArena, set_wgrad_arena
    lib = _lib.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(42)
    model = TwoViewXFMambaTop(in_channels=1, outputs=2, type="tiny").to(dev).train()
    buckets = GradBuckets(model, bucket_mb=48.0)
    wcache = WeightCache(model)
    opt = FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-5, weight_cache=wcache)
    arena = WgradArena(model.parameters())
    set_wgrad_arena(arena)
    defer_partial_sums(True)
    crit = torch.nn.CrossEntropyLoss()
    B = 32
    xa = torch.randn(B, 1, 224, 224, device=dev)
    xb = torch.randn(B, 1, 224, 224, device=dev)
    lab = torch.randint(0, 2, (B,), device=dev)

    def step():
        buckets.zero_grad()
        arena.zero()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(xa, xb)
            loss = crit(out.float(), lab)
        loss.backward()
        buckets.finish()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    real = lib.xfm_wgrad
    rec = []

    def traced(a, b, dw, M, N, Bt, L, a_bs, b_bs, ap, bp, stream):
        s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(s)
        rc = real(a, b, dw, M, N, Bt, L, a_bs, b_bs, ap, bp, stream)
        e1.record(s)
        torch.cuda.synchronize()
        rec.append(((M, N, Bt, L, ap, bp), e0.elapsed_time(e1) * 1e3))
        return rc

    lib.xfm_wgrad = traced
    os.environ["XFM_WGRAD_SIDE"] = "0"
    nrep = 3
    for _ in range(nrep):
        step()
    lib.xfm_wgrad = real
    agg = collections.defaultdict(list)
    for k, t in rec:
        agg[k].append(t)
    tot = 0.0
    print(f"{'M':>5} {'N':>5} {'batch':>5} {'L':>6} ap bp  calls/step  median us   GB/s(unique)  TFLOP/s   us/step")
    for k, ts in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        M, N, Bt, L, ap, bp = k
        ts.sort()
        med = ts[len(ts) // 2]
        n = len(ts) / nrep
        gb = (Bt * L * (M + N) * 2 + M * N * 4) / med / 1e3
        tf = 2.0 * Bt * L * M * N / med / 1e6
        tot += med * n
        print(f"{M:5d} {N:5d} {Bt:5d} {L:6d} {ap:2d} {bp:2d}  {n:9.1f}  {med:9.1f}  {gb:12.0f}  {tf:7.1f}  {med * n:8.1f}")
    print(f"total {tot / 1e3:.3f} ms/step (isolated launches: events include ~3 us of launch latency each)")


if __name__ == "__main__":
    main()
