"""Op-level attribution of one training step (torch.profiler, eager launches).

    python tools/opprof.py [--batch 32] [--top 70] [--shapes]

Prints GPU time per aten / autograd op (optionally split by input shapes) for the bench.py step, so the
torch-side overhead around the hand-written kernels can be traced back to the module code that causes it.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--top", type=int, default=70)
    ap.add_argument("--shapes", action="store_true")
    ap.add_argument("--width", type=int, default=60)
    ap.add_argument("--stack", action="store_true")
    ap.add_argument("--find", action="store_true", help="MIOpen find mode (as bench.py)")
    ap.add_argument("--ops-only", action="store_true", help="list framework ops only (no kernel rows)")
    a = ap.parse_args()
    from xfmamba_amd import _lib
    from xfmamba_amd.dp import GradBuckets
    from xfmamba_amd.net_fusionmamba import TwoViewXFMambaTop
    _lib.lib()
    dev = torch.device("cuda", 0)
    if a.find:
        torch.backends.cudnn.benchmark = True
    torch.manual_seed(42)
    model = TwoViewXFMambaTop(in_channels=1, outputs=2, type="tiny").to(dev).train()
    from xfmamba_amd.amp import WeightCache
    buckets = GradBuckets(model, bucket_mb=48.0)
    wcache = WeightCache(model)                                   # as bench.py: bf16 weight shadows, one refresh per step
    from xfmamba_amd.optim import FusedAdam
    from xfmamba_amd.proj import WgradArena, set_wgrad_arena
    opt = FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-5, weight_cache=wcache)      # as bench.py
    arena = WgradArena(model.parameters())
    set_wgrad_arena(arena)
    crit = torch.nn.CrossEntropyLoss()
    B = a.batch
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

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=a.shapes,
                 with_stack=a.stack) as prof:
        step()
        torch.cuda.synchronize()
    ka = prof.key_averages(group_by_input_shape=a.shapes, group_by_stack_n=6 if a.stack else 0)
    rows = sorted(ka, key=lambda e: -e.self_device_time_total)
    tot = sum(e.self_device_time_total for e in rows)
    print(f"total self device time {tot / 1e3:.2f} ms")
    if a.ops_only:
        rows = [e for e in rows if e.key.startswith("aten::") or "Hip" in e.key or e.key.startswith("Optimizer")]
    for e in rows[:a.top]:
        shp = str(e.input_shapes)[:150] if a.shapes else ""
        print(f"{e.self_device_time_total / 1e3:8.3f} ms  n={e.count:4d}  {e.key[:a.width]:{a.width}s} {shp}")
        if a.stack and e.stack:
            for s in e.stack[:6]:
                if "xfmamba_amd" in s or "bench" in s:
                    print("              ", s[-110:])


if __name__ == "__main__":
    main()
