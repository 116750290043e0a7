"""Where do the framework copy / cast / cat / add kernels of the bench step come from?  (torch.profiler with stacks; eager step)

    python tools/copyprof.py [--top 60]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--top", type=int, default=60)
    a = ap.parse_args()
    from xfmamba_amd import _lib
    from xfmamba_amd.amp import WeightCache
    from xfmamba_amd.deferred import defer_partial_sums
    from xfmamba_amd.dp import GradBuckets
    from xfmamba_amd.net_fusionmamba import TwoViewXFMambaTop
    from xfmamba_amd.optim import FusedAdam
    from xfmamba_amd.proj import WgradArena, set_wgrad_arena
    _lib.lib()
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

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    want = ("aten::copy_", "aten::cat", "aten::add", "aten::add_", "aten::mul", "aten::sum", "aten::fill_", "aten::zero_",
            "aten::neg", "aten::exp", "aten::silu", "aten::sigmoid", "aten::mean", "aten::native_layer_norm", "aten::div",
            "aten::native_layer_norm_backward", "aten::silu_backward", "aten::sigmoid_backward", "aten::_foreach_copy_")
    agg = collections.defaultdict(lambda: [0.0, 0])
    for e in prof.events():
        if e.name not in want or e.self_device_time_total <= 0:
            continue
        frames = [s for s in (e.stack or []) if "xfmamba_amd" in s or "bench" in s or "tools/" in s]
        where = " <- ".join(f.split("/")[-1][:60] for f in frames[:3]) or "(autograd / library)"
        key = (e.name, str(e.input_shapes)[:70], where)
        agg[key][0] += e.self_device_time_total
        agg[key][1] += 1
    tot = sum(v[0] for v in agg.values())
    print(f"framework elementwise / copy ops: {tot / 1e3:.3f} ms per step (profiler timing)")
    for (name, shp, where), (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.top]:
        print(f"{t / 1e3:7.3f} ms n={n:3d} {name:22s} {shp:70s} {where}")


if __name__ == "__main__":
    main()
