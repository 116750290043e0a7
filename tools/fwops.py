#!/usr/bin/env python
"""Framework (aten) operators of one eager XFMamba-T training step, grouped by operator and input shapes: which element-wise /
copy / reduce launches are left, how long they run and -- for the forward -- from which line of the package they come
(torch.profiler; batch 32, autocast bf16).  XFM_FWOPS_STACK=1 adds the innermost package frame of forward-side operators."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from torch.profiler import profile, ProfilerActivity
    from xfmamba_amd import _lib
    from xfmamba_amd.net_fusionmamba import TwoViewXFMambaTop
    _lib.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(42)
    model = TwoViewXFMambaTop(in_channels=1, outputs=2, type="tiny").to(dev).train()
    B = 32
    xa = torch.randn(B, 1, 224, 224, device=dev)
    xb = torch.randn(B, 1, 224, 224, device=dev)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            model(xa.expand(-1, 3, -1, -1) if False else xa, xb).float().sum().backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    stack = os.environ.get("XFM_FWOPS_STACK", "0") == "1"
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=stack) as prof:
        step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=8 if stack else 0):
        t = e.self_device_time_total
        if t <= 0 or not e.key.startswith("aten::"):
            continue
        where = ""
        if stack:
            for fr in e.stack:
                if "xfmamba_amd" in fr:
                    where = fr.split("xfmamba_amd/")[-1][:60]
                    break
        rows.append((t, e.count, e.key, str(e.input_shapes)[:110], where))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print(f"aten operators with device time: {tot / 1e3:.3f} ms in {sum(r[1] for r in rows)} calls")
    for t, n, k, shp, where in rows[:90]:
        print(f"{t:9.1f} us {n:4d}  {k:28s} {shp}  {where}")


if __name__ == "__main__":
    main()
