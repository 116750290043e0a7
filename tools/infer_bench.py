"""Inference entry with a synchronised FPS measurement.

Mirrors the statistics the reference's ``2_inference_*.py`` writes to ``inference_timing.txt`` (total / per-image /
mean / std / median / min / max batch time, FPS = 1 / per-image time; ``2_inference_chexpert.py:130-263``) for the
eval forward of ``ModelWrapper(TwoViewXFMambaTop)`` on synthetic two-view batches -- but with the device synchronised
around every timed batch (the reference reads ``time.time()`` around an asynchronous launch).

    python tools/infer_bench.py [--batch 32] [--batches 50] [--dtype bf16|fp32] [--model tiny|small|base] [--no-graph]

Prints one JSON line.  With the hipGraph option (default) the eval forward is captured once and replayed per batch.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--batches", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--model", default="tiny", choices=["tiny", "small", "base"])
    ap.add_argument("--no-graph", action="store_true")
    a = ap.parse_args()
    from xfmamba_amd import _lib
    from xfmamba_amd.amp import WeightCache
    from xfmamba_amd.net_fusionmamba import ModelWrapper, TwoViewXFMambaTop
    _lib.lib()
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = True
    torch.manual_seed(42)
    kw = dict(hidden_dim=1024) if a.model == "base" else {}
    net = ModelWrapper(TwoViewXFMambaTop(in_channels=1, outputs=2, type=a.model, **kw)).to(dev).eval()
    cache = WeightCache(net) if a.dtype == "bf16" else None          # weights are frozen: shadows stay current
    x = torch.randn(a.batch, 2, a.size, a.size, device=dev)          # ModelWrapper splits the channel axis into views

    def fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=a.dtype == "bf16"):
            return net(x)

    for _ in range(a.warmup):
        fwd()
    torch.cuda.synchronize()
    graph, out = None, None
    if not a.no_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fwd()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = fwd()
    times = []
    for _ in range(a.batches):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if graph is not None:
            graph.replay()
        else:
            out = fwd()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    assert torch.isfinite(out.float()).all()
    total = sum(times)
    images = a.batch * a.batches
    per_image = total / images
    print(json.dumps({
        "metric": "two-view samples/sec, eval forward", "value": round(1.0 / per_image, 2),
        "unit": "two-view samples/s (1 sample = 2 images)", "fps_single_view_images": round(2.0 / per_image, 2),
        "per_sample_ms": round(per_image * 1e3, 4), "mean_batch_ms": round(statistics.mean(times) * 1e3, 3),
        "std_batch_ms": round(statistics.pstdev(times) * 1e3, 3), "median_batch_ms": round(statistics.median(times) * 1e3, 3),
        "min_batch_ms": round(min(times) * 1e3, 3), "max_batch_ms": round(max(times) * 1e3, 3),
        "config": {"workload": f"XFMamba-{a.model[0].upper()} ({a.dtype}), 2x{a.size}x{a.size}, batch {a.batch}, eval",
                   "batches": a.batches, "launch_mode": "hipGraph" if graph is not None else "eager"},
        "data": "synthetic", "dtype": a.dtype}))
    if cache is not None:
        cache.close()


if __name__ == "__main__":
    main()
