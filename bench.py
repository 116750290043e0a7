#!/usr/bin/env python
"""bench.py -- XFMamba-T two-view training step on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = forward + backward (+ RCCL gradient all-reduce for N > 1) + Adam update of
``TwoViewXFMambaTop(type='tiny')`` on one synthetic batch of 32 two-view samples per GPU
(2 x 224 x 224, already resident in HBM), bf16 compute (autocast GEMMs, bf16 scan I/O, fp32 scan
state / norms / master weights / optimizer), model in train() mode with the reference's DropPath.
Semantics of the step follow the reference loop (libs/training.py:181-195, 1_train_model.py:134-141:
CrossEntropyLoss, Adam lr 1e-4 wd 1e-5).

Prints ONE JSON line (rank 0).  ``roofline`` is for the dominant hand-written kernel, timed with HIP
events on its own launch stream inside the timed region; ``cpu_baseline`` is the CPU oracle
(faithful restatement of the reference's CPU path, checked against reference goldens) timed on
this host for BASELINE configs[0].
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)       # >= 10 warm-up + >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="two-view samples per GPU")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--model", default="tiny", choices=["tiny", "small", "base"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--ss2d", default=None, choices=["fused", "unfused"])
    ap.add_argument("--graph-scope", default=None, choices=["step", "fwdbwd"],
                    help="what the hipGraph captures (default: whole step at N=1, forward+backward at N>1)")
    ap.add_argument("--stream", default=None, choices=["tokens", "planes"], help="residual-stream layout of the trunk")
    ap.add_argument("--fp8", action="store_true", help="BASELINE configs[4]: fp8 (e4m3) weights for the SS2D x_proj / out_proj "
                    "on the fp8 matrix cores, scan in bf16")
    ap.add_argument("--torch-adam", action="store_true", help="library optimizer instead of the fused multi-tensor kernel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--drop-path", type=float, default=None, help="override the reference's DropPath rates (e.g. 0)")
    ap.add_argument("--comm-dtype", default=None, choices=["bf16", "fp32"],
                    help="dtype of the gradient all-reduce at N > 1 (default: the step's --dtype)")
    ap.add_argument("--dp-cut", type=int, default=1, help="N > 1 with a captured step: the backward pass is cut after this "
                    "trunk stage into two hipGraphs, the late layers' all-reduce runs under the second (-1: one graph, "
                    "all-reduce after it)")
    ap.add_argument("--no-deferred-sums", action="store_true",
                    help="a finish kernel per LayerNorm / bias gradient (default: one fold launch per step, xfmamba_amd/deferred.py)")
    ap.add_argument("--no-wgrad-arena", action="store_true",
                    help="a fresh zero-filled tensor per weight gradient (default: one arena, zeroed once per step)")
    ap.add_argument("--wgrad-stream", action="store_true",
                    help="weight-gradient kernels on a side stream = a parallel branch of the graph (measured: 1520 vs 1571 "
                         "samples/s on the main stream -- the branch competes for the CUs it was meant to fill; off by default)")
    ap.add_argument("--aten-profile", action="store_true", help="development: after the warm-up, one eager step under "
                    "torch.profiler; prints the framework (aten) operators that still launch kernels, by input shapes and "
                    "innermost package frame, to stderr")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one captured hipGraph per step")
    ap.add_argument("--report-conv-kernels", action="store_true", help="development: after the JSON line's measurements, one extra "
                    "eager step under torch.profiler to list the MIOpen / CK convolution solvers the find pass chose (off by "
                    "default: it nests a profiler inside rocprofv3 runs and adds an eager step to their kernel statistics)")
    ap.add_argument("--no-miopen-find", action="store_true", help="leave MIOpen's default solver heuristics (default: "
                    "torch.backends.cudnn.benchmark = True, i.e. MIOpen's own find pass during warm-up).  Four convolution "
                    "launches per step are left on the library since round 5 and the search still matters for them: on a fresh "
                    "box the heuristics picked a 244 us forward solver where the search finds a 48 us one (0.54 vs 0.23 ms per "
                    "step).  (A same-box A/B cannot show this: the first run's search result stays in the library's user "
                    "database and the run without the search finds it there.)")
    return ap.parse_args()


def cpu_baseline(batch=2, size=224):
    """Oracle model (pure torch CPU, sequential scan exactly as reference models/csms6s.py:25-68),
    one fwd+bwd step of BASELINE configs[0]; ~20-30 s of CPU work."""
    from oracle import xfm_oracle as O
    from xfmamba_amd.net_fusionmamba import TwoViewXFMambaTop
    # The sequential scan is thousands of tiny ops: more threads than ~16 only add contention (measured:
    # 250 s/step with torch's default 128 threads on the GPU host vs ~25 s with 8-16).
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    torch.set_num_threads(max(1, min(16, ncpu)))
    torch.manual_seed(42)
    sd = {k: v.detach().clone() for k, v in TwoViewXFMambaTop(1, 2, type="tiny").state_dict().items()}
    params = {k: v.requires_grad_() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    sd.update(params)
    xa, xb = torch.randn(batch, 1, size, size), torch.randn(batch, 1, size, size)
    lab = torch.randint(0, 2, (batch,))
    nstep = 2
    t0 = time.perf_counter()
    for _ in range(nstep):
        for v in params.values():
            v.grad = None
        out = O.xfmamba_top_ref(sd, xa, xb, True, O.selective_scan_ref)
        torch.nn.functional.cross_entropy(out, lab).backward()
    dt = (time.perf_counter() - t0) / nstep
    return dict(value=batch / dt, unit="two-view samples/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{nstep} fwd+bwd steps, XFMamba-T fp32, batch {batch}, 2x{size}x{size} (BASELINE configs[0]); "
                       f"oracle = restatement of the reference CPU selective-scan path; {dt:.1f} s/step")


def _library_conv_kernels(step):
    """The convolution kernels the library (MIOpen / composable_kernel) launches in one eager step -- the solvers its find pass
    chose for the four strided 3x3 convolutions and the stem on this machine -- as {kernel name: launches}; None on failure."""
    try:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        out = {}
        for e in prof.key_averages():
            n = e.key
            if any(k in n for k in ("igemm_", "_ZN2ck", "ck::", "miopen", "MIOpen", "naive_conv", "gemm_conv", "Conv")):
                short = n.split("(")[0][:96]
                out[short] = out.get(short, 0) + e.count
        return out
    except Exception:                                       # noqa: BLE001
        return None


def _aten_profile(step):
    """Which framework operators still launch kernels in one eager step (development aid for the element-wise tail)."""
    from torch.profiler import profile, ProfilerActivity
    step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
        t = e.self_device_time_total
        if t <= 0 or not e.key.startswith("aten::"):
            continue
        where = next((fr.split("xfmamba_amd/")[-1][:70] for fr in e.stack if "xfmamba_amd" in fr), "")
        rows.append((t, e.count, e.key, str(e.input_shapes)[:100], where))
    rows.sort(reverse=True)
    print(f"[aten-profile] {sum(r[0] for r in rows) / 1e3:.3f} ms in {sum(r[1] for r in rows)} calls", file=sys.stderr)
    for t, n, k, shp, where in rows[:120]:
        print(f"[aten-profile] {t:8.1f} us {n:4d}  {k:30s} {shp}  {where}", file=sys.stderr)


# dense forward MACs per two-view sample (scan excluded): SURVEY.md section 6 / BASELINE.md section 2, measured on the reference
_DENSE_GMAC = {("tiny", 224): 10.25, ("small", 224): 17.63, ("base", 224): 31.24, ("base", 384): 91.81}
_SCAN_MELEM = {("tiny", 224): 13.698048, ("small", 224): 34.771968, ("base", 384): 136.249344}   # sum K D L per sample (x 1e6)


def roofline_step(a, B, world, sec_per_step, kernels, ksteps):
    """The whole step against both roofs: dense FLOPs of forward + backward (3 x the forward MACs x 2;
    the reference's own count) against the bf16 matrix-core peak, and the bytes the hand-written kernels
    report at their own boundaries (every launch of the eager re-run, HIP-event timed) against the HBM peak -- the latter a
    LOWER bound of the step's traffic: library GEMMs / convolutions and framework element-wise kernels carry no byte count."""
    gmac = _DENSE_GMAC.get((a.model, a.size))
    out = {"ms_per_step": round(1e3 * sec_per_step, 3)}
    if gmac is not None:
        flops = 3 * 2 * gmac * 1e9 * B
        scan = _SCAN_MELEM.get((a.model, a.size))
        if scan is not None:
            flops += 3 * (9 + 1) * scan * 1e6 * B      # 9 K D L N + K D L per scan call (N = 1 trunk; the N = 16 blocks are < 3 %)
        out.update(dense_flops_per_step=int(flops), achieved_tflops=round(flops / sec_per_step / 1e12, 1), peak_tflops=2500.0,
                   frac_mfma=round(flops / sec_per_step / 1e12 / 2500.0, 4))
    if kernels:
        nb = sum(k["bytes"] for k in kernels.values()) / max(ksteps, 1)
        tk = sum(k["total_ms"] for k in kernels.values()) / max(ksteps, 1)
        out.update(own_kernel_bytes_per_step=int(nb), own_kernel_ms_per_step=round(tk, 3),
                   own_kernels_GBps=round(nb / (tk * 1e-3) / 1e9, 1) if tk > 0 else None,
                   step_GBps_lower_bound=round(nb / sec_per_step / 1e9, 1), frac_hbm_lower_bound=round(nb / sec_per_step / 1e9 / HBM_PEAK_GBS, 4))
    return out


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_with_watchdog(cmd, env, limit, what="the child"):
    """Run ``cmd`` as a fresh child process in its own process group and return its exit code; when it has not finished within
    ``limit`` seconds -- a rendezvous that never completes because a rank died before init_process_group -- end the whole GROUP
    (the launcher and its ranks; by id, never by pattern) and return 124."""
    import signal
    import subprocess
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)
    try:
        return proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        print(f"[bench] {what} did not finish within {limit:.0f} s: ending process group {proc.pid}", file=sys.stderr)
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
                proc.wait(timeout=20)
                break
            except ProcessLookupError:
                break
            except subprocess.TimeoutExpired:
                continue
        return 124


def self_launch(a):
    """``python bench.py --gpus N`` typed without a launcher: start the N ranks as a fresh child process
    (``python -m torch.distributed.run``; never an exec -- nothing in THIS process has touched the GPU yet and nothing will),
    forward its output and exit with its return code."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return _run_with_watchdog(cmd, env, float(os.environ.get("XFM_BENCH_LAUNCH_TIMEOUT", "1500")), f"the {a.gpus} ranks")


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:           # before ANY torch.cuda call
        sys.exit(self_launch(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # (development hook: XFM_BENCH_BACKEND=gloo runs the N > 1 flow -- captured forward/backward/packing, eager
    #  all-reduce + Adam -- with every rank on GPU 0 of a one-GPU box; never used for a reported number)
    backend = os.environ.get("XFM_BENCH_BACKEND", "nccl")
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run, or without a launcher)"
    cuda_ok = torch.cuda.is_available()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend != "nccl" and not cuda_ok:
            # a machine without GPUs (the CPU test of the launcher): the host-side rendezvous still proves that the ranks came
            # up; the assert below then stops them.  (With a GPU the group is created after the device is set, as for RCCL.)
            dist.init_process_group(backend)
            if rank == 0:
                print(f"[bench] ranks: {dist.get_world_size()} ({backend})", file=sys.stderr, flush=True)
    assert cuda_ok, "bench.py needs MI355X GPUs"
    if backend != "nccl":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)      # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)
        if rank == 0:
            print(f"[bench] {'RCCL ' if backend == 'nccl' else ''}ranks: {dist.get_world_size()}"
                  f"{'' if backend == 'nccl' else ' (' + backend + ')'}", file=sys.stderr, flush=True)

    from xfmamba_amd import _lib, fusion_vmamba
    from xfmamba_amd.amp import WeightCache
    from xfmamba_amd.optim import FusedAdam
    from xfmamba_amd.dp import GradBuckets, PhasedGrads, broadcast_parameters
    from xfmamba_amd.proj import WgradArena, join_wgrad_stream, set_wgrad_arena, wgrad_stream
    wgrad_stream(a.wgrad_stream)
    from xfmamba_amd.deferred import defer_partial_sums
    defer_partial_sums(not a.no_deferred_sums)       # (every gradient reader below goes through join_wgrad_stream() first)
    from xfmamba_amd.net_fusionmamba import TwoViewXFMambaTop
    _lib.lib()                                                   # fail loudly if the HIP extension is missing
    if a.fp8:
        from xfmamba_amd import fp8 as _fp8
        assert a.dtype == "bf16", "--fp8 rides on the bf16 configuration"
        _fp8.ENABLED = True
    if a.ss2d:
        fusion_vmamba.SS2D_MODE = a.ss2d
    if a.stream:
        fusion_vmamba.STREAM_LAYOUT = a.stream

    if not a.no_miopen_find:
        torch.backends.cudnn.benchmark = True
    torch.manual_seed(42)                                        # libs/config.py:22
    kw = dict(hidden_dim=1024) if a.model == "base" else {}
    model = TwoViewXFMambaTop(in_channels=1, outputs=2, type=a.model, **kw).to(dev).train()
    if a.drop_path is not None:
        for m in model.modules():
            if hasattr(m, "drop_prob"):
                m.drop_prob = a.drop_path
    if world > 1:
        broadcast_parameters(model)
    use_graph = not a.no_graph
    # What the hipGraph holds: the whole step at N = 1 ("step"); with several ranks (or --graph-scope fwdbwd) the
    # forward + backward + gradient packing, while the RCCL all-reduce of the flat buckets, Adam and the weight-shadow
    # refresh -- a few dozen launches -- are issued eagerly after each replay, so no collective is captured.
    scope = a.graph_scope or ("step" if world == 1 else "fwdbwd")
    assert not (world > 1 and scope == "step"), "RCCL collectives are not captured: use --graph-scope fwdbwd with N > 1"
    # gradients cross xGMI in the step's compute dtype: bf16 buckets for the bf16 step (half the bytes of the ring
    # all-reduce; the optimizer still reads fp32), fp32 for --dtype fp32 or --comm-dtype fp32
    comm = torch.bfloat16 if (a.comm_dtype or a.dtype) == "bf16" else None
    use_phased = world > 1 and use_graph and scope != "step" and a.dp_cut >= 0
    buckets = GradBuckets(model, bucket_mb=48.0, overlap=not use_graph, comm_dtype=comm, world=1 if use_phased else None)
    crit = torch.nn.CrossEntropyLoss()
    wcache = WeightCache(model) if a.dtype == "bf16" else None     # bf16 shadows of the GEMM weights
    # Adam of the reference loop (1_train_model.py:141) for all parameters in ONE launch that also rewrites the shadows
    # (csrc/adam.hip); --torch-adam: torch.optim.Adam(fused=True) + a multi-tensor shadow refresh, for A/B runs
    if a.torch_adam:
        opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-5, fused=True,
                               capturable=use_graph and scope == "step")
    else:
        opt = FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-5, weight_cache=wcache)

    torch.manual_seed(42 + rank)
    B = a.batch
    xa = torch.randn(B, 1, a.size, a.size, device=dev)           # synthetic, resident in HBM
    xb = torch.randn(B, 1, a.size, a.size, device=dev)
    lab = torch.randint(0, 2, (B,), device=dev)
    use_bf16 = a.dtype == "bf16"

    # weight-gradient accumulators: ONE zero fill per step instead of ~60 (the loop drops .grad before every backward)
    arena = None
    if not a.no_wgrad_arena and use_bf16:
        arena = WgradArena(model.parameters())
        set_wgrad_arena(arena)

    # N > 1 with a captured step: backward in two pieces around the activation entering trunk stage dp_cut + 1, one flat
    # wire bucket per piece (bf16 for the bf16 step), the late piece's all-reduce under the early piece's graph, Adam reading
    # the summed wire values in place (dp.PhasedGrads)
    phased = None
    names = {id(p_): n_ for n_, p_ in model.named_parameters()}
    if use_phased:
        phased = PhasedGrads(model, wire_dtype=comm or torch.float32)
        model.mamba_feature_extrac.cut_after = a.dp_cut

    def forward_loss():
        buckets.zero_grad()
        if arena is not None:
            arena.zero()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_bf16):
            out = model(xa, xb)
            loss = crit(out.float(), lab)
        return loss

    def fwd_bwd():
        loss = forward_loss()
        loss.backward()
        return loss

    def phase_a():                                   # graph A: forward, loss, backward down to the cut, pack bucket 0
        if os.environ.get("XFM_BENCH_FAIL_PHASED"):  # (development hook: exercises the one-graph fallback below)
            raise RuntimeError("forced by XFM_BENCH_FAIL_PHASED")
        loss = forward_loss()
        phased.backward_late(loss, model.mamba_feature_extrac.cut_tensor)
        return loss

    def phased_update():
        phased.wait()
        if a.torch_adam:
            phased.materialize()
            update()
        else:
            opt.step(grads=phased.grads(), grad_scale=phased.grad_scale)

    def phased_step():                               # eager form of the two-graph step (warm-up, kernel timing)
        if os.environ.get("XFM_BENCH_NANCHECK"):     # development: the same step with a finiteness check behind every phase
            st = phased_step.__dict__.setdefault("n", 0) + 1
            phased_step.__dict__["n"] = st

            def chk(tag, t):
                torch.cuda.synchronize()
                if not phased_step.__dict__.get("seen") and not bool(torch.isfinite(t.float()).all()):
                    phased_step.__dict__["seen"] = True
                    print(f"[bench] rank {rank} eager step {st}: first non-finite values in {tag} "
                          f"({int((~torch.isfinite(t.float())).sum())} of {t.numel()})", file=sys.stderr)
            loss = phase_a()
            chk("loss", loss)
            for pp, vv in zip(phased.pieces[0], phased.views[0]):
                chk("bucket 0 slot of " + names[id(pp)], vv)
            phased.reduce(0, overlap=False)
            phased.wait()
            chk("bucket 0 after its all-reduce", phased.flat[0])
            phased.backward_early()
            torch.cuda.synchronize()
            bad1 = [names[id(pp)] for pp, vv in zip(phased.pieces[1], phased.views[1]) if not bool(torch.isfinite(vv.float()).all())]
            if bad1 and not phased_step.__dict__.get("seen"):
                order = [names[id(pp)] for pp in phased.pieces[1]]
                print(f"[bench] rank {rank} eager step {st}: {len(bad1)} of {len(order)} early gradients non-finite; slots in bucket order: "
                      + ", ".join(f"{i}:{n}" for i, n in enumerate(order) if n in bad1)[:3000], file=sys.stderr)
            for pp, vv in zip(phased.pieces[1], phased.views[1]):
                chk("bucket 1 slot of " + names[id(pp)], vv)
            phased.reduce(1, overlap=False)
            phased_update()
            chk("bucket 1 after its all-reduce", phased.flat[1])
            for n_, p_ in model.named_parameters():
                chk("parameter " + n_ + " after the update", p_)
            return loss
        loss = phase_a()
        phased.reduce(0)
        phased.backward_early()
        phased.reduce(1)
        phased_update()
        return loss

    def update():
        join_wgrad_stream()                          # (FusedAdam joins by itself; torch's Adam does not know the side stream)
        opt.step()
        if wcache is not None and a.torch_adam:
            wcache.refresh()

    def step():                                      # one eager training step
        if phased is not None:
            return phased_step()
        loss = fwd_bwd()
        buckets.finish()
        update()
        return loss

    def captured_part():                             # what goes into the graph
        loss = fwd_bwd()
        if scope == "step":
            buckets.finish()                         # world == 1: nothing to reduce
            update()
        elif world > 1:
            buckets.pack_all()
        return loss

    def after_replay():                              # eager tail of a replayed step
        if scope != "step":
            if world > 1:
                buckets.reduce_all()
            update()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the whole step (fwd + bwd + Adam) as ONE hipGraph: ~2800 launches per step are replayed by the
    # runtime instead of being issued one by one from Python ("HIP graphs instead of a tracing compiler")
    graph = None
    graph_b = None
    loss_static = None
    dp_fallback = None                               # why the two-graph data-parallel step is not the one being timed

    def capture():
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(3, min(a.warmup, 5))):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g, gb = torch.cuda.CUDAGraph(), None
        if phased is not None:
            # two graphs over one memory pool, always replayed A then B: B reads the activations A's forward saved
            with torch.cuda.graph(g):
                ls = phase_a()
            gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gb, pool=g.pool()):
                phased.backward_early()
        else:
            with torch.cuda.graph(g):
                ls = captured_part()
        return g, gb, ls

    if use_graph:
        try:
            graph, graph_b, loss_static = capture()
        except Exception as e:                       # noqa: BLE001
            torch.cuda.synchronize()
            from xfmamba_amd import deferred as _deferred
            _deferred.release_capture_tables(dev)    # the abandoned capture's job tables go back to the pool
            if phased is not None:
                # the two-graph data-parallel step did not come up (every rank runs the same code and fails alike): fall
                # back to ONE graph for forward + backward + packing and the all-reduce after it
                print(f"[bench] two-graph step failed ({type(e).__name__}: {e}); one graph + all-reduce after it", file=sys.stderr)
                dp_fallback = f"fallback: one graph (two-graph capture failed: {type(e).__name__}: {str(e)[:120]})"
                phased = None
                model.mamba_feature_extrac.cut_after = None
                buckets = GradBuckets(model, bucket_mb=48.0, overlap=False, comm_dtype=comm)
                try:
                    graph, graph_b, loss_static = capture()
                except Exception as e2:              # noqa: BLE001  (report and fall back to eager launches)
                    print(f"[bench] graph capture failed, running eager: {type(e2).__name__}: {e2}", file=sys.stderr)
                    dp_fallback = f"fallback: eager launches (graph capture failed: {type(e2).__name__}: {str(e2)[:120]})"
                    graph = graph_b = None
                    torch.cuda.synchronize()
            else:
                print(f"[bench] graph capture failed, running eager: {type(e).__name__}: {e}", file=sys.stderr)
                graph = graph_b = None

    nan_check = bool(os.environ.get("XFM_BENCH_NANCHECK"))     # development: where a non-finite value first appears in the two-graph step
    nan_state = {"step": 0, "seen": False}

    def _finite(tag, t):
        torch.cuda.synchronize()
        if not nan_state["seen"] and not bool(torch.isfinite(t.float()).all()):
            nan_state["seen"] = True
            bad = int((~torch.isfinite(t.float())).sum())
            print(f"[bench] rank {rank} step {nan_state['step']}: first non-finite values in {tag} ({bad} of {t.numel()})", file=sys.stderr)

    def run_step():
        if graph_b is not None and nan_check:
            nan_state["step"] += 1
            graph.replay()
            _finite("loss after graph A", loss_static)
            _finite("bucket 0 after graph A (packed late gradients)", phased.flat[0])
            phased.reduce(0, overlap=False)
            phased.wait()
            _finite("bucket 0 after its all-reduce", phased.flat[0])
            graph_b.replay()
            _finite("bucket 1 after graph B (packed early gradients)", phased.flat[1])
            phased.reduce(1, overlap=False)
            phased_update()
            _finite("bucket 1 after its all-reduce", phased.flat[1])
            for n_, p_ in model.named_parameters():
                if not nan_state["seen"]:
                    _finite("parameter " + n_ + " after the update", p_)
            return loss_static
        if graph_b is not None:
            graph.replay()
            phased.reduce(0)                         # communication stream: runs under graph B
            graph_b.replay()
            phased.reduce(1)
            phased_update()
            return loss_static
        if graph is not None:
            graph.replay()
            after_replay()
            return loss_static
        return step()

    for _ in range(a.warmup):
        run_step()
    if a.aten_profile and world == 1:
        _aten_profile(step)
    timer = None
    if not a.no_kernel_timer and graph is None:
        timer = _lib.KernelTimer()
        _lib.set_timer(timer)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = run_step()
    fence()
    dt = time.perf_counter() - t0
    _lib.set_timer(None)
    if graph is not None and not a.no_kernel_timer:
        # per-kernel HIP-event timing cannot live inside a captured graph: the same step is run eagerly, with
        # events around every hand-written launch, right after the timed region (not part of `value`)
        timer = _lib.KernelTimer()
        _lib.set_timer(timer)
        ksteps = min(a.steps, 5)
        for _ in range(ksteps):
            step()
        torch.cuda.synchronize()
        _lib.set_timer(None)
    else:
        ksteps = a.steps
    conv_kernels = None
    if world == 1 and a.report_conv_kernels:         # (one process only: an extra step on rank 0 alone would hang the collectives)
        conv_kernels = _library_conv_kernels(step)   # (after the timed region: which MIOpen / CK solvers the find pass chose)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(loss.detach()).item(), "loss diverged"

    if rank == 0:
        value = a.steps * B * world / dt
        metric_name = f"two-view images/sec fwd+bwd, XFMamba-{a.model[0].upper()} {a.size}^2, batch {B}/GPU"
        if (a.model, a.size, B) == ("tiny", 224, 32):            # the headline configuration: BASELINE.json's own wording
            try:
                metric_name = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
            except Exception:                                   # noqa: BLE001
                pass
        cfg_label = {("tiny", 224, 32, "bf16"): " (BASELINE configs[4]: fp8 x_proj / out_proj weights)" if a.fp8 else " (BASELINE configs[1])", ("small", 224, 32, "bf16"): " (BASELINE configs[2], one GPU of it)",
                     ("base", 384, 16, "bf16"): " (BASELINE configs[3])"}.get((a.model, a.size, B, a.dtype), "")
        roof = None
        roofs = {}
        mfma = None
        kernels = {}
        if timer is not None:
            kernels = timer.summary()
            if kernels:
                import glob
                tj, tf = {}, None
                try:
                    tf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))[-1]
                    tj = json.load(open(tf))
                except Exception:      # noqa: BLE001
                    pass
                # timer name -> kernel key of tools/pmc_traffic.py (HBM bytes per launch from the PMC counters of this very
                # command: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate --pmc passes; MI355X_MICROARCH.md "HBM")
                pmc_key = {"ss2d_bwd": "ss2d_w_bwd_kernel", "ss2d_fwd": "ss2d_w_fwd_kernel",
                           "ss2dc_bwd": "chan1::bwd_kernel", "ss2dc_fwd": "chan1::fwd_kernel",
                           "ss2dc16_bwd": "deep_bwd_kernel<3, 4>", "ss2dc16_fwd": "ss2dc_fwd_kernel_n16",
                           "ss2dc16s_bwd": "deep_bwd_kernel<3, 1>", "ss2dc16s_fwd": "deep_fwd1_kernel"}

                def roof_of(name):
                    k = kernels[name]
                    ach = k["bytes"] / (k["total_ms"] * 1e-3) / 1e9
                    traffic = (tj.get(pmc_key.get(name, name + "_kernel")) or {}).get("hbm_bytes_per_launch")
                    src = None
                    if traffic is not None:
                        src = (f"offline rocprofv3 PMC passes of this command (FETCH_SIZE x2 + WRITE_SIZE), "
                               f"profiles/{os.path.basename(tf)}")
                    r = dict(bound="hbm", kernel=name, achieved=round(ach, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic, traffic_source=src,
                             launches=k["launches"], avg_launch_us=round(k["avg_us"], 2),
                             algorithmic_bytes_per_launch=int(k["bytes"] / k["launches"]),
                             ms_per_step=round(k["total_ms"] / ksteps, 3))
                    if "bytes_alt" in k:
                        # the kernel keeps the step sizes (dts / ddts) in HBM; at SURVEY 8(d)'s "dt_proj fused as well"
                        # boundary those bytes do not count: the same time against the smaller byte count
                        ach_f = k["bytes_alt"] / (k["total_ms"] * 1e-3) / 1e9
                        r["dt_proj_fused_boundary"] = dict(achieved=round(ach_f, 1), frac=round(ach_f / HBM_PEAK_GBS, 4),
                                                           algorithmic_bytes_per_launch=int(k["bytes_alt"] / k["launches"]))
                    return r

                # the north-star kernel family: the fused SS2D scans (wide-map kernels at 56x56 / 28x28, channel-lane kernels at
                # 14x14 / 7x7, d_state 16 variants in the fusion blocks), all in `roofline_scan_kernels`.  `roofline` is the kernel
                # VERDICT.md names -- the wide-map backward (csrc/ss2d_w.hpp; BASELINE.json's ">= 40 % in the scan kernel") --
                # whichever scan family costs most per step (since round 6 the 14x14 channel-lane backward does, by a few
                # percent: the rule "most ms per step" would flip between the two from box to box)
                scan = [k for k in kernels if k.startswith("ss2d") and not k.endswith("_finish")]
                roofs = {k: roof_of(k) for k in scan}
                if scan:
                    roof = roofs["ss2d_bwd"] if "ss2d_bwd" in roofs else roofs[max(scan, key=lambda k: kernels[k]["total_ms"])]
                    roof["costliest_scan_kernel_per_step"] = max(scan, key=lambda k: kernels[k]["total_ms"])
                    if "ss2d_bwd_finish" in kernels:
                        roof["timing"] = ("HIP events recorded by the library right around the scan kernel (xfm_prof_main_kernel); the sum of "
                                          "the workgroups' partial dB / dC rows behind it is `kernels.ss2d_bwd_finish`")
                try:                                   # matrix-core utilisation of the GEMM kernels, offline PMC pass
                    mf = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mfma_util.json")))[-1]
                    mj = json.load(open(mf))
                    top = sorted(mj.items(), key=lambda kv: -kv[1]["launches"] * kv[1]["avg_kernel_cycles"])[:12]
                    mfma = dict(source=f"offline rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs)), "
                                       f"profiles/{os.path.basename(mf)}",
                                kernels={k: v["mfma_util"] for k, v in top})
                except Exception:      # noqa: BLE001
                    pass
        line = {
            "metric": metric_name,
            "value": round(value, 2), "unit": "two-view samples/s (1 sample = 2 images)",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"XFMamba-{a.model[0].upper()} ({a.dtype}), 2x{a.size}x{a.size}, batch {B}/GPU, "
                                   f"fwd+bwd+Adam, train mode" + cfg_label,
                       "note": "outnorm0-2 of the trunk are skipped: the reference computes them and discards the "
                               "results (net_fusionmamba.py:200-201); they carry no gradient",
                       "global_batch": B * world, "parallelism": f"dp{world}",
                       "ranks": (dist.get_world_size() if world > 1 else 1), "collective_backend": ("RCCL" if backend == "nccl" else backend) if world > 1 else None, "grad_allreduce": ("bf16" if comm is not None else "fp32") if world > 1 else None,
                       "dp_step": (f"two hipGraphs cut after trunk stage {a.dp_cut}: the late layers' all-reduce runs under the early "
                                   f"layers' backward; Adam reads the summed wire bucket in place" if graph_b is not None else
                                   (None if world == 1 else (dp_fallback or ("one graph (forward + backward + packing), all-reduce after it"
                                                                             if graph is not None else "eager: bucket all-reduce overlapped with backward")))),
                       "ss2d_mode": fusion_vmamba.SS2D_MODE, "fp8_proj": bool(a.fp8),
                       "single_view_images_per_s": round(2 * value, 2)},
            "roofline": roof,
            "roofline_step": roofline_step(a, B, world, dt / a.steps, kernels, ksteps),
            "roofline_scan_kernels": roofs,
            "mfma_util": mfma,
            "kernels": {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 2),
                            "GBps": round(v["bytes"] / (v["total_ms"] * 1e-3) / 1e9, 1),
                            "ms_per_step": round(v["total_ms"] / ksteps, 3)} for k, v in kernels.items()},
            "launch_mode": (f"hipGraph({scope})" if graph is not None else "eager"),
            "library_conv_kernels": conv_kernels,
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
