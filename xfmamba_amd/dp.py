"""Batch-sharded data parallelism for the XFMamba hot path: gradients only, over RCCL/xGMI.

The reference has no distributed code; this is new work required by
BASELINE.json's north star.  One process per GPU, identical replicas, per-rank batches.

Design for MI355X's point-to-point xGMI fabric (7 links per GPU): few, LARGE collectives.
Autograd writes every gradient into a fresh tensor (``zero_grad`` drops the old ones, so there is neither a
zero-fill nor an accumulate kernel per parameter); when the last gradient of a bucket has arrived the bucket is
packed with ONE multi-tensor copy into a flat fp32 buffer, the parameters' ``.grad`` are re-pointed at views of that
buffer, and the bucket's all-reduce is issued on a side stream, overlapping the rest of the backward pass.
With a single process nothing is packed at all.  Parameters that never receive a gradient (``outnorm0-2``,
``Cross_SS2Dv5.in_proj``) are the same on every rank: their slots are zero-filled at
``finish()``.  BatchNorm statistics of the shallow fusion block stay per-rank (the reference has no SyncBN);
``broadcast_buffers`` is offered for checkpoint time.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist

__all__ = ["GradBuckets", "PhasedGrads", "broadcast_parameters", "broadcast_buffers"]


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """Rank ``src``'s weights to every rank.  The parameter itself is the collective's output (under ``no_grad``), so
    its version counter moves and low-precision weight shadows (``amp.WeightCache``) of the old values are not
    served afterwards; they are dropped explicitly as well."""
    from .amp import invalidate_shadows
    with torch.no_grad():
        for p in module.parameters():
            dist.broadcast(p, src)
    invalidate_shadows(module)


def broadcast_buffers(module: torch.nn.Module, src: int = 0) -> None:
    for b in module.buffers():
        dist.broadcast(b, src)


class GradBuckets:
    """Gradient buckets with overlap-capable all-reduce (average)."""

    def __init__(self, module: torch.nn.Module, bucket_mb: float = 48.0, process_group=None, overlap: bool = True,
                 world: int = None, comm_dtype: torch.dtype = None):
        """``comm_dtype=torch.bfloat16``: the collective moves bf16 (half the bytes on the xGMI links).  Each rank's
        gradient is scaled by 1/world in fp32, rounded once to bf16 into a staging buffer, summed by the all-reduce
        in bf16 and widened back into the fp32 flat buffer the optimizer reads; ``None`` / fp32 reduces the fp32
        buffer in place."""
        self.group = process_group
        self.comm_dtype = None if comm_dtype in (None, torch.float32) else comm_dtype
        self._stage: List[torch.Tensor] = []
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        if world is not None:                               # (tests: exercise the packing path without peers)
            self.world = int(world)
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.buckets: List[torch.Tensor] = []
        self._groups: List[List[torch.nn.Parameter]] = []
        self._views: List[List[torch.Tensor]] = []
        self._bucket_of = {}
        self._work = []
        self._stream = None
        self.overlap = overlap and self.world > 1
        if self.world == 1:
            return
        order = list(reversed(self.params))                # gradients become ready roughly in reverse order
        cap = int(bucket_mb * (1 << 20) / 4)
        cur: List[torch.nn.Parameter] = []
        cur_n = 0
        for p in order:
            if cur and cur_n + p.numel() > cap:
                self._groups.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self._groups.append(cur)
        for bi, grp in enumerate(self._groups):
            flat = torch.zeros(sum(p.numel() for p in grp), dtype=torch.float32, device=grp[0].device)
            views, off = [], 0
            for p in grp:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
                self._bucket_of[p] = bi
            self.buckets.append(flat)
            self._views.append(views)
            if self.comm_dtype is not None:
                self._stage.append(torch.zeros_like(flat, dtype=self.comm_dtype))
        self._sizes = [len(g) for g in self._groups]
        self._pending = list(self._sizes)
        self._launched = [False] * len(self._groups)
        if self.overlap:
            if self.buckets[0].is_cuda:
                self._stream = torch.cuda.Stream()
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    # ---- backward-time hook -----------------------------------------------------------------
    def _on_grad(self, p):
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def _pack(self, bi):
        """Copy the bucket's gradients into its flat buffer (one multi-tensor kernel) and re-point ``.grad``."""
        from .proj import join_wgrad_stream
        # (weight gradients launched on the side stream; deferred column sums.  A bucket hook runs in the middle of the pass)
        join_wgrad_stream(end_of_pass=False)
        grp, views = self._groups[bi], self._views[bi]
        have = [(v, p.grad) for v, p in zip(views, grp) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v, p in zip(views, grp):
            if p.grad is None:
                v.zero_()                                   # never-used parameter: same on every rank
            p.grad = v

    def _reduce(self, bi, async_op=True):
        flat = self.buckets[bi]
        if self.comm_dtype is None:
            flat.div_(self.world)
            w = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            if async_op:
                self._work.append(w)
            return
        stage = self._stage[bi]
        torch.mul(flat, 1.0 / self.world, out=stage)        # scale in fp32, ONE rounding to the wire dtype
        w = dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if async_op:
            self._work.append((w, bi))
        else:
            flat.copy_(stage)

    def _wait(self, w):
        if isinstance(w, tuple):                            # reduced-precision wire: widen back once the sum is in
            w[0].wait()
            self.buckets[w[1]].copy_(self._stage[w[1]])
        else:
            w.wait()

    def _launch(self, bi):
        if self._launched[bi]:
            return
        self._launched[bi] = True
        self._pack(bi)
        if self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                self._reduce(bi)
        else:
            self._reduce(bi)

    # ---- split form for a captured forward/backward: pack inside the graph, reduce outside -------------------
    def pack_all(self):
        """Pack every bucket (no communication).  Safe to capture in a hipGraph: after the capture ``.grad`` of
        every parameter is a view of a flat buffer the replayed pack kernels refill."""
        for bi in range(len(self.buckets)):
            self._pack(bi)

    def reduce_all(self):
        """All-reduce (average) the packed buckets on the current stream."""
        for bi in range(len(self.buckets)):
            self._reduce(bi, async_op=False)

    # ---- call after loss.backward() ------------------------------------------------------------
    def finish(self):
        """Flush buckets whose parameters did not all fire (unused parameters) and wait."""
        if self.world == 1:
            return
        for bi in range(len(self.buckets)):
            self._launch(bi)
        if self._stream is not None:
            with torch.cuda.stream(self._stream):           # (the widening copies queue behind their collectives)
                for w in self._work:
                    self._wait(w)
            torch.cuda.current_stream().wait_stream(self._stream)
        else:
            for w in self._work:
                self._wait(w)
        self._work.clear()
        self._pending = list(self._sizes)
        self._launched = [False] * len(self.buckets)
        from .proj import join_wgrad_stream
        join_wgrad_stream()                                 # end of the pass

    def zero_grad(self):
        """Drop the gradients: the next backward pass assigns fresh tensors instead of accumulating."""
        for p in self.params:
            p.grad = None


# ---------------------------------------------------------------------------------------------------------------------------
# Two-piece backward for captured steps: the all-reduce of the late layers' gradients runs under the early layers' backward
# ---------------------------------------------------------------------------------------------------------------------------
def _reachable_leaves(roots, stop=None):
    """Parameters (AccumulateGrad leaves) reachable from the autograd nodes ``roots`` without passing through ``stop``."""
    seen, out, stack = set(), [], [r for r in roots if r is not None]
    while stack:
        fn = stack.pop()
        if fn is None or fn is stop or fn in seen:
            continue
        seen.add(fn)
        var = getattr(fn, "variable", None)
        if var is not None:
            out.append(var)
        stack.extend(nf for nf, _ in fn.next_functions)
    return out


class PhasedGrads:
    """Data-parallel gradients of a step whose backward pass is cut in two at one activation (``cut``).

    A captured step (hipGraph) has no hooks to hang collectives on, and one graph for the whole backward pass leaves the
    all-reduce exposed after it.  XFMamba's parameters sit almost entirely in the LATE layers (768-channel stage, the
    fusion blocks: > 90 % of the bytes) while the EARLY layers (56 x 56 / 28 x 28 maps) take about half of the backward
    time -- so the step is captured as two graphs::

        graph A:  forward, loss, backward from the loss down to ``cut``          -> late gradients, packed into bucket 0
        (eager)   all-reduce of bucket 0 on the communication stream                (RCCL ring over xGMI)
        graph B:  backward from ``cut`` to the input                             -> early gradients, packed into bucket 1
        (eager)   all-reduce of bucket 1, join, optimizer

    and bucket 0's ring runs under graph B.  The buckets ARE the wire: one multi-tensor copy rounds the fp32 gradients into
    a flat bf16 buffer (``wire_dtype``), RCCL sums it in place, and ``FusedAdam.step(grads=..., grad_scale=1 / world)``
    reads the summed bf16 values where they lie -- no fp32 staging copy before and no widening copy after the collective.

    ``backward_late(loss, cut)`` / ``backward_early()`` are plain ``torch.autograd.grad`` calls over disjoint parts of
    the graph (the first call partitions the parameters by walking the autograd graph: late = reachable from the loss
    without passing ``cut``; a parameter feeding both sides would make the split unsound and raises).  Parameters no
    piece reaches keep a zero slot and are left out of ``grads()`` (torch's optimizers skip ``grad is None`` too).

    Capturing: as for any whole-network capture, run the warm-up steps -- INCLUDING the very first one, which plans the
    pieces -- on a side stream before ``torch.cuda.graph`` (seen on ROCm 7.2: a first backward pass issued on the default
    stream makes ``capture_end`` of graph B crash); capture B with ``pool=graph_a.pool()`` and always replay A then B.
    """

    def __init__(self, module: torch.nn.Module, process_group=None, wire_dtype: torch.dtype = torch.bfloat16, world: int = None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        if world is not None:
            self.world = int(world)
        self.wire_dtype = wire_dtype
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.pieces = None                   # [late params, early params] in gradient-arrival order
        self.flat: List[torch.Tensor] = []
        self.views: List[List[torch.Tensor]] = []
        self._view_of = {}
        self._gcut = None
        self._cut = None
        self._work = []
        self._stream = None

    # ---- planning (first step) ---------------------------------------------------------------------------------------
    def _plan(self, loss, cut):
        if cut.grad_fn is None:
            raise RuntimeError("PhasedGrads: the cut tensor must be an activation inside the autograd graph")
        mine = {id(p) for p in self.params}
        post = [p for p in _reachable_leaves([loss.grad_fn], stop=cut.grad_fn) if id(p) in mine]
        pre = [p for p in _reachable_leaves([cut.grad_fn]) if id(p) in mine]
        both = {id(p) for p in post} & {id(p) for p in pre}
        if both:
            raise RuntimeError(f"PhasedGrads: {len(both)} parameter(s) feed both sides of the cut; choose another cut")
        self.pieces = [post, pre]
        dev = self.params[0].device
        for grp in self.pieces:
            flat = torch.zeros(max(1, sum((p.numel() + 7) // 8 * 8 for p in grp)), dtype=self.wire_dtype, device=dev)
            views, off = [], 0
            for p in grp:
                v = flat[off:off + p.numel()].view(p.shape)
                views.append(v)
                self._view_of[p] = v
                off += (p.numel() + 7) // 8 * 8           # 16-byte aligned slots (vector loads of the optimizer)
            self.flat.append(flat)
            self.views.append(views)
        if dev.type == "cuda":
            self._stream = torch.cuda.Stream(device=dev)

    def _pack(self, i, grads):
        have = [(v, g) for v, g in zip(self.views[i], grads) if g is not None]
        if have:
            torch._foreach_copy_([v for v, _ in have], [g for _, g in have])
        for v, g in zip(self.views[i], grads):
            if g is None:
                v.zero_()                                  # (a parameter the graph reaches but this step does not use)

    # ---- the two pieces ------------------------------------------------------------------------------------------------
    def backward_late(self, loss: torch.Tensor, cut: torch.Tensor):
        """Backward from ``loss`` down to ``cut``; the late gradients land in bucket 0."""
        from .proj import join_wgrad_stream
        if self.pieces is None:
            self._plan(loss, cut)
        late = self.pieces[0]
        out = torch.autograd.grad(loss, [cut] + late, allow_unused=True)
        self._cut, self._gcut = cut, out[0]
        join_wgrad_stream(end_of_pass=False)
        self._pack(0, out[1:])

    def backward_early(self):
        """Backward from the cut to the inputs; the early gradients land in bucket 1."""
        from .proj import join_wgrad_stream
        early = self.pieces[1]
        if early and self._gcut is not None:
            out = torch.autograd.grad(self._cut, early, grad_outputs=self._gcut, allow_unused=True)
        else:
            out = [None] * len(early)
        self._cut = self._gcut = None
        join_wgrad_stream()
        self._pack(1, out)

    # ---- collectives ----------------------------------------------------------------------------------------------------
    def reduce(self, i: int, overlap: bool = True):
        """SUM all-reduce of bucket ``i`` in place.  With ``overlap`` it is queued on the communication stream behind the
        work issued so far on the current stream and ``wait()`` joins it; later launches on the current stream (graph B)
        run beside it."""
        if self.world == 1:
            return
        if overlap and self._stream is not None:
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                self._work.append(dist.all_reduce(self.flat[i], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self._work.append(dist.all_reduce(self.flat[i], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        if self._stream is not None and self._work:
            with torch.cuda.stream(self._stream):
                for w in self._work:
                    w.wait()
            torch.cuda.current_stream().wait_stream(self._stream)
        else:
            for w in self._work:
                w.wait()
        self._work.clear()

    # ---- consumers --------------------------------------------------------------------------------------------------------
    @property
    def grad_scale(self) -> float:
        return 1.0 / self.world

    def grads(self):
        """{parameter: view of its wire slot} (the SUM over the ranks after ``reduce`` + ``wait``; multiply by
        ``grad_scale``) -- what ``FusedAdam.step(grads=..., grad_scale=...)`` takes."""
        return self._view_of

    def materialize(self):
        """``p.grad`` = averaged fp32 gradient for every reached parameter (library optimizers, clipping, tests)."""
        for p, v in self._view_of.items():
            p.grad = v.to(torch.float32) * self.grad_scale

    def zero_grad(self):
        for p in self.params:
            p.grad = None
