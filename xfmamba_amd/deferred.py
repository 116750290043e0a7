"""Deferred column sums: one fold kernel per step for the partial rows of every LayerNorm / bias gradient.

The row-LayerNorm backward, bias+GELU backward and the bias column sums end in a small "finish" kernel that folds
per-workgroup partial rows into the parameter gradient: ~50 launches of 6-7 us per XFMamba-T step whose results nobody
reads before the optimizer.  With ``defer_partial_sums(True)`` the producers leave their partial rows in their workspaces
(null result pointer at the C ABI), register a job here, and ONE ``xfm_partial_sums_multi`` launch folds them all when
``flush()`` runs -- ``proj.join_wgrad_stream()`` calls it, i.e. ``FusedAdam.step`` and the data-parallel packers do before
they read a gradient.

Opt-in (``bench.py`` turns it on): between ``backward()`` and the flush the affected ``.grad`` tensors are allocated but
NOT yet filled, so code that reads gradients straight after ``backward()`` must call ``flush()`` (or leave this off).

What is deferred, and what is not (``add_job`` decides; a refused job runs the producer's own finish kernel):
* only results that are the gradient of an **fp32 ``nn.Parameter`` whose ``.grad`` is None** (``set_to_none`` semantics):
  autograd then ADOPTS the result tensor as ``.grad`` and the fold fills it in place.  With an existing ``.grad``
  (``zero_grad(set_to_none=False)``, micro-batch accumulation) autograd would add the unfilled tensor at once, and for a
  reduced-precision parameter the producer's ``.to(dtype)`` would copy it -- neither is deferred;
* a parameter that receives a second gradient in the same pass (shared weights) is flushed before its second producer
  runs and that producer is not deferred, so autograd never adds unfilled tensors.

Job tables.  The fold kernel reads a job table and a block table that are uploaded from pinned host memory.  A memcpy node
of a captured graph re-reads its pinned source at EVERY replay, so every flush issued under capture takes a table pair of
its own from a pool (allocated outside captures, topped up at every eager flush) and that pair is never written again:
two graphs of one step (``dp.PhasedGrads``) or several flushes inside one graph each replay their own job set.
"""
from __future__ import annotations

import os

import torch

from . import _lib

__all__ = ["defer_partial_sums", "deferring", "add_job", "flush", "reserve_capture_tables"]

_MAX_JOBS, _MAX_BLOCKS = 512, 1 << 15
_CAP_POOL_MIN = 8                           # free table pairs kept ready for flushes under capture
_S = {"on": False, "jobs": [], "keys": set(), "eager": {}, "cap_free": {}, "cap_used": {}}


def defer_partial_sums(enable: bool) -> None:
    if not enable:
        flush()
    _S["on"] = bool(enable)


def deferring() -> bool:
    return _S["on"]


def _new_tables(device):
    return dict(jobs_h=torch.empty(6 * _MAX_JOBS, dtype=torch.int64).pin_memory(),
                blocks_h=torch.empty(_MAX_BLOCKS, dtype=torch.int32).pin_memory(),
                jobs_d=torch.empty(6 * _MAX_JOBS, dtype=torch.int64, device=device),
                blocks_d=torch.empty(_MAX_BLOCKS, dtype=torch.int32, device=device), key=None, evt=None, nblocks=0, nrows=0)


def reserve_capture_tables(device, n: int = _CAP_POOL_MIN) -> None:
    """Make sure ``n`` unused table pairs exist for flushes under capture (call outside a capture; every eager flush does)."""
    if torch.cuda.is_current_stream_capturing():
        return
    free = _S["cap_free"].setdefault(str(device), [])
    while len(free) < n:
        free.append(_new_tables(device))


def release_capture_tables(device=None) -> None:
    """Hand the table pairs of captured flushes back to the pool.  Call it when the graphs that recorded them are gone -- a
    capture that failed, a graph that was destroyed or is about to be re-captured -- and never while such a graph can still be
    replayed (its memcpy nodes read the pinned tables at every replay)."""
    for dev, used in _S["cap_used"].items():
        if device is None or dev == str(device):
            for sl in used:
                sl["key"] = None
            _S["cap_free"].setdefault(dev, []).extend(used)
            used.clear()


def _tables(device, capturing, key=None):
    """Eager flushes share one table pair per device (rewritten when the job set changes, behind an event).  A flush under
    capture takes a pair from the pool; the pair stays with the graph (its memcpy nodes read it at replay) until
    ``release_capture_tables``.  A captured flush of a job set that an earlier capture already recorded (the same step captured
    again) reuses that pair -- same contents, read-only at replay -- instead of taking another one."""
    dev = str(device)
    if not capturing:
        sl = _S["eager"].get(dev)
        if sl is None:
            sl = _S["eager"][dev] = _new_tables(device)
        reserve_capture_tables(device)
        return sl
    for sl in _S["cap_used"].get(dev, ()):
        if key is not None and sl["key"] == key:
            return sl
    free = _S["cap_free"].get(dev)
    if not free:
        raise RuntimeError("xfmamba_amd.deferred: no job table left for a flush under capture -- run a warm-up step with "
                           "deferred sums (or deferred.reserve_capture_tables(device, n)) before capturing")
    sl = free.pop()
    _S["cap_used"].setdefault(dev, []).append(sl)
    return sl


def _deferrable(p) -> bool:
    return isinstance(p, torch.nn.Parameter) and p.dtype == torch.float32 and p.grad is None


def add_job(part: torch.Tensor, outs, nblk: int, C: int, nparts: int, params=()) -> bool:
    """Register ``outs[k][c] = sum_j part[(j * nparts + k) * C + c]`` for the next flush.  ``params[k]``: the parameter
    ``outs[k]`` is the gradient of (None where ``outs[k]`` is None).  False: not deferred (the caller runs its own finish
    kernel) -- deferral is off, a result is not the gradient of an fp32 parameter without ``.grad``, or a parameter already
    got a gradient in this pass."""
    if not _S["on"]:
        return False
    params = list(params) + [None] * (len(outs) - len(params))
    for o, p in zip(outs, params):
        if o is not None and not _deferrable(p):
            return False
    keys = [id(p) for o, p in zip(outs, params) if o is not None]
    if any(k in _S["keys"] for k in keys):
        # second gradient of the same parameter in one pass (shared weights, two trunk calls): autograd adds the two as
        # soon as this one is returned, so the first must be complete NOW and this one is not deferred
        flush(_end_of_pass=False)
        return False
    if len(_S["jobs"]) >= _MAX_JOBS:
        flush(_end_of_pass=False)
    # only the ADDRESSES of the results are kept: a second reference to a gradient tensor would make autograd's
    # AccumulateGrad clone it instead of adopting it as .grad, and the fold would fill the orphan.  The tensors stay alive as
    # the .grad of their parameters (or in the tuple torch.autograd.grad returns) until the flush.
    ptrs = [0 if o is None else o.data_ptr() for o in outs] + [0] * (3 - len(outs))
    _S["jobs"].append((part, ptrs, int(nblk), int(C), int(nparts)))
    _S["keys"].update(keys)
    return True


@torch.no_grad()
def flush(_end_of_pass: bool = True) -> None:
    """Fold every registered job (one launch on the current stream).  No-op without jobs.  (The flushes ``add_job`` issues
    in the middle of a backward pass keep the set of parameters seen in this pass: a later second gradient of any of them
    must not be deferred either.)"""
    jobs = _S["jobs"]
    if _end_of_pass:
        _S["keys"] = set()
    if not jobs:
        return
    _S["jobs"] = []
    dev = jobs[0][0].device
    capturing = torch.cuda.is_current_stream_capturing()
    key = tuple((p.data_ptr(), tuple(outs), nblk, C, nparts) for p, outs, nblk, C, nparts in jobs)
    sl = _tables(dev, capturing, key)
    if key != sl["key"]:
        if sl["evt"] is not None:
            sl["evt"].synchronize()                  # the previous upload of this pinned table may still be in flight
        jh, bh = sl["jobs_h"], sl["blocks_h"]
        rows, blocks = [], []
        for ji, (p, outs, nblk, C, nparts) in enumerate(jobs):
            rows += [p.data_ptr()] + list(outs) + [nblk | (C << 32), nparts]
            blocks += [ji | (cb << 16) for cb in range((nparts * C + 63) // 64)]
        if len(blocks) > _MAX_BLOCKS:
            raise RuntimeError("xfmamba_amd.deferred: too many column blocks in one flush")
        jh[:len(rows)] = torch.tensor(rows, dtype=torch.int64)
        bh[:len(blocks)] = torch.tensor(blocks, dtype=torch.int32)
        sl["jobs_d"][:len(rows)].copy_(jh[:len(rows)], non_blocking=True)
        sl["blocks_d"][:len(blocks)].copy_(bh[:len(blocks)], non_blocking=True)
        sl["key"], sl["nblocks"], sl["nrows"] = key, len(blocks), len(rows)
        if not capturing:
            if sl["evt"] is None:
                sl["evt"] = torch.cuda.Event()
            sl["evt"].record()
    elif capturing:
        # a pair that an EARLIER capture recorded: its device tables are written only by that graph's memcpy nodes, i.e. only
        # if that graph is replayed first.  Record the uploads in this graph too (same pinned contents, read-only): the new
        # graph then never depends on another one having run.
        nr, nb = sl["nrows"], sl["nblocks"]
        sl["jobs_d"][:nr].copy_(sl["jobs_h"][:nr], non_blocking=True)
        sl["blocks_d"][:nb].copy_(sl["blocks_h"][:nb], non_blocking=True)
    with torch.cuda.device(dev), _lib.timed("partial_sums", 0):
        _lib.check(_lib.lib().xfm_partial_sums_multi(sl["jobs_d"].data_ptr(), sl["blocks_d"].data_ptr(), sl["nblocks"],
                                                     _lib.stream_ptr()), "partial_sums_multi")
    # (the workspaces of `jobs` stay referenced until here: the launch is queued behind their producers)
