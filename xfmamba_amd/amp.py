"""Low-precision shadows of the GEMM weights for mixed-precision training.

Under autocast every weight is cast fp32 -> bf16 by its own small kernel on every step (about 150 launches for
XFMamba-T).  ``WeightCache`` keeps one bf16 copy per parameter and refreshes ALL of them with a single multi-tensor
copy after the optimizer step; the hand-written projection / Mlp nodes (``proj.py``, ``mlp_tokens.py``) pick the copy
up through ``cast_weight``.  A copy is only used while the parameter's version counter still equals the one recorded
at refresh time: after an in-place update OF THE PARAMETER that was not followed by ``refresh()`` the cast falls back to
``weight.to(dtype)``.

Caveat: writes through ``p.data`` (``p.data.copy_()``, ``dist.broadcast(p.data)``, EMA / clipping code that goes
through ``.data``) do NOT bump the version counter of ``p``, so they are invisible to that check.  Code that writes
weights that way must call ``WeightCache.refresh()`` or ``invalidate_shadows()`` afterwards; this package's own
writers do (``dp.broadcast_parameters`` mutates the parameter itself under ``no_grad`` and invalidates;
``load_state_dict`` copies into the parameter, which bumps the version).
"""
from __future__ import annotations

import weakref

import torch

__all__ = ["WeightCache", "cast_weight", "invalidate_shadows", "padded_shadow", "adopt_padded", "refresh_derived"]

_SHADOWS = {}          # id(parameter) -> (weakref(parameter), shadow tensor, version at refresh)
# Row-padded copies of shadows (the channel-lane SS2D kernels read x_proj_weight with every route's rows padded from
# R + 2N to a multiple of 8: ss2d_chan.py).  Padding the bf16 shadow inside every forward pass is a fill + a copy per block
# and step; a registered padded copy is instead rewritten for ALL blocks by one multi-tensor copy right after the optimizer
# step (refresh_derived), in place, so a captured step keeps reading the same storage.
_PADDED = {}           # id(parameter) -> [weakref(parameter), padded (K, C2p, D) tensor, parameter version it was made from]


def padded_shadow(w: torch.Tensor, rows_padded: int, dtype: torch.dtype):
    """The registered (K, rows_padded, D) copy of the 3-D parameter ``w`` (rows beyond w.shape[1] zero), or None when there
    is none or it is stale (the parameter was modified in place since)."""
    ent = _PADDED.get(id(w))
    if ent is None or ent[0]() is not w or ent[2] != w._version or ent[1].dtype != dtype or ent[1].shape[1] != rows_padded:
        return None
    return ent[1]


def adopt_padded(w: torch.Tensor, padded: torch.Tensor) -> None:
    """Register ``padded`` -- just computed from the CURRENT value of parameter ``w`` -- as its padded copy.  Only parameters
    with a registered shadow are adopted: their copy is kept current by ``refresh_derived`` after every optimizer step."""
    sh = _SHADOWS.get(id(w))
    if sh is None or sh[0]() is not w or sh[2] != w._version or padded.dtype != sh[1].dtype:
        return
    _PADDED[id(w)] = [weakref.ref(w), padded, w._version]


@torch.no_grad()
def refresh_derived() -> None:
    """Rewrite every padded copy from its parameter's shadow: ONE multi-tensor copy (per-route blocks are contiguous on both
    sides).  Called by the writers of the shadows (FusedAdam.step, WeightCache.refresh)."""
    dst, src = [], []
    for key, ent in list(_PADDED.items()):
        w = ent[0]()
        sh = _SHADOWS.get(key)
        if w is None or sh is None or sh[0]() is not w or sh[2] != w._version or sh[1].dtype != ent[1].dtype:
            del _PADDED[key]                         # no current shadow to copy from: the next forward pads again
            continue
        K, C2 = w.shape[0], w.shape[1]
        for k in range(K):
            dst.append(ent[1][k, :C2])
            src.append(sh[1][k])
        ent[2] = w._version
    if dst:
        torch._foreach_copy_(dst, src)


def cast_weight(w: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """``w.to(dtype)``, served from a registered shadow when one is current (``w`` may be a reshaped view)."""
    if w.dtype == dtype:
        return w
    src = w._base if w._base is not None else w
    ent = _SHADOWS.get(id(src))
    if ent is not None:
        ref, shadow, version = ent
        if ref() is src and shadow.dtype == dtype and src._version == version:
            if w is src:
                return shadow
            if w.is_contiguous() and w.numel() == src.numel() and src.is_contiguous():
                return shadow.view(w.shape)
    return w.to(dtype)


def invalidate_shadows(module: torch.nn.Module = None) -> None:
    """Forget the registered shadows (of ``module``'s parameters, or all): the next ``cast_weight`` re-casts from the
    fp32 master until ``WeightCache.refresh()`` registers fresh copies.  For code that wrote weights through ``.data``."""
    if module is None:
        _SHADOWS.clear()
        _PADDED.clear()
        return
    for p in module.parameters():
        _SHADOWS.pop(id(p), None)
        _PADDED.pop(id(p), None)


class WeightCache:
    def __init__(self, module: torch.nn.Module, dtype: torch.dtype = torch.bfloat16):
        self.params = [p for p in module.parameters() if p.is_floating_point() and p.dtype != dtype]
        self.shadows = [torch.empty_like(p, dtype=dtype) for p in self.params]
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        """Re-cast every parameter (one multi-tensor kernel); call right after ``optimizer.step()``."""
        if self.params:
            torch._foreach_copy_(self.shadows, self.params)
        for p, s in zip(self.params, self.shadows):
            _SHADOWS[id(p)] = (weakref.ref(p), s, p._version)
        refresh_derived()

    def mark_current(self, written=None):
        """Register the shadows as current WITHOUT copying (they were just written by ``optim.FusedAdam``'s kernel).
        ``written``: the parameters the kernel actually updated; the others (frozen, no gradient this step) keep their
        entry, so a shadow made stale by an in-place change stays subject to the version check."""
        ids = None if written is None else {id(p) for p in written}
        for p, s in zip(self.params, self.shadows):
            if ids is None or id(p) in ids:
                _SHADOWS[id(p)] = (weakref.ref(p), s, p._version)
        refresh_derived()

    def close(self):
        for p in self.params:
            ent = _SHADOWS.get(id(p))
            if ent is not None and ent[0]() is p:
                del _SHADOWS[id(p)]
            _PADDED.pop(id(p), None)
