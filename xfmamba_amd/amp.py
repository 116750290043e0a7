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

__all__ = ["WeightCache", "cast_weight", "invalidate_shadows"]

_SHADOWS = {}          # id(parameter) -> (weakref(parameter), shadow tensor, version at refresh)


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
        return
    for p in module.parameters():
        _SHADOWS.pop(id(p), None)


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

    def mark_current(self, written=None):
        """Register the shadows as current WITHOUT copying (they were just written by ``optim.FusedAdam``'s kernel).
        ``written``: the parameters the kernel actually updated; the others (frozen, no gradient this step) keep their
        entry, so a shadow made stale by an in-place change stays subject to the version check."""
        ids = None if written is None else {id(p) for p in written}
        for p, s in zip(self.params, self.shadows):
            if ids is None or id(p) in ids:
                _SHADOWS[id(p)] = (weakref.ref(p), s, p._version)

    def close(self):
        for p in self.params:
            ent = _SHADOWS.get(id(p))
            if ent is not None and ent[0]() is p:
                del _SHADOWS[id(p)]
