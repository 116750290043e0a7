"""Fused multi-tensor Adam for the XFMamba training step (``xfm_adam_multi``, ``csrc/adam.hip``).

The reference trains with ``torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-5)`` (``1_train_model.py:141``)
and steps it once per batch (``libs/training.py:195``).  ``FusedAdam`` applies exactly that update to ALL parameters in one
launch and, when given the ``amp.WeightCache`` of the model, writes the bf16 weight shadows in the same pass (no separate
multi-tensor cast after the step).  The step count lives on the device, so the whole update is graph-capturable.
Parameters without a gradient (the reference's never-trained ``outnorm0-2`` / ``in_proj``) are skipped, as torch does.
ONE step count serves all parameters (torch keeps one per parameter: the same thing unless a parameter receives
gradients only on some steps, which this model never does).
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

__all__ = ["FusedAdam"]

_CHUNK = 16384


class FusedAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, weight_cache=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params or any(p.dtype != torch.float32 or not p.is_cuda for p in self.params):
            raise RuntimeError("FusedAdam: fp32 parameters on an MI355X device expected")
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        dev = self.params[0].device
        self.exp_avg = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        self.step_count = torch.zeros(1, dtype=torch.float32, device=dev)
        self.shadow_of = {}
        if weight_cache is not None:
            self.shadow_of = {id(p): s for p, s in zip(weight_cache.params, weight_cache.shadows) if s.dtype == torch.bfloat16}
        self._cache = weight_cache
        # pointer-table slots (pinned host + device), allocated HERE: nothing may be allocated on the host side while a
        # hipGraph capture is running
        self._nchunks_max = sum((p.numel() + _CHUNK - 1) // _CHUNK for p in self.params)
        self._slots = {}
        for name in ("eager", "capture"):
            tot = 6 * len(self.params) + self._nchunks_max
            self._slots[name] = dict(host=torch.empty(tot, dtype=torch.int64).pin_memory(),
                                     dev=torch.empty(tot, dtype=torch.int64, device=dev), key=None)

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None

    def _slot(self, capturing):
        """Pointer tables live in a pinned host buffer + a device buffer.  Filling them is a host write and ONE async
        copy, which a hipGraph capture records as a memcpy node (replays re-read the pinned buffer, whose addresses are
        the captured allocations).  A captured step has its own slot so later eager steps cannot overwrite it."""
        return self._slots["capture" if capturing else "eager"]

    @torch.no_grad()
    def step(self, grads=None, grad_scale: float = 1.0):
        """``grads``: {parameter: gradient tensor} to use instead of ``.grad`` -- fp32 or bf16, contiguous, e.g. the views
        of ``dp.PhasedGrads``' wire buckets holding the all-reduced SUM over the ranks, with ``grad_scale = 1 / world``:
        the kernel reads the bucket in place (parameters missing from the dict are skipped, like ``.grad is None``)."""
        from .proj import join_wgrad_stream
        join_wgrad_stream()                          # weight gradients launched on the side stream (proj.wgrad_stream)
        if grads is not None:
            active = [(p, grads[p], m, v) for p, m, v in zip(self.params, self.exp_avg, self.exp_avg_sq) if p in grads]
        else:
            # (gradients of the 3x3 convolutions may arrive channels_last: bring those few to the parameter's layout)
            active = [(p, p.grad if p.grad.is_contiguous() else p.grad.contiguous(), m, v)
                      for p, m, v in zip(self.params, self.exp_avg, self.exp_avg_sq) if p.grad is not None]
        if not active:
            return
        n = len(active)
        capturing = torch.cuda.is_current_stream_capturing()
        sl = self._slot(capturing)
        key = tuple((p.data_ptr(), g.data_ptr(), g.dtype) for p, g, _, _ in active)
        if key != sl["key"]:                        # rebuilt only when a tensor moved (eager steps allocate fresh gradients)
            h = sl["host"]
            # the previous upload of this pinned table may still be in flight (the CPU can run a step ahead of the GPU):
            # wait for it before the table is rewritten, or that step's kernel reads the NEXT step's pointers
            if sl.get("evt") is not None and not capturing:
                sl["evt"].synchronize()
            ct, co = [], []
            for ti, (p, g, m, v) in enumerate(active):
                if not p.is_contiguous() or not g.is_contiguous() or g.dtype not in (torch.float32, torch.bfloat16) \
                        or g.numel() != p.numel():
                    raise RuntimeError("FusedAdam: contiguous fp32 parameters and contiguous fp32 / bf16 gradients expected")
                sh = self.shadow_of.get(id(p))
                h[ti], h[n + ti], h[2 * n + ti], h[3 * n + ti] = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
                h[4 * n + ti] = sh.data_ptr() if sh is not None else 0
                h[5 * n + ti] = p.numel() | ((1 << 62) if g.dtype == torch.bfloat16 else 0)      # bit 62: bf16 gradient
                for c in range((p.numel() + _CHUNK - 1) // _CHUNK):
                    ct.append(ti)
                    co.append(c)
            # chunk tables as int32 pairs packed into the int64 tail
            tail = torch.tensor([a | (b << 32) for a, b in zip(ct, co)], dtype=torch.int64)
            h[6 * n:6 * n + len(ct)] = tail
            sl["nchunks"] = len(ct)
            sl["key"] = key
            sl["n"] = n
            sl["dev"][:6 * n + len(ct)].copy_(h[:6 * n + len(ct)], non_blocking=True)
            if not capturing:
                if sl.get("evt") is None:
                    sl["evt"] = torch.cuda.Event()
                sl["evt"].record()
        d, nn = sl["dev"], sl["n"]
        base = d.data_ptr()
        with torch.cuda.device(active[0][0].device):
            _lib.check(_lib.lib().xfm_adam_multi_scaled(base, base + 8 * nn, base + 16 * nn, base + 24 * nn, base + 32 * nn,
                                                        base + 40 * nn, base + 48 * nn, sl["nchunks"], _CHUNK,
                                                        self.step_count.data_ptr(), self.lr, self.betas[0], self.betas[1],
                                                        self.eps, self.weight_decay, float(grad_scale), _lib.stream_ptr()),
                       "adam_multi")
        # (the kernel wrote the parameters AND their bf16 shadows through raw pointers: the parameters' version counters
        #  did not move, so the shadows registered in amp.WeightCache stay the ones cast_weight serves -- and they are current)
        if self._cache is not None:
            self._cache.mark_current([p for p, _, _, _ in active])
        from . import fp8 as _fp8
        _fp8.invalidate()                            # quantised copies of the masters are stale now

    def state_dict(self):
        """torch.optim.Adam-compatible layout (state per parameter index)."""
        state = {i: dict(step=self.step_count.clone(), exp_avg=m, exp_avg_sq=v)
                 for i, (m, v) in enumerate(zip(self.exp_avg, self.exp_avg_sq))}
        group = dict(lr=self.lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay, params=list(range(len(self.params))))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        for i, st in sd["state"].items():
            self.exp_avg[int(i)].copy_(st["exp_avg"])
            self.exp_avg_sq[int(i)].copy_(st["exp_avg_sq"])
            self.step_count.copy_(torch.as_tensor(st["step"], dtype=torch.float32).reshape(1))
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.weight_decay = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
