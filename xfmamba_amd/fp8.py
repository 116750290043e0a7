"""BASELINE.json configs[4]: fp8 (OCP e4m3fn) weights for the SS2D ``x_proj`` / ``out_proj`` projections on the CDNA4 fp8
matrix cores, everything else (scan, norms, Mlp) unchanged in bf16.

Reference call sites: ``models/fusion_vmamba.py:1147-1150`` (x_proj, ``F.conv1d`` on the four routes) and ``:1205``
(out_proj, a 1x1 ``Linear2d``).  The reference has no fp8 path; the recipe here is this build's:
  * weight: per-tensor scale ``s_w = amax|W| / 448``, ``W_q = rne_e4m3(W / s_w)`` (quantised from the fp32 master weight
    once per optimizer step and cached);
  * activation: quantised inside the kernel while it is staged -- clamp to +-448, round to nearest even, no scale (the
    inputs are SiLU / LayerNorm outputs of order 1);
  * product on ``v_mfma_f32_32x32x16_fp8_fp8`` with fp32 accumulation, output bf16 tokens;
  * backward: straight-through -- data gradient with the de-quantised weight, weight gradient of the fp32 master, both as
    bf16 GEMMs.
``ENABLED`` switches the SS2D blocks over (``bench.py --fp8``).  Oracle: fp32 matmul of the e4m3-rounded operands.
"""
from __future__ import annotations

import weakref

import torch

from . import _lib

__all__ = ["fp8_planes_linear", "quantize_weight", "invalidate", "ENABLED", "usable"]

ENABLED = False
FP8_MAX = 448.0
_WQ = {}          # id(weight) -> (weakref, version, epoch, wq, scale tensor, dequantised bf16)
_EPOCH = 0        # bumped by every optimizer step that writes parameters through raw pointers (optim.FusedAdam)


def invalidate() -> None:
    """Drop every cached quantised weight: ``optim.FusedAdam`` updates the masters through raw pointers, which does not
    move ``Tensor._version`` -- the version check alone would keep serving the weights of the first forward pass."""
    global _EPOCH
    _EPOCH += 1


def quantize_weight(w: torch.Tensor):
    """(W_q as float8_e4m3fn, scale as python-free 0-d fp32 tensor, de-quantised bf16 copy) of a (M, K) weight."""
    wf = w.detach().float()
    scale = (wf.abs().amax().clamp_min(1e-12) / FP8_MAX).reshape(1)
    wq = (wf / scale).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return wq, scale, (wq.float() * scale).to(torch.bfloat16)


def _cached(w: torch.Tensor):
    src = w._base if w._base is not None else w
    ent = _WQ.get(id(src))
    # while a stream is being captured the weight is ALWAYS re-quantised: the graph must contain the quantisation, or every
    # replay would multiply with the weights of the capture step
    if (ent is not None and ent[0]() is src and ent[1] == src._version and ent[2] == _EPOCH and ent[3].shape == w.shape
            and not torch.cuda.is_current_stream_capturing()):
        return ent[3], ent[4], ent[5]
    wq, scale, wdq = quantize_weight(w)
    _WQ[id(src)] = (weakref.ref(src), src._version, _EPOCH, wq, scale, wdq)
    return wq, scale, wdq


def usable(x: torch.Tensor, K: int, M: int) -> bool:
    return bool(ENABLED and x.is_cuda and x.dtype == torch.bfloat16 and _lib.lib().xfm_fp8_planes_gemm_supported(K, M))


class Fp8PlanesLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        # x (B, K, L) bf16 planes, weight (M, K) master -> y (B, L, M) bf16 tokens
        _lib.require_cuda(x, weight)
        x = x.contiguous()
        B, K, L = x.shape
        M = weight.shape[0]
        wq, scale, wdq = _cached(weight)
        y = torch.empty((B, L, M), dtype=torch.bfloat16, device=x.device)
        with torch.cuda.device(x.device), _lib.timed("fp8_planes_gemm", B * L * (K + M) * 2 + M * K):
            _lib.check(_lib.lib().xfm_fp8_planes_gemm(x.data_ptr(), wq.data_ptr(), scale.data_ptr(), y.data_ptr(), B, K, L, M,
                                                      _lib.stream_ptr()), "fp8_planes_gemm")
        ctx.save_for_backward(x, wdq)
        ctx.wdtype = weight.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        from .proj import _bmm_f32, wgrad_mfma
        x, wdq = ctx.saved_tensors
        B, K, L = x.shape
        M = wdq.shape[0]
        dy = dy.contiguous() if dy.dtype == wdq.dtype else dy.to(wdq.dtype).contiguous()
        dx = torch.bmm(wdq.t().unsqueeze(0).expand(B, K, M), dy.transpose(1, 2))                 # (B, K, L)
        dw = wgrad_mfma(dy, False, x, True)                                                       # dy tokens (B, L, M), x planes
        if dw is None:
            dw = _bmm_f32(dy.transpose(1, 2), x.transpose(1, 2)).sum(0)
        dw = dw.to(ctx.wdtype)                                                                    # (M, K)
        return dx, dw


def fp8_planes_linear(x, weight):
    """``y[b, l, :] = W_q-dequantised @ q(x[b, :, l])``: x (B, K, L) bf16 planes, weight (M, K) -> (B, L, M) bf16 tokens."""
    return Fp8PlanesLinear.apply(x, weight)
