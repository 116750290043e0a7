"""Fused SS2D core: cross-scan + 4-route selective scan + cross-merge in one gfx950 kernel.

Replaces the operator chain of ``SS2Dv2.forward_corev2`` (reference
``models/fusion_vmamba.py:1145-1174``: ``cross_scan_fn`` -> ``selective_scan_fn`` ->
``cross_merge_fn``) with ``xfm_ss2d_fwd`` / ``xfm_ss2d_bwd`` (``include/xfm_hip.h``).  All operands
are in the feature map's natural row-major order; route k (0 row-major, 1 column-major, 2/3
their reversals, ``models/csm_triton.py:25-29``) is an index walk inside the kernel, so neither
the (B,4,D,L) scan inputs nor the (B,4,D,L) fp32 scan outputs ever reach HBM.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib

__all__ = ["ss2d_core_fn", "SS2DCoreHip"]


def _fill(p, x, dts, A, Bs, Cs, D, bias, H, W, out_dtype):
    Bt, Dm, L = x.shape
    p.batch, p.d_inner, p.H, p.W, p.dstate = Bt, Dm, H, W, A.shape[1]
    p.delta_softplus = 1
    p.in_dtype, p.out_dtype = _lib.dtype_code(x.dtype), _lib.dtype_code(out_dtype)
    p.x, p.dts, p.Bs, p.Cs = x.data_ptr(), dts.data_ptr(), Bs.data_ptr(), Cs.data_ptr()
    p.A, p.D, p.delta_bias = A.data_ptr(), D.data_ptr(), bias.data_ptr()


class SS2DCoreHip(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, dts, A, Bs, Cs, D, bias, H, W):
        _lib.require_cuda(x, dts, A, Bs, Cs, D, bias)
        Bt, Dm, L = x.shape
        if L != H * W or dts.shape != (Bt, 4, Dm, L) or Bs.shape != Cs.shape or Bs.shape[:2] != (Bt, 4):
            raise RuntimeError("ss2d_core: x (B,D,H*W), dts (B,4,D,H*W), Bs/Cs (B,4,N,H*W) expected")
        if not (x.dtype == dts.dtype == Bs.dtype == Cs.dtype):
            raise RuntimeError("ss2d_core: x, dts, Bs, Cs must share one dtype")
        x, dts, Bs, Cs = x.contiguous(), dts.contiguous(), Bs.contiguous(), Cs.contiguous()
        A, D, bias = A.float().contiguous(), D.float().contiguous(), bias.float().contiguous()
        y = torch.empty((Bt, Dm, L), dtype=torch.float32, device=x.device)   # oflex: fp32 out
        p = _lib.SS2DParams()
        _fill(p, x, dts, A, Bs, Cs, D, bias, H, W, torch.float32)
        p.y = y.data_ptr()
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().xfm_ss2d_fwd(ctypes.byref(p), _lib.stream_ptr()), "ss2d_fwd")
        ctx.hw = (H, W)
        ctx.save_for_backward(x, dts, A, Bs, Cs, D, bias)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, dts, A, Bs, Cs, D, bias = ctx.saved_tensors
        H, W = ctx.hw
        dev = x.device
        dy = dy.contiguous().float()
        dx = torch.empty_like(x)
        ddts = torch.empty_like(dts)
        dBs = torch.zeros(Bs.shape, dtype=torch.float32, device=dev)
        dCs = torch.zeros(Cs.shape, dtype=torch.float32, device=dev)
        dA, dD, dbias = torch.zeros_like(A), torch.zeros_like(D), torch.zeros_like(bias)
        p = _lib.SS2DParams()
        _fill(p, x, dts, A, Bs, Cs, D, bias, H, W, torch.float32)
        p.dy, p.dx, p.ddts = dy.data_ptr(), dx.data_ptr(), ddts.data_ptr()
        p.dBs, p.dCs, p.dA, p.dD, p.ddelta_bias = (dBs.data_ptr(), dCs.data_ptr(), dA.data_ptr(), dD.data_ptr(),
                                                   dbias.data_ptr())
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().xfm_ss2d_bwd(ctypes.byref(p), _lib.stream_ptr()), "ss2d_bwd")
        return dx, ddts, dA, dBs.to(Bs.dtype), dCs.to(Cs.dtype), dD, dbias, None, None


def ss2d_core_fn(x, dts, A, Bs, Cs, D, bias, H, W):
    """x (B,D,L), dts (B,4,D,L), A (4D,N), Bs/Cs (B,4,N,L), D/bias (4D,) -> y (B,D,L) fp32."""
    return SS2DCoreHip.apply(x, dts, A, Bs, Cs, D, bias, H, W)
