"""Mlp of a VSS block on the token-major stream: library GEMMs + the HIP kernels ``xfm_bias_gelu_fwd/_bwd`` and
``xfm_colsum`` for everything between them (``models/fusion_vmamba.py:135-153``: fc1 -> GELU -> drop -> fc2 -> drop).

fc1's bias add and the exact GELU are one pass; the backward pass of that kernel also emits fc1's bias gradient, and
fc2's bias gradient is a column-sum kernel -- no framework reductions are left on the path.
"""
from __future__ import annotations

import os

import torch

from . import _lib
from . import deferred as _deferred
from .amp import cast_weight
from .proj import split_k_wgrad, wgrad_slot

__all__ = ["bias_gelu_fn", "colsum_fn", "linear_tokens_fn", "mlp_tokens_fn"]


def _rows(t):
    C = t.shape[-1]
    return t.numel() // C, C


def colsum_fn(x: torch.Tensor, grad_of=None) -> torch.Tensor:
    """Sum over all axes but the last of a contiguous tensor -> (C,) fp32.  ``grad_of``: the result is the gradient of that
    parameter and nothing reads it before the optimizer -- with deferred column sums (deferred.py) it is filled at the flush."""
    _lib.require_cuda(x)
    x = x.contiguous()
    rows, C = _rows(x)
    lib = _lib.lib()
    code = _lib.dtype_code(x.dtype)
    nblk = lib.xfm_colsum_blocks(rows, C, code)
    if nblk <= 0:
        raise RuntimeError(f"xfmamba_amd: colsum does not support width {C} / dtype {x.dtype}")
    out = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = torch.empty(nblk * C, dtype=torch.float32, device=x.device)
    later = grad_of is not None and _deferred.add_job(ws, [out], nblk, C, 1, params=(grad_of,))
    with torch.cuda.device(x.device), _lib.timed("colsum", x.numel() * x.element_size()):
        _lib.check(lib.xfm_colsum(x.data_ptr(), None if later else out.data_ptr(), ws.data_ptr(), rows, C, code,
                                  _lib.stream_ptr()), "colsum")
    return out


class BiasGeluHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, bias):
        _lib.require_cuda(z, bias)
        z = z.contiguous()
        rows, C = _rows(z)
        b = None if bias is None else bias.float().contiguous()
        g = torch.empty_like(z)
        with torch.cuda.device(z.device), _lib.timed("bias_gelu_fwd", 2 * z.numel() * z.element_size()):
            _lib.check(_lib.lib().xfm_bias_gelu_fwd(z.data_ptr(), _lib.ptr(b), g.data_ptr(), rows, C,
                                                    _lib.dtype_code(z.dtype), _lib.stream_ptr()), "bias_gelu_fwd")
        ctx.save_for_backward(z, b)
        ctx.bdtype = None if bias is None else bias.dtype
        ctx.bparam = bias                                   # (identity only: what deferred.add_job checks)
        return g

    @staticmethod
    def backward(ctx, dg):
        z, b = ctx.saved_tensors
        rows, C = _rows(z)
        lib = _lib.lib()
        code = _lib.dtype_code(z.dtype)
        dg = dg.contiguous() if dg.dtype == z.dtype else dg.to(z.dtype).contiguous()
        dz = torch.empty_like(z)
        db = torch.empty(C, dtype=torch.float32, device=z.device)
        nblk = lib.xfm_colsum_blocks(rows, C, code)
        ws = torch.empty(nblk * C, dtype=torch.float32, device=z.device)
        later = b is not None and _deferred.add_job(ws, [db], nblk, C, 1, params=(ctx.bparam,))
        with torch.cuda.device(z.device), _lib.timed("bias_gelu_bwd", 3 * z.numel() * z.element_size()):
            _lib.check(lib.xfm_bias_gelu_bwd(z.data_ptr(), _lib.ptr(b), dg.data_ptr(), dz.data_ptr(),
                                             None if later else db.data_ptr(), ws.data_ptr(), rows, C, code,
                                             _lib.stream_ptr()), "bias_gelu_bwd")
        return dz, (None if ctx.bdtype is None else db.to(ctx.bdtype))


def bias_gelu_fn(z, bias=None):
    """``gelu(z + bias)`` (exact erf form, ``nn.GELU()``), bias over the last axis."""
    return BiasGeluHip.apply(z, bias)


# XFM_TOKENS_GEMM=0: library GEMMs everywhere.  Read ONCE at import: set it before importing xfmamba_amd.
_SKINNY = os.environ.get("XFM_TOKENS_GEMM", "1") == "1"


def _skinny_ok(x, w, transposed=False):
    """bf16 token-major product whose (contraction, output) widths the MFMA kernel of csrc/tokens_gemm.hip covers."""
    if not _SKINNY or x.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or not x.is_cuda or not w.is_contiguous():
        return False
    con, out = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    if x.data_ptr() % 16 or w.data_ptr() % 16 or not x.is_contiguous():     # 16-byte vector loads
        return False
    return x.shape[-1] == con and x.numel() >= con * 4096 and bool(_lib.lib().xfm_tokens_gemm_supported(con, out))


def _skinny(x, w, bias, out, transposed):
    """``x @ w.T`` (w: (out, con)) or, transposed, ``x @ w`` (w: (con, out)) through ``xfm_tokens_gemm``."""
    con = x.shape[-1]
    T = x.numel() // con
    y = torch.empty(*x.shape[:-1], out, dtype=x.dtype, device=x.device)
    b = None if bias is None else bias.float().contiguous()
    nbytes = T * (con + out) * 2
    with torch.cuda.device(x.device), _lib.timed("tokens_gemm", nbytes):
        _lib.check(_lib.lib().xfm_tokens_gemm(x.data_ptr(), w.data_ptr(), _lib.ptr(b), y.data_ptr(), T, con, out,
                                              1 if transposed else 0, _lib.stream_ptr()), "tokens_gemm")
    return y


# square token GEMMs on the tiled kernel of xfm_tokens_gemm2 (epilogue 0); XFM_TILED_LINEAR=0: the library
_TILED_LINEAR = os.environ.get("XFM_TILED_LINEAR", "1") == "1"


class LinearTokens(torch.autograd.Function):
    """``F.linear`` on (..., K) tokens whose bias gradient comes from the column-sum kernel."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        cd = x.dtype
        w = cast_weight(weight, cd)
        if _skinny_ok(x, w):
            y = _skinny(x, w, bias, w.shape[0], False)
        elif _TILED_LINEAR and w.shape[0] == w.shape[1] and _gemm2_ok(x, w.shape[1], w.shape[0]) and w.is_contiguous():
            # square products of the 14 x 14 / 7 x 7 stages (384 -> 384, 768 -> 768) on the tiled LDS-direct kernel
            # (tools/gemm3probe.py: 9.8 vs 12.9 us for the library at 12544 x 384 -> 384)
            y = _gemm2(x.reshape(-1, x.shape[-1]), w, bias, w.shape[0], False, 0)[0].view(*x.shape[:-1], w.shape[0])
        else:
            y = torch.nn.functional.linear(x, w, None if bias is None else cast_weight(bias, cd))
        ctx.save_for_backward(x, w)
        ctx.meta = (weight.dtype, None if bias is None else bias.dtype)
        ctx.wparam = weight if isinstance(weight, torch.nn.Parameter) else None      # (identity only: the arena's slot key)
        ctx.bparam = bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        wdtype, bdtype = ctx.meta
        dy = dy.contiguous() if dy.dtype == w.dtype else dy.to(w.dtype).contiguous()
        dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if _skinny_ok(dy2, w, transposed=True):
                dx = _skinny(dy2, w, None, w.shape[1], True).view(x.shape)
            elif _TILED_LINEAR and w.shape[0] == w.shape[1] and _gemm2_ok(dy2, w.shape[0], w.shape[1]) and w.is_contiguous():
                dx = _gemm2(dy2, w, None, w.shape[1], True, 0)[0].view(x.shape)
            else:
                dx = torch.mm(dy2, w).view(x.shape)
        if ctx.needs_input_grad[1]:
            slot = wgrad_slot(ctx.wparam, w.shape[0], w.shape[1]) if wdtype == torch.float32 else None
            dw = split_k_wgrad(dy2, x2, deferred=wdtype == torch.float32, out=slot).to(wdtype)
        if bdtype is not None and ctx.needs_input_grad[2]:
            db = colsum_fn(dy2, grad_of=ctx.bparam).to(bdtype)
        return dx, dw, db


def linear_tokens_fn(x, weight, bias=None):
    return LinearTokens.apply(x, weight, bias)


# ---------------------------------------------------------------------------------------------------------------------------
# Mlp with the GELU inside the products (xfm_tokens_gemm2): fc1's forward epilogue emits z and gelu(z + b1), fc2's data
# gradient comes out already multiplied by gelu'(z + b1) -- no pass over the hidden activation between the GEMMs, in either
# direction.  XFM_MLP_FUSED=0 keeps the three-node chain (A/B switch, read once at import).
# ---------------------------------------------------------------------------------------------------------------------------
_FUSED = os.environ.get("XFM_MLP_FUSED", "1") == "1"
# the Mlp's deep-k / narrow-out products (fc2 forward, fc1 data gradient) on the tiled kernel too (experiment switch, read once)
_OWN_NARROW = os.environ.get("XFM_MLP_OWN_NARROW", "0") == "1"


def _gemm2_ok(x, con, out):
    # (con % 64: the kernel also takes 96 channels -- trunk stage 0 -- through a zero-filled last k-stage, but the fused Mlp
    #  loses there: 77 M GELU evaluations per launch make the erf arithmetic of the epilogue the bound, 1918 vs 1936 samples/s)
    return bool(_FUSED and con % 64 == 0 and x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] == con and x.numel() >= con * 1024
                and x.is_contiguous() and x.data_ptr() % 16 == 0 and _lib.lib().xfm_tokens_gemm2_supported(con, out))


def _gemm2(x2, w, bias, out, transposed, epi, zin=None):
    """xfm_tokens_gemm2 on (T, con) rows: epi 1 -> (z, g), epi 2 -> (dz, None), epi 0 -> (y, None)."""
    T, con = x2.shape
    # (the kernel's 16-byte vector loads: a gradient or weight VIEW at an odd offset is copied to a fresh -- 256-byte aligned --
    #  allocation instead of failing the step; never taken on the model's own tensors)
    if x2.data_ptr() % 16 or not x2.is_contiguous():
        x2 = x2.clone(memory_format=torch.contiguous_format)
    if w.data_ptr() % 16:
        w = w.clone()
    if zin is not None and zin.data_ptr() % 16:
        zin = zin.clone()
    y = torch.empty((T, out), dtype=x2.dtype, device=x2.device)
    y2 = torch.empty_like(y) if epi == 1 else None
    b = None if bias is None else bias.float().contiguous()
    nbytes = T * (con + out * (2 if epi else 1)) * 2
    with torch.cuda.device(x2.device), _lib.timed(("tokens_gemm2", "mlp_fc1_gelu", "mlp_fc2_dgrad_gelu")[epi], nbytes):
        _lib.check(_lib.lib().xfm_tokens_gemm2(x2.data_ptr(), w.data_ptr(), _lib.ptr(b), y.data_ptr(), _lib.ptr(y2), _lib.ptr(zin),
                                               T, con, out, 1 if transposed else 0, epi, _lib.stream_ptr()), "tokens_gemm2")
    return y, y2


class MlpFusedHip(torch.autograd.Function):
    """fc2(gelu(fc1(x))) on token-major ``x`` (..., C) (reference models/fusion_vmamba.py:135-153) as ONE autograd node."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        cd = x.dtype
        w1c, w2c = cast_weight(w1, cd), cast_weight(w2, cd)          # (H, C), (C, H)
        H, C = w1c.shape
        x2 = x.reshape(-1, C)
        z, g = _gemm2(x2, w1c, b1, H, False, 1)
        if _OWN_NARROW and _gemm2_ok(g, H, C):
            y, _ = _gemm2(g, w2c, b2, C, False, 0)
        else:
            y = torch.nn.functional.linear(g, w2c, None if b2 is None else cast_weight(b2, cd))
        ctx.save_for_backward(x2, z, g, w1c, w2c, b1)
        ctx.meta = (x.shape, w1.dtype, w2.dtype, None if b1 is None else b1.dtype, None if b2 is None else b2.dtype)
        ctx.params = (w1 if isinstance(w1, torch.nn.Parameter) else None, w2 if isinstance(w2, torch.nn.Parameter) else None,
                      b1, b2)                                          # (identity only: arena slots, deferred column sums)
        return y.view(*x.shape[:-1], C)

    @staticmethod
    def backward(ctx, dy):
        x2, z, g, w1c, w2c, b1 = ctx.saved_tensors
        xshape, w1dt, w2dt, b1dt, b2dt = ctx.meta
        pw1, pw2, pb1, pb2 = ctx.params
        H, C = w1c.shape
        dy2 = dy.reshape(-1, C)
        dy2 = dy2.contiguous() if dy2.dtype == w2c.dtype else dy2.to(w2c.dtype).contiguous()
        # d z = (dy W2) * gelu'(z + b1): fc2's weight (C, H) IS the (con, out) layout of the transposed product
        # (the tiled form of the kernel also leaves the column sums of dz -- fc1's bias gradient -- as one partial row per
        #  128-token tile: no second pass over the hidden-width tensor)
        dx = dw1 = db1 = dw2 = db2 = None
        want_db1 = b1dt is not None and ctx.needs_input_grad[2]
        lib = _lib.lib()
        nblk = lib.xfm_tokens_gemm2_parts_blocks(dy2.shape[0], C, H) if want_db1 and dy2.data_ptr() % 16 == 0 else 0
        if nblk > 0:
            T = dy2.shape[0]
            dz = torch.empty((T, H), dtype=dy2.dtype, device=dy2.device)
            part = torch.empty(nblk * H, dtype=torch.float32, device=dy2.device)
            bf = None if b1 is None else b1.float().contiguous()
            with torch.cuda.device(dy2.device), _lib.timed("mlp_fc2_dgrad_gelu", T * (C + 2 * H) * 2):
                _lib.check(lib.xfm_tokens_gemm2_parts(dy2.data_ptr(), w2c.data_ptr(), _lib.ptr(bf), dz.data_ptr(), z.data_ptr(),
                                                      part.data_ptr(), T, C, H, 1, _lib.stream_ptr()), "tokens_gemm2_parts")
            db1 = torch.empty(H, dtype=torch.float32, device=dy2.device)
            if not _deferred.add_job(part, [db1], nblk, H, 1, params=(pb1,)):
                db1 = part.view(nblk, H).sum(0)
            db1 = db1.to(b1dt)
            want_db1 = False
        else:
            dz, _ = _gemm2(dy2, w2c, b1, H, True, 2, zin=z)
        if ctx.needs_input_grad[0]:
            if _OWN_NARROW and _gemm2_ok(dz, H, C):
                dx = _gemm2(dz, w1c, None, C, True, 0)[0].view(xshape)
            else:
                dx = torch.mm(dz, w1c).view(xshape)
        if ctx.needs_input_grad[1]:
            slot = wgrad_slot(pw1, H, C) if w1dt == torch.float32 else None
            dw1 = split_k_wgrad(dz, x2, deferred=w1dt == torch.float32, out=slot).to(w1dt)
        if want_db1:
            db1 = colsum_fn(dz, grad_of=pb1).to(b1dt)
        if ctx.needs_input_grad[3]:
            slot = wgrad_slot(pw2, C, H) if w2dt == torch.float32 else None
            dw2 = split_k_wgrad(dy2, g, deferred=w2dt == torch.float32, out=slot).to(w2dt)
        if b2dt is not None and ctx.needs_input_grad[4]:
            db2 = colsum_fn(dy2, grad_of=pb2).to(b2dt)
        return dx, dw1, db1, dw2, db2


def mlp_tokens_fn(x, w1, b1, w2, b2, drop=None, defer_bias=False):
    """fc2(drop(gelu(fc1(x)))) on token-major ``x`` (..., C); weights are the (out, in) Linear2d / nn.Linear weights.
    ``defer_bias``: leave fc2's bias out -- the caller adds it inside the residual-add + LayerNorm kernel, whose backward
    pass then also yields its gradient (no column-sum pass over the fc2 output gradient)."""
    if drop is None and b1 is not None and _gemm2_ok(x, w1.shape[1], w1.shape[0]) and tuple(w2.shape) == (w1.shape[1], w1.shape[0]):
        return MlpFusedHip.apply(x, w1, b1, w2, None if defer_bias else b2)
    z = linear_tokens_fn(x, w1, None)
    g = bias_gelu_fn(z, b1)
    if drop is not None:
        g = drop(g)
    y = linear_tokens_fn(g, w2, None if defer_bias else b2)
    return y if drop is None else drop(y)
