"""Residual add + DropPath + LayerNorm on a token-major (B, H, W, C) stream: ``xfm_add_layernorm_rows_fwd/_bwd``.

``add_layernorm_rows_fn(x, y, scale, weight, bias)`` returns ``(x + scale[b] * y, LayerNorm_C(x + scale[b] * y))`` --
the ``x = x + self.drop_path(branch(x))`` of ``VSSBlock._forward`` (``models/fusion_vmamba.py:1325-1337``) together
with the norm that reads the sum next -- in one pass over HBM; ``layernorm_rows_fn`` is the plain norm.  The residual
stream is fp32; ``y`` and the normalised output are in ``out_dtype`` (the consumer GEMM's dtype under autocast).
"""
from __future__ import annotations

import torch

from . import _lib
from . import deferred as _deferred

__all__ = ["residual_settle_fn", "add_layernorm_rows_fn", "layernorm_rows_fn", "layernorm_rows_gelu_fn", "rows_supported"]


def rows_supported(C: int) -> bool:
    return bool(_lib.lib().xfm_add_layernorm_rows_supported(int(C)))


def _fwd(x, y, scale, w, b, eps, out_dtype, pre=None):
    B, C = x.shape[0], x.shape[-1]
    rps = x.numel() // (B * C)
    h = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    x_new = torch.empty_like(x) if y is not None else None
    mean = torch.empty(B * rps, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    nbytes = x.numel() * (x.element_size() + h.element_size() + (0 if y is None else 4 + y.element_size()))
    with torch.cuda.device(x.device), _lib.timed("add_layernorm_rows_fwd", nbytes):
        _lib.check(_lib.lib().xfm_add_layernorm_rows_fwd(
            x.data_ptr(), _lib.ptr(y), _lib.ptr(scale), _lib.ptr(pre), w.data_ptr(), _lib.ptr(b), _lib.ptr(x_new), h.data_ptr(),
            mean.data_ptr(), rstd.data_ptr(), B, rps, C, float(eps), _lib.dtype_code(x.dtype), _lib.dtype_code(out_dtype),
            _lib.stream_ptr()),
            "add_layernorm_rows_fwd")
    return x_new, h, mean, rstd


def _bwd(x_new, w, dh, dres, mean, rstd, scale, want_dy, has_bias, dtype, pre=None, params=()):
    """``params``: the (weight, bias, pre_bias) PARAMETERS the three column sums are gradients of -- deferral (deferred.py)
    needs fp32 parameters without a ``.grad`` so that the callers' ``.to(dtype)`` is the identity and autograd adopts the
    result tensors; anything else takes the finish kernel."""
    B, C = x_new.shape[0], x_new.shape[-1]
    rps = x_new.numel() // (B * C)
    lib = _lib.lib()
    dx = torch.empty_like(x_new)
    dy = torch.empty(x_new.shape, dtype=dtype, device=x_new.device) if want_dy else None
    dw = torch.empty_like(w)
    db = torch.empty_like(w) if has_bias else None
    dpre = torch.empty_like(w) if pre is not None else None
    nblk = lib.xfm_add_layernorm_rows_bwd_blocks(B * rps, C)
    ws = torch.empty(3 * C * nblk, dtype=torch.float32, device=x_new.device)
    nbytes = x_new.numel() * (2 * x_new.element_size() + dh.element_size() + (0 if dres is None else 4)
                              + (dy.element_size() if want_dy else 0))
    # with deferred column sums the kernel leaves its partial rows in ws and ONE launch per step folds them (deferred.py)
    nparts = 3 if pre is not None else 2
    later = _deferred.add_job(ws, [dw, db, dpre], nblk, C, nparts, params=params)
    with torch.cuda.device(x_new.device), _lib.timed("add_layernorm_rows_bwd", nbytes):
        _lib.check(lib.xfm_add_layernorm_rows_bwd(
            x_new.data_ptr(), _lib.ptr(pre), w.data_ptr(), dh.data_ptr(), _lib.ptr(dres), mean.data_ptr(), rstd.data_ptr(),
            _lib.ptr(scale), dx.data_ptr(), _lib.ptr(dy), None if later else dw.data_ptr(), _lib.ptr(db), _lib.ptr(dpre),
            ws.data_ptr(), B, rps, C, _lib.dtype_code(x_new.dtype), _lib.dtype_code(dtype), _lib.stream_ptr()),
            "add_layernorm_rows_bwd")
    return dx, dy, dw, db, dpre


def _prep(x, weight, bias, out_dtype, need_f32=False):
    _lib.require_cuda(x, weight, bias)
    if x.dtype != torch.float32 and (need_f32 or x.dtype != torch.bfloat16):
        raise RuntimeError(f"xfmamba_amd: token-major LayerNorm input must be fp32 (bf16 without a residual add), not {x.dtype}")
    out_dtype = out_dtype or x.dtype
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"xfmamba_amd: add_layernorm_rows emits fp32 or bf16, not {out_dtype}")
    w = weight.float().contiguous()
    b = None if bias is None else bias.float().contiguous()
    return x.contiguous(), w, b, out_dtype


class LayerNormRowsHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype, pre_bias):
        x, w, b, out_dtype = _prep(x, weight, bias, out_dtype)
        pre = None if pre_bias is None else pre_bias.float().contiguous()
        _, h, mean, rstd = _fwd(x, None, None, w, b, eps, out_dtype, pre)
        ctx.save_for_backward(x, w, mean, rstd, pre)
        ctx.meta = (bias is not None, weight.dtype, out_dtype, None if pre_bias is None else pre_bias.dtype)
        ctx.params = (weight, bias, pre_bias)              # (identity only: what deferred.add_job checks)
        return h

    @staticmethod
    def backward(ctx, dh):
        x, w, mean, rstd, pre = ctx.saved_tensors
        has_bias, wdtype, dtype, pdtype = ctx.meta
        dh = dh.contiguous() if dh.dtype == dtype else dh.to(dtype).contiguous()
        dx, _, dw, db, dpre = _bwd(x, w, dh, None, mean, rstd, None, False, has_bias, dtype, pre, ctx.params)
        return (dx, dw.to(wdtype), (None if db is None else db.to(wdtype)), None, None,
                (None if dpre is None else dpre.to(pdtype)))


class LayerNormRowsGeluHip(torch.autograd.Function):
    """``gelu(LayerNorm(x + pre_bias))`` (exact erf GELU) in one pass each way: ``xfm_layernorm_rows_gelu_fwd/_bwd`` -- the
    norm -> GELU pair of the patch embedding (reference ``models/fusion_vmamba.py:1504-1518``).  Nothing but x, mean, rstd is
    kept: the backward kernel recomputes the pre-activation."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype, pre_bias):
        x, w, b, out_dtype = _prep(x, weight, bias, out_dtype)
        pre = None if pre_bias is None else pre_bias.float().contiguous()
        C = x.shape[-1]
        rows = x.numel() // C
        h = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        with torch.cuda.device(x.device), _lib.timed("layernorm_rows_gelu_fwd", x.numel() * (x.element_size() + h.element_size())):
            _lib.check(_lib.lib().xfm_layernorm_rows_gelu_fwd(
                x.data_ptr(), _lib.ptr(pre), w.data_ptr(), _lib.ptr(b), h.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, C,
                float(eps), _lib.dtype_code(x.dtype), _lib.dtype_code(out_dtype), _lib.stream_ptr()), "layernorm_rows_gelu_fwd")
        ctx.save_for_backward(x, w, b, mean, rstd, pre)
        ctx.meta = (weight.dtype, out_dtype, None if pre_bias is None else pre_bias.dtype)
        ctx.params = (weight, bias, pre_bias)
        return h

    @staticmethod
    def backward(ctx, dh):
        x, w, b, mean, rstd, pre = ctx.saved_tensors
        wdtype, dtype, pdtype = ctx.meta
        dh = dh.contiguous() if dh.dtype == dtype else dh.to(dtype).contiguous()
        C = x.shape[-1]
        rows = x.numel() // C
        lib = _lib.lib()
        dx = torch.empty_like(x)
        dw = torch.empty_like(w)
        db = torch.empty_like(w) if b is not None else None
        dpre = torch.empty_like(w) if pre is not None else None
        nblk = lib.xfm_add_layernorm_rows_bwd_blocks(rows, C)
        ws = torch.empty(3 * C * nblk, dtype=torch.float32, device=x.device)
        later = _deferred.add_job(ws, [dw, db, dpre], nblk, C, 3 if pre is not None else 2, params=ctx.params)
        with torch.cuda.device(x.device), _lib.timed("layernorm_rows_gelu_bwd", x.numel() * (2 * x.element_size() + dh.element_size())):
            _lib.check(lib.xfm_layernorm_rows_gelu_bwd(
                x.data_ptr(), _lib.ptr(pre), w.data_ptr(), _lib.ptr(b), dh.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                None if later else dw.data_ptr(), _lib.ptr(db), _lib.ptr(dpre), ws.data_ptr(), rows, C, _lib.dtype_code(x.dtype),
                _lib.dtype_code(dtype), _lib.stream_ptr()), "layernorm_rows_gelu_bwd")
        return (dx, dw.to(wdtype), (None if db is None else db.to(wdtype)), None, None, (None if dpre is None else dpre.to(pdtype)))


class AddLayerNormRowsHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, scale, weight, bias, eps, out_dtype, y_bias):
        x, w, b, out_dtype = _prep(x, weight, bias, out_dtype, need_f32=True)
        _lib.require_cuda(y)
        ctx.ydtype = y.dtype
        y = y.contiguous() if y.dtype == out_dtype else y.to(out_dtype).contiguous()
        if y.shape != x.shape:
            raise RuntimeError(f"xfmamba_amd: residual {tuple(x.shape)} vs branch {tuple(y.shape)}")
        s = None if scale is None else scale.float().contiguous()
        yb = None if y_bias is None else y_bias.float().contiguous()
        x_new, h, mean, rstd = _fwd(x, y, s, w, b, eps, out_dtype, yb)
        ctx.save_for_backward(x_new, w, mean, rstd, s, yb)
        ctx.meta = (bias is not None, weight.dtype, out_dtype, None if y_bias is None else y_bias.dtype)
        ctx.params = (weight, bias, y_bias)
        return x_new, h

    @staticmethod
    def backward(ctx, dres, dh):
        x_new, w, mean, rstd, s, yb = ctx.saved_tensors
        has_bias, wdtype, dtype, ybdtype = ctx.meta
        if dh is None:                       # the norm output was not used: only the sum carries a gradient
            dx = dres
            dy = dres if s is None else dres * s.view(-1, *([1] * (dres.ndim - 1)))
            dyb = None if yb is None else dy.reshape(-1, dy.shape[-1]).sum(0).to(ybdtype)
            return dx, dy.to(ctx.ydtype), None, None, None, None, None, dyb
        dh = dh.contiguous() if dh.dtype == dtype else dh.to(dtype).contiguous()
        if dres is not None:
            dres = dres.float().contiguous()
        dx, dy, dw, db, dyb = _bwd(x_new, w, dh, dres, mean, rstd, s, True, has_bias, dtype, yb, ctx.params)
        if dy.dtype != ctx.ydtype:
            dy = dy.to(ctx.ydtype)
        return (dx, dy, None, dw.to(wdtype), (None if db is None else db.to(wdtype)), None, None,
                (None if dyb is None else dyb.to(ybdtype)))


class LayerNormRowsPassHip(torch.autograd.Function):
    """``(x, LayerNorm(x))`` with ``x`` handed through as an OUTPUT of the node: the residual stream of the first block of a
    stage feeds both this LayerNorm and the next add + LayerNorm kernel, and as two consumers of one tensor autograd summed
    their two gradients with a stream-sized add; as outputs of one node both gradients arrive here and the backward kernel
    adds the pass-through one itself (its ``dres`` operand)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        x, w, b, out_dtype = _prep(x, weight, bias, out_dtype, need_f32=True)
        _, h, mean, rstd = _fwd(x, None, None, w, b, eps, out_dtype, None)
        ctx.save_for_backward(x, w, mean, rstd)
        ctx.meta = (bias is not None, weight.dtype, out_dtype)
        ctx.params = (weight, bias)
        return x.view_as(x), h

    @staticmethod
    def backward(ctx, dres, dh):
        x, w, mean, rstd = ctx.saved_tensors
        has_bias, wdtype, dtype = ctx.meta
        if dh is None:
            return dres, None, None, None, None
        dh = dh.contiguous() if dh.dtype == dtype else dh.to(dtype).contiguous()
        if dres is not None:
            dres = dres.float().contiguous()
        dx, _, dw, db, _ = _bwd(x, w, dh, dres, mean, rstd, None, False, has_bias, dtype, None, ctx.params)
        return dx, dw.to(wdtype), (None if db is None else db.to(wdtype)), None, None


def layernorm_rows_pass_fn(x, weight, bias, eps=1e-5, out_dtype=None):
    """``(x, LayerNorm(x))`` for the fp32 token-major stream, see LayerNormRowsPassHip."""
    return LayerNormRowsPassHip.apply(x, weight, bias, eps, out_dtype)


def layernorm_rows_fn(x, weight, bias, eps=1e-5, out_dtype=None, pre_bias=None):
    """LayerNorm over the last axis of a contiguous fp32 / bf16 (B, ..., C) tensor; output in ``out_dtype``.
    ``pre_bias`` (C,) is added to ``x`` first (the bias of the convolution that produced ``x``)."""
    return LayerNormRowsHip.apply(x, weight, bias, eps, out_dtype, pre_bias)


def layernorm_rows_gelu_fn(x, weight, bias, eps=1e-5, out_dtype=None, pre_bias=None):
    """``gelu(LayerNorm(x + pre_bias))`` over the last axis (exact erf GELU), one kernel each way."""
    return LayerNormRowsGeluHip.apply(x, weight, bias, eps, out_dtype, pre_bias)


def add_layernorm_rows_fn(x, y, scale, weight, bias, eps=1e-5, out_dtype=None, y_bias=None):
    """``x_new = x + scale[b] * (y + y_bias)`` (scale (B,) or None; ``y_bias`` (C,) = the deferred bias of the linear
    layer that produced ``y``, or None) and ``LayerNorm(x_new)``: returns ``(x_new, h)``."""
    return AddLayerNormRowsHip.apply(x, y, scale, weight, bias, eps, out_dtype, y_bias)


# ---------------------------------------------------------------------------------------------------------------------------
# end-of-stage residual settle: x + scale[b] * (y + y_bias) in the consumer's dtype (xfm_residual_settle_fwd/_bwd)
# ---------------------------------------------------------------------------------------------------------------------------
class ResidualSettleHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, scale, y_bias, out_dtype):
        _lib.require_cuda(x, y, scale, y_bias)
        if x.dtype != torch.float32 or y.dtype not in (torch.float32, torch.bfloat16) or x.shape != y.shape:
            raise RuntimeError("residual_settle: x fp32 and y (bf16 / fp32) of the same (B, ..., C) shape expected")
        x, y = x.contiguous(), y.contiguous()
        B, C = x.shape[0], x.shape[-1]
        rps = x.numel() // (B * C)
        sc = None if scale is None else scale.float().contiguous()
        yb = None if y_bias is None else y_bias.float().contiguous()
        out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        nbytes = x.numel() * (4 + y.element_size() + out.element_size())
        with torch.cuda.device(x.device), _lib.timed("residual_settle_fwd", nbytes):
            _lib.check(_lib.lib().xfm_residual_settle_fwd(x.data_ptr(), y.data_ptr(), _lib.ptr(sc), _lib.ptr(yb), out.data_ptr(),
                                                          B, rps, C, _lib.dtype_code(y.dtype), _lib.dtype_code(out_dtype),
                                                          _lib.stream_ptr()), "residual_settle_fwd")
        ctx.save_for_backward(sc)
        ctx.meta = (B, rps, C, y.dtype, out_dtype, None if y_bias is None else y_bias.dtype)
        ctx.bparam = y_bias
        return out

    @staticmethod
    def backward(ctx, dout):
        (sc,) = ctx.saved_tensors
        B, rps, C, ydt, odt, bdt = ctx.meta
        dout = dout.contiguous() if dout.dtype == odt else dout.to(odt).contiguous()
        dx = torch.empty(dout.shape, dtype=torch.float32, device=dout.device)
        dy = torch.empty(dout.shape, dtype=ydt, device=dout.device)
        nbytes = dout.numel() * (dout.element_size() + 4 + dy.element_size())
        with torch.cuda.device(dout.device), _lib.timed("residual_settle_bwd", nbytes):
            _lib.check(_lib.lib().xfm_residual_settle_bwd(dout.data_ptr(), _lib.ptr(sc), dx.data_ptr(), dy.data_ptr(), B, rps, C,
                                                          _lib.dtype_code(ydt), _lib.dtype_code(odt), _lib.stream_ptr()),
                       "residual_settle_bwd")
        db = None
        if bdt is not None and ctx.needs_input_grad[3]:
            from .mlp_tokens import colsum_fn
            db = colsum_fn(dy, grad_of=ctx.bparam).to(bdt)
        return dx, dy, None, db, None


def residual_settle_fn(x, y, scale=None, y_bias=None, out_dtype=None):
    """``x + scale[b] * (y + y_bias)`` for the token-major fp32 stream ``x`` (B, ..., C) and a branch output ``y`` of the same
    shape, emitted in ``out_dtype`` (default fp32): one pass instead of bias add, scale, add (and the consumer's cast)."""
    return ResidualSettleHip.apply(x, y, scale, y_bias, out_dtype or torch.float32)
