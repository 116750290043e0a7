"""LayerNorm over the channel axis of NCHW maps on the HIP kernel ``xfm_layernorm2d_fwd/_bwd``.

Replaces the body of the reference's ``LayerNorm2d.forward`` (``models/fusion_vmamba.py:52-57``:
permute -> ``F.layer_norm`` -> permute).  The output may be emitted directly in a narrower dtype
(``out_dtype``) when the consumer is a GEMM/conv that would cast it anyway under autocast.
"""
from __future__ import annotations

import os

import torch

from . import _lib
from . import deferred as _deferred
from .proj import zeros_f32

__all__ = ["layernorm2d_fn", "LayerNorm2dHip"]

_PARTS = os.environ.get("XFM_LN2D_PARTS", "1") == "1"     # weight / bias gradient as partial rows of the dx kernel


class LayerNorm2dHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        _lib.require_cuda(x, weight, bias)
        B, C = x.shape[0], x.shape[1]
        L = x.numel() // (B * C)
        x = x.contiguous()
        w = weight.float().contiguous()
        b = None if bias is None else bias.float().contiguous()
        out_dtype = out_dtype or x.dtype
        y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        mean = torch.empty((B, L), dtype=torch.float32, device=x.device)
        rstd = torch.empty((B, L), dtype=torch.float32, device=x.device)
        nbytes = x.numel() * (x.element_size() + y.element_size())
        slab_dt = x.dtype in (torch.float32, torch.bfloat16) and out_dtype in (torch.float32, torch.bfloat16)
        nws = _lib.lib().xfm_layernorm2d_ws_floats(B, C, L) if slab_dt else 0      # (the slab form is built for fp32 / bf16)
        if nws > 0:
            # 7 x 7 maps with wide rows: the slab form (two kernels, partial statistics per 64-channel slab between them)
            ws = torch.empty(nws, dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device), _lib.timed("layernorm2d_fwd", nbytes):
                _lib.check(_lib.lib().xfm_layernorm2d_fwd_ws(x.data_ptr(), w.data_ptr(), _lib.ptr(b), y.data_ptr(), mean.data_ptr(),
                                                             rstd.data_ptr(), ws.data_ptr(), B, C, L, float(eps),
                                                             _lib.dtype_code(x.dtype), _lib.dtype_code(out_dtype),
                                                             _lib.stream_ptr()), "layernorm2d_fwd_ws")
        else:
            with torch.cuda.device(x.device), _lib.timed("layernorm2d_fwd", nbytes):
                _lib.check(_lib.lib().xfm_layernorm2d_fwd(x.data_ptr(), w.data_ptr(), _lib.ptr(b), y.data_ptr(), mean.data_ptr(),
                                                          rstd.data_ptr(), B, C, L, float(eps), _lib.dtype_code(x.dtype),
                                                          _lib.dtype_code(out_dtype), _lib.stream_ptr()), "layernorm2d_fwd")
        ctx.save_for_backward(x, w, mean, rstd)
        ctx.has_bias = bias is not None
        ctx.params = (weight, bias)                        # (identity only: what deferred.add_job checks)
        ctx.wdtype = weight.dtype
        ctx.ydtype = out_dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        B, C = x.shape[0], x.shape[1]
        L = x.numel() // (B * C)
        dy = dy.contiguous()
        if dy.dtype != ctx.ydtype:
            dy = dy.to(ctx.ydtype)
        dx = torch.empty_like(x)
        lib = _lib.lib()
        xc, yc = _lib.dtype_code(x.dtype), _lib.dtype_code(ctx.ydtype)
        slab_dt = x.dtype in (torch.float32, torch.bfloat16) and ctx.ydtype in (torch.float32, torch.bfloat16)
        nws = lib.xfm_layernorm2d_bwd_ws_floats(B, C, L) if ctx.has_bias and slab_dt and _PARTS else 0
        if nws > 0:
            # the slab form (7 x 7 maps: partial rows of the weight / bias gradient per sample) or the split form (14 x 14 maps
            # with 384 channels: per 64 positions); two kernels with a workspace between them
            nrow = lib.xfm_layernorm2d_bwd_ws_blocks(B, C, L)
            ws = torch.empty(nws, dtype=torch.float32, device=x.device)
            part = torch.empty(nrow * 2 * C, dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device), _lib.timed("layernorm2d_bwd", x.numel() * (2 * x.element_size() + dy.element_size())):
                _lib.check(lib.xfm_layernorm2d_bwd_parts_ws(x.data_ptr(), w.data_ptr(), dy.data_ptr(), mean.data_ptr(),
                                                            rstd.data_ptr(), dx.data_ptr(), part.data_ptr(), ws.data_ptr(), B, C, L,
                                                            xc, yc, _lib.stream_ptr()), "layernorm2d_bwd_parts_ws")
            dw, db = torch.empty(C, dtype=torch.float32, device=x.device), torch.empty(C, dtype=torch.float32, device=x.device)
            if not _deferred.add_job(part, [dw, db], nrow, C, 2, params=ctx.params):
                pr = part.view(nrow, 2, C).sum(0)
                dw, db = pr[0], pr[1]
            return dx, dw.to(ctx.wdtype), db.to(ctx.wdtype), None, None
        # (the partial-row kernels exist for fp32 / bf16 only: fp16 maps keep xfm_layernorm2d_bwd below)
        nblk = lib.xfm_layernorm2d_bwd_parts_blocks(B, C, L, xc, yc) if ctx.has_bias and slab_dt and _PARTS else 0
        if nblk > 0:
            # the dx kernel leaves the weight / bias gradient as one partial row pair per workgroup: no second kernel reading x and
            # dy again; folded with all the other column sums of the step (deferred.py) or summed here
            part = torch.empty(nblk * 2 * C, dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device), _lib.timed("layernorm2d_bwd", x.numel() * (2 * x.element_size() + dy.element_size())):
                _lib.check(lib.xfm_layernorm2d_bwd_parts(x.data_ptr(), w.data_ptr(), dy.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                         dx.data_ptr(), part.data_ptr(), B, C, L, xc, yc, _lib.stream_ptr()),
                           "layernorm2d_bwd_parts")
            dw, db = torch.empty(C, dtype=torch.float32, device=x.device), torch.empty(C, dtype=torch.float32, device=x.device)
            if not _deferred.add_job(part, [dw, db], nblk, C, 2, params=ctx.params):
                pr = part.view(nblk, 2, C).sum(0)
                dw, db = pr[0], pr[1]
            return dx, dw.to(ctx.wdtype), db.to(ctx.wdtype), None, None
        acc = zeros_f32(2 * w.numel() if ctx.has_bias else w.numel(), w.device)
        dw = acc[:w.numel()]                                             # one fill for both accumulators
        db = acc[w.numel():] if ctx.has_bias else None
        nbytes = x.numel() * (2 * x.element_size() + dy.element_size())
        with torch.cuda.device(x.device), _lib.timed("layernorm2d_bwd", nbytes):
            _lib.check(_lib.lib().xfm_layernorm2d_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                      dx.data_ptr(), dw.data_ptr(), _lib.ptr(db), B, C, L,
                                                      _lib.dtype_code(x.dtype), _lib.dtype_code(ctx.ydtype), _lib.stream_ptr()),
                       "layernorm2d_bwd")
        return dx, dw.to(ctx.wdtype), (None if db is None else db.to(ctx.wdtype)), None, None


def layernorm2d_fn(x, weight, bias, eps=1e-5, out_dtype=None):
    """x (B, C, H, W) -> LayerNorm over C; statistics in fp32; output dtype ``out_dtype`` or x's."""
    return LayerNorm2dHip.apply(x, weight, bias, eps, out_dtype)
