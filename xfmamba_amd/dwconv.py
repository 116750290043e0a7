"""Depthwise 3x3 convolution fused with SiLU -- the ``conv2d`` -> ``act`` pair of every SS2D block.

Replaces ``self.act(self.conv2d(x))`` (reference ``models/fusion_vmamba.py:1198-1201``, ``:594-601``,
``:853-857``; ``nn.Conv2d(D, D, 3, padding=1, groups=D)`` + ``nn.SiLU``) with ``xfm_dwconv3x3_fwd/_bwd``.
The parameters stay where the reference keeps them (``conv2d.weight`` (D,1,3,3), ``conv2d.bias``).
"""
from __future__ import annotations

import torch

from . import _lib
from .proj import zeros_f32

__all__ = ["dwconv3x3_silu_fn", "DWConv3x3SiLUHip", "dwconv3x3_silu_tokens_fn", "dwconv_tokens_supported"]


class DWConv3x3SiLUHip(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias, silu):
        _lib.require_cuda(x, weight, bias)
        B, D, H, W = x.shape
        if weight.shape != (D, 1, 3, 3):
            raise RuntimeError("dwconv3x3: weight must be (D, 1, 3, 3)")
        x = x.contiguous()
        w = weight.float().contiguous()
        b = None if bias is None else bias.float().contiguous()
        y = torch.empty_like(x)
        nbytes = 2 * x.numel() * x.element_size()
        with torch.cuda.device(x.device), _lib.timed("dwconv3x3_fwd", nbytes):
            _lib.check(_lib.lib().xfm_dwconv3x3_fwd(x.data_ptr(), w.data_ptr(), _lib.ptr(b), y.data_ptr(), B, D, H, W,
                                                    _lib.dtype_code(x.dtype), int(silu), _lib.stream_ptr()), "dwconv3x3_fwd")
        ctx.silu = int(silu)
        ctx.wdtype = weight.dtype
        ctx.save_for_backward(x, w, b)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        B, D, H, W = x.shape
        dy = dy.contiguous().to(x.dtype)
        dx = torch.empty_like(x)
        acc = zeros_f32(w.numel() + (b.numel() if b is not None else 0), w.device)
        dw = acc[:w.numel()].view(w.shape)                              # one fill for both accumulators
        db = acc[w.numel():] if b is not None else None
        nbytes = 3 * x.numel() * x.element_size()
        with torch.cuda.device(x.device), _lib.timed("dwconv3x3_bwd", nbytes):
            _lib.check(_lib.lib().xfm_dwconv3x3_bwd(x.data_ptr(), w.data_ptr(), _lib.ptr(b), dy.data_ptr(), dx.data_ptr(),
                                                    dw.data_ptr(), _lib.ptr(db), B, D, H, W, _lib.dtype_code(x.dtype),
                                                    ctx.silu, _lib.stream_ptr()), "dwconv3x3_bwd")
        return dx, dw.to(ctx.wdtype), (None if db is None else db.to(ctx.wdtype)), None


def dwconv3x3_silu_fn(x, weight, bias=None, silu=True):
    """x (B,D,H,W), weight (D,1,3,3), bias (D,)|None -> silu(conv(x) + bias) in x's dtype."""
    return DWConv3x3SiLUHip.apply(x, weight, bias, silu)


def dwconv_tokens_supported(x: torch.Tensor) -> bool:
    """Token-major x (B, H, W, C) bf16 that ``xfm_dwconv3x3_tokens_fwd/_bwd`` covers (14 x 14 / 7 x 7 maps, C % 8 == 0)."""
    return bool(x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16
                and _lib.lib().xfm_dwconv3x3_tokens_supported(x.shape[1], x.shape[2], x.shape[3]))


class DWConv3x3SiLUTokensHip(torch.autograd.Function):
    """``silu(conv2d(x) + bias)`` (depthwise 3 x 3, padding 1; reference models/fusion_vmamba.py:1198-1201) on TOKEN-MAJOR
    maps x (B, H, W, C) bf16 -- the short-map stages keep the SS2D block in the token layout (csrc/dwconv_tok.hip)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias):
        _lib.require_cuda(x, weight)
        B, H, W, C = x.shape
        if weight.shape != (C, 1, 3, 3):
            raise RuntimeError("dwconv3x3 (tokens): weight must be (C, 1, 3, 3)")
        x = x.contiguous()
        w = weight.float().contiguous()
        b = None if bias is None else bias.float().contiguous()
        y = torch.empty_like(x)
        with torch.cuda.device(x.device), _lib.timed("dwconv3x3_fwd", 2 * x.numel() * 2):
            _lib.check(_lib.lib().xfm_dwconv3x3_tokens_fwd(x.data_ptr(), w.data_ptr(), _lib.ptr(b), y.data_ptr(), B, H, W, C,
                                                           _lib.stream_ptr()), "dwconv3x3_tokens_fwd")
        ctx.wdtype = weight.dtype
        ctx.save_for_backward(x, w, b)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        B, H, W, C = x.shape
        dy = dy.contiguous() if dy.dtype == x.dtype else dy.to(x.dtype).contiguous()
        dx, dz = torch.empty_like(x), torch.empty_like(x)
        part = torch.empty(B * H * 10 * C, dtype=torch.float32, device=x.device)
        acc = zeros_f32(C * 9 + (C if b is not None else 0), x.device)       # one fill for both results (the fold adds)
        dw = acc[:C * 9]
        db = acc[C * 9:] if b is not None else None
        with torch.cuda.device(x.device), _lib.timed("dwconv3x3_bwd", 5 * x.numel() * 2):
            _lib.check(_lib.lib().xfm_dwconv3x3_tokens_bwd(x.data_ptr(), w.data_ptr(), _lib.ptr(b), dy.data_ptr(), dz.data_ptr(),
                                                           dx.data_ptr(), part.data_ptr(), dw.data_ptr(), _lib.ptr(db), B, H, W, C,
                                                           _lib.stream_ptr()), "dwconv3x3_tokens_bwd")
        return dx, dw.view(C, 1, 3, 3).to(ctx.wdtype), (None if db is None else db.to(ctx.wdtype))


def dwconv3x3_silu_tokens_fn(x, weight, bias=None):
    """x (B, H, W, C) bf16 token-major, weight (C, 1, 3, 3), bias (C,) | None -> silu(conv(x) + bias), same layout."""
    return DWConv3x3SiLUTokensHip.apply(x, weight, bias)
