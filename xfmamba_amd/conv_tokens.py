"""3x3 / stride-2 / padding-1 convolutions of the patch embedding and the downsampling layers on the token-major
(channels-last) stream, as implicit GEMMs on the matrix cores (``csrc/tile_gemm.hip``).

The reference runs them as ``nn.Conv2d`` on NCHW maps (``models/fusion_vmamba.py:1362-1390``); MIOpen, handed the same
maps channels-last, spends most of its time around the convolution proper (layout transposes, casts, fp32 Winograd
kernels for the weight gradient).  Here the forward and the data gradient gather their operand rows straight from the
(B, H, W, C) map inside the GEMM's loader, and the weight gradient contracts ``dy`` with the gathered taps (one gather
kernel + the split-K token-contracting product the Mlp weights use)."""
from __future__ import annotations

import os

import torch

from . import _lib
from .amp import cast_weight
from .proj import split_k_wgrad

__all__ = ["conv3x3s2_tokens_ok", "conv3x3s2_tokens_fn", "conv3x3s2_tokens_enabled"]

# Off by default: measured on the XFMamba-T step (hipGraph, batch 32) the implicit-GEMM path runs 1500 samples/s against
# 1547 with MIOpen's tuned NHWC igemm kernels (the tile kernel reaches ~12 % of the bf16 MFMA peak on these shapes, the
# library ~20-30 %).  XFM_CONV_TOKENS=1 selects it (read once at import).
ENABLED = os.environ.get("XFM_CONV_TOKENS", "0") == "1"


def conv3x3s2_tokens_enabled() -> bool:
    return ENABLED


def conv3x3s2_tokens_ok(conv: torch.nn.Conv2d, t: torch.Tensor) -> bool:
    """``t``: (B, H, W, C) bf16 on the GPU; the convolution 3x3, stride 2, padding 1, dense, channels multiples of 8,
    even map sides (the data gradient walks the four pixel-parity classes)."""
    return bool(ENABLED and t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 4
                and conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.padding == (1, 1)
                and conv.dilation == (1, 1) and conv.groups == 1 and conv.in_channels % 8 == 0
                and conv.out_channels % 8 == 0 and t.shape[1] % 2 == 0 and t.shape[2] % 2 == 0
                and t.shape[3] == conv.in_channels)


class Conv3x3S2Tokens(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, weight):
        t = t.contiguous()
        B, H, W, C = t.shape
        N = weight.shape[0]
        w = cast_weight(weight, t.dtype)                               # (N, C, 3, 3) bf16 (shadow when cached)
        w9 = w.permute(0, 2, 3, 1).contiguous()                        # (N, 3, 3, C): tap-major rows
        y = torch.empty((B, H // 2, W // 2, N), dtype=t.dtype, device=t.device)
        with torch.cuda.device(t.device), _lib.timed("conv3x3s2_fwd", (t.numel() + y.numel()) * 2):
            _lib.check(_lib.lib().xfm_conv3x3s2_fwd(t.data_ptr(), w9.data_ptr(), None, y.data_ptr(), B, H, W, C, N,
                                                    _lib.stream_ptr()), "conv3x3s2_fwd")
        ctx.save_for_backward(t, w)
        ctx.wdtype = weight.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        t, w = ctx.saved_tensors
        B, H, W, C = t.shape
        N = w.shape[0]
        dy = dy.contiguous() if dy.dtype == t.dtype else dy.to(t.dtype).contiguous()
        lib = _lib.lib()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wt = w.permute(2, 3, 1, 0).contiguous()                    # (3, 3, C, N)
            dx = torch.empty_like(t)
            with torch.cuda.device(t.device), _lib.timed("conv3x3s2_dgrad", (t.numel() + dy.numel()) * 2):
                _lib.check(lib.xfm_conv3x3s2_dgrad(dy.data_ptr(), wt.data_ptr(), dx.data_ptr(), B, H, W, C, N,
                                                   _lib.stream_ptr()), "conv3x3s2_dgrad")
        if ctx.needs_input_grad[1]:
            T = B * (H // 2) * (W // 2)
            col = torch.empty((T, 9 * C), dtype=t.dtype, device=t.device)
            with torch.cuda.device(t.device), _lib.timed("im2col3x3s2", (t.numel() + col.numel()) * 2):
                _lib.check(lib.xfm_im2col3x3s2(t.data_ptr(), col.data_ptr(), B, H, W, C, _lib.stream_ptr()), "im2col3x3s2")
            dw9 = split_k_wgrad(dy.view(T, N), col)                    # (N, 9 C) fp32, rows (kh, kw, c)
            dw = dw9.view(N, 3, 3, C).permute(0, 3, 1, 2).to(ctx.wdtype)
        return dx, dw


def conv3x3s2_tokens_fn(t: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """(B, H, W, C) -> (B, H/2, W/2, N), no bias (the callers add it inside the LayerNorm that follows)."""
    return Conv3x3S2Tokens.apply(t, weight)
