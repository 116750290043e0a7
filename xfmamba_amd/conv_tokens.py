"""3 x 3, stride-2, padding-1 convolution on token-major maps -- the second convolution of the patch embedding and the three
downsample layers of the trunk (reference ``models/fusion_vmamba.py:1504-1518``, ``:1531-1538``: ``nn.Conv2d(dim, out_dim, 3,
2, 1)``) -- through ``xfm_conv3x3s2_tokens_fwd/_bwd_data/_bwd_weight`` (csrc/conv_tok.hip): the 3 x 3 neighbourhoods as rows,
then the library's own MFMA GEMM kernels.  No convolution library, no solver search.  The parameter stays where the reference
keeps it (``weight`` (O, C, 3, 3)); the bias is NOT applied here (the caller folds it into the LayerNorm that follows).
"""
from __future__ import annotations

import os

import torch

from . import _lib
from .amp import cast_weight
from .proj import wgrad_slot, zeros_f32

__all__ = ["conv3x3s2_tokens_fn", "conv3x3s2_tokens_supported", "conv3x3s2_wgrad_from_map", "Conv3x3S2TokensHip",
           "conv3x3s2_gray_fn", "conv3x3s2_gray_supported", "Conv3x3S2GrayHip"]

# XFM_CONV_OWN=0: the strided convolutions stay on the convolution library (A/B switch, read once).
# XFM_CONV_OWN_MIN_C: the fewest input channels the own path takes.  Measured per pass at batch 64 (tools/convprobe.py, us, own /
# library): 112 x 112 x 48 -> 96: 104 / 48 forward, 124 / 88 data gradient, 38 / 66 weight gradient; 56 x 56 x 96 -> 192: 62 / 36,
# 60 / 60, 33 / 65; 28 x 28 x 192 -> 384: 48 / 43, 44 / 54, 29 / 60; 14 x 14 x 384 -> 768: 58 / 53, 40 / 68, 34 / 59.  The
# neighbourhood rows cost 9/4 of the input in HBM traffic each way: at the two large maps that is more than the library's implicit
# GEMM spends in total, so those two layers stay on it.
_OWN = os.environ.get("XFM_CONV_OWN", "1") == "1"
_OWN_MIN_C = int(os.environ.get("XFM_CONV_OWN_MIN_C", "192"))
# XFM_CONV_WGRAD_X=0: the layers that stay on the convolution library also take their weight gradient from it
_WGRAD_X = os.environ.get("XFM_CONV_WGRAD_X", "1") == "1"
# XFM_CONV_GRAY=0: the first convolution of the patch embedding stays on the convolution library even when its input channels are
# replicas of one channel
_GRAY = os.environ.get("XFM_CONV_GRAY", "1") == "1"


def conv3x3s2_tokens_supported(t: torch.Tensor, conv: torch.nn.Conv2d) -> bool:
    """Token-major t (B, H, W, C) bf16 (or fp32 under bf16 autocast) and a convolution ``xfm_conv3x3s2_tokens_*`` covers."""
    if not (_OWN and t.is_cuda and t.dim() == 4 and conv.kernel_size == (3, 3) and conv.stride == (2, 2)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros"):
        return False
    cd = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else conv.weight.dtype
    if cd != torch.bfloat16:
        return False
    B, H, W, C = t.shape
    return C >= _OWN_MIN_C and bool(_lib.lib().xfm_conv3x3s2_tokens_supported(C, conv.out_channels, H, W))


class Conv3x3S2TokensHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, weight):
        _lib.require_cuda(t, weight)
        B, H, W, C = t.shape
        O = weight.shape[0]
        if weight.shape != (O, C, 3, 3):
            raise RuntimeError("conv3x3s2_tokens: weight must be (O, C, 3, 3)")
        ctx.t_dtype, ctx.w_dtype = t.dtype, weight.dtype
        x = t.contiguous() if t.dtype == torch.bfloat16 else t.to(torch.bfloat16).contiguous()
        # (O, 3, 3, C): the channels_last memory of the parameter's bf16 shadow
        w = cast_weight(weight, torch.bfloat16).permute(0, 2, 3, 1).contiguous()
        OH, OW = H // 2, W // 2
        T = B * OH * OW
        col = torch.empty(T, 9 * C, dtype=torch.bfloat16, device=t.device)
        y = torch.empty(B, OH, OW, O, dtype=torch.bfloat16, device=t.device)
        nbytes = x.numel() * 2 + 2 * col.numel() * 2 + y.numel() * 2
        with torch.cuda.device(t.device), _lib.timed("conv3x3s2_fwd", nbytes):
            _lib.check(_lib.lib().xfm_conv3x3s2_tokens_fwd(x.data_ptr(), w.data_ptr(), col.data_ptr(), y.data_ptr(), B, H, W, C, O,
                                                           _lib.stream_ptr()), "conv3x3s2_tokens_fwd")
        ctx.shape = (B, H, W, C, O)
        ctx.weight = weight
        ctx.save_for_backward(col, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        col, w = ctx.saved_tensors
        B, H, W, C, O = ctx.shape
        dy = dy.contiguous() if dy.dtype == torch.bfloat16 else dy.to(torch.bfloat16).contiguous()
        lib = _lib.lib()
        dx = dw = None
        with torch.cuda.device(dy.device):
            if ctx.needs_input_grad[1]:
                slot = wgrad_slot(ctx.weight, O, 9 * C)
                acc = slot if slot is not None else zeros_f32(O * 9 * C, dy.device).view(O, 9 * C)
                with _lib.timed("conv3x3s2_wgrad", (dy.numel() + col.numel()) * 2):
                    _lib.check(lib.xfm_conv3x3s2_tokens_bwd_weight(dy.data_ptr(), col.data_ptr(), acc.data_ptr(), B, H, W, C, O,
                                                                   _lib.stream_ptr()), "conv3x3s2_tokens_bwd_weight")
                # (O, 3, 3, C) sums seen as the parameter's (O, C, 3, 3): a channels_last gradient (optim.py brings it to the
                # parameter's layout, as it did for the convolution library's)
                dw = acc.view(O, 3, 3, C).permute(0, 3, 1, 2)
                if dw.dtype != ctx.w_dtype:
                    dw = dw.to(ctx.w_dtype)
            if ctx.needs_input_grad[0]:
                dcol = torch.empty_like(col)
                dx = torch.empty(B, H, W, C, dtype=torch.bfloat16, device=dy.device)
                with _lib.timed("conv3x3s2_dgrad", (dy.numel() + 2 * dcol.numel() + dx.numel()) * 2):
                    _lib.check(lib.xfm_conv3x3s2_tokens_bwd_data(dy.data_ptr(), w.data_ptr(), dcol.data_ptr(), dx.data_ptr(), B, H, W,
                                                                 C, O, _lib.stream_ptr()), "conv3x3s2_tokens_bwd_data")
                if dx.dtype != ctx.t_dtype:
                    dx = dx.to(ctx.t_dtype)
        return dx, dw


class Conv3x3S2GrayHip(torch.autograd.Function):
    """The first convolution of the patch embedding on ONE replicated channel: x1 (B, H, W) bf16, weight (O, CI, 3, 3) ->
    (B, H/2, W/2, O) bf16 = conv2d(x1 expanded to CI channels, weight, stride 2, padding 1), no bias."""

    @staticmethod
    def forward(ctx, x1, weight):
        _lib.require_cuda(x1, weight)
        B, H, W = x1.shape
        O, CI = weight.shape[0], weight.shape[1]
        w = cast_weight(weight, torch.bfloat16).contiguous()
        y = torch.empty(B, H // 2, W // 2, O, dtype=torch.bfloat16, device=x1.device)
        with torch.cuda.device(x1.device), _lib.timed("conv3x3s2_gray_fwd", x1.numel() * 2 + y.numel() * 2):
            _lib.check(_lib.lib().xfm_conv3x3s2_gray_fwd(x1.data_ptr(), w.data_ptr(), y.data_ptr(), B, H, W, CI, O,
                                                         _lib.stream_ptr()), "conv3x3s2_gray_fwd")
        ctx.w_dtype, ctx.w_shape = weight.dtype, tuple(weight.shape)
        ctx.save_for_backward(x1)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x1,) = ctx.saved_tensors
        B, H, W = x1.shape
        O, CI = ctx.w_shape[0], ctx.w_shape[1]
        dw = None
        if ctx.needs_input_grad[1]:
            dy = dy.contiguous() if dy.dtype == torch.bfloat16 else dy.to(torch.bfloat16).contiguous()
            nws = _lib.lib().xfm_conv3x3s2_gray_ws_floats(O)
            acc = zeros_f32(O * 9 + nws, dy.device)                    # one fill: the sums and the kernel's replicas of them
            dw9, ws = acc[:O * 9], acc[O * 9:]
            with torch.cuda.device(dy.device), _lib.timed("conv3x3s2_gray_wgrad", x1.numel() * 2 + dy.numel() * 2):
                _lib.check(_lib.lib().xfm_conv3x3s2_gray_bwd_weight(dy.data_ptr(), x1.data_ptr(), dw9.data_ptr(), ws.data_ptr(), B, H,
                                                                    W, O, _lib.stream_ptr()), "conv3x3s2_gray_bwd_weight")
            # every input-channel slice of the parameter saw the same image: the same gradient
            dw = dw9.view(O, 1, 3, 3).expand(O, CI, 3, 3)
            if dw.dtype != ctx.w_dtype:
                dw = dw.to(ctx.w_dtype)
        return None, dw


def conv3x3s2_gray_supported(x: torch.Tensor, conv: torch.nn.Conv2d) -> bool:
    """x (B, CI, H, W) is a stride-0 broadcast of one channel, needs no gradient, and ``conv`` is a 3 x 3 stride-2 padding-1
    convolution ``xfm_conv3x3s2_gray_*`` covers (bf16 compute)."""
    if not (_OWN and _GRAY and x.is_cuda and x.dim() == 4 and x.shape[1] > 1 and x.stride(1) == 0 and not x.requires_grad
            and conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.padding == (1, 1) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros" and conv.in_channels == x.shape[1]):
        return False
    cd = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else conv.weight.dtype
    if cd != torch.bfloat16:
        return False
    return bool(_lib.lib().xfm_conv3x3s2_gray_supported(conv.out_channels, x.shape[2], x.shape[3]))


def conv3x3s2_gray_fn(x1: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    return Conv3x3S2GrayHip.apply(x1, weight)


def conv3x3s2_wgrad_from_map(dy: torch.Tensor, x: torch.Tensor, weight: torch.Tensor):
    """Weight gradient of the convolution from its token-major input map: dy (B, H/2, W/2, O), x (B, H, W, C) bf16 ->
    the (O, C, 3, 3) gradient (a channels_last view of fp32 (O, 3, 3, C) sums) through ``xfm_conv3x3s2_tokens_bwd_weight_x``,
    or None when the kernel does not cover the call (the caller then asks the convolution library)."""
    if not (_OWN and _WGRAD_X and dy.is_cuda and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.dim() == 4
            and x.dim() == 4 and dy.is_contiguous() and x.is_contiguous()):
        return None
    B, H, W, C = x.shape
    O = dy.shape[-1]
    if dy.shape != (B, H // 2, W // 2, O) or weight.shape != (O, C, 3, 3):
        return None
    lib = _lib.lib()
    if not lib.xfm_conv3x3s2_tokens_bwd_weight_x_supported(B, H, W, C, O):
        return None
    slot = wgrad_slot(weight, O, 9 * C)
    acc = slot if slot is not None else zeros_f32(O * 9 * C, dy.device).view(O, 9 * C)
    with torch.cuda.device(dy.device), _lib.timed("conv3x3s2_wgrad", (dy.numel() + x.numel() * 9 // 4) * 2):
        _lib.check(lib.xfm_conv3x3s2_tokens_bwd_weight_x(dy.data_ptr(), x.data_ptr(), acc.data_ptr(), B, H, W, C, O,
                                                         _lib.stream_ptr()), "conv3x3s2_tokens_bwd_weight_x")
    return acc.view(O, 3, 3, C).permute(0, 3, 1, 2)


def conv3x3s2_tokens_fn(t: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """t (B, H, W, C), weight (O, C, 3, 3) -> conv2d(t, weight, stride 2, padding 1) as (B, H/2, W/2, O) bf16 (no bias)."""
    return Conv3x3S2TokensHip.apply(t, weight)
