"""``cross_scan_fn`` / ``cross_merge_fn`` and the two-view swap -- drop-ins for the reference.

Mirrors ``models/csm_triton.py:501-517`` (dispatchers), ``:182-273`` (CrossScanF/CrossMergeF
autograd pairing: the backward of a scan is a merge and vice versa) and
``models/fusion_vmamba.py:189-241`` (SwappingScan_multiview / SwappingMerge_multiview) of
XZheng0427/XFMamba.  Only the configuration the hot path exercises is built
(``scans=0``, channel-first in and out, ``one_by_one=False``);
anything else raises ``NotImplementedError`` instead of falling back.
"""
from __future__ import annotations

import torch

from . import _lib

__all__ = ["cross_scan_fn", "cross_merge_fn", "CrossScanHip", "CrossMergeHip",
           "SwappingScan_multiview", "SwappingMerge_multiview", "SwappingScanStacked"]


def _scan(x: torch.Tensor) -> torch.Tensor:
    _lib.require_cuda(x)
    B, C, H, W = x.shape
    x = x.contiguous()
    y = torch.empty((B, 4, C, H * W), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device), _lib.timed("cross_scan", 5 * x.numel() * x.element_size()):
        _lib.check(_lib.lib().xfm_cross_scan(x.data_ptr(), y.data_ptr(), B, C, H, W, _lib.dtype_code(x.dtype),
                                             _lib.stream_ptr()), "cross_scan")
    return y


def _merge(ys: torch.Tensor, H: int, W: int, out_dtype=None) -> torch.Tensor:
    _lib.require_cuda(ys)
    B, K, C = ys.shape[0], ys.shape[1], ys.shape[2]
    assert K == 4
    ys = ys.contiguous()
    out_dtype = out_dtype or ys.dtype
    x = torch.empty((B, C, H * W), dtype=out_dtype, device=ys.device)
    with torch.cuda.device(ys.device), _lib.timed("cross_merge", ys.numel() * ys.element_size() + x.numel() * x.element_size()):
        _lib.check(_lib.lib().xfm_cross_merge(ys.data_ptr(), x.data_ptr(), B, C, H, W, _lib.dtype_code(ys.dtype),
                                              _lib.dtype_code(out_dtype), _lib.stream_ptr()), "cross_merge")
    return x


class CrossScanHip(torch.autograd.Function):
    """(B, C, H, W) -> (B, 4, C, H*W); backward = cross merge (csm_triton.py:207-225)."""

    @staticmethod
    def forward(ctx, x: torch.Tensor):
        ctx.hw = x.shape[2:]
        return _scan(x)

    @staticmethod
    def backward(ctx, ys: torch.Tensor):
        H, W = ctx.hw
        B, K, C, L = ys.shape
        return _merge(ys, H, W).view(B, C, H, W)


class CrossMergeHip(torch.autograd.Function):
    """(B, 4, C, H, W) -> (B, C, H*W); backward = cross scan (csm_triton.py:248-273)."""

    @staticmethod
    def forward(ctx, ys: torch.Tensor):
        B, K, C, H, W = ys.shape
        ctx.shape = (B, C, H, W)
        return _merge(ys, H, W)

    @staticmethod
    def backward(ctx, x: torch.Tensor):
        B, C, H, W = ctx.shape
        return _scan(x.reshape(B, C, H, W)).view(B, 4, C, H, W)


def _only_hot_path(in_channel_first, out_channel_first, one_by_one, scans, force_torch):
    if not (in_channel_first and out_channel_first) or one_by_one or scans != 0:
        raise NotImplementedError(
            "xfmamba_amd builds only scans=0, channel-first, one_by_one=False (the XFMamba hot path; "
            "reference call sites models/fusion_vmamba.py:483,517,548,1145,1174)")
    if force_torch:
        raise NotImplementedError("force_torch: the torch implementation is test infrastructure (oracle/), not shipped")


def cross_scan_fn(x: torch.Tensor, in_channel_first=True, out_channel_first=True, one_by_one=False, scans=0,
                  force_torch=False):
    """Same signature as the reference (models/csm_triton.py:501-507)."""
    _only_hot_path(in_channel_first, out_channel_first, one_by_one, scans, force_torch)
    return CrossScanHip.apply(x)


def cross_merge_fn(y: torch.Tensor, in_channel_first=True, out_channel_first=True, one_by_one=False, scans=0,
                   force_torch=False):
    """Same signature as the reference (models/csm_triton.py:511-517)."""
    _only_hot_path(in_channel_first, out_channel_first, one_by_one, scans, force_torch)
    return CrossMergeHip.apply(y)


class SwappingScan_multiview(torch.autograd.Function):
    """Even channels exchanged between the two views -> (B, 2, C, H*W).

    The backward is the reference's un-swapped pass-through (fusion_vmamba.py:217-221), kept
    bit-for-meaning: it is NOT the adjoint of the forward."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, x2: torch.Tensor):
        _lib.require_cuda(x, x2)
        B, C, H, W = x.shape
        ctx.shape = (B, C, H, W)
        x, x2 = x.contiguous(), x2.contiguous()
        out = torch.empty((B, 2, C, H * W), dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().xfm_swap_scan(x.data_ptr(), x2.data_ptr(), out.data_ptr(), B, C, H * W,
                                                _lib.dtype_code(x.dtype), _lib.stream_ptr()), "swap_scan")
        return out

    @staticmethod
    def backward(ctx, ys: torch.Tensor):
        B, C, H, W = ctx.shape
        return ys[:, 0].reshape(B, -1, H, W), ys[:, 1].reshape(B, -1, H, W)


class SwappingScanStacked(torch.autograd.Function):
    """``SwappingScan_multiview`` on the two views stacked along the batch axis, ``x`` = [view 1 | view 2] (2B, C, H, W) ->
    (B, 2, C, H*W).  Same forward kernel and the same pass-through backward (fusion_vmamba.py:217-221: route k's gradient
    goes to view k un-swapped), returned as ONE (2B, C, H, W) tensor instead of two slice gradients."""

    @staticmethod
    def forward(ctx, x: torch.Tensor):
        _lib.require_cuda(x)
        B2, C, H, W = x.shape
        B = B2 // 2
        ctx.shape = (B, C, H, W)
        x = x.contiguous()
        out = torch.empty((B, 2, C, H * W), dtype=x.dtype, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().xfm_swap_scan(x.data_ptr(), x[B:].data_ptr(), out.data_ptr(), B, C, H * W,
                                                _lib.dtype_code(x.dtype), _lib.stream_ptr()), "swap_scan")
        return out

    @staticmethod
    def backward(ctx, ys: torch.Tensor):
        B, C, H, W = ctx.shape
        return ys.transpose(0, 1).reshape(2 * B, C, H, W)


class SwappingMerge_multiview(torch.autograd.Function):
    """(B, 2, C, L) -> two (B, C, L) tensors, no un-swap (fusion_vmamba.py:224-241)."""

    @staticmethod
    def forward(ctx, ys: torch.Tensor):
        return ys[:, 0].contiguous(), ys[:, 1].contiguous()

    @staticmethod
    def backward(ctx, x: torch.Tensor, x2: torch.Tensor):
        return torch.stack([x, x2], dim=1)
