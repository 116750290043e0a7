"""``selective_scan_fn`` -- drop-in for the reference operator of the same name.

Mirrors ``models/csms6s.py:71-126`` of XZheng0427/XFMamba (``SelectiveScanCuda`` +
``selective_scan_fn``): same signature, same argument meaning, same autograd contract
(7 gradients, ``None`` for the three non-tensor arguments, dB/dC returned in B/C's dtype), but
the work is done by the hand-written gfx950 kernels of ``libxfm_hip.so``
(``xfm_selective_scan_fwd/_bwd``, see ``include/xfm_hip.h``).

Differences, on purpose:
  * one backend.  ``backend`` in {None, "oflex", "core", "mamba", "hip"} all select the HIP
    kernels; ``backend="torch"`` (the reference's CPU loop, csms6s.py:25-68) is NOT shipped --
    it lives under ``oracle/`` as test infrastructure, and asking for it here raises.
  * no silent fallback: CPU tensors or a missing extension raise ``RuntimeError``.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .proj import zeros_f32

__all__ = ["selective_scan_fn", "SelectiveScanHip"]


def _row_major_last(t: torch.Tensor) -> torch.Tensor:
    return t if t.stride(-1) == 1 else t.contiguous()


def _fill_common(p: _lib.ScanParams, u, delta, A, B, C, D, delta_bias, delta_softplus, out_dtype):
    Bt, KD, L = u.shape
    _, K, N, _ = B.shape
    p.batch, p.dim, p.seqlen, p.dstate, p.n_groups = Bt, KD, L, N, K
    p.delta_softplus = int(bool(delta_softplus))
    p.in_dtype = _lib.dtype_code(u.dtype)
    p.out_dtype = _lib.dtype_code(out_dtype)
    p.u, p.delta, p.A, p.B, p.C = u.data_ptr(), delta.data_ptr(), A.data_ptr(), B.data_ptr(), C.data_ptr()
    p.D, p.delta_bias = _lib.ptr(D), _lib.ptr(delta_bias)
    p.u_batch_stride, p.u_d_stride = u.stride(0), u.stride(1)
    p.delta_batch_stride, p.delta_d_stride = delta.stride(0), delta.stride(1)
    p.A_d_stride = A.stride(0)
    p.B_batch_stride, p.B_group_stride, p.B_dstate_stride = B.stride(0), B.stride(1), B.stride(2)
    p.C_batch_stride, p.C_group_stride, p.C_dstate_stride = C.stride(0), C.stride(1), C.stride(2)


def _check_args(u, delta, A, B, C, D, delta_bias):
    _lib.require_cuda(u, delta, A, B, C, D, delta_bias)
    if u.dim() != 3 or delta.shape != u.shape:
        raise RuntimeError("selective_scan: u and delta must both be (batch, dim, seqlen)")
    if B.dim() != 4 or C.shape != B.shape or B.shape[0] != u.shape[0] or B.shape[3] != u.shape[2]:
        raise RuntimeError("selective_scan: B and C must both be (batch, n_groups, dstate, seqlen)")
    if A.shape != (u.shape[1], B.shape[2]):
        raise RuntimeError("selective_scan: A must be (dim, dstate)")
    if u.shape[1] % B.shape[1] != 0:
        raise RuntimeError("selective_scan: dim must be a multiple of n_groups")
    if not (u.dtype == delta.dtype == B.dtype == C.dtype):
        raise RuntimeError("selective_scan: u, delta, B, C must share one dtype (fp32, fp16 or bf16)")
    for t in (D, delta_bias):
        if t is not None and t.shape != (u.shape[1],):
            raise RuntimeError("selective_scan: D and delta_bias must be (dim,)")


class SelectiveScanHip(torch.autograd.Function):
    """Counterpart of ``SelectiveScanCuda`` (models/csms6s.py:71-109)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, oflex=True, backend=None):
        _check_args(u, delta, A, B, C, D, delta_bias)
        u, delta, B, C = map(_row_major_last, (u, delta, B, C))
        A = A.float().contiguous()
        D = None if D is None else D.float().contiguous()
        delta_bias = None if delta_bias is None else delta_bias.float().contiguous()
        Bt, KD, L = u.shape
        K, N = B.shape[1], B.shape[2]
        plan = _lib.scan_plan(Bt, KD, L, N, K)
        out_dtype = torch.float32 if oflex else u.dtype
        out = torch.empty((Bt, KD, L), dtype=out_dtype, device=u.device)
        x = (torch.empty((Bt, KD, plan.n_chunks, N), dtype=torch.float32, device=u.device)
             if plan.n_chunks > 1 else None)
        p = _lib.ScanParams()
        _fill_common(p, u, delta, A, B, C, D, delta_bias, delta_softplus, out_dtype)
        p.out, p.out_batch_stride, p.out_d_stride = out.data_ptr(), out.stride(0), out.stride(1)
        p.x = _lib.ptr(x)
        isz, osz = u.element_size(), out.element_size()
        nbytes = Bt * KD * L * (2 * isz + osz) + 2 * Bt * K * N * L * isz + KD * (N + 2) * 4
        with torch.cuda.device(u.device), _lib.timed("selective_scan_fwd", nbytes):
            _lib.check(_lib.lib().xfm_selective_scan_fwd(ctypes.byref(p), _lib.stream_ptr()), "selective_scan_fwd")
        ctx.delta_softplus = bool(delta_softplus)
        ctx.out_dtype = out_dtype
        ctx.save_for_backward(u, delta, A, B, C, D, delta_bias, x)
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dout, *args):
        u, delta, A, B, C, D, delta_bias, x = ctx.saved_tensors
        if dout.stride(-1) != 1:                      # csms6s.py:94-95
            dout = dout.contiguous()
        if dout.dtype != ctx.out_dtype:
            dout = dout.to(ctx.out_dtype)
        dev = u.device
        du, ddelta = torch.empty(u.shape, dtype=u.dtype, device=dev), torch.empty(u.shape, dtype=u.dtype, device=dev)
        # fp32 accumulators (selective_scan.cpp:332-333,360), carved out of ONE zero-filled buffer: one fill kernel
        sizes = [A.numel(), B.numel(), C.numel(), D.numel() if D is not None else 0,
                 delta_bias.numel() if delta_bias is not None else 0]
        acc = zeros_f32(sum(sizes), dev)
        parts = torch.split(acc, sizes)
        dA, dB, dC = parts[0].view(A.shape), parts[1].view(B.shape), parts[2].view(C.shape)
        dD = parts[3].view(D.shape) if D is not None else None
        dbias = parts[4].view(delta_bias.shape) if delta_bias is not None else None
        p = _lib.ScanParams()
        _fill_common(p, u, delta, A, B, C, D, delta_bias, ctx.delta_softplus, ctx.out_dtype)
        p.x = _lib.ptr(x)
        p.dout, p.dout_batch_stride, p.dout_d_stride = dout.data_ptr(), dout.stride(0), dout.stride(1)
        p.du, p.ddelta = du.data_ptr(), ddelta.data_ptr()
        p.dA, p.dB, p.dC, p.dD, p.ddelta_bias = dA.data_ptr(), dB.data_ptr(), dC.data_ptr(), _lib.ptr(dD), _lib.ptr(dbias)
        Bt, KD, L = u.shape
        K, N = B.shape[1], B.shape[2]
        isz, osz = u.element_size(), dout.element_size()
        nbytes = Bt * KD * L * (4 * isz + osz) + 2 * Bt * K * N * L * (isz + 4) + 2 * KD * (N + 2) * 4
        with torch.cuda.device(dev), _lib.timed("selective_scan_bwd", nbytes):
            _lib.check(_lib.lib().xfm_selective_scan_bwd(ctypes.byref(p), _lib.stream_ptr()), "selective_scan_bwd")
        return du, ddelta, dA, dB.to(B.dtype), dC.to(C.dtype), dD, dbias, None, None, None


def selective_scan_fn(
    u: torch.Tensor,                     # (B, K * C, L)
    delta: torch.Tensor,                 # (B, K * C, L)
    A: torch.Tensor,                     # (K * C, N)
    B: torch.Tensor,                     # (B, K, N, L)
    C: torch.Tensor,                     # (B, K, N, L)
    D: torch.Tensor = None,              # (K * C)
    delta_bias: torch.Tensor = None,     # (K * C)
    delta_softplus=True,
    oflex=True,
    backend=None,
):
    """Same contract as the reference ``selective_scan_fn`` (models/csms6s.py:112-126)."""
    if backend == "torch":
        raise NotImplementedError(
            "xfmamba_amd ships only the HIP selective scan; the sequential torch restatement "
            "(reference models/csms6s.py:25-68) is test infrastructure under oracle/ and is never a fallback")
    if backend not in (None, "oflex", "core", "mamba", "hip"):
        raise ValueError(f"unknown selective-scan backend {backend!r}")
    return SelectiveScanHip.apply(u, delta, A, B, C, D, delta_bias, delta_softplus, oflex, backend)
