"""ctypes binding + autograd wrapper for ``oracle/scan_oracle.c``.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see the header of scan_oracle.c).
``selective_scan_c`` has the call signature of the reference's ``selective_scan_fn``
(models/csms6s.py:112-123) minus ``backend`` and runs on CPU tensors only.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libxfm_scan_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.xfm_oracle_scan_fwd.restype = ctypes.c_int
        _LIB.xfm_oracle_scan_bwd.restype = ctypes.c_int
    return _LIB


def _p(t):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _f(t):
    return None if t is None else t.detach().float().contiguous()


def scan_fwd_c(u, delta, A, B, C, D, delta_bias, delta_softplus):
    u, delta, A, B, C, D, delta_bias = map(_f, (u, delta, A, B, C, D, delta_bias))
    Bt, K, N, L = B.shape
    KD = u.shape[1]
    out = torch.empty_like(u)
    rc = lib().xfm_oracle_scan_fwd(_p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(delta_bias),
                                   int(bool(delta_softplus)), _p(out), Bt, KD, K, N, L)
    assert rc == 0, rc
    return out


def scan_bwd_c(u, delta, A, B, C, D, delta_bias, dout, delta_softplus):
    u, delta, A, B, C, D, delta_bias, dout = map(_f, (u, delta, A, B, C, D, delta_bias, dout))
    Bt, K, N, L = B.shape
    KD = u.shape[1]
    du, dd = torch.empty_like(u), torch.empty_like(u)
    dA, dB, dC = torch.empty_like(A), torch.empty_like(B), torch.empty_like(C)
    dD = torch.empty_like(D) if D is not None else None
    db = torch.empty_like(delta_bias) if delta_bias is not None else None
    rc = lib().xfm_oracle_scan_bwd(_p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(delta_bias), _p(dout),
                                   int(bool(delta_softplus)), _p(du), _p(dd), _p(dA), _p(dB), _p(dC), _p(dD),
                                   _p(db), Bt, KD, K, N, L)
    assert rc == 0, rc
    return du, dd, dA, dB, dC, dD, db


class _ScanC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, B, C, D, delta_bias, delta_softplus, oflex):
        ctx.sp = delta_softplus
        ctx.save_for_backward(u, delta, A, B, C, D, delta_bias)
        out = scan_fwd_c(u, delta, A, B, C, D, delta_bias, delta_softplus)
        return out if oflex else out.to(u.dtype)

    @staticmethod
    def backward(ctx, dout):
        u, delta, A, B, C, D, delta_bias = ctx.saved_tensors
        du, dd, dA, dB, dC, dD, db = scan_bwd_c(u, delta, A, B, C, D, delta_bias, dout, ctx.sp)
        return du.to(u.dtype), dd.to(delta.dtype), dA, dB.to(B.dtype), dC.to(C.dtype), dD, db, None, None


def selective_scan_c(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=True, oflex=True):
    return _ScanC.apply(u, delta, A, B, C, D, delta_bias, delta_softplus, oflex)
