"""Channel-lane fused SS2D core for short maps (``xfm_ss2dc_fwd/_bwd``, ``csrc/ss2d_chan.hip``).

One autograd node for ``x_proj -> split -> dt_proj -> softplus -> 4-route selective scan -> cross-merge`` of
``SS2Dv2.forward_corev2`` (reference ``models/fusion_vmamba.py:1145-1174``) on maps of at most 14 x 14: the x_proj of all
routes is one dense GEMM on the natural map whose result is kept TOKEN-MAJOR ``(B, L, 4*C2p)`` (padded so every operand
block starts on a 16-byte boundary); dt_proj runs on MFMA inside the scan kernel, which therefore never reads or writes a
``(B, 4, D, L)`` step-size tensor.  The backward kernel returns ``dx``, the gradient of the raw step size (bf16, token-major,
consumed by two small dense products for ``d x_dbl`` and the dt_proj weight gradient) and the B / C column gradients.
"""
from __future__ import annotations

import ctypes
import os

import torch

from . import _lib
from . import fp8 as _fp8
from . import amp as _amp
from .amp import cast_weight
from .proj import zeros_f32, _transpose_raw, _transpose_short_ok

__all__ = ["ytokens_supported", "ss2d_chan_fn", "ss2d_chan_swap_fn", "swap_views_stacked", "chan_supported", "SS2DChanHip", "SS2DChanSwapHip"]

ENABLED = os.environ.get("XFM_SS2D_CHAN", "1") == "1"       # read once at import (A/B switch of the benches)


def chan_supported(x: torch.Tensor, H: int, W: int, N: int, n_routes: int, D: int, R: int) -> bool:
    return bool(ENABLED and x.is_cuda and x.dtype == torch.bfloat16
                and _lib.lib().xfm_ss2dc_supported(H, W, N, n_routes, D, R))


def _col_layout(R: int, N: int):
    """Columns of one route in the padded x_proj row -> (C2p, index of every original column)."""
    Rp8 = (R + 7) // 8 * 8
    NBo = 1 if N == 1 else N
    C2p = Rp8 + (8 if N == 1 else 2 * N)
    idx = list(range(R)) + [Rp8 + n for n in range(N)] + [Rp8 + NBo + n for n in range(N)]
    return Rp8, NBo, C2p, idx


# x_proj (and its accumulating data gradient) on the tiled layout-changing kernel where it covers the shape; 0: the library
_XPROJ_TILED = os.environ.get("XFM_XPROJ_TILED", "1") == "1"

_IDX_CACHE = {}


def _row_index(K: int, R: int, N: int, device):
    key = (K, R, N, str(device))
    t = _IDX_CACHE.get(key)
    if t is None:
        _, _, C2p, idx = _col_layout(R, N)
        t = torch.tensor([k * C2p + j for k in range(K) for j in idx], dtype=torch.long, device=device)
        _IDX_CACHE[key] = t
    return t


_ZEROS = {}


def _zeros(device):
    z = _ZEROS.get(str(device))
    if z is None:
        z = _ZEROS[str(device)] = torch.zeros(1024, dtype=torch.uint8, device=device)
    return z


def ytokens_supported(H: int, W: int, N: int, n_routes: int = 4) -> bool:
    """The kernels of this shape can write y / read dy TOKEN-MAJOR (B, L, D) (second-generation d_state-1 kernels)."""
    return bool(ENABLED and _YTOK and _lib.lib().xfm_ss2dc_ytokens_supported(H, W, N, n_routes))


# y / dy of the channel-lane core token-major where the kernels can (14 x 14, 7 x 7; d_state 1): out_norm is then the row
# LayerNorm and out_proj a plain token GEMM -- no LayerNorm2d, no layout-changing projection, no plane-major weight-gradient
# operand behind the scan.  XFM_SS2D_YTOK=0: planes (the A/B switch).
_YTOK = os.environ.get("XFM_SS2D_YTOK", "1") == "1"


def _params(x, xdbl, wdt, A, D, bias, H, W, N, R, n_routes, c_mod, c_off, wdiv, chk, y_tokens=False, x_tokens=False):
    p = _lib.SS2DCParams()
    p.y_tokens = 1 if y_tokens else 0
    p.x_tokens = 1 if x_tokens else 0
    p.zeros = _zeros(x.device).data_ptr()
    p.batch, p.d_inner, p.H, p.W, p.dstate, p.dt_rank, p.n_routes = x.shape[0], (x.shape[2] if x_tokens else x.shape[1]), H, W, N, R, n_routes
    p.c_mod, p.c_off, p.wdiv = c_mod, c_off, wdiv
    p.x, p.xdbl, p.wdt = x.data_ptr(), xdbl.data_ptr(), wdt.data_ptr()
    p.A, p.D, p.delta_bias = A.data_ptr(), D.data_ptr(), bias.data_ptr()
    p.chk = chk.data_ptr()
    return p


class SS2DChanHip(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, x_proj_w, dt_w, A, D, bias, H, W, c_mod=0, c_off=0, y_tokens=False, x_tokens=False):
        _lib.require_cuda(x, x_proj_w, dt_w, A, D, bias)
        if x_tokens:
            Bt, L, Dm = x.shape                  # x (B, L, D) TOKEN-MAJOR (a token-major depthwise convolution in front)
        else:
            Bt, Dm, L = x.shape
        K, C2, _ = x_proj_w.shape
        ctx.ytok, ctx.xtok = bool(y_tokens), bool(x_tokens)
        R, N = dt_w.shape[2], A.shape[1]
        if L != H * W or K != 4 or C2 != R + 2 * N or x.dtype != torch.bfloat16:
            raise RuntimeError("ss2d_chan: x (B,D,H*W) bf16, x_proj_weight (4,R+2N,D), dt_projs_weight (4,D,R) expected")
        x = x.contiguous()
        xt = None
        Rp8, NBo, C2p, _ = _col_layout(R, N)
        XC = K * C2p
        rows = _row_index(K, R, N, x.device)
        wdt = cast_weight(dt_w, x.dtype)
        wdt = (torch.nn.functional.pad(wdt, (0, Rp8 - R)) if Rp8 != R else wdt).contiguous()    # (4, D, Rp8): no copy when R % 8 == 0
        plain = Rp8 == R                         # the dt_proj block needs no inner padding: rows only grow at the END of a route

        def padded(w3):                          # (K, C2, D) -> (K * C2p, D) in the column layout of the kernels
            if plain:
                return (torch.nn.functional.pad(w3, (0, 0, 0, C2p - C2)) if C2p != C2 else w3).reshape(XC, Dm)
            out = torch.zeros((XC, Dm), dtype=w3.dtype, device=w3.device)
            return out.index_copy_(0, rows, w3.reshape(K * C2, Dm))

        if x_tokens:
            # x_proj on the token-major x: one plain token GEMM (the tiled kernel where it covers the widths)
            from .mlp_tokens import _gemm2, _gemm2_ok
            xw3 = _amp.padded_shadow(x_proj_w, C2p, x.dtype) if plain else None
            if xw3 is not None:
                xw_pad = xw3.view(XC, Dm)
            else:
                xw_pad = padded(cast_weight(x_proj_w, x.dtype)).contiguous()
                if plain and isinstance(x_proj_w, torch.nn.Parameter):
                    _amp.adopt_padded(x_proj_w, xw_pad.view(K, C2p, Dm))
            x2 = x.view(Bt * L, Dm)
            if _gemm2_ok(x2, Dm, XC) and xw_pad.is_contiguous():
                xdbl = _gemm2(x2, xw_pad, None, XC, False, 0)[0].view(Bt, L, XC)
            else:
                xdbl = torch.matmul(x, xw_pad.t())
            xt = x
        elif _fp8.usable(x, Dm, XC):
            # BASELINE configs[4]: x_proj with fp8 weights on the fp8 matrix cores (straight-through backward with the
            # de-quantised weight)
            wq, scale, xw_pad = _fp8.quantize_weight(padded(x_proj_w.detach().float()))
            xdbl = torch.empty((Bt, L, XC), dtype=x.dtype, device=x.device)
            with torch.cuda.device(x.device), _lib.timed("fp8_planes_gemm", Bt * L * (Dm + XC) * 2):
                _lib.check(_lib.lib().xfm_fp8_planes_gemm(x.data_ptr(), wq.data_ptr(), scale.data_ptr(), xdbl.data_ptr(), Bt, Dm,
                                                          L, XC, _lib.stream_ptr()), "fp8_planes_gemm")
        else:
            # the padded bf16 weight: a registered copy kept current by the optimizer step (amp.refresh_derived), else padded
            # here (a fill + a copy) and handed over for the following steps
            xw3 = _amp.padded_shadow(x_proj_w, C2p, x.dtype) if plain else None
            if xw3 is not None:
                xw_pad = xw3.view(XC, Dm)
            else:
                xw_pad = padded(cast_weight(x_proj_w, x.dtype)).contiguous()
                if plain and isinstance(x_proj_w, torch.nn.Parameter):
                    _amp.adopt_padded(x_proj_w, xw_pad.view(K, C2p, Dm))
            from .proj import _mfma_proj
            # (the tiled layout-changing projection where it covers the shape -- 14 x 14: 384 -> 128 -- else the library)
            xdbl = _mfma_proj(x, xw_pad, None, False, True, False) if _XPROJ_TILED else None      # (B, L, XC) token-major
            if xdbl is None and x.dtype == torch.bfloat16 and xw_pad.dtype == torch.bfloat16 and _transpose_short_ok(x, L, Dm):
                # short maps (7 x 7): x to token-major by the streaming transpose, then ONE GEMM over all B * L rows (the
                # per-sample products of bmm: 28 us for the deep block against 5 + 8); the backward pass reuses xt
                xt = _transpose_raw(x, False)                                                    # (B, L, D)
                xdbl = torch.matmul(xt, xw_pad.t())
            if xdbl is None:
                xdbl = torch.bmm(x.transpose(1, 2), xw_pad.t().unsqueeze(0).expand(Bt, Dm, XC))
        A, D, bias = A.float().contiguous(), D.float().contiguous(), bias.float().contiguous()
        lib = _lib.lib()
        nst = lib.xfm_ss2dc_nsteps(H, W, N)
        chk = torch.empty((Bt, K, nst, N, Dm), dtype=torch.float32, device=x.device)
        y = torch.empty((Bt, L, Dm) if y_tokens else (Bt, Dm, L), dtype=torch.float32, device=x.device)
        p = _params(x, xdbl, wdt, A, D, bias, H, W, N, R, 4, c_mod, c_off, 1, chk, y_tokens, x_tokens)
        p.y = y.data_ptr()
        nbytes = Bt * Dm * L * (2 + 4) + xdbl.numel() * 2
        with torch.cuda.device(x.device), _lib.timed("ss2dc_fwd" if N == 1 else "ss2dc16_fwd", nbytes):
            _lib.check(lib.xfm_ss2dc_fwd(ctypes.byref(p), _lib.stream_ptr()), "ss2dc_fwd")
        ctx.hw = (H, W)
        ctx.cmod = (c_mod, c_off)
        ctx.meta = (x_proj_w.dtype, tuple(x_proj_w.shape), dt_w.dtype)
        ctx.save_for_backward(x, xdbl, xw_pad, wdt, A, D, bias, chk)
        ctx.xt = xt                                        # token-major x of the short-map path, or None
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        from .proj import _bmm_f32, wgrad_mfma
        x, xdbl, xw_pad, wdt, A, D, bias, chk = ctx.saved_tensors
        H, W = ctx.hw
        xw_dtype, xw_shape, dtw_dtype = ctx.meta
        dev = x.device
        if ctx.xtok:
            Bt, L, Dm = x.shape
        else:
            Bt, Dm, L = x.shape
        K, C2 = xw_shape[0], xw_shape[1]
        N = A.shape[1]
        R = C2 - 2 * N
        Rp8, NBo, C2p, _ = _col_layout(R, N)
        XC = K * C2p
        dy = dy.contiguous().float()
        dx = torch.empty_like(x)
        ddts = torch.empty((Bt, K, L, Dm), dtype=x.dtype, device=dev)
        nbc, na, nd, nw = Bt * K * 2 * N * L, A.numel(), D.numel(), K * Dm * R
        acc = zeros_f32(nbc + na + 2 * nd + nw, dev)       # ONE fill for all accumulators
        dBC = acc[:nbc].view(Bt, K, 2, N, L)
        dA = acc[nbc:nbc + na].view(A.shape)
        dD, dbias = acc[nbc + na:nbc + na + nd], acc[nbc + na + nd:nbc + na + 2 * nd]
        dwdt = acc[nbc + na + 2 * nd:].view(K, Dm, R)
        lib = _lib.lib()
        p = _params(x, xdbl, wdt, A, D, bias, H, W, N, R, 4, ctx.cmod[0], ctx.cmod[1], 1, chk, ctx.ytok, ctx.xtok)
        p.dy, p.dx, p.ddts = dy.data_ptr(), dx.data_ptr(), ddts.data_ptr()
        p.dBC, p.dA, p.dD, p.ddelta_bias = dBC.data_ptr(), dA.data_ptr(), dD.data_ptr(), dbias.data_ptr()
        nbytes = Bt * Dm * L * (2 + 4 + 2 + 2 * K) + xdbl.numel() * 2
        with torch.cuda.device(dev), _lib.timed("ss2dc_bwd" if N == 1 else "ss2dc16_bwd", nbytes):
            _lib.check(lib.xfm_ss2dc_bwd(ctypes.byref(p), _lib.stream_ptr()), "ss2dc_bwd")
        # ---- d x_dbl (dt_proj columns from ddts, B / C columns from the scan kernel) and the dt_proj weight gradient:
        # two MFMA kernels, each reading ddts once
        dxdbl = torch.empty((Bt, L, XC), dtype=x.dtype, device=dev)
        with torch.cuda.device(dev), _lib.timed("ss2dc_post", 2 * ddts.numel() * 2):
            _lib.check(lib.xfm_ss2dc_post(ddts.data_ptr(), xdbl.data_ptr(), wdt.data_ptr(), dBC.data_ptr(), dxdbl.data_ptr(),
                                          dwdt.data_ptr(), Bt, Dm, L, R, N, _lib.stream_ptr()), "ss2dc_post")
        # x_proj backward on the natural map
        lib2 = _lib.lib()
        if ctx.xtok:
            # token-major x: dx (B, L, D) += d x_dbl . Wx as one token GEMM with beta = 1; dWx = d x_dbl^T . x, tokens x tokens
            dx.view(Bt * L, Dm).addmm_(dxdbl.view(Bt * L, XC), xw_pad)
        elif (_XPROJ_TILED and XC % 64 == 0 and Dm % 128 == 0 and Bt * L >= 4096 and lib2.xfm_proj_gemm_supported(XC, Dm, L)
                and xw_pad.dtype == torch.bfloat16 and dxdbl.dtype == torch.bfloat16 and dx.dtype == torch.bfloat16
                and dx.is_contiguous()
                and dxdbl.data_ptr() % 16 == 0 and xw_pad.data_ptr() % 16 == 0 and xw_pad.is_contiguous()):
            # dx += Wx^T . d x_dbl^T: tokens in, planes out, accumulating; xw_pad (XC, D) IS the (con, out) layout
            with torch.cuda.device(dev), _lib.timed("proj_gemm", Bt * L * (XC + 2 * Dm) * 2):
                _lib.check(lib2.xfm_proj_gemm_accumulate(dxdbl.data_ptr(), xw_pad.data_ptr(), dx.data_ptr(), Bt, L, XC, Dm, 1,
                                                         _lib.stream_ptr()), "proj_gemm_accumulate")
        elif ctx.xt is not None and dx.dtype == torch.bfloat16 and dx.is_contiguous() and dx.data_ptr() % 16 == 0:
            t = torch.matmul(dxdbl, xw_pad)                                               # (B, L, D) token-major
            with torch.cuda.device(dev), _lib.timed("transpose_short", t.numel() * 6):
                _lib.check(lib2.xfm_transpose_short_add_bf16(t.data_ptr(), dx.data_ptr(), Bt, L, Dm, _lib.stream_ptr()),
                           "transpose_short_add")                                         # dx += t^T
        else:
            dx.baddbmm_(xw_pad.t().unsqueeze(0).expand(Bt, Dm, XC), dxdbl.transpose(1, 2))   # dx += Wx^T . d x_dbl^T
        if ctx.xt is not None:
            dxw_pad = wgrad_mfma(dxdbl, False, ctx.xt, False)                             # (XC, D) fp32, tokens x tokens
            ctx.xt = None
        else:
            dxw_pad = wgrad_mfma(dxdbl, False, x, True)                                   # (XC, D) fp32
        if dxw_pad is None:
            dxw_pad = _bmm_f32(dxdbl.transpose(1, 2), x.transpose(1, 2)).sum(0)
        if Rp8 == R:
            dxw = dxw_pad.view(K, C2p, Dm)[:, :C2].to(xw_dtype)
        else:
            dxw = dxw_pad.index_select(0, _row_index(K, R, N, dev)).view(xw_shape).to(xw_dtype)
        return dx, dxw, dwdt.to(dtw_dtype), dA, dD, dbias, None, None, None, None, None, None


def ss2d_chan_fn(x, x_proj_weight, dt_projs_weight, A, D, bias, H, W, c_mod=0, c_off=0, y_tokens=False, x_tokens=False):
    """x (B,D,L) bf16 natural; x_proj_weight (4,R+2N,D); dt_projs_weight (4,D,R); A (4D,N); D/bias (4D,) -> y (B,D,L) fp32.
    ``c_mod > 0``: sample sb reads its C operand from sample ``c_off + sb % c_mod`` (the deep fusion block's three streams
    as one batch [view 1 | view 2 | fused]: the view streams read through the fused stream's C, reference
    models/fusion_vmamba.py:536-538, 567-569).  ``y_tokens``: y comes out (and its gradient goes in) TOKEN-MAJOR (B, L, D)
    (``ytokens_supported``)."""
    return SS2DChanHip.apply(x, x_proj_weight, dt_projs_weight, A, D, bias, H, W, c_mod, c_off, y_tokens, x_tokens)


# ---------------------------------------------------------------------------------------------------------------------
# the shallow swap block (reference models/fusion_vmamba.py:189-241, 808-845): K = 2 FORWARD-ONLY row-major routes over the
# two channel-swapped views.  With the views stacked [view 1 | view 2] along the batch axis the two routes ARE 2B samples
# with one route each (route k = rows k B .. (k + 1) B - 1, weight set k): xfm_ss2dc_fwd/_bwd with n_routes == 1, wdiv == B --
# the single-route kernels of csrc/ss2d_chan.hip (namespace deep), dt_proj inside, no (B, 2, D, L) step-size tensor, no
# swap_scan -> matmul -> matmul -> contiguous -> rowscan operator chain.
# ---------------------------------------------------------------------------------------------------------------------
class SwapViewsStacked(torch.autograd.Function):
    """x = [view 1 | view 2] (2B, C, L) -> [route 0 | route 1] (2B, C, L): even channels exchanged between the views
    (``SwappingScan_multiview``, reference fusion_vmamba.py:189-221).  ``xfm_swap_scan`` with the batch folded into the channel
    axis (C even, so channel parity survives) writes the two routes one after the other.  The backward is the reference's
    un-swapped pass-through (:217-221): route k's gradient goes to view k -- in this ordering the identity."""

    @staticmethod
    def forward(ctx, x):
        _lib.require_cuda(x)
        B2, C, L = x.shape
        if B2 % 2 or C % 2:
            raise RuntimeError("swap_views_stacked: an even batch [view 1 | view 2] and an even channel count expected")
        x = x.contiguous()
        out = torch.empty_like(x)
        half = x.numel() // 2
        with torch.cuda.device(x.device), _lib.timed("swap_scan", 2 * x.numel() * x.element_size()):
            _lib.check(_lib.lib().xfm_swap_scan(x.data_ptr(), x.data_ptr() + half * x.element_size(), out.data_ptr(), 1,
                                                (B2 // 2) * C, L, _lib.dtype_code(x.dtype), _lib.stream_ptr()), "swap_scan")
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy


def swap_views_stacked(x):
    return SwapViewsStacked.apply(x)


class SS2DChanSwapHip(torch.autograd.Function):
    """x_proj -> split -> dt_proj -> softplus -> scan of the two swap routes (fusion_vmamba.py:813-833) as one node.
    x (2B, D, L) bf16 = [route 0 | route 1] planes; x_proj_weight (2, R + 2N, D); dt_projs_weight (2, D, R); A (2D, N);
    D / bias (2D,) -> ys (2B, D, L) fp32 = [route 0 | route 1]."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, x_proj_w, dt_w, A, D, bias, H, W):
        _lib.require_cuda(x, x_proj_w, dt_w, A, D, bias)
        Bt, Dm, L = x.shape
        K, C2, _ = x_proj_w.shape
        R, N = dt_w.shape[2], A.shape[1]
        if L != H * W or K != 2 or Bt % 2 or C2 != R + 2 * N or x.dtype != torch.bfloat16:
            raise RuntimeError("ss2d_chan_swap: x (2B,D,H*W) bf16, x_proj_weight (2,R+2N,D), dt_projs_weight (2,D,R) expected")
        x = x.contiguous()
        B = Bt // 2
        Rp8, NBo, C2p, idx = _col_layout(R, N)
        wdt = cast_weight(dt_w, x.dtype)
        wdt = (torch.nn.functional.pad(wdt, (0, Rp8 - R)) if Rp8 != R else wdt).contiguous()    # (2, D, Rp8)
        xw = cast_weight(x_proj_w, x.dtype)
        if Rp8 == R:
            xw_pad = (torch.nn.functional.pad(xw, (0, 0, 0, C2p - C2)) if C2p != C2 else xw).contiguous()
        else:
            xw_pad = torch.zeros((K, C2p, Dm), dtype=x.dtype, device=x.device)
            xw_pad[:, torch.tensor(idx, device=x.device)] = xw
        xt = _transpose_raw(x, False)                                                        # (2B, L, D) token-major
        xdbl = torch.bmm(xt.view(K, B * L, Dm), xw_pad.transpose(1, 2)).view(Bt, L, C2p)     # one product per weight set
        A, D, bias = A.float().contiguous(), D.float().contiguous(), bias.float().contiguous()
        lib = _lib.lib()
        nst = lib.xfm_ss2dc_nsteps(H, W, N)
        chk = torch.empty((Bt, 1, nst, N, Dm), dtype=torch.float32, device=x.device)
        y = torch.empty((Bt, Dm, L), dtype=torch.float32, device=x.device)
        p = _params(x, xdbl, wdt, A, D, bias, H, W, N, R, 1, 0, 0, B, chk)
        p.y = y.data_ptr()
        with torch.cuda.device(x.device), _lib.timed("ss2dc16s_fwd", Bt * Dm * L * (2 + 4) + xdbl.numel() * 2):
            _lib.check(lib.xfm_ss2dc_fwd(ctypes.byref(p), _lib.stream_ptr()), "ss2dc_fwd (n_routes 1)")
        ctx.hw = (H, W)
        ctx.meta = (x_proj_w.dtype, tuple(x_proj_w.shape), dt_w.dtype)
        ctx.save_for_backward(x, xt, xdbl, xw_pad, wdt, A, D, bias, chk)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        from .proj import _bmm_f32
        x, xt, xdbl, xw_pad, wdt, A, D, bias, chk = ctx.saved_tensors
        H, W = ctx.hw
        xw_dtype, xw_shape, dtw_dtype = ctx.meta
        dev = x.device
        Bt, Dm, L = x.shape
        K, C2 = xw_shape[0], xw_shape[1]
        B = Bt // 2
        N = A.shape[1]
        R = C2 - 2 * N
        Rp8, NBo, C2p, idx = _col_layout(R, N)
        dy = dy.contiguous().float()
        dx = torch.empty_like(x)
        ddts = torch.empty((Bt, 1, L, Dm), dtype=x.dtype, device=dev)
        nbc, na, nd = Bt * 2 * N * L, A.numel(), D.numel()
        acc = zeros_f32(nbc + na + 2 * nd, dev)                # ONE fill for all accumulators
        dBC = acc[:nbc].view(Bt, 1, 2, N, L)
        dA = acc[nbc:nbc + na].view(A.shape)
        dD, dbias = acc[nbc + na:nbc + na + nd], acc[nbc + na + nd:]
        lib = _lib.lib()
        p = _params(x, xdbl, wdt, A, D, bias, H, W, N, R, 1, 0, 0, B, chk)
        p.dy, p.dx, p.ddts = dy.data_ptr(), dx.data_ptr(), ddts.data_ptr()
        p.dBC, p.dA, p.dD, p.ddelta_bias = dBC.data_ptr(), dA.data_ptr(), dD.data_ptr(), dbias.data_ptr()
        with torch.cuda.device(dev), _lib.timed("ss2dc16s_bwd", Bt * Dm * L * (2 + 4 + 2 + 2) + xdbl.numel() * 2):
            _lib.check(lib.xfm_ss2dc_bwd(ctypes.byref(p), _lib.stream_ptr()), "ss2dc_bwd (n_routes 1)")
        # d x_dbl: the dt_proj columns from ddts (one small product per weight set), the B / C columns from the scan kernel
        g2 = ddts.view(K, B * L, Dm)
        dxr = torch.bmm(g2, wdt)                                                             # (2, B L, Rp8), zero beyond R
        dbc = dBC.view(Bt, 2 * N, L).transpose(1, 2).to(x.dtype)                             # (2B, L, 2N): [dB | dC]
        dxdbl = torch.cat([dxr.view(Bt, L, Rp8), dbc], dim=2)                                # (2B, L, C2p)
        dwdt = _bmm_f32(g2.transpose(1, 2), xdbl.view(K, B * L, C2p)[:, :, :R])              # (2, D, R) fp32
        # x_proj backward: dx += (d x_dbl . Wx)^T through the streaming transpose, dWx = d x_dbl^T . x (token-major x)
        d2 = dxdbl.view(K, B * L, C2p)
        t = torch.bmm(d2, xw_pad).view(Bt, L, Dm)
        if dx.data_ptr() % 16 == 0 and t.is_contiguous() and _transpose_short_ok(t, L, Dm):
            with torch.cuda.device(dev), _lib.timed("transpose_short", t.numel() * 6):
                _lib.check(lib.xfm_transpose_short_add_bf16(t.data_ptr(), dx.data_ptr(), Bt, L, Dm, _lib.stream_ptr()),
                           "transpose_short_add")                                            # dx += t^T
        else:
            dx += t.transpose(1, 2)
        dxw_pad = _bmm_f32(d2.transpose(1, 2), xt.view(K, B * L, Dm))                        # (2, C2p, D) fp32
        if Rp8 == R:
            dxw = dxw_pad[:, :C2].to(xw_dtype)
        else:
            dxw = dxw_pad[:, torch.tensor(idx, device=dev)].to(xw_dtype)
        return dx, dxw, dwdt.to(dtw_dtype), dA, dD, dbias, None, None


def ss2d_chan_swap_fn(x, x_proj_weight, dt_projs_weight, A, D, bias, H, W):
    """x (2B, D, L) bf16 = [route 0 | route 1] (``swap_views_stacked``); weights of the two routes stacked -> (2B, D, L) fp32."""
    return SS2DChanSwapHip.apply(x, x_proj_weight, dt_projs_weight, A, D, bias, H, W)
