"""Channel projections (1x1 convolutions) as batched library GEMMs that change the activation layout for free.

The trunk keeps its residual stream TOKEN-major, (B, H*W, C): LayerNorm and the Mlp GEMMs then run on plain
row-major matrices.  The scan path wants PLANES, (B, D, H*W).  ``batched_proj`` evaluates ``y[b] = W @ x[b]`` with
either operand layout on either side; the transposition is a BLAS operand flag (a strided view handed to
``torch.bmm``), so no activation is ever copied or permuted: in_proj reads tokens and writes planes, out_proj reads
planes and writes tokens, x_proj stays on planes.  This replaces ``Linear2d.forward`` = ``F.conv2d`` with a 1x1
kernel (``models/fusion_vmamba.py:42-45``), for which MIOpen transposes NCHW <-> NHWC around an implicit GEMM.
Weight gradients are per-sample partial products (fp32) summed over the batch.
"""
from __future__ import annotations

import os

import torch

from .amp import cast_weight

__all__ = ["batched_proj", "split_k_wgrad", "mfma_planes", "wgrad_mfma", "wgrad_stream", "join_wgrad_stream", "WgradArena",
           "set_wgrad_arena", "wgrad_slot", "zeros_f32"]

_F32_OUT = [None]      # does torch.bmm accept out_dtype on this build?  probed once


def _bmm_f32(a, b):
    if a.dtype not in (torch.bfloat16, torch.float16):
        return torch.bmm(a, b)
    if _F32_OUT[0] is None:
        try:
            r = torch.bmm(a, b, out_dtype=torch.float32)
            _F32_OUT[0] = True
            return r
        except (TypeError, RuntimeError):
            _F32_OUT[0] = False
    if _F32_OUT[0]:
        return torch.bmm(a, b, out_dtype=torch.float32)
    return torch.bmm(a, b).float()


# XFM_WGRAD=0: weight gradients through the library (per-sample / per-slice partial products + a sum).  Read once.
_WGRAD = os.environ.get("XFM_WGRAD", "1") == "1"


# ---- weight gradients on a side stream ------------------------------------------------------------------------------------
# A weight gradient has no consumer before the optimizer, and the token-contracting kernel runs one workgroup per CU (its
# atomic tail sets the slice count): next to it the CUs have room for the element-wise / LayerNorm / scan kernels of the
# main backward chain.  With ``wgrad_stream(True)`` the launches whose result IS the returned gradient go to one side
# stream (forked from the current stream, so they are captured into the step's hipGraph as a parallel branch) and the
# caller joins it -- ``join_wgrad_stream()`` -- before anything reads ``.grad`` (FusedAdam.step and GradBuckets do).  Off by
# default: code that reads gradients straight after ``backward()`` must not need to know about it -- and measured on the
# XFMamba-T step it LOSES (1520 vs 1571 samples/s: the branch takes CUs from kernels that were already filling them).
_SIDE = {"on": False, "stream": None, "pending": False}


def wgrad_stream(enable: bool) -> None:
    _SIDE["on"] = bool(enable)


def join_wgrad_stream(end_of_pass: bool = True) -> None:
    """Make the current stream wait for the weight-gradient launches issued so far (no-op when there are none), and fold
    the deferred column sums (deferred.py): every reader of parameter gradients calls this first."""
    from . import deferred
    deferred.flush(_end_of_pass=end_of_pass)         # (False: a reader in the MIDDLE of a backward pass, e.g. a bucket hook)
    if _SIDE["pending"]:
        torch.cuda.current_stream().wait_stream(_SIDE["stream"])
        _SIDE["pending"] = False


# ---- one zero fill per step for the weight-gradient accumulators -----------------------------------------------------------
# xfm_wgrad ACCUMULATES into dw (fp32 atomics), so every launch needs a zeroed (M, N) tensor: ~60 fill kernels of a few us
# each per step.  A ``WgradArena`` is one flat fp32 buffer with a slot per registered weight; the training loop zeroes it
# once before ``backward()`` (``arena.zero()``: one fill) and the weight-gradient launches accumulate straight into their
# slots, which autograd then adopts as ``.grad``.  Opt-in (``set_wgrad_arena``): the gradients alias a buffer the NEXT
# ``arena.zero()`` wipes, which is only sound for loops that drop ``.grad`` before every backward pass (bench.py does).  A
# weight that receives a second gradient in the same pass (shared weights, two sequential trunk calls) falls back to a fresh
# tensor for it -- the slot must not be handed to autograd twice.
#
# The same buffer serves the OTHER zero-initialised fp32 accumulators of the backward nodes (dA / dD / dbias / dB / dC sums
# of the scan kernels, the weight / bias sums of the LayerNorm and depthwise-convolution kernels, weight gradients without a
# slot: ~70 more fills per step) from a scratch region behind the slots: ``zeros_f32(n)`` hands out consecutive pieces of
# it (a bump pointer reset by ``zero()``).  The region is sized by what the previous step asked for -- the first step
# after construction falls back to ``torch.zeros`` and only records its demand -- and the same lifetime rule applies: what
# a node returns from it lives until the next ``zero()``.
class WgradArena:
    def __init__(self, params):
        ps = [p for p in params if p.requires_grad and p.dtype == torch.float32 and p.dim() >= 2 and p.is_cuda]
        self.offsets, n = {}, 0
        for p in ps:
            self.offsets[id(p)] = (n, p.shape[0], p.numel() // p.shape[0])
            n += (p.numel() + 63) // 64 * 64                    # 256-byte aligned slots
        self.n_slots = n
        self.device = ps[0].device if ps else torch.device("cpu")
        self.buf = torch.zeros(max(n, 1), dtype=torch.float32, device=self.device)
        self.used = set()
        self.scratch_cap = 0          # elements behind the slots
        self.scratch_off = 0          # bump pointer of this step
        self.scratch_need = 0         # demand of this step (granted or not)
        self.frozen = False           # a captured graph holds raw addresses of ``buf``: it is never reallocated afterwards

    def zero(self):
        if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            self.frozen = True
        if self.scratch_need > self.scratch_cap and not self.frozen:
            self.scratch_cap = self.scratch_need + self.scratch_need // 8
            self.buf = torch.empty(self.n_slots + self.scratch_cap, dtype=torch.float32, device=self.device)
        self.buf.zero_()
        self.used.clear()
        self.scratch_off = self.scratch_need = 0

    def scratch(self, n: int):
        """``n`` zeroed fp32 elements (256-byte aligned) from the scratch region, or None when it is exhausted."""
        n64 = (n + 63) // 64 * 64
        self.scratch_need += n64
        if self.scratch_off + n64 > self.scratch_cap:
            return None
        o = self.n_slots + self.scratch_off
        self.scratch_off += n64
        return self.buf[o:o + n]

    def slot(self, weight, M, N):
        """The (M, N) fp32 slot of ``weight``, or None (unregistered, other shape, or already used in this pass)."""
        o = self.offsets.get(id(weight))
        if o is None or (o[1], o[2]) != (M, N) or id(weight) in self.used:
            return None
        self.used.add(id(weight))
        return self.buf[o[0]:o[0] + M * N].view(M, N)


_ARENA = [None]


def set_wgrad_arena(arena) -> None:
    _ARENA[0] = arena


def wgrad_slot(weight, M, N):
    return None if _ARENA[0] is None or weight is None else _ARENA[0].slot(weight, M, N)


def zeros_f32(n: int, device) -> torch.Tensor:
    """A zero-filled flat fp32 accumulator of ``n`` elements: a piece of the arena's scratch region when an arena is set
    (valid until its next ``zero()``), else a fresh ``torch.zeros``."""
    a = _ARENA[0]
    if a is not None and a.device == device:
        t = a.scratch(n)
        if t is not None:
            return t
    return torch.zeros(n, dtype=torch.float32, device=device)


def wgrad_mfma(a: torch.Tensor, a_planes: bool, b: torch.Tensor, b_planes: bool, out: torch.Tensor = None,
               deferred: bool = False):
    """``dW[m, n] = sum_{b, l} A[b, l, m] B[b, l, n]`` -> (M, N) fp32 through ``xfm_wgrad`` (csrc/wgrad_gemm.hip), or None
    when the kernel does not cover the call (the caller then uses the library).

    ``a`` / ``b``: 3-D bf16 tensors, token-major (batch, L, C) or -- ``x_planes`` -- plane-major (batch, C, L); the last two
    axes contiguous, any sample stride.  ``out``: an fp32 (M, N) tensor to ACCUMULATE into (e.g. a view of a buffer the
    caller zero-fills together with other accumulators); by default a fresh zero-filled one.  ``deferred``: the result
    is only read after ``join_wgrad_stream()`` (see above) -- the launch may then go to the side stream."""
    if not _WGRAD or a.dtype != torch.bfloat16 or b.dtype != torch.bfloat16 or not a.is_cuda or a.dim() != 3 or b.dim() != 3:
        return None
    from . import _lib
    Bt = a.shape[0]
    L, M = (a.shape[2], a.shape[1]) if a_planes else (a.shape[1], a.shape[2])
    Lb, N = (b.shape[2], b.shape[1]) if b_planes else (b.shape[1], b.shape[2])
    if Lb != L or b.shape[0] != Bt:
        return None
    if L % 4 and Bt * L * max(M, N) >= (1 << 20):
        # 7 x 7 maps (49 tokens): the kernel reads ragged plane rows element by element (88 us for 64 x 49 x 768 x 768).
        # Transposing such an operand to token-major first (a 5 MB copy) and contracting the samples as ONE token run
        # on the LDS-direct token x token kernel is more than twice as fast.
        if a_planes:
            a, a_planes = (_transpose_raw(a, False) if _transpose_short_ok(a, L, M) else a.transpose(1, 2).contiguous()), False
        if b_planes:
            b, b_planes = (_transpose_raw(b, False) if _transpose_short_ok(b, L, N) else b.transpose(1, 2).contiguous()), False
    if not a_planes and not b_planes and Bt > 1 and a.is_contiguous() and b.is_contiguous() and (Bt * L) % 64 == 0:
        a, b, L, Bt = a.view(1, Bt * L, M), b.view(1, Bt * L, N), Bt * L, 1       # one long token run

    def dense(t):
        return t.stride(2) == 1 and t.stride(1) == t.shape[2]

    lib = _lib.lib()
    if not (dense(a) and dense(b)) or a.data_ptr() % 16 or b.data_ptr() % 16 \
            or not lib.xfm_wgrad_supported(M, N, L, int(a_planes), int(b_planes)):
        return None
    a_bs, b_bs = (a.stride(0) if Bt > 1 else a.shape[1] * a.shape[2]), (b.stride(0) if Bt > 1 else b.shape[1] * b.shape[2])
    rag = L % 4 != 0                                        # ragged plane rows (7 x 7 maps) are read element-wise
    if (a_bs % ((1 if rag else 4) if a_planes else 8)) or (b_bs % ((1 if rag else 4) if b_planes else 8)):
        return None
    dw = zeros_f32(M * N, a.device).view(M, N) if out is None else out

    def launch():
        with torch.cuda.device(a.device), _lib.timed("wgrad", (a.numel() + b.numel()) * 2):
            _lib.check(lib.xfm_wgrad(a.data_ptr(), b.data_ptr(), dw.data_ptr(), M, N, Bt, L, a_bs, b_bs, int(a_planes),
                                     int(b_planes), _lib.stream_ptr()), "wgrad")

    if deferred and _SIDE["on"]:
        if _SIDE["stream"] is None:
            _SIDE["stream"] = torch.cuda.Stream(device=a.device)
        side = _SIDE["stream"]
        side.wait_stream(torch.cuda.current_stream())      # operands (and the zero fill) are ready
        with torch.cuda.stream(side):
            launch()
        for t in (a, b, dw):                                # allocated on the main stream, in use on the side stream
            t.record_stream(side)
        _SIDE["pending"] = True
    else:
        launch()
    return dw


def _k_slices(rows: int, target: int = 2048, cap: int = 128) -> int:
    """Number of equal row slices (a divisor of ``rows``) whose length is closest to ``target``."""
    want = max(1, min(cap, rows // target))
    best = 1
    for s in range(1, min(cap, rows) + 1):
        if rows % s == 0 and abs(s - want) < abs(best - want):
            best = s
    return best


def split_k_wgrad(dy2: torch.Tensor, x2: torch.Tensor, deferred: bool = False, out: torch.Tensor = None) -> torch.Tensor:
    """``dy2^T @ x2`` for tall operands (rows, M), (rows, K) -> (M, K) fp32.

    A weight gradient contracts over every token (rows = B*H*W up to 2e5) into a small (M, K) result; handed to the
    GEMM library as one product it runs on the handful of workgroups that tile (M, K).  Cut into row slices it is a
    batched GEMM that fills the chip, with fp32 partial products summed afterwards."""
    dy2, x2 = dy2.contiguous(), x2.contiguous()
    rows = dy2.shape[0]
    dw = wgrad_mfma(dy2.unsqueeze(0), False, x2.unsqueeze(0), False, out=out, deferred=deferred)   # one launch, no partials
    if dw is not None:
        return dw
    if out is not None:                               # (the library formulation below writes a fresh tensor)
        return out.add_(split_k_wgrad(dy2, x2))
    S = _k_slices(rows)
    if S == 1:
        return _bmm_f32(dy2.t().unsqueeze(0), x2.unsqueeze(0))[0]
    return _bmm_f32(dy2.view(S, rows // S, -1).transpose(1, 2), x2.view(S, rows // S, -1)).sum(0)


# XFM_TOKENS_GEMM=0: library GEMMs everywhere.  Read ONCE at import: set it before importing xfmamba_amd.
_MFMA = os.environ.get("XFM_TOKENS_GEMM", "1") == "1"


def _mfma_proj(x, w, bias, in_tokens, out_tokens, transposed):
    """The layout-changing projection through ``xfm_proj_gemm`` (csrc/tokens_gemm.hip), or None when it does not cover
    the call: bf16, exactly one plane-major side, widths built, contiguous 16-byte aligned operands,
    L % 8 == 0 and (B * L) % 32 == 0 (the kernel's own guard; samples may end inside 32-token tiles).  ``transposed``: ``w`` is (con, out)."""
    if not _MFMA or in_tokens == out_tokens or x.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or not x.is_cuda:
        return None
    from . import _lib
    B = x.shape[0]
    L, con = (x.shape[1], x.shape[2]) if in_tokens else (x.shape[2], x.shape[1])
    out = w.shape[1] if transposed else w.shape[0]
    if (w.shape[0] if transposed else w.shape[1]) != con or B * L < 4096 or (B * L) % 32 or not w.is_contiguous() \
            or not x.is_contiguous() or x.data_ptr() % 16 or w.data_ptr() % 16:              # 16-byte vector accesses
        return None
    lib = _lib.lib()
    if not lib.xfm_proj_gemm_supported(con, out, L):
        return None
    y = torch.empty((B, L, out) if out_tokens else (B, out, L), dtype=x.dtype, device=x.device)
    b = None if bias is None else bias.float().contiguous()
    with torch.cuda.device(x.device), _lib.timed("proj_gemm", B * L * (con + out) * 2):
        _lib.check(lib.xfm_proj_gemm(x.data_ptr(), w.data_ptr(), _lib.ptr(b), y.data_ptr(), B, L, con, out,
                                     0 if in_tokens else 1, 1 if transposed else 0, _lib.stream_ptr()), "proj_gemm")
    return y


def mfma_planes(x, w, out, transposed=False, accumulate_into=None):
    """Plane-major product through ``xfm_planes_gemm``: ``W @ x[b]`` for x (B, con, L) -> (B, out, L); ``w`` is
    (out, con), or (con, out) with ``transposed``; ``accumulate_into``: add into that (B, out, L) tensor instead.
    Returns None when the kernel does not cover the call (the caller then uses the library)."""
    if not _MFMA or x.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or not x.is_cuda:
        return None
    from . import _lib
    B, con, L = x.shape
    y = accumulate_into
    if B * L < 4096 or (B * L) % 32 or not (x.is_contiguous() and w.is_contiguous()) or x.data_ptr() % 16 \
            or w.data_ptr() % 16 or (y is not None and (not y.is_contiguous() or y.dtype != x.dtype or y.data_ptr() % 16)):
        return None
    lib = _lib.lib()
    if not lib.xfm_planes_gemm_supported(con, out, L):
        return None
    if y is None:
        y = torch.empty(B, out, L, dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device), _lib.timed("planes_gemm", B * L * (con + out * (2 if accumulate_into is not None else 1)) * 2):
        _lib.check(lib.xfm_planes_gemm(x.data_ptr(), w.data_ptr(), None, y.data_ptr(), B, L, con, out,
                                       1 if transposed else 0, 0 if accumulate_into is None else 1, _lib.stream_ptr()),
                   "planes_gemm")
    return y


_CAST_IN = os.environ.get("XFM_PROJ_CAST_IN", "1") == "1"


# XFM_TRANSPOSE_SHORT=0: the framework's permute + contiguous for the 7 x 7 layout changes (A/B switch, read once)
_TSHORT = os.environ.get("XFM_TRANSPOSE_SHORT", "1") == "1"


def _transpose_short_ok(x: torch.Tensor, R: int, C: int) -> bool:
    if not (_TSHORT and x.is_cuda and x.dim() == 3 and x.element_size() == 2 and x.is_contiguous() and x.data_ptr() % 16 == 0):
        return False
    from . import _lib
    return bool(_lib.lib().xfm_transpose_short_supported(R, C))


def _transpose_raw(x: torch.Tensor, to_planes: bool) -> torch.Tensor:
    """``xfm_transpose_short`` on a contiguous 3-D tensor the caller checked with ``_transpose_short_ok`` (no autograd)."""
    from . import _lib
    B = x.shape[0]
    R, Cc = (x.shape[1], x.shape[2]) if to_planes else (x.shape[2], x.shape[1])
    y = torch.empty((B, Cc, R) if to_planes else (B, R, Cc), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device), _lib.timed("transpose_short", x.numel() * 4):
        _lib.check(_lib.lib().xfm_transpose_short(x.data_ptr(), y.data_ptr(), B, R, Cc, 1 if to_planes else 0,
                                                  _lib.stream_ptr()), "transpose_short")
    return y


class TransposeShort(torch.autograd.Function):
    """(B, R, C) tokens -> (B, C, R) planes (``to_planes``) or back, through ``xfm_transpose_short``; the gradient is the
    opposite move."""

    @staticmethod
    def forward(ctx, x, to_planes):
        ctx.to_planes = to_planes
        return _transpose_raw(x, to_planes)

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        R, Cc = (dy.shape[2], dy.shape[1]) if ctx.to_planes else (dy.shape[1], dy.shape[2])
        if _transpose_short_ok(dy, R, Cc):
            return TransposeShort.apply(dy, not ctx.to_planes), None
        return dy.transpose(1, 2).contiguous(), None


class PooledTokensToPlanes(torch.autograd.Function):
    """t (B, L, C) tokens -> (planes (B, C, L), pooled (B, C) = mean over the positions), bf16, through
    ``xfm_pooled_transpose_fwd/_bwd``: the squeeze pooling read off the tile the transpose holds anyway."""

    @staticmethod
    def forward(ctx, t):
        from . import _lib
        B, L, C = t.shape
        planes = torch.empty((B, C, L), dtype=t.dtype, device=t.device)
        pooled = torch.empty((B, C), dtype=t.dtype, device=t.device)
        with torch.cuda.device(t.device), _lib.timed("transpose_short", t.numel() * 4):
            _lib.check(_lib.lib().xfm_pooled_transpose_fwd(t.data_ptr(), planes.data_ptr(), pooled.data_ptr(), B, L, C,
                                                           _lib.stream_ptr()), "pooled_transpose_fwd")
        ctx.shape = (B, L, C)
        return planes, pooled

    @staticmethod
    def backward(ctx, dplanes, dpooled):
        from . import _lib
        B, L, C = ctx.shape
        dplanes = torch.zeros((B, C, L), dtype=dpooled.dtype, device=dpooled.device) if dplanes is None else dplanes.contiguous()
        dpooled = torch.zeros((B, C), dtype=dplanes.dtype, device=dplanes.device) if dpooled is None else dpooled.contiguous()
        dt = torch.empty((B, L, C), dtype=dplanes.dtype, device=dplanes.device)
        with torch.cuda.device(dplanes.device), _lib.timed("transpose_short", dt.numel() * 4):
            _lib.check(_lib.lib().xfm_pooled_transpose_bwd(dplanes.data_ptr(), dpooled.data_ptr(), dt.data_ptr(), B, L, C,
                                                           _lib.stream_ptr()), "pooled_transpose_bwd")
        return dt


def tokens_to_planes_pooled(t: torch.Tensor):
    """``(t.transpose(1, 2).contiguous(), t.mean(1))`` for (B, L, C) tokens."""
    if t.dtype == torch.bfloat16 and t.is_contiguous() and _transpose_short_ok(t, t.shape[1], t.shape[2]):
        return PooledTokensToPlanes.apply(t)
    return t.transpose(1, 2).contiguous(), t.mean(1)


class GatedPlanesToTokens(torch.autograd.Function):
    """``(yy * gate[:, :, None]).transpose(1, 2)`` for yy (B, C, L) planes and gate (B, C), bf16 -> (B, L, C) tokens, through
    ``xfm_gated_transpose_fwd/_bwd`` (one kernel each way; the backward also sums d gate over the positions)."""

    @staticmethod
    def forward(ctx, yy, gate):
        from . import _lib
        B, C, L = yy.shape
        out = torch.empty((B, L, C), dtype=yy.dtype, device=yy.device)
        with torch.cuda.device(yy.device), _lib.timed("gated_transpose", yy.numel() * 4):
            _lib.check(_lib.lib().xfm_gated_transpose_fwd(yy.data_ptr(), gate.data_ptr(), out.data_ptr(), B, L, C, _lib.stream_ptr()),
                       "gated_transpose_fwd")
        ctx.save_for_backward(yy, gate)
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        yy, gate = ctx.saved_tensors
        B, C, L = yy.shape
        g = g.contiguous() if g.dtype == yy.dtype else g.to(yy.dtype).contiguous()
        dyy, dgate = torch.empty_like(yy), torch.empty_like(gate)
        with torch.cuda.device(yy.device), _lib.timed("gated_transpose", yy.numel() * 6):
            _lib.check(_lib.lib().xfm_gated_transpose_bwd(g.data_ptr(), yy.data_ptr(), gate.data_ptr(), dyy.data_ptr(),
                                                          dgate.data_ptr(), B, L, C, _lib.stream_ptr()), "gated_transpose_bwd")
        return dyy, dgate


def gated_planes_to_tokens(yy: torch.Tensor, gate: torch.Tensor) -> torch.Tensor:
    """``yy (B, C, L) * gate (B, C)`` as contiguous (B, L, C) tokens."""
    if (yy.dtype == torch.bfloat16 and gate.dtype == torch.bfloat16 and yy.is_contiguous() and yy.shape[2] >= 8
            and _transpose_short_ok(yy, yy.shape[2], yy.shape[1]) and gate.data_ptr() % 16 == 0):
        return GatedPlanesToTokens.apply(yy, gate.contiguous())
    return planes_to_tokens(yy * gate.unsqueeze(-1))


def tokens_to_planes(t: torch.Tensor) -> torch.Tensor:
    """(B, L, C) token-major -> contiguous (B, C, L) plane-major (``t.transpose(1, 2).contiguous()``)."""
    if t.is_contiguous() and _transpose_short_ok(t, t.shape[1], t.shape[2]):
        return TransposeShort.apply(t, True)
    return t.transpose(1, 2).contiguous()


def planes_to_tokens(p: torch.Tensor) -> torch.Tensor:
    """(B, C, L) plane-major -> contiguous (B, L, C) token-major."""
    if p.is_contiguous() and _transpose_short_ok(p, p.shape[2], p.shape[1]):
        return TransposeShort.apply(p, False)
    return p.transpose(1, 2).contiguous()


class BatchedProj(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, in_tokens, out_tokens):
        # x: (B, L, K) tokens or (B, K, L) planes; weight (M, K); result (B, L, M) tokens or (B, M, L) planes
        if _CAST_IN and x.is_cuda and x.dtype == torch.float32 and torch.is_autocast_enabled():
            # an fp32 operand under autocast: the products below would cast it per call (and the backward pass, which
            # runs outside autocast, would contract the SAVED fp32 tensors in fp32): cast once, save the cast
            x = x.to(torch.get_autocast_gpu_dtype())
        x = x.contiguous()
        B = x.shape[0]
        cd = x.dtype
        w = cast_weight(weight, cd)
        M, K = w.shape
        y = _mfma_proj(x, w, bias, in_tokens, out_tokens, False)
        L = x.shape[1] if in_tokens else x.shape[2]
        short = (y is None and in_tokens != out_tokens and _transpose_short_ok(x, L, K)
                 and _transpose_short_ok(x, L, M))
        if short:
            # short maps (7 x 7): ONE GEMM over all B * L token rows with the layout change as a streaming transpose next to it
            # (64 per-sample products through bmm: 28 us for 64 x 49 x 768 x 768 against 8 + 3)
            xt = x if in_tokens else _transpose_raw(x, False)                 # (B, L, K)
            y = torch.nn.functional.linear(xt, w, None if bias is None else bias.to(cd))
            if not out_tokens:
                y = _transpose_raw(y, True)
            x = xt
        elif y is not None:
            pass
        elif out_tokens:
            xt = x if in_tokens else x.transpose(1, 2)                       # (B, L, K)
            y = torch.bmm(xt, w.t().unsqueeze(0).expand(B, K, M))            # (B, L, M)
            if bias is not None:
                y = y + bias.to(cd)
        else:
            xp = x.transpose(1, 2) if in_tokens else x                       # (B, K, L)
            y = torch.bmm(w.unsqueeze(0).expand(B, M, K), xp)                # (B, M, L)
            if bias is not None:
                y = y + bias.to(cd)[:, None]
        ctx.save_for_backward(x, w)
        ctx.short = short                                  # (x is saved token-major then, whatever layout it came in)
        ctx.meta = (in_tokens, out_tokens, weight.dtype, bias is not None and bias.dtype)
        ctx.wparam = weight if isinstance(weight, torch.nn.Parameter) else None      # (identity only: the arena's slot key)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        in_tokens, out_tokens, wdtype, bdtype = ctx.meta
        B = x.shape[0]
        M, K = w.shape
        dy = dy.contiguous() if dy.dtype == w.dtype else dy.to(w.dtype).contiguous()
        dx = dw = db = None
        if ctx.short:
            dyt = dy if out_tokens else _transpose_raw(dy, False)             # (B, L, M); x was saved as (B, L, K)
            if ctx.needs_input_grad[0]:
                dx = torch.matmul(dyt, w)                                     # (B, L, K)
                if not in_tokens:
                    dx = _transpose_raw(dx, True)
            if ctx.needs_input_grad[1]:
                slot = wgrad_slot(ctx.wparam, M, K) if wdtype == torch.float32 else None
                dw = wgrad_mfma(dyt, False, x, False, out=slot, deferred=wdtype == torch.float32)
                if dw is None and slot is not None:
                    ctx.wparam = None
                if dw is None:
                    dw = _bmm_f32(dyt.reshape(1, -1, M).transpose(1, 2), x.reshape(1, -1, K))[0]
                dw = dw.to(wdtype)
            if bdtype is not False and ctx.needs_input_grad[2]:
                db = dyt.sum((0, 1), dtype=torch.promote_types(dy.dtype, torch.float32)).to(bdtype)
            return dx, dw, db, None, None
        dyp = dy.transpose(1, 2) if out_tokens else dy                        # (B, M, L)
        if ctx.needs_input_grad[0]:
            dx = _mfma_proj(dy, w, None, out_tokens, in_tokens, True)        # dy has the output's layout, dx the input's
            if dx is not None:
                pass
            elif in_tokens:
                dx = torch.bmm(dyp.transpose(1, 2), w.unsqueeze(0).expand(B, M, K))          # (B, L, K)
            else:
                dx = torch.bmm(w.t().unsqueeze(0).expand(B, K, M), dyp)                      # (B, K, L)
        if ctx.needs_input_grad[1]:
            # (fp32 parameter: the kernel's output IS the gradient, nothing reads it before the optimizer)
            slot = wgrad_slot(ctx.wparam, M, K) if wdtype == torch.float32 else None
            dw = wgrad_mfma(dy, not out_tokens, x, not in_tokens, out=slot, deferred=wdtype == torch.float32)
            if dw is None and slot is not None:
                ctx.wparam = None                                                       # (slot untouched: still all zeros)
            if dw is None:
                xt = x if in_tokens else x.transpose(1, 2)                                    # (B, L, K)
                dw = _bmm_f32(dyp, xt).sum(0)
            dw = dw.to(wdtype)                                                                # (M, K)
        if bdtype is not False and ctx.needs_input_grad[2]:
            db = dy.sum((0, 1) if out_tokens else (0, 2), dtype=torch.promote_types(dy.dtype, torch.float32)).to(bdtype)
        return dx, dw, db, None, None


def batched_proj(x, weight, bias=None, in_tokens=False, out_tokens=False):
    """``y[b] = weight @ x[b]`` (+ bias) over the channel axis, reading / writing token- or plane-major tensors."""
    return BatchedProj.apply(x, weight, bias, in_tokens, out_tokens)
