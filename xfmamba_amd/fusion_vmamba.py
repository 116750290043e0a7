"""nn.Module layer of the XFMamba hot path on MI355X.

Same class names, constructor signatures, attribute names and ``state_dict`` keys as
``models/fusion_vmamba.py`` of XZheng0427/XFMamba, so reference /
VMamba checkpoints load unchanged and ``TwoViewXFMambaTop`` (``net_fusionmamba.py``) can be
wired exactly like upstream.  Only what ``TwoViewXFMambaTop`` reaches is built
(rows a4-a10 of SURVEY.md section 8); everything the reference file carries besides that
(PatchMerging2D, gMlp, cascade scans, the Mamba-2 path ...) is out of scope.

Inside the blocks, ``cross_scan_fn`` / ``selective_scan_fn`` / ``cross_merge_fn`` and the swap are
served by the gfx950 kernels in ``libxfm_hip.so``; the dense contractions go to hipBLASLt / MIOpen through
PyTorch, which is the MFMA path for plain library GEMMs.

Two switches select equivalent evaluation orders (same parameters, same results; the tests run both):
  * ``SS2D_MODE``: ``"fused"`` -- x_proj and dt_proj are evaluated once on the feature map in its NATURAL
    row-major order (route k's projection of the permuted sequence equals the permuted projection), and one
    kernel (``xfm_ss2d_fwd``) walks the four routes, scans and merges without materialising the (B,4,D,L)
    tensors; ``"unfused"`` -- the reference's operator sequence cross_scan -> x_proj -> dt_proj ->
    selective_scan -> cross_merge (fusion_vmamba.py:1145-1174), each on its own kernel (always used for 7x7 maps).
  * ``STREAM_LAYOUT``: ``"tokens"`` -- the trunk's residual stream is token-major (B, H, W, C) fp32: residual add +
    DropPath + LayerNorm are one row kernel, Mlp / 1x1 projections are plain GEMMs, 3x3 convolutions run
    channels_last, and the (B, D, L) planes of the scan path come out of in_proj's batched GEMM; ``"planes"`` -- NCHW
    modules exactly as the reference wires them.
"""
from __future__ import annotations

import math
import os
from collections import OrderedDict
from functools import partial
from typing import Any, Callable

import torch
import torch.nn as nn
import torch.nn.functional as F

from .csm import SwappingMerge_multiview, SwappingScan_multiview, SwappingScanStacked, cross_merge_fn, cross_scan_fn
from .conv_tokens import (conv3x3s2_gray_fn, conv3x3s2_gray_supported, conv3x3s2_tokens_fn, conv3x3s2_tokens_supported,
                          conv3x3s2_wgrad_from_map)
from .csms6s import selective_scan_fn
from .dwconv import dwconv3x3_silu_fn, dwconv3x3_silu_tokens_fn, dwconv_tokens_supported
from .layernorm2d import layernorm2d_fn
from .mlp_tokens import bias_gelu_fn, linear_tokens_fn, mlp_tokens_fn
from .proj import batched_proj, gated_planes_to_tokens, planes_to_tokens, tokens_to_planes, tokens_to_planes_pooled
from .rowln import (residual_settle_fn, add_layernorm_rows_fn, layernorm_rows_fn, layernorm_rows_gelu_fn, layernorm_rows_pass_fn,
                    rows_supported)
from .ss2d import ss2d_core_fn, ss2d_xproj_core_fn, to_route_order
from .ss2d_chan import chan_supported, ss2d_chan_fn, ytokens_supported
from . import fp8 as _fp8
from .amp import cast_weight

SS2D_MODE = "fused"          # "fused" | "unfused"
# Layout of the trunk's residual stream between VSS blocks.  "tokens": (B, H, W, C) fp32 -- LayerNorm (+ residual add
# + DropPath) is one row kernel, Mlp / in_proj / out_proj are plain hipBLASLt GEMMs, the scan path gets its (B, D, L)
# planes from in_proj's GEMM epilogue layout.  "planes": NCHW as the reference's channel_first blocks
# (1x1 convs through MIOpen).  Same parameters, same results.
STREAM_LAYOUT = "tokens"
# the shallow block's swap scan as one kernel each way (xfm_ss2dc_fwd/_bwd, n_routes 1); XFM_SHALLOW_KERNEL=0: the
# swap_scan -> matmul -> matmul -> selective_scan_fn operator chain (the A/B switch of the tests)
SHALLOW_KERNEL = os.environ.get("XFM_SHALLOW_KERNEL", "1") == "1"
# SS2D blocks of the short-map stages (14 x 14, 7 x 7) ENTIRELY token-major: in_proj / x_proj as token GEMMs, the token-major
# depthwise convolution (csrc/dwconv_tok.hip), the scan reading x token-major.  Built, parity-tested and measured at break-even
# with the default (planes between in_proj and the scan, token-major only BEHIND the scan): 13.60 vs 13.62 ms per step -- the
# layout-changing projections it removes (-0.69 ms) come back as plain token GEMMs (+0.59 ms) and the token-major depthwise
# kernels are 5 / 11 us per block slower than the plane-major ones (14.3 / 31.5 vs 9.1 / 20.6 us).  Opt in: XFM_TOKEN_SS2D=1.
TOKEN_SS2D = os.environ.get("XFM_TOKEN_SS2D", "0") == "1"


def trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, std=std)


class DropPath(nn.Module):
    """Per-sample stochastic depth (what the reference imports from timm 0.4.12: Bernoulli(keep) mask, divided by
    keep).  The scaled mask comes out of ONE dropout kernel on a cached vector of ones (same distribution)."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = float(drop_prob)
        self.scale_by_keep = scale_by_keep
        self._ones = {}
        self._preset = None          # factors sampled ahead by _DropPathBank (one launch for all layers of a trunk)

    def sample_scale(self, batch: int, device):
        """The per-sample factor ``forward`` multiplies by, as a (B,) fp32 vector (None when it is the identity)."""
        if self.drop_prob == 0.0 or not self.training:
            return None
        if self._preset:                                  # rows sampled ahead for this pass: one per use of the module
            pre = self._preset.pop()
            if pre.shape[0] == batch and pre.device.type == torch.device(device).type:
                return pre
        keep = 1.0 - self.drop_prob
        key = (batch, str(device))
        ones = self._ones.get(key)
        if ones is None:
            ones = self._ones[key] = torch.ones(batch, dtype=torch.float32, device=device)
        if keep > 0.0 and self.scale_by_keep:
            return F.dropout(ones, p=self.drop_prob, training=True)          # Bernoulli(keep) / keep
        return torch.empty(batch, dtype=torch.float32, device=device).bernoulli_(keep)

    def forward(self, x):
        mask = self.sample_scale(x.shape[0], x.device)
        if mask is None:
            return x
        return x * mask.view((x.shape[0],) + (1,) * (x.ndim - 1)).to(x.dtype)

    def extra_repr(self):
        return f"drop_prob={self.drop_prob}"


class Linear2d(nn.Linear):
    """1x1 convolution stored as an (out, in) linear weight (fusion_vmamba.py:42-49)."""

    def forward(self, x: torch.Tensor):
        # (measured: issuing this as a broadcast matmul on the NCHW planes is slower than MIOpen's implicit GEMM
        # here -- 65.6 vs 56.8 ms/step -- so the library convolution stays)
        return F.conv2d(x, self.weight[:, :, None, None], self.bias)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        k = prefix + "weight"
        if k in state_dict:
            state_dict[k] = state_dict[k].view(self.weight.shape)   # accept (out,in,1,1) conv weights
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class LayerNorm2d(nn.LayerNorm):
    """LayerNorm over the channel axis of an NCHW tensor (fusion_vmamba.py:52-57)."""

    cast_out = False      # True: under autocast emit the autocast dtype (the consumer is a GEMM that casts anyway)

    def forward(self, x: torch.Tensor, out_dtype=None):
        if out_dtype is None and self.cast_out and torch.is_autocast_enabled():
            out_dtype = torch.get_autocast_dtype("cuda")
        return layernorm2d_fn(x, self.weight, self.bias, self.eps, out_dtype)


class Permute(nn.Module):
    def __init__(self, *args):
        super().__init__()
        self.args = args

    def forward(self, x):
        return x.permute(*self.args)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0,
                 channels_first=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        Linear = Linear2d if channels_first else nn.Linear
        self.fc1 = Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))

    def forward_tokens(self, x):
        """Same Mlp on a token-major (B, H, W, C) tensor: two plain GEMMs (Linear2d weights are (out, in))."""
        return self.forward_tokens_deferred(x, defer=False)[0]

    def forward_tokens_deferred(self, x, defer=True):
        """(y, deferred_bias): with ``defer`` (and no dropout after fc2) fc2's bias is handed back instead of added,
        for the residual-add + LayerNorm kernel that consumes ``y``."""
        drop = self.drop if self.drop.p > 0.0 else None
        defer = defer and drop is None and self.fc2.bias is not None
        if isinstance(self.act, nn.GELU) and self.act.approximate == "none":
            y = mlp_tokens_fn(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, drop, defer_bias=defer)
        else:
            x = self.drop(self.act(linear_tokens_fn(x, self.fc1.weight, self.fc1.bias)))
            y = self.drop(linear_tokens_fn(x, self.fc2.weight, None if defer else self.fc2.bias))
        return y, (self.fc2.bias if defer else None)


class mamba_init:
    """Initialisers of the S6 parameters (statistics of fusion_vmamba.py:289-356)."""

    @staticmethod
    def dt_init(dt_rank, d_inner, dt_scale=1.0, dt_init="random", dt_min=0.001, dt_max=0.1, dt_init_floor=1e-4):
        proj = nn.Linear(dt_rank, d_inner, bias=True)
        std = dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(proj.weight, std)
        elif dt_init == "random":
            nn.init.uniform_(proj.weight, -std, std)
        else:
            raise NotImplementedError(dt_init)
        lo, hi = math.log(dt_min), math.log(dt_max)
        dt = torch.exp(torch.rand(d_inner) * (hi - lo) + lo).clamp(min=dt_init_floor)
        with torch.no_grad():
            proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))      # softplus^-1(dt)
        return proj

    @staticmethod
    def A_log_init(d_state, d_inner, copies=-1, device=None, merge=True):
        A_log = torch.log(torch.arange(1, d_state + 1, dtype=torch.float32, device=device)).repeat(d_inner, 1)
        if copies > 0:
            A_log = A_log[None].repeat(copies, 1, 1)
            if merge:
                A_log = A_log.flatten(0, 1)
        A_log = nn.Parameter(A_log.contiguous())
        A_log._no_weight_decay = True
        return A_log

    @staticmethod
    def D_init(d_inner, copies=-1, device=None, merge=True):
        D = torch.ones(d_inner, device=device)
        if copies > 0:
            D = D[None].repeat(copies, 1)
            if merge:
                D = D.flatten(0, 1)
        D = nn.Parameter(D.contiguous())
        D._no_weight_decay = True
        return D

    @classmethod
    def init_dt_A_D(cls, d_state, dt_rank, d_inner, dt_scale, dt_init, dt_min, dt_max, dt_init_floor, k_group=4):
        projs = [cls.dt_init(dt_rank, d_inner, dt_scale, dt_init, dt_min, dt_max, dt_init_floor)
                 for _ in range(k_group)]
        dt_projs_weight = nn.Parameter(torch.stack([t.weight for t in projs], dim=0))   # (K, inner, rank)
        dt_projs_bias = nn.Parameter(torch.stack([t.bias for t in projs], dim=0))       # (K, inner)
        A_logs = cls.A_log_init(d_state, d_inner, copies=k_group, merge=True)           # (K*D, N)
        Ds = cls.D_init(d_inner, copies=k_group, merge=True)                            # (K*D)
        return A_logs, Ds, dt_projs_weight, dt_projs_bias


# ---------------------------------------------------------------------------------------------
# SS2D core shared by the backbone block and the deep fusion block
# ---------------------------------------------------------------------------------------------
def _dwconv_act(conv: nn.Conv2d, act: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """``act(conv(x))``; the 3x3 depthwise + SiLU case every reference block uses runs on the fused HIP kernel."""
    if (isinstance(act, nn.SiLU) and conv.kernel_size == (3, 3) and conv.padding == (1, 1) and conv.stride == (1, 1)
            and conv.groups == conv.in_channels == conv.out_channels):
        return dwconv3x3_silu_fn(x, conv.weight, conv.bias, True)
    return act(conv(x))


def _ss2d_core(x, x_proj_weight, dt_projs_weight, A_logs, Ds, dt_projs_bias, Cs_override=None, want_Cs=False, As=None):
    """x: (B, D, H, W) -> (y: (B, D, H*W) fp32, Cs in the layout of the active mode).

    ``Cs_override`` lets the view streams of Cross_SS2Dv5 read their state through the fused
    stream's C (fusion_vmamba.py:537,568); ``want_Cs`` makes the call return its own C for that purpose
    (otherwise the second result is None on the fused path)."""
    B, D, H, W = x.shape
    L = H * W
    K, _, R = dt_projs_weight.shape
    N = A_logs.shape[1]
    if As is None:
        As = -A_logs.float().exp()                   # (the trunk hands in a precomputed A: _NegExpAll)
    Dsf = Ds.float()
    bias = dt_projs_bias.reshape(-1).float()
    cd = x.dtype
    # short square maps (14 x 14 and below) in bf16: the channel-lane kernel -- dt_proj on MFMA inside the scan, one lane
    # per channel; no (B,4,D,L) step-size tensor, no cross_scan / cross_merge copies (csrc/ss2d_chan.hip)
    if SS2D_MODE == "fused" and Cs_override is None and not want_Cs and chan_supported(x, H, W, N, K, D, R):
        return ss2d_chan_fn(x.reshape(B, D, L), x_proj_weight, dt_projs_weight, As, Dsf, bias, H, W), None
    # 7x7 maps (trunk stage 3, both fusion blocks): rows of 49 are too short for a parallel scan to pay; the
    # operator chain with the one-lane-per-row scan kernel is faster there and the (B,4,D,49) tensors are tiny.
    # (96 x 96 of XFMamba-B at 384^2 in 16-bit I/O runs the one-plane-per-tile lean variants; other maps beyond the lean
    #  kernel's LDS plan take the operator chain: the generic fused tile kernel measured 32.5 ms per block forward +
    #  backward at 96 x 96, the chain 14.1 ms)
    big_lean = N == 1 and cd in (torch.bfloat16, torch.float16) and 8704 < L <= 9216 and L % 8 == 0
    if SS2D_MODE == "fused" and (64 < L <= 4096 or big_lean):
        # x_proj of all K routes as ONE dense GEMM on the map in natural order (route k's projection of
        # the permuted sequence is the permuted projection).  Routes 1/3 walk columns, so their slice of
        # the small x_dbl tensor is transposed to column-major here; dt_proj (batched GEMM) then emits
        # dts for those routes directly in the order the kernel walks them -- the big (B,4,D,L) tensor is
        # written once, contiguous per route, and never permuted.
        C2 = R + 2 * N
        if Cs_override is None and not want_Cs:
            # x_proj, route split, dt_proj, scan and merge as one autograd node (the (B,4,.,L) tensors stay inside)
            return ss2d_xproj_core_fn(x.reshape(B, D, L), x_proj_weight, dt_projs_weight, As, Dsf, bias, H, W), None
        x_dbl = batched_proj(x.reshape(B, D, L), x_proj_weight.reshape(K * C2, D))             # (B, K*C2, L)
        x_dbl = to_route_order(x_dbl.view(B, K, C2, L), H, W)
        dts = torch.matmul(dt_projs_weight.to(cd), x_dbl[:, :, :R])                            # (B, K, D, L)
        Bs = x_dbl[:, :, R:R + N].contiguous()
        Cs = x_dbl[:, :, R + N:].contiguous() if Cs_override is None else Cs_override
        y = ss2d_core_fn(x.reshape(B, D, L), dts, As, Bs, Cs, Dsf, bias, H, W)
        return y, Cs
    xs = cross_scan_fn(x, in_channel_first=True, out_channel_first=True, scans=0)           # (B, 4, D, L)
    x_dbl = torch.matmul(x_proj_weight.to(cd), xs)                                          # (B, K, R+2N, L)
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
    dts = torch.matmul(dt_projs_weight.to(cd), dts).view(B, -1, L)
    Bs = Bs.contiguous()
    Cs = Cs.contiguous() if Cs_override is None else Cs_override
    ys = selective_scan_fn(xs.view(B, -1, L), dts, As, Bs, Cs, Dsf, bias, True, True, None)
    y = cross_merge_fn(ys.view(B, K, -1, H, W), in_channel_first=True, out_channel_first=True, scans=0)
    return y, Cs


_HIP_FORWARD_TYPES = ("v05", "v04", "v03", "v3")     # all: no fp32 up-cast, oflex output, cross2d routes


class SS2Dv2(nn.Module):
    """2-D selective scan block (fusion_vmamba.py:923-1254).  Built for the configurations the
    reference reaches with ``forward_type="v05_noz"`` (and its ``_noz``-less sibling)."""

    def __init__(self, d_model=96, d_state=16, ssm_ratio=2.0, dt_rank="auto", act_layer=nn.SiLU,
                 d_conv=3, conv_bias=True, dropout=0.0, bias=False,
                 dt_min=0.001, dt_max=0.1, dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, initialize="v0",
                 forward_type="v2", channel_first=False, **kwargs):
        super().__init__()
        self.k_group = 4
        self.d_model = int(d_model)
        self.d_state = int(d_state)
        self.d_inner = int(ssm_ratio * d_model)
        self.dt_rank = int(math.ceil(self.d_model / 16) if dt_rank == "auto" else dt_rank)
        self.channel_first = channel_first
        self.with_dconv = d_conv > 1
        Linear = Linear2d if channel_first else nn.Linear

        def cut(tag, value):
            hit = value.endswith(tag)
            return hit, (value[:-len(tag)] if hit else value)

        self.disable_z, forward_type = cut("_noz", forward_type)
        self.disable_z_act, forward_type = cut("_nozact", forward_type)
        if forward_type not in _HIP_FORWARD_TYPES:
            raise NotImplementedError(f"forward_type {forward_type!r}: xfmamba_amd builds the cross2d/oflex core only "
                                      f"({_HIP_FORWARD_TYPES}, optionally with _noz)")
        self.out_norm = (LayerNorm2d if channel_first else nn.LayerNorm)(self.d_inner)
        self._As_pre = None          # A = -exp(A_logs) computed for all blocks of the trunk at once (transient)

        self.in_proj = Linear(self.d_model, self.d_inner if self.disable_z else self.d_inner * 2, bias=bias)
        self.act = act_layer()
        if self.with_dconv:
            self.conv2d = nn.Conv2d(self.d_inner, self.d_inner, kernel_size=d_conv, padding=(d_conv - 1) // 2,
                                    groups=self.d_inner, bias=conv_bias)
        self.x_proj_weight = nn.Parameter(torch.stack(
            [nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False).weight for _ in range(self.k_group)],
            dim=0))                                                                      # (K, R+2N, D)
        self.out_act = nn.Identity()
        self.out_proj = Linear(self.d_inner, self.d_model, bias=bias)
        self.dropout = nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()
        if initialize != "v0":
            raise NotImplementedError("only the 'v0' S6 initialiser is built")
        self.A_logs, self.Ds, self.dt_projs_weight, self.dt_projs_bias = mamba_init.init_dt_A_D(
            self.d_state, self.dt_rank, self.d_inner, dt_scale, dt_init, dt_min, dt_max, dt_init_floor,
            k_group=self.k_group)

    def forward_core(self, x: torch.Tensor):
        B, D, H, W = x.shape
        y, _ = _ss2d_core(x, self.x_proj_weight, self.dt_projs_weight, self.A_logs, self.Ds, self.dt_projs_bias,
                          As=self._As_pre)
        y = y.view(B, -1, H, W)
        if self.channel_first:
            return self.out_norm(y, out_dtype=x.dtype)          # LayerNorm2d kernel emits x's dtype directly
        return self.out_norm(y.permute(0, 2, 3, 1)).to(x.dtype)

    def forward(self, x: torch.Tensor, **kwargs):
        x = self.in_proj(x)
        z = None
        if not self.disable_z:
            x, z = x.chunk(2, dim=(1 if self.channel_first else -1))
            if not self.disable_z_act:
                z = self.act(z)
        if not self.channel_first:
            x = x.permute(0, 3, 1, 2).contiguous()
        x = _dwconv_act(self.conv2d, self.act, x) if self.with_dconv else self.act(x)
        y = self.out_act(self.forward_core(x))
        if z is not None:
            y = y * z
        return self.dropout(self.out_proj(y))

    def forward_tokens(self, h: torch.Tensor):
        """``forward`` for a token-major (B, H, W, C) input / output (channel_first blocks only): in_proj writes the
        (B, D, L) planes the scan path wants, out_proj reads planes and writes tokens -- layout changes ride on the
        GEMMs' operand flags."""
        B, H, W, C = h.shape
        L = H * W
        D, N, R = self.d_inner, self.d_state, self.dt_rank
        if (TOKEN_SS2D and self.disable_z and SS2D_MODE == "fused" and self.with_dconv and isinstance(self.act, nn.SiLU)
                and isinstance(self.out_act, nn.Identity) and isinstance(self.out_norm, LayerNorm2d) and h.dtype == torch.bfloat16
                and not _fp8.ENABLED and self.conv2d.kernel_size == (3, 3) and self.conv2d.padding == (1, 1)
                and self.conv2d.groups == D and rows_supported(D) and chan_supported(h, H, W, N, 4, D, R)
                and ytokens_supported(H, W, N) and dwconv_tokens_supported(h.new_empty((1, H, W, D)))):
            # short maps (14 x 14, 7 x 7): the whole block stays TOKEN-MAJOR (round 5) -- in_proj, x_proj, out_proj and their data
            # gradients are plain token GEMMs, every weight gradient is tokens x tokens, the depthwise convolution walks
            # (B, H, W, D) maps (csrc/dwconv_tok.hip), the scan reads x / writes y token-major, out_norm is the row LayerNorm
            xt = linear_tokens_fn(h.view(B, L, C), self.in_proj.weight, self.in_proj.bias)               # (B, L, D)
            xt = dwconv3x3_silu_tokens_fn(xt.view(B, H, W, D), self.conv2d.weight, self.conv2d.bias)
            As = self._As_pre if self._As_pre is not None else -self.A_logs.float().exp()
            yt = ss2d_chan_fn(xt.view(B, L, D), self.x_proj_weight, self.dt_projs_weight, As, self.Ds.float(),
                              self.dt_projs_bias.reshape(-1).float(), H, W, y_tokens=True, x_tokens=True)  # (B, L, D) fp32
            yn = layernorm_rows_fn(yt.view(B, H, W, D), self.out_norm.weight, self.out_norm.bias, self.out_norm.eps, h.dtype)
            return self.dropout(linear_tokens_fn(yn, self.out_proj.weight, self.out_proj.bias))
        x = batched_proj(h.view(B, L, C), self.in_proj.weight, self.in_proj.bias, in_tokens=True, out_tokens=False)
        z = None
        if not self.disable_z:
            x, z = x.chunk(2, dim=1)
            if not self.disable_z_act:
                z = self.act(z)
        x = x.reshape(B, -1, H, W)
        x = _dwconv_act(self.conv2d, self.act, x) if self.with_dconv else self.act(x)
        D, N, R = x.shape[1], self.d_state, self.dt_rank
        if (z is None and SS2D_MODE == "fused" and isinstance(self.out_act, nn.Identity) and isinstance(self.out_norm, LayerNorm2d)
                and x.dtype == torch.bfloat16 and not _fp8.ENABLED and rows_supported(D) and chan_supported(x, H, W, N, 4, D, R)
                and ytokens_supported(H, W, N)):
            # short maps (14 x 14, 7 x 7): the scan writes y TOKEN-MAJOR, so out_norm (the reference's channel-last nn.LayerNorm,
            # models/fusion_vmamba.py:1186-1188) is the row LayerNorm and out_proj (:1205) a plain token GEMM -- no LayerNorm2d
            # pass over planes, no layout-changing projection, token x token weight gradients (round 5)
            As = self._As_pre if self._As_pre is not None else -self.A_logs.float().exp()
            yt = ss2d_chan_fn(x.reshape(B, D, L), self.x_proj_weight, self.dt_projs_weight, As, self.Ds.float(),
                              self.dt_projs_bias.reshape(-1).float(), H, W, y_tokens=True)                   # (B, L, D) fp32
            yn = layernorm_rows_fn(yt.view(B, H, W, D), self.out_norm.weight, self.out_norm.bias, self.out_norm.eps, x.dtype)
            return self.dropout(linear_tokens_fn(yn, self.out_proj.weight, self.out_proj.bias))
        y = self.out_act(self.forward_core(x)).view(B, -1, L)
        if z is not None:
            y = y * z
        if self.out_proj.bias is None and _fp8.usable(y, y.shape[1], self.out_proj.weight.shape[0]):
            out = _fp8.fp8_planes_linear(y, self.out_proj.weight)       # BASELINE configs[4]: fp8 weights, fp8 MFMA
        else:
            out = batched_proj(y, self.out_proj.weight, self.out_proj.bias, in_tokens=False, out_tokens=True)
        return self.dropout(out.view(B, H, W, C))


def _tokens_dtype(ref: torch.Tensor):
    """Dtype the token-major path computes its GEMM operands in (autocast dtype, else the weights' dtype), or None
    when the row kernels do not emit it (fp16): the caller then stays on the NCHW modules."""
    d = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else ref.dtype
    return d if d in (torch.float32, torch.bfloat16) else None


def _norm_tokens(norm: nn.LayerNorm, x, pend):
    """LayerNorm of the token-major stream, folding in a pending ``x += scale * y``.  Returns (x, normalised)."""
    out_dtype = _tokens_dtype(norm.weight)
    if pend is None:
        if x.is_cuda and x.dtype == torch.float32 and x.requires_grad and torch.is_grad_enabled():
            # (the stream goes on to the next add + LayerNorm kernel: hand it through this node, see LayerNormRowsPassHip)
            return layernorm_rows_pass_fn(x, norm.weight, norm.bias, norm.eps, out_dtype)
        return x, layernorm_rows_fn(x, norm.weight, norm.bias, norm.eps, out_dtype)
    return add_layernorm_rows_fn(x, pend[0], pend[1], norm.weight, norm.bias, norm.eps, out_dtype, pend[2])


def _settle(x, pend, out_dtype=None):
    """Apply a pending ``x += scale * y`` with nothing to fuse it into (end of a stage).  ``out_dtype``: what the only
    consumer reads (the downsample convolution under autocast: bf16) -- the HIP kernel then emits that directly."""
    if pend is None:
        return x if out_dtype is None else x.to(out_dtype)
    y, s, yb = pend
    if (x.is_cuda and x.dtype == torch.float32 and y.dtype in (torch.float32, torch.bfloat16) and y.shape == x.shape
            and x.shape[-1] % 8 == 0):
        return residual_settle_fn(x, y, s, yb, out_dtype)
    if yb is not None:
        y = y + yb.to(y.dtype)
    if s is not None:
        y = y * s.view(-1, *([1] * (y.ndim - 1))).to(y.dtype)
    x = x + y
    return x if out_dtype is None else x.to(out_dtype)


class VSSBlock(nn.Module):
    def __init__(self, hidden_dim: int = 0, drop_path: float = 0, norm_layer: nn.Module = nn.LayerNorm,
                 channel_first=False, ssm_d_state: int = 16, ssm_ratio=2.0, ssm_dt_rank: Any = "auto",
                 ssm_act_layer=nn.SiLU, ssm_conv: int = 3, ssm_conv_bias=True, ssm_drop_rate: float = 0,
                 ssm_init="v0", forward_type="v0", mlp_ratio=4.0, mlp_act_layer=nn.GELU, mlp_drop_rate: float = 0.0,
                 gmlp=False, use_checkpoint: bool = False, post_norm: bool = False, **kwargs):
        super().__init__()
        if gmlp or use_checkpoint:
            raise NotImplementedError("gMlp / activation checkpointing are outside the XFMamba hot path")
        self.ssm_branch = ssm_ratio > 0
        self.mlp_branch = mlp_ratio > 0
        self.post_norm = post_norm
        if self.ssm_branch:
            self.norm = norm_layer(hidden_dim)
            if isinstance(self.norm, LayerNorm2d):
                self.norm.cast_out = not post_norm               # feeds in_proj (a GEMM)
            self.op = SS2Dv2(d_model=hidden_dim, d_state=ssm_d_state, ssm_ratio=ssm_ratio, dt_rank=ssm_dt_rank,
                             act_layer=ssm_act_layer, d_conv=ssm_conv, conv_bias=ssm_conv_bias, dropout=ssm_drop_rate,
                             initialize=ssm_init, forward_type=forward_type, channel_first=channel_first)
        self.drop_path = DropPath(drop_path)
        if self.mlp_branch:
            self.norm2 = norm_layer(hidden_dim)
            if isinstance(self.norm2, LayerNorm2d):
                self.norm2.cast_out = not post_norm              # feeds mlp.fc1 (a GEMM)
            self.mlp = Mlp(in_features=hidden_dim, hidden_features=int(hidden_dim * mlp_ratio),
                           act_layer=mlp_act_layer, drop=mlp_drop_rate, channels_first=channel_first)

    def forward(self, x: torch.Tensor):
        if self.ssm_branch:
            x = x + self.drop_path(self.norm(self.op(x)) if self.post_norm else self.op(self.norm(x)))
        if self.mlp_branch:
            x = x + self.drop_path(self.norm2(self.mlp(x)) if self.post_norm else self.mlp(self.norm2(x)))
        return x

    def tokens_ok(self) -> bool:
        """Can this block run on the token-major stream (pre-norm LayerNorm2d blocks of a supported width)?"""
        norms = [m for m in (getattr(self, "norm", None), getattr(self, "norm2", None)) if m is not None]
        return (not self.post_norm and all(isinstance(m, LayerNorm2d) and rows_supported(m.normalized_shape[0])
                                           for m in norms)
                and (not self.ssm_branch or self.op.channel_first))

    def forward_tokens(self, x: torch.Tensor, pend=None, defer_bias=True):
        """x: (B, H, W, C) fp32 residual stream; ``pend`` = (y, scale, y_bias) is a branch output not yet added to it.
        Every ``x + drop_path(branch)`` is deferred into the LayerNorm kernel that reads the sum next."""
        B = x.shape[0]
        if self.ssm_branch:
            x, h = _norm_tokens(self.norm, x, pend)
            pend = (self.op.forward_tokens(h), self.drop_path.sample_scale(B, x.device), None)
        if self.mlp_branch:
            x, h = _norm_tokens(self.norm2, x, pend)
            # (defer fc2's bias into the next add + LayerNorm kernel -- unless nothing follows in this stage: the
            #  plain settle would leave its gradient to a framework reduction)
            y, yb = self.mlp.forward_tokens_deferred(h, defer=defer_bias)
            pend = (y, self.drop_path.sample_scale(B, x.device), yb)
        return x, pend


class _NegExpAll(torch.autograd.Function):
    """``A_i = -exp(A_log_i)`` for a list of tensors with two multi-tensor kernels (and one for the backward) instead
    of an exp, a neg and their two backward kernels per SS2D block (models/fusion_vmamba.py:1161)."""

    @staticmethod
    def forward(ctx, *logs):
        outs = torch._foreach_neg(torch._foreach_exp([t.float() for t in logs]))
        ctx.save_for_backward(*outs)
        ctx.dtypes = [t.dtype for t in logs]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        outs = ctx.saved_tensors
        idx = [i for i, g in enumerate(grads) if g is not None]
        res = [None] * len(grads)
        if idx:
            prod = torch._foreach_mul([grads[i] for i in idx], [outs[i] for i in idx])      # dA_log = dA * A
            for i, p in zip(idx, prod):
                res[i] = p.to(ctx.dtypes[i])
        return tuple(res)


class _DropPathBank:
    """Context: sample the stochastic-depth factors of every ``DropPath`` under ``root`` for one forward pass with ONE
    uniform draw (a (layers, B) matrix compared with the keep probabilities) instead of a dropout kernel per layer (26
    launches in XFMamba-T).  Same distribution per layer and sample: Bernoulli(keep) / keep, independent across both."""

    USES = 2

    def __init__(self, root: nn.Module, batch: int, device):
        mods = root.__dict__.get("_droppath_mods")
        if mods is None:
            mods = [m for m in root.modules() if isinstance(m, DropPath)]
            root.__dict__["_droppath_mods"] = mods
        self.mods = [m for m in mods if m.training and 0.0 < m.drop_prob < 1.0 and m.scale_by_keep]
        self.root, self.batch, self.device = root, batch, device

    def __enter__(self):
        if len(self.mods) > 1 and self.device.type == "cuda":
            key = (tuple(m.drop_prob for m in self.mods), str(self.device))
            keep = self.root.__dict__.get("_droppath_keep")          # (layers, 1) keep probabilities, uploaded once
            if keep is None or keep[0] != key:
                if torch.cuda.is_current_stream_capturing():
                    return self                                      # (no upload inside a capture: per-layer kernels)
                keep = (key, torch.tensor([1.0 - p for p in key[0]], dtype=torch.float32).to(self.device)[:, None])
                self.root.__dict__["_droppath_keep"] = keep
            # a VSSBlock applies its DropPath twice per pass (SS2D branch and Mlp branch): USES independent rows per module
            n = len(self.mods)
            u = torch.rand(self.USES, n, self.batch, dtype=torch.float32, device=self.device)
            bank = (u < keep[1]).to(torch.float32) / keep[1]
            for i, m in enumerate(self.mods):
                m._preset = [bank[j, i] for j in range(self.USES)]
        return self

    def __exit__(self, *exc):
        for m in self.mods:
            m._preset = None
        return False


class _PrecomputedA:
    """Context: hand every SS2Dv2 of ``root`` its ``A = -exp(A_logs)`` from one batched evaluation."""

    def __init__(self, root: nn.Module, split_after=None):
        """``split_after``: index of the child of ``root`` (a stage) after which the backward pass will be cut
        (``Backbone_VSSM.cut_after``): the blocks on either side get their own batched evaluation, so that no autograd
        node spans the cut (dp.PhasedGrads)."""
        mods = root.__dict__.get("_ss2d_mods")           # (cached on the container: the module tree is static)
        if mods is None:
            mods = [[m for m in child.modules() if isinstance(m, SS2Dv2)] for child in root.children()]
            root.__dict__["_ss2d_mods"] = mods
        if split_after is None:
            self.groups = [[m for g in mods for m in g]]
        else:
            self.groups = [[m for g in mods[:split_after + 1] for m in g], [m for g in mods[split_after + 1:] for m in g]]
        self.mods = [m for g in self.groups for m in g]

    def __enter__(self):
        for grp in self.groups:
            if grp and grp[0].A_logs.is_cuda:
                for m, a in zip(grp, _NegExpAll.apply(*[m.A_logs for m in grp])):
                    m._As_pre = a
        return self

    def __exit__(self, *exc):
        for m in self.mods:
            m._As_pre = None
        return False


def _blocks_tokens_ok(blocks) -> bool:
    return len(blocks) > 0 and all(isinstance(b, VSSBlock) and b.tokens_ok() for b in blocks)


def _blocks_tokens(blocks, t, out_dtype=None):
    """VSSBlocks of a stage on the token-major stream; the last residual add is settled here (in ``out_dtype`` when the
    stage output has a single reduced-precision reader)."""
    pend = None
    for i, b in enumerate(blocks):
        t, pend = b.forward_tokens(t, pend, defer_bias=True)
    return _settle(t, pend, out_dtype)


def _run_blocks(blocks: nn.Sequential, x: torch.Tensor):
    """A stage's VSSBlocks on an NCHW map; internally on the token-major stream when ``STREAM_LAYOUT == "tokens"``."""
    if (STREAM_LAYOUT != "tokens" or not x.is_cuda or not _blocks_tokens_ok(blocks)
            or _tokens_dtype(next(blocks.parameters())) is None):
        return blocks(x)
    t = x.permute(0, 2, 3, 1).float().contiguous()
    return _blocks_tokens(blocks, t).permute(0, 3, 1, 2).contiguous()


# XFM_LN_GELU=0: the patch embedding's norm -> GELU as two kernels each way (A/B switch, read once)
_LN_GELU = os.environ.get("XFM_LN_GELU", "1") == "1"

# XFM_CONV_CL=0: the strided convolutions through F.conv2d (A/B switch, read once)
_CONV_CL = os.environ.get("XFM_CONV_CL", "1") == "1"


class _ConvChannelsLast(torch.autograd.Function):
    """A bias-free ``F.conv2d`` on a channels_last map with the weight's layout copies taken out.  Through ``F.conv2d`` under
    autocast every strided convolution of the trunk pays five weight-sized copies per step (autocast's cast, the
    channels_last copy for MIOpen's NHWC kernels in the forward pass and again in the backward pass, the weight gradient
    back to OIHW, its cast to fp32); here the weight's compute-dtype shadow (``amp.cast_weight``) is made channels_last once,
    kept for the backward pass, and the weight gradient goes from MIOpen's layout to the parameter's in ONE cast + permute."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding, dilation, groups):
        cd = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else weight.dtype
        ctx.x_dtype, ctx.w_dtype = x.dtype, weight.dtype
        x = x if x.dtype == cd else x.to(cd)
        w = cast_weight(weight, cd).contiguous(memory_format=torch.channels_last)
        ctx.conv = (tuple(stride), tuple(padding), tuple(dilation), groups)
        ctx.weight = weight
        with torch.autocast("cuda", enabled=False):
            y = torch.ops.aten.convolution(x, w, None, ctx.conv[0], ctx.conv[1], ctx.conv[2], False, [0, 0], groups)
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, padding, dilation, groups = ctx.conv
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        gw, need_w = None, ctx.needs_input_grad[1]
        if need_w and stride == (2, 2) and padding == (1, 1) and dilation == (1, 1) and groups == 1 and w.shape[2:] == (3, 3):
            # the weight gradient on this repository's token x token kernel, straight from the channels_last map (no rows
            # written): 38 / 33 us against the library's 66 / 65 at the two large maps of the trunk
            xt, gt = x.permute(0, 2, 3, 1), gy.permute(0, 2, 3, 1)
            if xt.is_contiguous():
                gw = conv3x3s2_wgrad_from_map(gt if gt.is_contiguous() else gt.contiguous(), xt, ctx.weight)
                if gw is not None:
                    need_w = False
                    if gw.dtype != ctx.w_dtype:
                        gw = gw.to(ctx.w_dtype)
        gx, gwl, _ = torch.ops.aten.convolution_backward(gy, x, w, None, stride, padding, dilation, False, [0, 0], groups,
                                                         [ctx.needs_input_grad[0], need_w, False])
        if gwl is not None:
            gw = gwl.to(dtype=ctx.w_dtype, memory_format=torch.contiguous_format)
        if gx is not None and gx.dtype != ctx.x_dtype:
            gx = gx.to(ctx.x_dtype)
        return gx, gw, None, None, None, None


def _conv_ln_tokens(conv: nn.Conv2d, norm: nn.Module, t: torch.Tensor, out_dtype=None) -> torch.Tensor:
    """``norm(conv(t))`` on a token-major (B, H, W, C) tensor -> (B, H', W', C').  The convolution is handed to
    MIOpen as a channels_last map (its implicit-GEMM kernels are NHWC-native, so no NCHW<->NHWC transposes run around
    them) WITHOUT its bias: the bias is added inside the LayerNorm kernel, whose backward pass also returns its
    gradient -- no per-channel reduction over the convolution output is left to the framework."""
    fused = isinstance(norm, LayerNorm2d)
    if t.dtype != conv.weight.dtype and not torch.is_autocast_enabled():
        t = t.to(conv.weight.dtype)                      # fp32 residual stream into a reduced-precision model
    if fused and conv3x3s2_tokens_supported(t, conv):
        # the library's own MFMA path (csrc/conv_tok.hip): every 3 x 3 stride-2 convolution with at least 8 input channels
        y = conv3x3s2_tokens_fn(t, conv.weight)
    else:
        if _CONV_CL and fused and t.is_cuda and conv.padding_mode == "zeros" and not isinstance(conv.padding, str):
            y = _ConvChannelsLast.apply(t.permute(0, 3, 1, 2), conv.weight, conv.stride, conv.padding, conv.dilation, conv.groups)
        else:
            y = F.conv2d(t.permute(0, 3, 1, 2), conv.weight, None if fused else conv.bias, conv.stride, conv.padding,
                         conv.dilation, conv.groups)
        y = y.permute(0, 2, 3, 1)
        y = y if y.is_contiguous() else y.contiguous()
    if not fused:
        return y if out_dtype is None else y.to(out_dtype)
    if y.dtype not in (torch.float32, torch.bfloat16):
        y = y.float()
    return layernorm_rows_fn(y, norm.weight, norm.bias, norm.eps, out_dtype, conv.bias)


def _ln_tokens(norm: nn.Module, t: torch.Tensor, out_dtype=None) -> torch.Tensor:
    if isinstance(norm, nn.Identity):
        return t if out_dtype is None else t.to(out_dtype)
    return layernorm_rows_fn(t, norm.weight, norm.bias, norm.eps, out_dtype)


def _norm_tokens_ok(norm: nn.Module) -> bool:
    return isinstance(norm, nn.Identity) or (isinstance(norm, LayerNorm2d) and rows_supported(norm.normalized_shape[0]))


class VSSM(nn.Module):
    """VMamba trunk in the one configuration XFMamba uses (patch-embed v2, downsample v3)."""

    # data-parallel runs with a captured step cut the backward pass after stage ``cut_after`` (dp.PhasedGrads): the
    # forward pass then leaves the activation entering the next stage in ``cut_tensor``.  None: no cut.
    cut_after = None
    cut_tensor = None

    def __init__(self, patch_size=4, in_chans=3, num_classes=2, depths=[2, 2, 9, 2], dims=[96, 192, 384, 768],
                 ssm_d_state=1, ssm_ratio=2.0, ssm_dt_rank="auto", ssm_act_layer="silu", ssm_conv=3,
                 ssm_conv_bias=False, ssm_drop_rate=0.0, ssm_init="v0", forward_type="v0",
                 mlp_ratio=4.0, mlp_act_layer="gelu", mlp_drop_rate=0.0, gmlp=False,
                 drop_path_rate=0.2, patch_norm=True, norm_layer="LN", downsample_version: str = "v3",
                 patchembed_version: str = "v2", use_checkpoint=False, posembed=False, imgsize=224, **kwargs):
        super().__init__()
        self.channel_first = norm_layer.lower() in ("bn", "ln2d")
        self.num_classes = num_classes
        self.num_layers = len(depths)
        if isinstance(dims, int):
            dims = [int(dims * 2 ** i) for i in range(self.num_layers)]
        self.num_features = dims[-1]
        self.dims = dims
        if patchembed_version != "v2" or downsample_version != "v3" or posembed or not self.channel_first:
            raise NotImplementedError("xfmamba_amd builds the XFMamba trunk only: patch-embed v2, downsample v3, "
                                      "channel-first norms, no positional embedding")
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        norm = dict(ln=nn.LayerNorm, ln2d=LayerNorm2d, bn=nn.BatchNorm2d)[norm_layer.lower()]
        acts = dict(silu=nn.SiLU, gelu=nn.GELU, relu=nn.ReLU, sigmoid=nn.Sigmoid)
        ssm_act, mlp_act = acts[ssm_act_layer.lower()], acts[mlp_act_layer.lower()]
        self.pos_embed = None
        self.patch_embed = self._make_patch_embed_v2(in_chans, dims[0], patch_size, patch_norm, norm, True)
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            down = (self._make_downsample_v3(dims[i], dims[i + 1], norm_layer=norm, channel_first=True)
                    if i < self.num_layers - 1 else nn.Identity())
            blocks = [VSSBlock(hidden_dim=dims[i], drop_path=dp, norm_layer=norm, channel_first=True,
                               ssm_d_state=ssm_d_state, ssm_ratio=ssm_ratio, ssm_dt_rank=ssm_dt_rank,
                               ssm_act_layer=ssm_act, ssm_conv=ssm_conv, ssm_conv_bias=ssm_conv_bias,
                               ssm_drop_rate=ssm_drop_rate, ssm_init=ssm_init, forward_type=forward_type,
                               mlp_ratio=mlp_ratio, mlp_act_layer=mlp_act, mlp_drop_rate=mlp_drop_rate, gmlp=gmlp,
                               use_checkpoint=use_checkpoint)
                      for dp in dpr[sum(depths[:i]):sum(depths[:i + 1])]]
            self.layers.append(nn.Sequential(OrderedDict(blocks=nn.Sequential(*blocks), downsample=down)))
        self.classifier = nn.Sequential(OrderedDict(
            norm=norm(self.num_features), permute=nn.Identity(), avgpool=nn.AdaptiveAvgPool2d(1),
            flatten=nn.Flatten(1), head=nn.Linear(self.num_features, num_classes)))
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m: nn.Module):
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    @staticmethod
    def _make_patch_embed_v2(in_chans=3, embed_dim=96, patch_size=4, patch_norm=True, norm_layer=nn.LayerNorm,
                             channel_first=False):
        s = patch_size // 2
        return nn.Sequential(
            nn.Conv2d(in_chans, embed_dim // 2, kernel_size=s + 1, stride=s, padding=1),
            nn.Identity(), (norm_layer(embed_dim // 2) if patch_norm else nn.Identity()), nn.Identity(),
            nn.GELU(),
            nn.Conv2d(embed_dim // 2, embed_dim, kernel_size=s + 1, stride=s, padding=1),
            nn.Identity(), (norm_layer(embed_dim) if patch_norm else nn.Identity()))

    @staticmethod
    def _make_downsample_v3(dim=96, out_dim=192, norm_layer=nn.LayerNorm, channel_first=False):
        return nn.Sequential(nn.Identity(), nn.Conv2d(dim, out_dim, kernel_size=3, stride=2, padding=1),
                             nn.Identity(), norm_layer(out_dim))

    # ---- token-major trunk: patch embedding, stages and downsampling without ever leaving (B, H, W, C) ----------
    def tokens_trunk_ok(self, x: torch.Tensor) -> bool:
        if STREAM_LAYOUT != "tokens" or not x.is_cuda or _tokens_dtype(self.patch_embed[0].weight) is None:
            return False
        pe = self.patch_embed
        ok = (len(pe) == 8 and isinstance(pe[0], nn.Conv2d) and isinstance(pe[5], nn.Conv2d)
              and _norm_tokens_ok(pe[2]) and _norm_tokens_ok(pe[7]))
        for layer in self.layers:
            ok = ok and _blocks_tokens_ok(layer.blocks)
            d = layer.downsample
            ok = ok and (isinstance(d, nn.Identity) or (len(d) == 4 and isinstance(d[1], nn.Conv2d)
                                                         and _norm_tokens_ok(d[3])))
        return ok

    def stem_tokens(self, x: torch.Tensor) -> torch.Tensor:
        """patch_embed (conv s2 -> LN -> GELU -> conv s2 -> LN) -> fp32 tokens (B, H/4, W/4, C0)."""
        pe = self.patch_embed
        act_dtype = _tokens_dtype(pe[0].weight)
        if act_dtype is not None and isinstance(pe[2], LayerNorm2d) and conv3x3s2_gray_supported(x, pe[0]):
            # the image is ONE channel replicated (net_fusionmamba.py: x.expand(-1, 3, -1, -1)): the convolution of that channel
            # with the weight summed over its input channels (csrc/conv_tok.hip: 9 taps, no replicated image, no library)
            y = conv3x3s2_gray_fn(x[:, 0].to(torch.bfloat16).contiguous(), pe[0].weight)
            if _LN_GELU and isinstance(pe[4], nn.GELU) and getattr(pe[4], "approximate", "none") == "none":
                # norm -> GELU as ONE kernel each way (xfm_layernorm_rows_gelu_fwd/_bwd): the activation pass over the
                # 112 x 112 x 48 map and its backward pass (154 + 231 MB per step) are gone
                t = layernorm_rows_gelu_fn(y, pe[2].weight, pe[2].bias, pe[2].eps, act_dtype, pe[0].bias)
            else:
                t = layernorm_rows_fn(y, pe[2].weight, pe[2].bias, pe[2].eps, act_dtype, pe[0].bias)
                t = bias_gelu_fn(t, None) if (isinstance(pe[4], nn.GELU) and getattr(pe[4], "approximate", "none") == "none"
                                              and t.dtype in (torch.float32, torch.bfloat16) and t.shape[-1] % 8 == 0) else pe[4](t)
            return _conv_ln_tokens(pe[5], pe[7], t, torch.float32)
        if x.shape[1] > 1 and x.stride(1) == 0 and act_dtype is not None and not x.requires_grad:
            # the image is a stride-0 broadcast of ONE channel (net_fusionmamba.py: x.expand(-1, 3, -1, -1)): cast the single
            # channel to the convolution's dtype first and replicate it as the LAST step -- 6 + 19 MB moved instead of a 77 MB
            # transposing copy of the broadcast followed by autocast's 58 MB cast
            t = x[:, 0].unsqueeze(-1).to(act_dtype).expand(-1, -1, -1, x.shape[1]).contiguous()
        else:
            t = x.permute(0, 2, 3, 1).contiguous()              # (B, H, W, 3): a channels_last image
        t = _conv_ln_tokens(pe[0], pe[2], t, act_dtype)         # norm output feeds GELU -> conv: the conv's dtype
        if isinstance(pe[4], nn.GELU) and getattr(pe[4], "approximate", "none") == "none" and t.is_cuda \
                and t.dtype in (torch.float32, torch.bfloat16) and t.shape[-1] % 8 == 0:
            t = bias_gelu_fn(t, None)                            # the exact-erf GELU kernel of the Mlp (streaming rate)
        else:
            t = pe[4](t)
        return _conv_ln_tokens(pe[5], pe[7], t, torch.float32)

    def stage_tokens(self, i: int, t: torch.Tensor, need_output: bool = True):
        """Stage i on tokens: returns (stage output, downsampled input of stage i+1 or None).  ``need_output=False``: the
        caller only wants the downsampled stream -- the stage output is then settled straight into the dtype the downsample
        convolution reads (bf16 under autocast) and NOT returned."""
        layer = self.layers[i]
        last = isinstance(layer.downsample, nn.Identity)
        cd = None
        if not need_output and not last and t.is_cuda and torch.is_autocast_enabled():
            cd = torch.get_autocast_gpu_dtype()
            cd = cd if cd == torch.bfloat16 else None
        o = _blocks_tokens(layer.blocks, t, cd)
        if last:
            return o, None
        return (None if cd is not None else o), _conv_ln_tokens(layer.downsample[1], layer.downsample[3], o, torch.float32)

    def forward(self, x: torch.Tensor):
        if self.tokens_trunk_ok(x):
            with _PrecomputedA(self.layers, self.cut_after), _DropPathBank(self.layers, x.shape[0], x.device):
                t = self.stem_tokens(x)
                for i in range(len(self.layers)):
                    o, t = self.stage_tokens(i, t, need_output=i + 1 == len(self.layers))
                    if i == self.cut_after:
                        self.cut_tensor = t              # (the stream entering stage i + 1: dp.PhasedGrads' cut)
            x = o.permute(0, 3, 1, 2)
            return self.classifier(x)
        x = self.patch_embed(x)
        for layer in self.layers:
            x = layer.downsample(_run_blocks(layer.blocks, x))
        return self.classifier(x)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """Accept checkpoints written by older VMamba code (key renames of fusion_vmamba.py:1607-1646)."""
        def rename(src, dst):
            key = prefix + src
            for k in [k for k in state_dict if k.startswith(key)]:
                state_dict[prefix + dst + k[len(key):]] = state_dict.pop(k)

        rename("patch_embed.proj", "patch_embed.0")
        rename("patch_embed.norm", "patch_embed.2")
        for i in range(len(self.layers)):
            for j in range(len(self.layers[i].blocks)):
                rename(f"layers.{i}.blocks.{j}.ln_1", f"layers.{i}.blocks.{j}.norm")
                rename(f"layers.{i}.blocks.{j}.self_attention", f"layers.{i}.blocks.{j}.op")
        if hasattr(self, "classifier"):
            rename("norm", "classifier.norm")
            rename("head", "classifier.head")
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class Backbone_VSSM(VSSM):
    """Feature-pyramid trunk; returns the normed outputs of ``out_indices`` (fusion_vmamba.py:1653-1724)."""

    def __init__(self, depths=[2, 2, 15, 2], dims=96, drop_path_rate=0.3, ssm_ratio=2.0, patch_size=4, in_chans=3,
                 num_classes=1000, ssm_d_state=1, ssm_dt_rank="auto", ssm_act_layer="silu", ssm_conv=3,
                 ssm_conv_bias=False, ssm_drop_rate=0.0, ssm_init="v0", forward_type="v05_noz",
                 mlp_ratio=4.0, mlp_act_layer="gelu", mlp_drop_rate=0.0, gmlp=False, patch_norm=True,
                 downsample_version="v3", patchembed_version="v2", use_checkpoint=False, posembed=False, imgsize=224,
                 out_indices=(0, 1, 2, 3), pretrained=None, norm_layer="ln2d", **kwargs):
        super().__init__(depths=depths, dims=dims, drop_path_rate=drop_path_rate, patch_size=patch_size,
                         in_chans=in_chans, num_classes=num_classes, ssm_d_state=ssm_d_state, ssm_ratio=ssm_ratio,
                         ssm_dt_rank=ssm_dt_rank, ssm_act_layer=ssm_act_layer, ssm_conv=ssm_conv,
                         ssm_conv_bias=ssm_conv_bias, ssm_drop_rate=ssm_drop_rate, ssm_init=ssm_init,
                         forward_type=forward_type, mlp_ratio=mlp_ratio, mlp_act_layer=mlp_act_layer,
                         mlp_drop_rate=mlp_drop_rate, gmlp=gmlp, patch_norm=patch_norm,
                         downsample_version=downsample_version, patchembed_version=patchembed_version,
                         use_checkpoint=use_checkpoint, posembed=posembed, imgsize=imgsize, norm_layer=norm_layer,
                         **kwargs)
        norm = dict(ln=nn.LayerNorm, ln2d=LayerNorm2d, bn=nn.BatchNorm2d)[norm_layer.lower()]
        self.out_indices = out_indices
        for i in out_indices:
            self.add_module(f"outnorm{i}", norm(self.dims[i]))
        del self.classifier
        self.load_pretrained(pretrained)

    def load_pretrained(self, ckpt=None, key="model"):
        if ckpt is None:
            return
        try:
            state = torch.load(open(ckpt, "rb"), map_location=torch.device("cpu"))
            print(f"Successfully load ckpt {ckpt}")
            print(self.load_state_dict(state[key], strict=False))
        except Exception as e:  # same tolerant behaviour as the reference (fusion_vmamba.py:1692-1702)
            print(f"Failed loading checkpoint form {ckpt}: {e}")

    def tokens_path_ok(self, x: torch.Tensor) -> bool:
        return self.tokens_trunk_ok(x) and all(_norm_tokens_ok(getattr(self, f"outnorm{i}")) for i in self.out_indices)

    def forward(self, x, only_last: bool = False, tokens_out: bool = False):
        """``only_last=True`` skips the out-norms whose results ``TwoViewXFMambaTop`` discards
        (outnorm0-2, net_fusionmamba.py:200-201); values of the last output are unchanged.  ``tokens_out`` (with the
        token-major trunk only, see ``tokens_trunk_ok``): the outputs stay (B, H, W, C) -- no NCHW copy."""
        last = len(self.layers) - 1
        if self.tokens_path_ok(x):
            outs = []
            with _PrecomputedA(self.layers, self.cut_after), _DropPathBank(self.layers, x.shape[0], x.device):
                t = self.stem_tokens(x)
                for i in range(len(self.layers)):
                    wanted = (i in self.out_indices and (not only_last or i == last)) or (i == last and not self.out_indices)
                    o, t = self.stage_tokens(i, t, need_output=wanted)
                    if i == self.cut_after:
                        self.cut_tensor = t              # (the stream entering stage i + 1: dp.PhasedGrads' cut)
                    if i in self.out_indices and (not only_last or i == last):
                        on = _ln_tokens(getattr(self, f"outnorm{i}"), o, torch.float32)
                        outs.append(on if tokens_out else on.permute(0, 3, 1, 2).contiguous())
            return outs if len(self.out_indices) else (o if tokens_out else o.permute(0, 3, 1, 2).contiguous())
        if tokens_out:
            raise RuntimeError("tokens_out needs the token-major trunk (tokens_trunk_ok)")
        x = self.patch_embed(x)
        outs = []
        for i, layer in enumerate(self.layers):
            o = _run_blocks(layer.blocks, x)
            x = layer.downsample(o)
            if i == self.cut_after:
                self.cut_tensor = x
            if i in self.out_indices and (not only_last or i == last):
                outs.append(getattr(self, f"outnorm{i}")(o).contiguous())
        if len(self.out_indices) == 0:
            return x
        return outs


def _linear_rows(lin: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """``lin(x)`` for token-major ``x`` on the path of the trunk's linear layers (``linear_tokens_fn``: the weight's bf16
    shadow instead of a cast per call, the weight gradient accumulated into the optimizer's fp32 arena): under autocast
    ``x`` is cast once here, as autocast would inside ``F.linear``."""
    if x.is_cuda and torch.is_autocast_enabled() and x.dtype == torch.float32:
        x = x.to(torch.get_autocast_dtype("cuda"))
    return linear_tokens_fn(x if x.is_contiguous() else x.contiguous(), lin.weight, lin.bias)


class _ViewsAvgStack(torch.autograd.Function):
    """n (2B, ...) fp32 = [view 1 | view 2] -> (3, 2B/2 ...) flattened [view 1 | view 2 | their mean] in ``out_dtype``
    (``xfm_views_avg_stack_fwd/_bwd``)."""

    @staticmethod
    def forward(ctx, n, out_dtype):
        from . import _lib
        M = n.numel() // 2
        out = torch.empty(3 * M, dtype=out_dtype, device=n.device)
        with torch.cuda.device(n.device), _lib.timed("views_avg_stack", M * (8 + 3 * out.element_size())):
            _lib.check(_lib.lib().xfm_views_avg_stack_fwd(n.data_ptr(), out.data_ptr(), M, _lib.dtype_code(out_dtype),
                                                          _lib.stream_ptr()), "views_avg_stack_fwd")
        ctx.shape = n.shape
        return out

    @staticmethod
    def backward(ctx, g):
        from . import _lib
        g = g.contiguous()
        if g.dtype not in (torch.float32, torch.bfloat16):
            g = g.float()
        M = g.numel() // 3
        dn = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device), _lib.timed("views_avg_stack", M * (8 + 3 * g.element_size())):
            _lib.check(_lib.lib().xfm_views_avg_stack_bwd(g.data_ptr(), dn.data_ptr(), M, _lib.dtype_code(g.dtype),
                                                          _lib.stream_ptr()), "views_avg_stack_bwd")
        return dn, None


# XFM_BN_TOKENS=0: BatchNorm of the shallow fusion block through F.batch_norm per view (A/B switch, read once)
_BN_TOKENS = os.environ.get("XFM_BN_TOKENS", "1") == "1"


class _BatchNormViews(torch.autograd.Function):
    """Training-mode ``nn.BatchNorm2d`` applied to V views one after the other (fusion_vmamba.py:906-907), on their token
    matrices ``x`` (V, N, C) fp32, through ``xfm_bn_tokens_fwd/_bwd``: batch statistics per view, the running statistics
    updated view after view, the output in ``out_dtype`` (the following GEMM's)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, out_dtype):
        from . import _lib
        V, N, C = x.shape
        lib = _lib.lib()
        y = torch.empty((V, N, C), dtype=out_dtype, device=x.device)
        mean = torch.empty((V, C), dtype=torch.float32, device=x.device)
        rstd = torch.empty((V, C), dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.xfm_bn_tokens_ws_floats(V, N, C), dtype=torch.float32, device=x.device)
        w = weight.float().contiguous()
        b = bias.float().contiguous()
        with torch.cuda.device(x.device), _lib.timed("bn_tokens_fwd", x.numel() * (8 + y.element_size())):
            _lib.check(lib.xfm_bn_tokens_fwd(x.data_ptr(), w.data_ptr(), b.data_ptr(), _lib.ptr(running_mean), _lib.ptr(running_var),
                                             float(momentum), float(eps), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                             ws.data_ptr(), V, N, C, _lib.dtype_code(out_dtype), _lib.stream_ptr()), "bn_tokens_fwd")
        ctx.save_for_backward(x, w, mean, rstd)
        ctx.dtypes = (weight.dtype, bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        x, w, mean, rstd = ctx.saved_tensors
        V, N, C = x.shape
        lib = _lib.lib()
        dy = dy.contiguous()
        if dy.dtype not in (torch.float32, torch.bfloat16):
            dy = dy.float()
        dx = torch.empty_like(x)
        dw = torch.empty(C, dtype=torch.float32, device=x.device)
        db = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = torch.empty(lib.xfm_bn_tokens_ws_floats(V, N, C), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device), _lib.timed("bn_tokens_bwd", x.numel() * (12 + 2 * dy.element_size())):
            _lib.check(lib.xfm_bn_tokens_bwd(x.data_ptr(), dy.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                                             dw.data_ptr(), db.data_ptr(), ws.data_ptr(), V, N, C, _lib.dtype_code(dy.dtype),
                                             _lib.stream_ptr()), "bn_tokens_bwd")
        return dx, dw.to(ctx.dtypes[0]), db.to(ctx.dtypes[1]), None, None, None, None, None


def _bn_views_ok(bn: nn.BatchNorm2d, x: torch.Tensor) -> bool:
    """The fused two-view form covers: training mode with tracked running statistics, fp32 token matrices on the GPU."""
    if not (_BN_TOKENS and bn.training and bn.track_running_stats and bn.momentum is not None and bn.affine and x.is_cuda
            and x.dtype == torch.float32 and bn.weight.dtype == torch.float32 and x.is_contiguous() and x.data_ptr() % 16 == 0):
        return False
    from . import _lib
    V, N, C = x.shape
    return bool(_lib.lib().xfm_bn_tokens_supported(V, N, C))


def _bn_rows(bn: nn.BatchNorm2d, rows: torch.Tensor) -> torch.Tensor:
    """``bn`` applied to the (B H W, C) token matrix of an NCHW map: what ``nn.BatchNorm2d.forward`` does (batch statistics
    in training mode, running statistics updated with ``momentum``, the step counter), on the 2-D view."""
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    use_batch = bn.training or (bn.running_mean is None and bn.running_var is None)
    keep = not bn.training or bn.track_running_stats
    return F.batch_norm(rows, bn.running_mean if keep else None, bn.running_var if keep else None, bn.weight, bn.bias,
                        use_batch, bn.momentum, bn.eps)


# ---------------------------------------------------------------------------------------------
# shallow fusion: channel-swapping SS2D between the two views (fusion_vmamba.py:693-920)
# ---------------------------------------------------------------------------------------------
class ShallowFuse_SS2Dv4(nn.Module):
    def __init__(self, d_model=96, d_state=16, ssm_ratio=2.0, dt_rank="auto", act_layer=nn.SiLU, d_conv=3,
                 conv_bias=True, dropout=0.0, bias=False, dt_min=0.001, dt_max=0.1, dt_init="random", dt_scale=1.0,
                 dt_init_floor=1e-4, channel_first=False, **kwargs):
        super().__init__()
        if channel_first:
            raise NotImplementedError("the reference only instantiates the channel-last variant")
        self.k_group = 2
        self.d_model = int(d_model)
        self.d_state = int(d_state)
        self.d_inner = int(ssm_ratio * d_model)
        self.dt_rank = int(math.ceil(self.d_model / 16) if dt_rank == "auto" else dt_rank)
        self.channel_first = channel_first
        self.with_dconv = d_conv > 1
        self.in_proj = nn.Linear(self.d_model, self.d_inner, bias=bias)
        self.act = act_layer()
        if self.with_dconv:
            self.conv2d = nn.Conv2d(self.d_inner, self.d_inner, kernel_size=d_conv, padding=(d_conv - 1) // 2,
                                    groups=self.d_inner, bias=conv_bias)
        self.x_proj_weight = nn.Parameter(torch.stack(
            [nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False).weight for _ in range(self.k_group)],
            dim=0))
        self.out_norm = nn.LayerNorm(self.d_inner)
        self.oact = False
        self.out_act = nn.Identity()
        self.out_proj = nn.Linear(self.d_inner, self.d_model, bias=bias)
        self.dropout = nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()
        self.A_logs, self.Ds, self.dt_projs_weight, self.dt_projs_bias = mamba_init.init_dt_A_D(
            self.d_state, self.dt_rank, self.d_inner, dt_scale, dt_init, dt_min, dt_max, dt_init_floor,
            k_group=self.k_group)
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Sequential(nn.Linear(self.d_inner, self.d_inner // 16, bias=False), nn.SiLU(inplace=True),
                                 nn.Linear(self.d_inner // 16, self.d_inner, bias=False), nn.Sigmoid())

    def forward_corev2(self, x: torch.Tensor, x2: torch.Tensor):
        B, D, H, W = x.shape
        L = H * W
        K, R, N = self.k_group, self.dt_rank, self.d_state
        xs = SwappingScan_multiview.apply(x, x2)                                         # (B, 2, D, L)
        x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, self.x_proj_weight.to(xs.dtype))
        dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
        dts = torch.einsum("bkrl,kdr->bkdl", dts, self.dt_projs_weight.to(xs.dtype))
        ys = selective_scan_fn(xs.view(B, -1, L), dts.contiguous().view(B, -1, L), -self.A_logs.float().exp(),
                               Bs.contiguous(), Cs.contiguous(), self.Ds.float(),
                               self.dt_projs_bias.reshape(-1).float(), True, True, None).view(B, K, -1, L)
        y, y2 = SwappingMerge_multiview.apply(ys)
        yy = self.out_norm(torch.cat([y, y2], dim=0).transpose(1, 2).reshape(2 * B, H, W, -1)).to(x.dtype)
        y, y2 = yy.chunk(2, dim=0)                      # (chunk, not slices: its backward is one cat, not zeros + adds)
        return y, y2

    def stacked_ok(self, n: torch.Tensor) -> bool:
        return (n.is_cuda and isinstance(self.out_act, nn.Identity) and self.out_proj.bias is None
                and isinstance(self.out_norm, nn.LayerNorm) and self.out_norm.elementwise_affine)

    def forward_stacked(self, n: torch.Tensor) -> torch.Tensor:
        """Both views as ONE token-major batch ``n`` = [view 1 | view 2] (2B, H, W, C) -> (B, 2, H, W, C): ``[:, 0]`` is the
        reference's first output, ``[:, 1]`` its second (fusion_vmamba.py:853-876).  Same operator chain as ``forward``
        with the layout copies taken out: the scan output (B, 2, D, L) IS a plane-major batch of 2B maps in (sample, view)
        order, so out_norm runs on it as LayerNorm2d (no merge copies, no transposes), the squeeze gates are re-ordered
        instead of the maps (a (2B, D) tensor), and out_proj reads planes and writes tokens."""
        B2, H, W, _ = n.shape
        B, L = B2 // 2, H * W
        K, R, N = self.k_group, self.dt_rank, self.d_state
        xp = _linear_rows(self.in_proj, n)                                                   # (2B, H, W, D)
        D = xp.shape[-1]
        xp, pooled = tokens_to_planes_pooled(xp.view(B2, L, D))                                # planes + the squeeze pooling (:866)
        xp = xp.view(B2, D, H, W)
        xc = _dwconv_act(self.conv2d, self.act, xp) if self.with_dconv else self.act(xp)
        from .ss2d_chan import chan_supported, ss2d_chan_swap_fn, swap_views_stacked
        f1 = self.fc1
        if len(f1) == 4 and isinstance(f1[1], nn.SiLU) and isinstance(f1[3], nn.Sigmoid):
            gate = torch.sigmoid(_linear_rows(f1[2], F.silu(_linear_rows(f1[0], pooled))))
        else:
            gate = f1(pooled)                                                                # [gate 1 | gate 2]
        if SHALLOW_KERNEL and xc.dtype == torch.bfloat16 and D % 2 == 0 and chan_supported(xc, H, W, N, 1, D, R):
            # the exchange as ONE kernel each way: the two swap routes are 2B single-route samples in
            # [view 1 | view 2] order, x_proj / dt_proj inside the node, the reference's pass-through swap gradient kept
            xs = swap_views_stacked(xc.view(B2, D, L))                                       # (2B, D, L) = [route 0 | route 1]
            ys = ss2d_chan_swap_fn(xs, self.x_proj_weight, self.dt_projs_weight, -self.A_logs.float().exp(), self.Ds.float(),
                                   self.dt_projs_bias.reshape(-1).float(), H, W)             # (2B, D, L) fp32, view order
            yy = layernorm2d_fn(ys.view(B2, D, H, W), self.out_norm.weight, self.out_norm.bias, self.out_norm.eps, xp.dtype)
            # view 1's map is gated by view 2's squeeze and the other way round (:870-871)
            gate = torch.cat([gate[B:], gate[:B]], dim=0)
            o = _linear_rows(self.out_proj, gated_planes_to_tokens(yy.view(B2, D, L), gate))
            return self.dropout(o).view(2, B, H, W, -1).transpose(0, 1)                      # (B, 2, H, W, C) as a view
        xs = SwappingScanStacked.apply(xc)                                                   # (B, 2, D, L)
        x_dbl = torch.matmul(self.x_proj_weight.to(xs.dtype), xs)                            # (B, 2, R + 2N, L)
        dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
        dts = torch.matmul(self.dt_projs_weight.to(xs.dtype), dts)                           # (B, 2, D, L)
        ys = selective_scan_fn(xs.view(B, -1, L), dts.view(B, -1, L), -self.A_logs.float().exp(), Bs.contiguous(),
                               Cs.contiguous(), self.Ds.float(), self.dt_projs_bias.reshape(-1).float(), True, True, None)
        # (slices of ys are what SwappingMerge_multiview returns, and their gradient is its stack)
        yy = layernorm2d_fn(ys.view(B * 2, D, H, W), self.out_norm.weight, self.out_norm.bias, self.out_norm.eps, xp.dtype)
        # view 1's map is gated by view 2's squeeze and the other way round (:870-871): (sample, view) order, swapped
        gate = torch.stack([gate[B:], gate[:B]], dim=1).view(B * 2, D)
        # (a transposing copy + one 3136-row GEMM: 64 per-sample products through batched_proj measured 41 vs 33 us here)
        o = _linear_rows(self.out_proj, gated_planes_to_tokens(yy.view(B * 2, D, L), gate))
        return self.dropout(o).view(B, 2, H, W, -1)

    def forward(self, x: torch.Tensor, x2: torch.Tensor):
        # both views share every weight here: they run as one batch of 2B wherever the reference makes two calls
        # (in_proj, conv, squeeze gate, out_norm, out_proj act per sample, so the values are the same)
        B = x.shape[0]
        if self.stacked_ok(x):
            o = self.forward_stacked(torch.cat([x, x2], dim=0))
            return o[:, 0], o[:, 1]
        xp = self.in_proj(torch.cat([x, x2], dim=0)).permute(0, 3, 1, 2).contiguous()        # (2B, D, H, W)
        xc = _dwconv_act(self.conv2d, self.act, xp) if self.with_dconv else self.act(xp)
        y1, y2 = self.forward_corev2(*xc.chunk(2, dim=0))
        y = self.out_act(torch.cat([y2, y1], dim=0))                                         # [view 2 | view 1]
        d = xp.shape[1]
        gate = self.fc1(self.avg_pool(xp).view(2 * B, d)).view(2 * B, 1, 1, d)               # [gate 1 | gate 2]
        o = self.dropout(self.out_proj(y * gate))       # each view is gated by the OTHER view's squeeze (:870-871)
        o2, o1 = o.chunk(2, dim=0)
        return o1, o2


class ShallowFusionBlock_v4(nn.Module):
    def __init__(self, hidden_dim: int = 0, drop_path: float = 0, norm_layer: Callable[..., nn.Module] = nn.BatchNorm2d,
                 attn_drop_rate: float = 0, d_state: int = 4, dt_rank: Any = "auto", ssm_ratio=2.0, **kwargs):
        super().__init__()
        self.norm = norm_layer(hidden_dim)
        self.shallowfuseSS2D = ShallowFuse_SS2Dv4(d_model=hidden_dim, d_state=d_state, ssm_ratio=ssm_ratio,
                                                  dt_rank=dt_rank, dropout=attn_drop_rate, **kwargs)
        self.drop_path = DropPath(drop_path)

    def stacked_ok(self, xt: torch.Tensor) -> bool:
        bn = self.norm
        return (isinstance(bn, nn.BatchNorm2d) and bn.momentum is not None and bn.affine
                and self.shallowfuseSS2D.stacked_ok(xt))

    def forward_stacked(self, xt: torch.Tensor) -> torch.Tensor:
        """[view 1 | view 2] as one token-major (2B, H, W, C) stream in and out.  BatchNorm2d over an NCHW map is a
        per-column normalisation of its (B H W, C) token matrix: the same module state (running statistics, view 1 then
        view 2, :906-907) through ``F.batch_norm`` on the 2-D view, no NCHW copies."""
        B2, H, W, C = xt.shape
        B = B2 // 2
        x2d = xt.reshape(2, B * H * W, C)
        bn = self.norm
        if _bn_views_ok(bn, x2d):
            if bn.num_batches_tracked is not None:
                bn.num_batches_tracked.add_(2)                                               # (one step per view)
            od = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else xt.dtype
            od = od if od in (torch.float32, torch.bfloat16) else xt.dtype
            n = _BatchNormViews.apply(x2d, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, od)
            n = n.view(B2, H, W, C)
        else:
            n = torch.cat([_bn_rows(bn, x2d[0]), _bn_rows(bn, x2d[1])], dim=0).view(B2, H, W, C)
        o = self.shallowfuseSS2D.forward_stacked(n)                                          # (B, 2, H, W, C)
        return (xt.view(2, B, H, W, C) + o.transpose(0, 1)).view(B2, H, W, C)            # x1 + o1 | x2 + o2 (:914)

    def forward(self, x1, x2):
        n1 = self.norm(x1).permute(0, 2, 3, 1)      # the same BatchNorm sees view 1 then view 2 (:906-907)
        n2 = self.norm(x2).permute(0, 2, 3, 1)
        o1, o2 = self.shallowfuseSS2D(n1, n2)       # drop_path on the tuple is the identity (p = 0, :903,912)
        return x1 + o1.permute(0, 3, 1, 2), x2 + o2.permute(0, 3, 1, 2)


# ---------------------------------------------------------------------------------------------
# deep fusion: three SS2D streams, view streams read through the fused stream's C (:360-690)
# ---------------------------------------------------------------------------------------------
class Cross_SS2Dv5(nn.Module):
    def __init__(self, d_model=96, d_state=16, ssm_ratio=2.0, dt_rank="auto", act_layer=nn.SiLU, d_conv=3,
                 conv_bias=True, dropout=0.0, bias=False, dt_min=0.001, dt_max=0.1, dt_init="random", dt_scale=1.0,
                 dt_init_floor=1e-4, initialize="v0", forward_type="v2", channel_first=False, **kwargs):
        super().__init__()
        if channel_first:
            raise NotImplementedError("the reference only instantiates the channel-last variant")
        self.k_group = 4
        self.d_model = int(d_model)
        self.d_state = int(d_state)
        self.d_inner = int(ssm_ratio * d_model)
        self.dt_rank = int(math.ceil(self.d_model / 16) if dt_rank == "auto" else dt_rank)
        self.channel_first = channel_first
        self.with_dconv = d_conv > 1
        self.in_proj = nn.Linear(self.d_model, self.d_inner * 2, bias=bias)   # present in checkpoints, never used
        self.in_proj_sec = nn.Linear(self.d_model, self.d_inner, bias=bias)
        self.act = act_layer()
        if self.with_dconv:
            self.conv2d = nn.Conv2d(self.d_inner, self.d_inner, kernel_size=d_conv, padding=(d_conv - 1) // 2,
                                    groups=self.d_inner, bias=conv_bias)
        self.x_proj_weight = nn.Parameter(torch.stack(
            [nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False).weight for _ in range(self.k_group)],
            dim=0))
        self.out_norm = nn.LayerNorm(self.d_inner)
        self.out_proj = nn.Linear(self.d_inner, self.d_model, bias=bias)
        self.dropout = nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()
        if initialize != "v0":
            raise NotImplementedError("only the 'v0' S6 initialiser is built")
        self.A_logs, self.Ds, self.dt_projs_weight, self.dt_projs_bias = mamba_init.init_dt_A_D(
            self.d_state, self.dt_rank, self.d_inner, dt_scale, dt_init, dt_min, dt_max, dt_init_floor,
            k_group=self.k_group)

    def forward_corev2(self, x, x2, x_fuse):
        B, D, H, W = x.shape
        w = (self.x_proj_weight, self.dt_projs_weight, self.A_logs, self.Ds, self.dt_projs_bias)

        def finish(y, like):
            return self.out_norm(y.transpose(1, 2).reshape(B, H, W, -1)).to(like.dtype)

        y_fuse, Cs_fuse = _ss2d_core(x_fuse, *w, want_Cs=True)
        y, _ = _ss2d_core(x, *w, Cs_override=Cs_fuse)
        y_2, _ = _ss2d_core(x2, *w, Cs_override=Cs_fuse)
        return finish(y, x), finish(y_2, x2), finish(y_fuse, x_fuse)

    def forward_core_batched(self, x3, planes_out=False):
        """The three streams [view 1 | view 2 | fused] as ONE batch of 3B through the operator chain (they share all
        weights); the view streams read their state through the fused stream's C (:537,568), i.e. the C operand of
        the whole batch is the fused third repeated.  Returns the out-normed (3B, H, W, D)."""
        B3, D, H, W = x3.shape
        B, L = B3 // 3, H * W
        K, _, R = self.dt_projs_weight.shape
        N = self.A_logs.shape[1]
        cd = x3.dtype
        if SS2D_MODE == "fused" and chan_supported(x3, H, W, N, K, D, R):
            # ONE kernel for the cross-fusion exchange: the three streams' four routes, dt_proj on MFMA inside, the view
            # streams reading their state through the fused stream's C rows (no cross_scan / expand / cross_merge copies)
            y = ss2d_chan_fn(x3.reshape(B3, D, L), self.x_proj_weight, self.dt_projs_weight, -self.A_logs.float().exp(),
                             self.Ds.float(), self.dt_projs_bias.reshape(-1).float(), H, W, c_mod=B, c_off=2 * B)
            if planes_out:
                # out_norm over the channel axis of the (3B, D, H, W) planes by the LayerNorm2d kernel (same maths as the
                # channel-last nn.LayerNorm of the reference on the transposed tensor, no transposing copy)
                return layernorm2d_fn(y.view(B3, D, H, W), self.out_norm.weight, self.out_norm.bias, self.out_norm.eps, cd)
            return self.out_norm(y.transpose(1, 2).reshape(B3, H, W, -1)).to(cd)
        xs = cross_scan_fn(x3, in_channel_first=True, out_channel_first=True, scans=0)           # (3B, 4, D, L)
        x_dbl = torch.matmul(self.x_proj_weight.to(cd), xs)                                       # (3B, K, R+2N, L)
        dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
        dts = torch.matmul(self.dt_projs_weight.to(cd), dts).view(B3, -1, L)
        Cs = torch.split(Cs, B, dim=0)[2].unsqueeze(0).expand(3, B, K, N, L).reshape(B3, K, N, L)
        ys = selective_scan_fn(xs.view(B3, -1, L), dts, -self.A_logs.float().exp(), Bs.contiguous(), Cs,
                               self.Ds.float(), self.dt_projs_bias.reshape(-1).float(), True, True, None)
        y = cross_merge_fn(ys.view(B3, K, -1, H, W), in_channel_first=True, out_channel_first=True, scans=0)
        return self.out_norm(y.view(B3, -1, L).transpose(1, 2).reshape(B3, H, W, -1)).to(cd)

    def forward_stacked(self, n: torch.Tensor) -> torch.Tensor:
        """``forward`` on the two views as one token-major batch ``n`` = [view 1 | view 2] (2B, H, W, C)."""
        B = n.shape[0] // 2
        if n.is_cuda and n.dtype == torch.float32 and n.is_contiguous() and n.numel() % 8 == 0 and n.data_ptr() % 16 == 0:
            od = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else n.dtype
            if od in (torch.float32, torch.bfloat16):
                # [view 1 | view 2 | (view 1 + view 2) / 2] in the GEMM's dtype by one kernel (mean + cat + cast otherwise)
                x3in = _ViewsAvgStack.apply(n, od).view(3 * B, *n.shape[1:])
                return self._from_x3(_linear_rows(self.in_proj_sec, x3in), B, n.shape[1], n.shape[2])
        # ((x + x2) / 2 as a mean over the view axis: slices would cost a zero fill + a copy + an add each in the backward pass)
        avg = n.view(2, B, *n.shape[1:]).mean(0)
        return self._from_x3(_linear_rows(self.in_proj_sec, torch.cat([n, avg], dim=0)), B, n.shape[1], n.shape[2])

    def forward(self, x, x2: torch.Tensor, **kwargs):
        B, H, W = x.shape[0], x.shape[1], x.shape[2]
        x3 = self.in_proj_sec(torch.cat([x, x2, (x + x2) / 2], dim=0))                 # one GEMM for the three streams
        return self._from_x3(x3, B, H, W)

    def _from_x3(self, x3: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
        tp = tokens_to_planes(x3.reshape(x3.shape[0], H * W, -1)).view(x3.shape[0], -1, H, W)   # (3B, D, H, W), pre-activation
        t = _dwconv_act(self.conv2d, self.act, tp) if self.with_dconv else self.act(tp)
        K, _, R = self.dt_projs_weight.shape
        if SS2D_MODE == "fused" and H * W > 64 and not chan_supported(t, H, W, self.A_logs.shape[1], K, t.shape[1], R):
            y, y2, y_fuse = self.forward_corev2(*torch.split(t, B, dim=0))
            z = self.act(torch.split(x3, B, dim=0)[2])
        else:
            if SS2D_MODE == "fused" and chan_supported(t, H, W, self.A_logs.shape[1], K, t.shape[1], R) \
                    and self.out_proj.bias is None:
                # planes all the way: scan -> LayerNorm2d kernel -> (y + y2 + y_fuse) * z on (B, D, L) -> out_proj as the
                # planes-in / tokens-out projection (y z + y2 z + y_fuse z of fusion_vmamba.py:604-608 with the gate
                # factored out: one reduction over the three streams, one product)
                y3 = self.forward_core_batched(t, planes_out=True)                       # (3B, D, H, W)
                zp = self.act(tp[2 * B:])                                               # the gate on planes
                g = y3.view(3, B, t.shape[1], H * W).sum(0) * zp.reshape(B, t.shape[1], H * W)
                return self.dropout(batched_proj(g, self.out_proj.weight, None, in_tokens=False, out_tokens=True)
                                    .view(B, H, W, -1))
            y3 = self.forward_core_batched(t)
            z = self.act(torch.split(x3, B, dim=0)[2])
            return self.dropout(self.out_proj(y3.view(3, B, *y3.shape[1:]).sum(0) * z))
        return self.dropout(self.out_proj((y + y2 + y_fuse) * z))


class FusionBlock_v5(nn.Module):
    def __init__(self, hidden_dim: int, drop_path: float, norm_layer: Callable[..., nn.Module], attn_drop_rate: float,
                 d_state: int, **kwargs):
        super().__init__()
        self.norm = norm_layer(hidden_dim)
        self.self_attention = Cross_SS2Dv5(d_model=hidden_dim, dropout=attn_drop_rate, d_state=d_state, **kwargs)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def stacked_ok(self, xt: torch.Tensor) -> bool:
        return (xt.is_cuda and xt.dtype == torch.float32 and isinstance(self.norm, LayerNorm2d)
                and rows_supported(self.norm.normalized_shape[0]))

    def forward_stacked(self, xt: torch.Tensor) -> torch.Tensor:
        """[view 1 | view 2] token-major (2B, H, W, C) fp32 -> (B, H, W, C): LayerNorm2d over the channels of an NCHW map is
        the row LayerNorm of its tokens (one launch for both views), the residual sum runs on tokens."""
        B = xt.shape[0] // 2
        n = layernorm_rows_fn(xt, self.norm.weight, self.norm.bias, self.norm.eps, xt.dtype)
        x = self.drop_path(self.self_attention.forward_stacked(n))
        # (fp32 + bf16 in one mixed-type add measured 64 us for 1.2 M elements: cast first)
        return xt.view(2, B, *xt.shape[1:]).sum(0) + x.to(xt.dtype)

    def forward(self, x1, x2):
        a = self.norm(x1).permute(0, 2, 3, 1)
        b = self.norm(x2).permute(0, 2, 3, 1)
        x = self.drop_path(self.self_attention(a, b)).permute(0, 3, 1, 2)
        return x1 + x2 + x


class CSSFVSSLayer_v5(nn.Module):
    def __init__(self, hidden_dim: int, depth: int = 1, drop_path=0.0, norm_layer: Callable[..., nn.Module] = LayerNorm2d,
                 attn_drop_rate: float = 0.0, d_state: int = 16, downsampling=None, **kwargs):
        super().__init__()
        self.blocks = nn.ModuleList([
            FusionBlock_v5(hidden_dim=hidden_dim, drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                           norm_layer=norm_layer, attn_drop_rate=attn_drop_rate, d_state=d_state, **kwargs)
            for i in range(depth)])
        self.downsampling = downsampling if downsampling != 1 else None

    def stacked_ok(self, xt: torch.Tensor) -> bool:
        return all(blk.stacked_ok(xt) for blk in self.blocks)

    def forward_stacked(self, xt: torch.Tensor) -> torch.Tensor:
        """[view 1 | view 2] token-major (2B, H, W, C) -> (B, H, W, C)."""
        B = xt.shape[0] // 2
        x1 = None
        for blk in self.blocks:
            x1 = blk.forward_stacked(xt if x1 is None else torch.cat([x1, xt[B:]], dim=0))
        return x1

    def forward(self, x1, x2):
        for blk in self.blocks:
            x1 = blk(x1, x2)
        return x1
