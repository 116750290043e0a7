"""CPU oracle for the XFMamba hot path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module, and only as the checker / the timed CPU baseline.  The
product package ``xfmamba_amd`` never imports it and has no CPU fallback.

Every function restates, in plain PyTorch on CPU tensors, what the reference
computes; the reference file:line it follows is cited per function (paths are
relative to the upstream tree XZheng0427/XFMamba).  The restatement is PINNED:
``tests/test_oracle_golden.py`` checks it against golden vectors produced by
importing the real reference in the build container
(``oracle/make_golden.py`` -> ``tests/golden/*.npz``).

The model-level functions are *functional*: they take a flat ``state_dict``
(the reference's own parameter names) plus inputs, so
they share no code with the product's ``nn.Module`` classes.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------------------
# a1  selective scan                                   models/csms6s.py:25-68 (forward)
# --------------------------------------------------------------------------------------


def selective_scan_ref(u: Tensor, delta: Tensor, A: Tensor, B: Tensor, C: Tensor,
                       D: Optional[Tensor] = None, delta_bias: Optional[Tensor] = None,
                       delta_softplus: bool = True, oflex: bool = True) -> Tensor:
    """Sequential selective scan, autograd-capable (restates ``selective_scan_torch``).

    u, delta: (B, K*Dg, L); A: (K*Dg, N); B, C: (B, K, N, L); D, delta_bias: (K*Dg,).
    ``delta + delta_bias`` then softplus (threshold 20) happen BEFORE the fp32 cast
    (csms6s.py:47-52); state and output are fp32; output keeps fp32 when ``oflex``
    (csms6s.py:67-68).  The D rows of group k share B[:, k], C[:, k] (csms6s.py:53-54).
    """
    dtype_in = u.dtype
    Bt, K, N, L = B.shape
    KD = u.shape[1]
    Dg = KD // K
    assert u.shape == (Bt, KD, L) and delta.shape == (Bt, KD, L)
    assert A.shape == (KD, N) and C.shape == B.shape
    if delta_bias is not None:
        delta = delta + delta_bias[..., None]
    if delta_softplus:
        delta = F.softplus(delta)
    u, delta, A, B, C = u.float(), delta.float(), A.float(), B.float(), C.float()
    Bx = B.view(Bt, K, 1, N, L).expand(Bt, K, Dg, N, L).reshape(Bt, KD, N, L)
    Cx = C.view(Bt, K, 1, N, L).expand(Bt, K, Dg, N, L).reshape(Bt, KD, N, L)
    dA = torch.exp(delta.unsqueeze(-1) * A.view(1, KD, 1, N))          # (B,KD,L,N)
    dBu = (delta * u).unsqueeze(-1) * Bx.permute(0, 1, 3, 2)           # (B,KD,L,N)
    h = A.new_zeros((Bt, KD, N))
    ys = []
    for t in range(L):
        h = dA[:, :, t] * h + dBu[:, :, t]
        ys.append((h * Cx[:, :, :, t]).sum(-1))
    y = torch.stack(ys, dim=2)
    out = y if D is None else y + u * D.float().unsqueeze(-1)
    return out if oflex else out.to(dtype_in)


def selective_scan_bwd_ref(u, delta, A, B, C, D, delta_bias, dout, delta_softplus=True):
    """Closed-form backward of the scan (what the reference's
    ``selective_scan_bwd_kernel.cuh:141-273`` evaluates).  All math in fp64 for use as an
    adjudicator; returns (du, ddelta, dA, dB, dC, dD, ddelta_bias) as fp64 tensors.
    Inputs are taken at their given precision (16-bit inputs are up-cast exactly).
    """
    f = torch.float64
    Bt, K, N, L = B.shape
    KD = u.shape[1]
    Dg = KD // K
    raw = delta.to(f)
    if delta_bias is not None:
        raw = raw + delta_bias.to(f)[None, :, None]
    if delta_softplus:
        dl = torch.where(raw <= 20.0, torch.log1p(torch.exp(torch.clamp(raw, max=20.0))), raw)
    else:
        dl = raw
    u_, A_, g = u.to(f), A.to(f), dout.to(f)
    Bx = B.to(f).view(Bt, K, 1, N, L).expand(Bt, K, Dg, N, L).reshape(Bt, KD, N, L)
    Cx = C.to(f).view(Bt, K, 1, N, L).expand(Bt, K, Dg, N, L).reshape(Bt, KD, N, L)
    a = torch.exp(dl.unsqueeze(2) * A_.view(1, KD, N, 1))               # (B,KD,N,L)
    b = (dl * u_).unsqueeze(2) * Bx
    h = torch.zeros(Bt, KD, N, L, dtype=f)
    hp = torch.zeros(Bt, KD, N, dtype=f)
    for t in range(L):
        hp = a[..., t] * hp + b[..., t]
        h[..., t] = hp
    dh = torch.zeros_like(h)
    nxt = torch.zeros(Bt, KD, N, dtype=f)
    for t in range(L - 1, -1, -1):
        cur = Cx[..., t] * g[:, :, None, t] + nxt
        dh[..., t] = cur
        nxt = a[..., t] * cur
    ah = h - b                                                          # a_t * h_{t-1}
    s1 = (dh * Bx).sum(2)                                               # (B,KD,L)
    du = dl * s1
    if D is not None:
        du = du + D.to(f)[None, :, None] * g
    ddl = u_ * s1 + (dh * ah * A_.view(1, KD, N, 1)).sum(2)
    if delta_softplus:
        ddelta = torch.where(raw <= 20.0, ddl * torch.sigmoid(raw), ddl)
    else:
        ddelta = ddl
    dA = (dh * ah * dl.unsqueeze(2)).sum((0, 3))
    dBx = dh * (dl * u_).unsqueeze(2)
    dCx = h * g.unsqueeze(2)
    dB = dBx.view(Bt, K, Dg, N, L).sum(2)
    dC = dCx.view(Bt, K, Dg, N, L).sum(2)
    dD = (g * u_).sum((0, 2)) if D is not None else None
    dbias = ddelta.sum((0, 2)) if delta_bias is not None else None
    return du, ddelta, dA, dB, dC, dD, dbias


# --------------------------------------------------------------------------------------
# a2/a3  cross scan / cross merge (scans=0, channel-first)   models/csm_triton.py:22-85
# --------------------------------------------------------------------------------------


def cross_scan_ref(x: Tensor) -> Tensor:
    """(B,C,H,W) -> (B,4,C,H*W): k0 row-major, k1 column-major, k2/k3 their reversals
    (csm_triton.py:25-29).  Plain torch ops: autograd gives the merge as its backward,
    which is what ``CrossScanF.backward`` does (csm_triton.py:207-225)."""
    B, C, H, W = x.shape
    y0 = x.flatten(2, 3)
    y1 = x.transpose(2, 3).flatten(2, 3)
    return torch.stack([y0, y1, y0.flip(-1), y1.flip(-1)], dim=1)


def cross_merge_ref(ys: Tensor) -> Tensor:
    """(B,4,C,H,W) -> (B,C,H*W): y0 + flip(y2) + T^-1(y1 + flip(y3)) (csm_triton.py:60-62)."""
    B, K, C, H, W = ys.shape
    y = ys.reshape(B, K, C, -1)
    y = y[:, 0:2] + y[:, 2:4].flip(-1)
    return y[:, 0] + y[:, 1].reshape(B, C, W, H).transpose(2, 3).reshape(B, C, -1)


# --------------------------------------------------------------------------------------
# a7  even-channel swap between the two views        models/fusion_vmamba.py:189-241
# --------------------------------------------------------------------------------------


class SwapScanRef(torch.autograd.Function):
    """fwd: out[:,0] = x with EVEN channels taken from x2, out[:,1] = x2 with EVEN channels
    from x (fusion_vmamba.py:198-213).  bwd is the reference's plain pass-through
    ys[:,0]->dx, ys[:,1]->dx2 (fusion_vmamba.py:217-221), NOT the true adjoint."""

    @staticmethod
    def forward(ctx, x, x2):
        B, C, H, W = x.shape
        ctx.shape = (B, C, H, W)
        xa, xb = x.reshape(B, C, -1), x2.reshape(B, C, -1)
        even = (torch.arange(C) % 2 == 0).view(1, C, 1)
        return torch.stack([torch.where(even, xb, xa), torch.where(even, xa, xb)], dim=1)

    @staticmethod
    def backward(ctx, g):
        B, C, H, W = ctx.shape
        return g[:, 0].reshape(B, C, H, W), g[:, 1].reshape(B, C, H, W)


def swap_scan_ref(x, x2):
    return SwapScanRef.apply(x, x2)


def swap_merge_ref(ys):
    """(B,2,C,L) -> two (B,C,L) tensors; no un-swap (fusion_vmamba.py:224-241)."""
    return ys[:, 0], ys[:, 1]


# --------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------


def _ln2d(x, w, b, eps=1e-5):
    """LayerNorm2d (fusion_vmamba.py:52-57): LayerNorm over C of an NCHW tensor."""
    return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, eps).permute(0, 3, 1, 2)


def _lin2d(x, w, b=None):
    """Linear2d (fusion_vmamba.py:42-45): 1x1 conv with an (out,in) weight."""
    return F.conv2d(x, w.view(w.shape[0], -1)[:, :, None, None], b)


def _proj_dt_B_C(xs, x_proj_w, dt_w, R, N):
    """x_proj -> split -> dt_proj (fusion_vmamba.py:1147-1150 grouped conv1d form and
    :490-495 einsum form; identical maths)."""
    x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, x_proj_w)
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
    dts = torch.einsum("bkrl,kdr->bkdl", dts, dt_w)
    return dts, Bs, Cs


# --------------------------------------------------------------------------------------
# a4  SS2Dv2 (forward_type v05_noz, channel-first)      fusion_vmamba.py:1035-1206
# --------------------------------------------------------------------------------------


def ss2d_v2_ref(sd: Dict[str, Tensor], p: str, x: Tensor, scan=selective_scan_ref) -> Tensor:
    w_in = sd[p + "in_proj.weight"]
    D = w_in.shape[0]
    x = _lin2d(x, w_in)                                                   # :1191
    x = F.conv2d(x, sd[p + "conv2d.weight"], sd.get(p + "conv2d.bias"), padding=1, groups=D)
    x = F.silu(x)                                                         # :1200-1201
    B, _, H, W = x.shape
    L = H * W
    xw, dtw = sd[p + "x_proj_weight"], sd[p + "dt_projs_weight"]
    K, _, R = dtw.shape
    N = sd[p + "A_logs"].shape[1]
    xs = cross_scan_ref(x)                                                # :1145
    dts, Bs, Cs = _proj_dt_B_C(xs, xw, dtw, R, N)
    As = -sd[p + "A_logs"].float().exp()                                  # :1161
    ys = scan(xs.reshape(B, -1, L), dts.reshape(B, -1, L), As, Bs.contiguous(), Cs.contiguous(),
              sd[p + "Ds"].float(), sd[p + "dt_projs_bias"].reshape(-1).float(), True, True)
    y = cross_merge_ref(ys.view(B, K, -1, H, W)).view(B, -1, H, W)        # :1174,1183
    y = _ln2d(y, sd[p + "out_norm.weight"], sd[p + "out_norm.bias"]).to(x.dtype)  # :1186-1188
    return _lin2d(y, sd[p + "out_proj.weight"])                           # :1205


# --------------------------------------------------------------------------------------
# a5  VSSBlock / Mlp (eval mode or drop_path 0)          fusion_vmamba.py:1325-1337, 135-153
# --------------------------------------------------------------------------------------


def vss_block_ref(sd, p, x, scan=selective_scan_ref, keep=None):
    """``keep``: None (eval mode / drop_path 0), or the pair of per-sample DropPath factors (B,) of the block's two branches
    as timm's ``drop_path`` forms them -- Bernoulli(1 - p) / (1 - p) per sample (fusion_vmamba.py:1327, 1335: ``x +
    self.drop_path(...)``): train-mode parity feeds the factors the implementation under test sampled."""
    h = _ln2d(x, sd[p + "norm.weight"], sd[p + "norm.bias"])
    y = ss2d_v2_ref(sd, p + "op.", h, scan)
    x = x + (y if keep is None or keep[0] is None else y * keep[0].view(-1, 1, 1, 1).to(y.dtype))
    h = _ln2d(x, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    h = _lin2d(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    h = _lin2d(F.gelu(h), sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + (h if keep is None or keep[1] is None else h * keep[1].view(-1, 1, 1, 1).to(h.dtype))


# --------------------------------------------------------------------------------------
# a6  Backbone_VSSM                                       fusion_vmamba.py:1504-1538,1704-1724
# --------------------------------------------------------------------------------------


def backbone_ref(sd, p, x, scan=selective_scan_ref, all_outs=False, drop=None):
    """``drop``: None, or a mapping (stage, block) -> (keep1, keep2) of per-sample DropPath factors (vss_block_ref)."""
    x = F.conv2d(x, sd[p + "patch_embed.0.weight"], sd[p + "patch_embed.0.bias"], stride=2, padding=1)
    x = F.gelu(_ln2d(x, sd[p + "patch_embed.2.weight"], sd[p + "patch_embed.2.bias"]))
    x = F.conv2d(x, sd[p + "patch_embed.5.weight"], sd[p + "patch_embed.5.bias"], stride=2, padding=1)
    x = _ln2d(x, sd[p + "patch_embed.7.weight"], sd[p + "patch_embed.7.bias"])
    outs = []
    i = 0
    while (p + f"layers.{i}.blocks.0.norm.weight") in sd:
        j = 0
        while (p + f"layers.{i}.blocks.{j}.norm.weight") in sd:
            x = vss_block_ref(sd, p + f"layers.{i}.blocks.{j}.", x, scan, None if drop is None else drop.get((i, j)))
            j += 1
        outs.append(_ln2d(x, sd[p + f"outnorm{i}.weight"], sd[p + f"outnorm{i}.bias"]))
        if (p + f"layers.{i}.downsample.1.weight") in sd:
            x = F.conv2d(x, sd[p + f"layers.{i}.downsample.1.weight"], sd[p + f"layers.{i}.downsample.1.bias"],
                         stride=2, padding=1)
            x = _ln2d(x, sd[p + f"layers.{i}.downsample.3.weight"], sd[p + f"layers.{i}.downsample.3.bias"])
        i += 1
    return outs if all_outs else outs[-1]


# --------------------------------------------------------------------------------------
# a8  ShallowFuse_SS2Dv4 / ShallowFusionBlock_v4          fusion_vmamba.py:777-920
# --------------------------------------------------------------------------------------


def shallow_block_ref(sd, p, x1, x2, training=False, scan=selective_scan_ref):
    """BatchNorm2d is applied to each view separately by the same module (:906-907): in
    training mode batch statistics are used and the running buffers in ``sd`` are updated
    twice, in this order."""
    q = p + "norm."

    def bn(x):
        return F.batch_norm(x, sd[q + "running_mean"], sd[q + "running_var"], sd[q + "weight"], sd[q + "bias"],
                            training, 0.1, 1e-5)

    n1, n2 = bn(x1), bn(x2)
    if training and (q + "num_batches_tracked") in sd:
        sd[q + "num_batches_tracked"] += 2
    s = p + "shallowfuseSS2D."
    a = F.linear(n1.permute(0, 2, 3, 1), sd[s + "in_proj.weight"])       # :848-849 (NHWC)
    b = F.linear(n2.permute(0, 2, 3, 1), sd[s + "in_proj.weight"])
    a_p, b_p = a.permute(0, 3, 1, 2).contiguous(), b.permute(0, 3, 1, 2).contiguous()
    D = a_p.shape[1]
    ca = F.silu(F.conv2d(a_p, sd[s + "conv2d.weight"], sd.get(s + "conv2d.bias"), padding=1, groups=D))
    cb = F.silu(F.conv2d(b_p, sd[s + "conv2d.weight"], sd.get(s + "conv2d.bias"), padding=1, groups=D))
    Bt, _, H, W = ca.shape
    L = H * W
    xw, dtw = sd[s + "x_proj_weight"], sd[s + "dt_projs_weight"]
    K, _, R = dtw.shape
    N = sd[s + "A_logs"].shape[1]
    xs = swap_scan_ref(ca, cb)                                            # :812
    dts, Bs, Cs = _proj_dt_B_C(xs, xw, dtw, R, N)
    ys = scan(xs.reshape(Bt, -1, L), dts.reshape(Bt, -1, L), -sd[s + "A_logs"].float().exp(),
              Bs.contiguous(), Cs.contiguous(), sd[s + "Ds"].float(),
              sd[s + "dt_projs_bias"].reshape(-1).float(), True, True).view(Bt, K, -1, L)
    y1, y2 = swap_merge_ref(ys)                                           # :835
    ow, ob = sd[s + "out_norm.weight"], sd[s + "out_norm.bias"]
    y1 = F.layer_norm(y1.transpose(1, 2).reshape(Bt, H, W, -1), (D,), ow, ob).to(ca.dtype)  # :839-845
    y2 = F.layer_norm(y2.transpose(1, 2).reshape(Bt, H, W, -1), (D,), ow, ob).to(cb.dtype)

    def gate(xp):                                                         # :865-869
        sq = xp.mean((2, 3))
        e = torch.sigmoid(F.linear(F.silu(F.linear(sq, sd[s + "fc1.0.weight"])), sd[s + "fc1.2.weight"]))
        return e.view(Bt, 1, 1, D)

    y1 = y1 * gate(b_p)                                                   # :870 (cross-view gate)
    y2 = y2 * gate(a_p)                                                   # :871
    o1 = F.linear(y1, sd[s + "out_proj.weight"]).permute(0, 3, 1, 2)
    o2 = F.linear(y2, sd[s + "out_proj.weight"]).permute(0, 3, 1, 2)
    return x1 + o1, x2 + o2                                               # :917-918


# --------------------------------------------------------------------------------------
# a9  Cross_SS2Dv5 / FusionBlock_v5                        fusion_vmamba.py:446-643
# --------------------------------------------------------------------------------------


def deep_block_ref(sd, p, x1, x2, scan=selective_scan_ref):
    n1 = _ln2d(x1, sd[p + "norm.weight"], sd[p + "norm.bias"]).permute(0, 2, 3, 1)
    n2 = _ln2d(x2, sd[p + "norm.weight"], sd[p + "norm.bias"]).permute(0, 2, 3, 1)
    s = p + "self_attention."
    w = sd[s + "in_proj_sec.weight"]                                      # in_proj is unused (:583-585)
    xf = (n1 + n2) / 2                                                    # :581
    a, b, f = F.linear(n1, w), F.linear(n2, w), F.linear(xf, w)
    z = F.silu(f)                                                         # :587
    D = w.shape[0]

    def dw(t):
        t = t.permute(0, 3, 1, 2).contiguous()
        return F.silu(F.conv2d(t, sd[s + "conv2d.weight"], sd.get(s + "conv2d.bias"), padding=1, groups=D))

    a, b, f = dw(a), dw(b), dw(f)
    Bt, _, H, W = a.shape
    L = H * W
    xw, dtw = sd[s + "x_proj_weight"], sd[s + "dt_projs_weight"]
    K, _, R = dtw.shape
    N = sd[s + "A_logs"].shape[1]
    As = -sd[s + "A_logs"].float().exp()
    Ds = sd[s + "Ds"].float()
    bias = sd[s + "dt_projs_bias"].reshape(-1).float()
    ow, ob = sd[s + "out_norm.weight"], sd[s + "out_norm.bias"]

    def stream(x, Cs_override=None):
        xs = cross_scan_ref(x)
        dts, Bs, Cs = _proj_dt_B_C(xs, xw, dtw, R, N)
        Cuse = Cs if Cs_override is None else Cs_override                 # view streams use Cs_fuse (:537,568)
        ys = scan(xs.reshape(Bt, -1, L), dts.reshape(Bt, -1, L), As, Bs.contiguous(), Cuse.contiguous(),
                  Ds, bias, True, True)
        y = cross_merge_ref(ys.view(Bt, K, -1, H, W))
        y = F.layer_norm(y.transpose(1, 2).reshape(Bt, H, W, -1), (D,), ow, ob).to(x.dtype)
        return y, Cs

    yf, Cs_fuse = stream(f)
    ya, _ = stream(a, Cs_fuse)
    yb, _ = stream(b, Cs_fuse)
    out = F.linear(ya * z + yb * z + yf * z, sd[s + "out_proj.weight"])   # :605-609
    return x1 + x2 + out.permute(0, 3, 1, 2)                              # :642


# --------------------------------------------------------------------------------------
# a11  TwoViewXFMambaTop                                   net_fusionmamba.py:192-210
# --------------------------------------------------------------------------------------


def xfmamba_top_ref(sd, x_a, x_b, training=False, scan=selective_scan_ref, drop_a=None, drop_b=None):
    """Whole model.  ``training`` switches the BatchNorm of the shallow block to batch statistics; DropPath is the
    identity unless ``drop_a`` / ``drop_b`` hand the trunk its per-sample factors for the two views (backbone_ref)."""
    x_a = x_a.expand(-1, 3, -1, -1)
    x_b = x_b.expand(-1, 3, -1, -1)
    z_a = backbone_ref(sd, "mamba_feature_extrac.", x_a, scan, drop=drop_a)
    z_b = backbone_ref(sd, "mamba_feature_extrac.", x_b, scan, drop=drop_b)
    z_a, z_b = shallow_block_ref(sd, "shallow_mamba_fusion.", z_a, z_b, training, scan)
    z = deep_block_ref(sd, "fusemamba.blocks.0.", z_a, z_b, scan)
    z = F.conv2d(z, sd["final_conv.weight"], sd["final_conv.bias"])
    z = z.mean((2, 3))
    return F.linear(z, sd["classifier.head.weight"], sd["classifier.head.bias"])


# --------------------------------------------------------------------------------------
# deterministic synthetic weights (shared by golden generation and tests)
# --------------------------------------------------------------------------------------


def synth_state_dict(shapes: Dict[str, tuple], seed: int = 0) -> Dict[str, Tensor]:
    """Fill a state_dict deterministically from its key names, independent of any module
    construction order.  Statistics mimic the reference initialisers (mamba_init,
    fusion_vmamba.py:289-356; VSSM._init_weights :1475-1482) so the network is in a
    realistic numeric regime, but the values are NOT the reference's RNG stream."""
    import zlib

    import numpy as np

    out = {}
    for key in sorted(shapes):
        shape = tuple(shapes[key])
        rng = np.random.default_rng([seed, zlib.crc32(key.encode())])
        leaf = key.rsplit(".", 1)[-1]
        n = int(np.prod(shape)) if len(shape) else 1
        if leaf == "num_batches_tracked":
            out[key] = torch.zeros(shape, dtype=torch.long)
            continue
        if leaf == "running_mean":
            v = 0.1 * rng.standard_normal(n)
        elif leaf == "running_var":
            v = 1.0 + 0.2 * rng.random(n)
        elif leaf == "A_logs":
            N = shape[1]
            v = (np.log(np.arange(1, N + 1, dtype=np.float64))[None, :] + 0.1 * rng.standard_normal(shape)).reshape(-1)
        elif leaf == "Ds":
            v = 1.0 + 0.1 * rng.standard_normal(n)
        elif leaf == "dt_projs_bias":
            dt = np.exp(rng.random(n) * (math.log(0.1) - math.log(0.001)) + math.log(0.001)).clip(min=1e-4)
            v = dt + np.log(-np.expm1(-dt))
        elif leaf == "dt_projs_weight":
            s = shape[-1] ** -0.5
            v = rng.uniform(-s, s, n)
        elif leaf == "bias":
            v = 0.02 * rng.standard_normal(n)
        elif leaf == "weight" and len(shape) == 1:
            v = 1.0 + 0.1 * rng.standard_normal(n)                       # norm scales
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else n
            v = rng.standard_normal(n) * min(0.05, 1.0 / math.sqrt(max(fan_in, 1))) * 1.5
        out[key] = torch.from_numpy(np.asarray(v, dtype=np.float32).reshape(shape).copy())
    return out
