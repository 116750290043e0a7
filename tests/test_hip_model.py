"""GPU parity of the module layer: the reference-named
nn.Modules running the HIP kernels vs golden vectors recorded from the real reference (G4 blocks,
G5 whole model) -- outputs, input gradients, parameter gradients, BatchNorm buffers."""
import pytest
import torch

from oracle.golden_inputs import g5_inputs
from oracle import xfm_oracle as O
from tests.helpers import assert_close, load_json, load_npz

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _build(kind):
    from xfmamba_amd import fusion_vmamba as fv
    if kind == "ss2dv2":
        return fv.SS2Dv2(d_model=16, d_state=1, ssm_ratio=1.0, dt_rank="auto", conv_bias=False,
                         forward_type="v05_noz", channel_first=True)
    if kind == "ss2dv2_r2":
        return fv.SS2Dv2(d_model=16, d_state=1, ssm_ratio=2.0, dt_rank="auto", conv_bias=False,
                         forward_type="v05_noz", channel_first=True)
    if kind == "vssblock":
        return fv.VSSBlock(hidden_dim=16, drop_path=0.0, norm_layer=fv.LayerNorm2d, channel_first=True,
                           ssm_d_state=1, ssm_ratio=1.0, ssm_dt_rank="auto", ssm_conv=3, ssm_conv_bias=False,
                           ssm_init="v0", forward_type="v05_noz", mlp_ratio=4.0)
    if kind.startswith("shallow"):
        return fv.ShallowFusionBlock_v4(hidden_dim=32, d_state=16)
    return fv.FusionBlock_v5(hidden_dim=32, drop_path=0.0, norm_layer=fv.LayerNorm2d, attn_drop_rate=0.0, d_state=16)


BLOCKS = [("ss2dv2", False), ("ss2dv2_r2", False), ("vssblock", True), ("shallow_train", True),
          ("shallow_eval", False), ("deep", True)]


@pytest.mark.parametrize("mode", ["unfused", "fused"])
@pytest.mark.parametrize("tag,training", BLOCKS, ids=[b[0] for b in BLOCKS])
def test_blocks_match_reference_golden(tag, training, mode):
    from xfmamba_amd import fusion_vmamba as fv
    z = load_npz("g4_blocks.npz")
    old = fv.SS2D_MODE
    fv.SS2D_MODE = mode
    try:
        m = _build(tag)
        pre = f"{tag}/sd/"
        m.load_state_dict({k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}, strict=True)
        m = m.to(DEV).train(training)
        ins = [torch.from_numpy(z[f"{tag}/in{i}"]).to(DEV).requires_grad_() for i in range(2) if f"{tag}/in{i}" in z.files]
        out = m(*ins)
        outs = out if isinstance(out, (tuple, list)) else (out,)
        for i, o in enumerate(outs):
            ref = torch.from_numpy(z[f"{tag}/out{i}"])
            assert_close(o.detach().cpu(), ref, 1e-3, 1e-3 * float(ref.abs().max()), f"out{i}")
        torch.autograd.backward(list(outs), [torch.from_numpy(z[f"{tag}/gout{i}"]).to(DEV) for i in range(len(outs))])
        for i, t in enumerate(ins):
            ref = torch.from_numpy(z[f"{tag}/din{i}"])
            assert_close(t.grad.cpu(), ref, 1e-3, 1e-3 * float(ref.abs().max()), f"din{i}")
        pre = f"{tag}/grad/"
        params = dict(m.named_parameters())
        for k in z.files:
            if k.startswith(pre):
                ref = torch.from_numpy(z[k])
                assert_close(params[k[len(pre):]].grad.cpu(), ref, 1e-3, 1e-3 * float(ref.abs().max()) + 1e-7, k)
        for k, p in params.items():
            assert (p.grad is not None) == ((pre + k) in z.files), k
        pre = f"{tag}/sd_after/"
        sd = m.state_dict()
        for k in z.files:
            if k.startswith(pre):
                assert_close(sd[k[len(pre):]].float().cpu(), torch.from_numpy(z[k]).float(), 1e-5, 1e-6, k)
    finally:
        fv.SS2D_MODE = old


def test_deep_fusion_block_golden_through_channel_lane_kernel():
    """G4 `deep` (FusionBlock_v5, hidden 32 -> d_inner 64, d_state 16, 5x5 maps) under bf16 autocast: the three streams'
    cross-fusion exchange runs as ONE launch of the channel-lane kernel (xfm_ss2dc_fwd/_bwd with the view streams reading
    the fused stream's C rows); output and gradients against the record of the real reference at the bf16 bound."""
    from xfmamba_amd import _lib
    z = load_npz("g4_blocks.npz")
    tag = "deep"
    m = _build(tag)
    pre = f"{tag}/sd/"
    m.load_state_dict({k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}, strict=True)
    m = m.to(DEV).train(True)
    ins = [torch.from_numpy(z[f"{tag}/in{i}"]).to(DEV).requires_grad_() for i in range(2)]
    timer = _lib.KernelTimer()
    _lib.set_timer(timer)
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = m(*ins)
        out.float().backward(torch.from_numpy(z[f"{tag}/gout0"]).to(DEV))
    finally:
        _lib.set_timer(None)
    names = set(timer.summary())
    assert {"ss2dc16_fwd", "ss2dc16_bwd"} <= names and not ({"cross_scan", "cross_merge", "selective_scan_fwd"} & names), names
    # Yardstick: BASELINE's 1e-2 is a bound on ONE 16-bit operator; this block chains five bf16 GEMMs, three
    # scans with bf16 operands and two LayerNorms under autocast.  What bf16 arithmetic itself does to it is measured by the CPU
    # oracle of the same block under torch.autocast(bfloat16) (fp32 scan): every tensor of the HIP path must sit within 1.5 x
    # the oracle's own distance from the fp32 record of the real reference (+ 5e-3 of the tensor scale), and never beyond the
    # caps 1e-2 (output, input gradients) and 2e-2 (parameter gradients; measured: 1.2e-3, 2.1e-3, <= 1.7e-2 with the oracle's own bf16 run at 1.2e-2 on that tensor, A_logs).
    sd = {k[len(f"{tag}/sd/"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(f"{tag}/sd/")}
    leaves = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    xin = [torch.from_numpy(z[f"{tag}/in{i}"]).clone().requires_grad_() for i in range(2)]
    with torch.autocast("cpu", dtype=torch.bfloat16):
        oo = O.deep_block_ref(leaves, "", xin[0], xin[1])
    oo.float().backward(torch.from_numpy(z[f"{tag}/gout0"]))

    def check(name, got, orc, ref, cap):
        scale = float(ref.abs().max()) + 1e-12
        d_hip, d_orc = float((got - ref).abs().max()) / scale, float((orc - ref).abs().max()) / scale
        assert d_hip <= 1.5 * d_orc + 5e-3, (name, d_hip, d_orc)
        assert d_hip <= cap, (name, d_hip)
        return d_hip, d_orc

    ref = torch.from_numpy(z[f"{tag}/out0"])
    rep = {"out": check("out", out.detach().float().cpu(), oo.detach().float(), ref, 1e-2)}
    for i, t in enumerate(ins):
        rep[f"din{i}"] = check(f"din{i}", t.grad.float().cpu(), xin[i].grad.float(), torch.from_numpy(z[f"{tag}/din{i}"]), 1e-2)
    pre = f"{tag}/grad/"
    params = dict(m.named_parameters())
    for k in z.files:
        if k.startswith(pre):
            n = k[len(pre):]
            rep[n] = check(n, params[n].grad.float().cpu(), leaves[n].grad.float(), torch.from_numpy(z[k]), 2e-2)
    print("deep block, bf16 autocast: (HIP error, oracle-autocast error) relative to each tensor's scale:",
          {k: (round(a, 4), round(b, 4)) for k, (a, b) in rep.items()})


def _lib_timer():
    from xfmamba_amd import _lib
    t = _lib.KernelTimer()
    _lib.set_timer(t)
    return t


def _lib_timer_stop(t):
    from xfmamba_amd import _lib
    _lib.set_timer(None)
    return set(t.summary())


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,H", [(96, 56), (192, 28), (384, 14), (768, 7)])
def test_vss_stage_tokens_stream_matches_planes_and_oracle(C, H, dt):
    """Two VSSBlocks (DropPath off) at a trunk width: the token-major stream (row LayerNorm + batched-GEMM layout
    changes) vs the NCHW module path vs the CPU oracle -- output, input gradient, every parameter gradient."""
    from xfmamba_amd import fusion_vmamba as fv
    torch.manual_seed(C)
    blocks = torch.nn.Sequential(*[
        fv.VSSBlock(hidden_dim=C, drop_path=0.0, norm_layer=fv.LayerNorm2d, channel_first=True, ssm_d_state=1,
                    ssm_ratio=1.0, ssm_dt_rank="auto", ssm_conv=3, ssm_conv_bias=False, ssm_init="v0",
                    forward_type="v05_noz", mlp_ratio=4.0) for _ in range(2)]).to(DEV).train()
    x = torch.randn(2, C, H, H, device=DEV)
    gy = torch.randn(2, C, H, H, device=DEV)
    res = {}
    old, old_full = fv.STREAM_LAYOUT, fv.TOKEN_SS2D
    # "tokens_full": the SS2D block of the short-map stages entirely token-major (fv.TOKEN_SS2D: token-major depthwise
    # convolution, the scan reading x token-major; opt-in) -- the same stream layout, one more kernel path to hold to the oracle
    layouts = ("planes", "tokens") + (("tokens_full",) if (dt == torch.bfloat16 and H <= 14) else ())
    try:
        for layout in layouts:
            fv.STREAM_LAYOUT = "tokens" if layout == "tokens_full" else layout
            fv.TOKEN_SS2D = layout == "tokens_full"
            blocks.zero_grad(set_to_none=True)
            xi = x.clone().requires_grad_()
            timer = _lib_timer()
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dt == torch.bfloat16):
                y = fv._run_blocks(blocks, xi)
            y.float().backward(gy)
            ran = _lib_timer_stop(timer)
            if layout == "tokens_full":
                assert "proj_gemm" not in ran and "layernorm2d_fwd" not in ran, sorted(ran)      # no layout-changing product left
            res[layout] = (y.detach().float().cpu(), xi.grad.cpu(),
                           {k: p.grad.detach().cpu().clone() for k, p in blocks.named_parameters()})
    finally:
        fv.STREAM_LAYOUT, fv.TOKEN_SS2D = old, old_full
    sd = {f"b.{k}": v.detach().cpu() for k, v in blocks.state_dict().items()}
    xr = x.cpu().clone().requires_grad_()
    pr = {k: v.clone().requires_grad_() for k, v in sd.items()}
    yr = xr
    for i in range(2):
        yr = O.vss_block_ref(pr, f"b.{i}.", yr)
    yr.backward(gy.cpu())
    tol = 1e-3 if dt == torch.float32 else 2e-2
    for layout in layouts:
        y, dx, grads = res[layout]
        assert_close(y, yr.detach(), tol, tol * float(yr.abs().max()), f"{layout} y")
        assert_close(dx, xr.grad, tol, tol * float(xr.grad.abs().max()), f"{layout} dx")
        for k, gref in ((k, pr[f"b.{k}"].grad) for k in grads):
            assert_close(grads[k], gref, 3 * tol, (tol if dt == torch.float32 else 1.5 * tol) * float(gref.abs().max()) + 1e-7, f"{layout} d{k}")


def test_vss_stage_tokens_stream_drop_path_matches_planes():
    """DropPath active (train mode): the token-major stream draws the same per-sample Bernoulli factors as the NCHW
    modules (same generator state) and folds them into the fused add+LayerNorm kernel -- outputs and gradients of
    the two layouts must agree, dropped samples included."""
    from xfmamba_amd import fusion_vmamba as fv
    torch.manual_seed(11)
    C, H, B = 96, 28, 8
    blocks = torch.nn.Sequential(*[
        fv.VSSBlock(hidden_dim=C, drop_path=0.5, norm_layer=fv.LayerNorm2d, channel_first=True, ssm_d_state=1,
                    ssm_ratio=1.0, ssm_dt_rank="auto", ssm_conv=3, ssm_conv_bias=False, ssm_init="v0",
                    forward_type="v05_noz", mlp_ratio=4.0) for _ in range(3)]).to(DEV).train()
    x = torch.randn(B, C, H, H, device=DEV)
    gy = torch.randn(B, C, H, H, device=DEV)
    res = {}
    old = fv.STREAM_LAYOUT
    try:
        for layout in ("planes", "tokens"):
            fv.STREAM_LAYOUT = layout
            blocks.zero_grad(set_to_none=True)
            xi = x.clone().requires_grad_()
            torch.manual_seed(123)                                    # same DropPath draws for both layouts
            y = fv._run_blocks(blocks, xi)
            y.backward(gy)
            res[layout] = (y.detach().cpu(), xi.grad.cpu(), {k: p.grad.cpu().clone() for k, p in blocks.named_parameters()})
    finally:
        fv.STREAM_LAYOUT = old
    (yp, dxp, gp), (yt, dxt, gt) = res["planes"], res["tokens"]
    assert float((yp - x.cpu()).abs().amax(dim=(1, 2, 3)).min()) >= 0          # (some samples may be fully dropped)
    assert_close(yt, yp, 1e-4, 1e-4 * float(yp.abs().max()), "y")
    assert_close(dxt, dxp, 1e-3, 1e-4 * float(dxp.abs().max()), "dx")
    for k in gp:
        assert_close(gt[k], gp[k], 2e-3, 2e-4 * float(gp[k].abs().max()) + 1e-8, f"d{k}")


# BASELINE.json model sizes: (golden tag, type, ctor kwargs, image size, batch) -- net_fusionmamba.py:150-159
MODEL_CFGS = {
    "tiny": ("g5", "tiny", {}, 224, 2),                                   # configs[0]/[1]
    "small": ("g5s", "small", {}, 224, 2),                                # configs[2]
    "base384": ("g5b", "base", dict(hidden_dim=1024), 384, 1),            # configs[3]
}


def _model_with_synth_weights(ty="tiny", kw=None):
    from xfmamba_amd.net_fusionmamba import TwoViewXFMambaTop
    shapes = load_json("g5_state_shapes.json")[ty]
    m = TwoViewXFMambaTop(in_channels=1, outputs=2, type=ty, **(kw or {}))
    m.load_state_dict(O.synth_state_dict(shapes, seed=0), strict=True)
    return m.to(DEV)


def _tiny_with_synth_weights():
    return _model_with_synth_weights("tiny")


def _check_model_against_golden(m, tag, size, batch, tol_logits, tol_gnorm, tol_grad, autocast=False):
    """eval logits, train logits, CE loss, which parameters get a gradient, every parameter-gradient norm, sampled
    gradients and the BatchNorm buffers of a whole TwoViewXFMambaTop vs the record of the real reference."""
    z = load_npz(f"{tag}_model.npz")
    names = load_json(f"{tag}_grad_names.json")
    xa, xb, lab = (t.to(DEV) for t in g5_inputs(batch, size))
    ac = dict(device_type="cuda", dtype=torch.bfloat16, enabled=autocast)
    m.eval()
    with torch.no_grad(), torch.autocast(**ac):
        logits = m(xa, xb)
    ref = torch.from_numpy(z["logits_eval"])
    assert_close(logits.float().cpu(), ref, tol_logits, tol_logits * float(ref.abs().max()), "eval logits")
    m.train()
    for mod in m.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    with torch.autocast(**ac):
        out = m(xa, xb)
    ref = torch.from_numpy(z["logits_train"])
    assert_close(out.detach().float().cpu(), ref, tol_logits, tol_logits * float(ref.abs().max()), "train logits")
    loss = torch.nn.functional.cross_entropy(out.float(), lab)
    assert abs(float(loss.detach()) - float(z["loss"])) < 2 * tol_logits * max(1.0, float(ref.abs().max()))
    loss.backward()
    params = dict(m.named_parameters())
    assert sorted(k for k, p in params.items() if p.grad is None) == sorted(names["no_grad"])
    if tol_gnorm is None:                                # (gradient values are checked elsewhere)
        return
    for k, row in zip(names["grad_names"], z["grad_stats"]):
        gn = float(params[k].grad.double().norm())
        assert abs(gn - row[2]) <= tol_gnorm * row[2] + 1e-6, (k, gn, row[2])
    for k in z.files:
        if k.startswith("grad/"):
            ref = torch.from_numpy(z[k])
            assert_close(params[k[5:]].grad.float().cpu(), ref, tol_grad, 0.4 * tol_grad * float(ref.abs().max()) + 1e-8, k)
        if k.startswith("bn_after/") and not autocast:
            assert_close(m.state_dict()[k[9:]].cpu(), torch.from_numpy(z[k]), 1e-4, 1e-5, k)


@pytest.mark.parametrize("mode,layout", [("unfused", "planes"), ("fused", "planes"), ("fused", "tokens")])
@pytest.mark.parametrize("merge_views", [True, False])
def test_model_tiny_fp32_matches_reference_golden(mode, layout, merge_views):
    """BASELINE config 0 (XFMamba-T, 2x224^2, batch 2, fp32): logits, loss, every parameter gradient."""
    from xfmamba_amd import fusion_vmamba as fv
    z = load_npz("g5_model.npz")
    names = load_json("g5_grad_names.json")
    old, old_layout = fv.SS2D_MODE, fv.STREAM_LAYOUT
    fv.SS2D_MODE, fv.STREAM_LAYOUT = mode, layout
    try:
        m = _tiny_with_synth_weights()
        m.merge_views = merge_views
        xa, xb, lab = (t.to(DEV) for t in g5_inputs())
        m.eval()
        with torch.no_grad():
            logits = m(xa, xb)
        ref = torch.from_numpy(z["logits_eval"])
        assert_close(logits.cpu(), ref, 1e-3, 1e-3 * float(ref.abs().max()), "eval logits")
        m.train()
        for mod in m.modules():
            if hasattr(mod, "drop_prob"):
                mod.drop_prob = 0.0
        out = m(xa, xb)
        ref = torch.from_numpy(z["logits_train"])
        assert_close(out.detach().cpu(), ref, 1e-3, 1e-3 * float(ref.abs().max()), "train logits")
        loss = torch.nn.functional.cross_entropy(out, lab)
        assert abs(float(loss.detach()) - float(z["loss"])) < 1e-3 * max(1.0, float(z["loss"]))
        loss.backward()
        params = dict(m.named_parameters())
        assert sorted(k for k, p in params.items() if p.grad is None) == sorted(names["no_grad"])
        for k, row in zip(names["grad_names"], z["grad_stats"]):
            gn = float(params[k].grad.double().norm())
            assert abs(gn - row[2]) <= 5e-3 * row[2] + 1e-6, (k, gn, row[2])
        for k in z.files:
            if k.startswith("grad/"):
                ref = torch.from_numpy(z[k])
                assert_close(params[k[5:]].grad.cpu(), ref, 5e-3, 2e-3 * float(ref.abs().max()) + 1e-8, k)
            if k.startswith("bn_after/"):
                assert_close(m.state_dict()[k[9:]].cpu(), torch.from_numpy(z[k]), 1e-4, 1e-5, k)
    finally:
        fv.SS2D_MODE, fv.STREAM_LAYOUT = old, old_layout


@pytest.mark.parametrize("cfg", ["small", "base384"])
@pytest.mark.parametrize("layout", ["tokens", "planes"])
def test_model_small_base_fp32_match_reference_golden(cfg, layout):
    """BASELINE configs[2] (XFMamba-S, 2x224^2) and configs[3] (XFMamba-B hidden_dim=1024, 2x384^2: 96x96 ... 12x12
    maps, d_inner 256 ... 2048) in fp32 against records of the real reference: other d_inner / dt_rank / map sizes
    select other plans of the fused scan, other GEMM kernels and the L = 9216 chunked rows."""
    from xfmamba_amd import fusion_vmamba as fv
    tag, ty, kw, size, batch = MODEL_CFGS[cfg]
    old = fv.STREAM_LAYOUT
    fv.STREAM_LAYOUT = layout
    try:
        m = _model_with_synth_weights(ty, kw)
        _check_model_against_golden(m, tag, size, batch, 1e-3, 5e-3, 5e-3)
    finally:
        fv.STREAM_LAYOUT = old


@pytest.mark.parametrize("cfg", ["small", "base384"])
def test_model_small_base_bf16_autocast(cfg):
    """The bench configuration (bf16 autocast, bf16 scan I/O) of configs[2]/[3]: every kernel of the bf16 path at
    these widths, whole-model logits and loss within the end-to-end bf16 bound, the no-gradient set.  The gradients are
    held to the oracle yardstick by test_model_bf16_autocast_gradients_track_the_oracle."""
    tag, ty, kw, size, batch = MODEL_CFGS[cfg]
    m = _model_with_synth_weights(ty, kw)
    _check_model_against_golden(m, tag, size, batch, 3e-2, None, None, autocast=True)


def test_model_tiny_bf16_autocast_within_tolerance():
    """bf16 compute (autocast GEMMs, bf16 scan I/O with fp32 state).  BASELINE's 1e-2 bound is an OPERATOR bound and is
    enforced per kernel in test_hip_ops.py / test_hip_chan.py.  End to end, through 24 residual blocks + 2 fusion blocks of
    bf16 GEMMs, the policy is pinned RELATIVE to the oracle: the CPU oracle run under bf16 autocast (same GEMM precision,
    fp32 scan) sits d_orc ~ 1.1e-2 of the logit scale from the fp32 reference record; the HIP path must be no farther
    than 1.5 d_orc + 2e-3 (and never beyond 2e-2)."""
    from oracle import c_scan
    z = load_npz("g5_model.npz")
    m = _tiny_with_synth_weights().eval()
    xa, xb, _ = g5_inputs()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(xa.to(DEV), xb.to(DEV))
    ref = torch.from_numpy(z["logits_eval"])
    scale = float(ref.abs().max())
    d_hip = float((logits.float().cpu() - ref).abs().max()) / scale
    sd = O.synth_state_dict(load_json("g5_state_shapes.json")["tiny"], seed=0)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        lo = O.xfmamba_top_ref(sd, xa, xb, False, c_scan.selective_scan_c)
    d_orc = float((lo.float() - ref).abs().max()) / scale
    assert 2e-3 < d_orc < 3e-2, d_orc                  # (the yardstick itself is a bf16 run: ~1e-2)
    assert d_hip <= 1.5 * d_orc + 2e-3, (d_hip, d_orc)
    assert d_hip < 2e-2, d_hip


def test_model_tiny_bf16_at_the_bench_batch_matches_oracle_samples():
    """BASELINE configs[1] at its REAL batch: XFMamba-T, bf16 autocast, 32 two-view samples in one forward pass (eval mode:
    samples are independent, BatchNorm reads its running statistics).  At batch 32 the kernels run in the launch modes the
    bench uses (XCD-local sample maps for batch % 8 == 0, the wide-map tile plans for 64 planes per launch).  Five of the 32
    samples are checked against the CPU oracle: fp32 oracle = truth, oracle under bf16 autocast = yardstick (same rule as the
    batch-2 test above: d_hip <= 1.5 d_orc + 2e-3).  The absolute cap is 3e-2 here, the cap the yardstick itself is held to:
    the maximum over five samples sits at 1.9e-2 ... 2.3e-2 depending on the box (which convolution solver MIOpen picks for
    the stem / downsample layers decides the rounding), per-sample 1.0e-2 ... 2.0e-2, while the oracle's own bf16 run reaches
    1.7e-2 on one of them; no kernel choice of this repository moves it (chunked / tiled / unfused Mlp, either channel-lane
    generation: 1.93e-2 ... 1.97e-2 on one box)."""
    from oracle import c_scan
    m = _tiny_with_synth_weights().eval()
    g = torch.Generator().manual_seed(42)
    xa, xb = torch.randn(32, 1, 224, 224, generator=g), torch.randn(32, 1, 224, 224, generator=g)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        logits = m(xa.to(DEV), xb.to(DEV)).float().cpu()
    pick = torch.tensor([0, 7, 8, 19, 31])
    sd = O.synth_state_dict(load_json("g5_state_shapes.json")["tiny"], seed=0)
    with torch.no_grad():
        ref = O.xfmamba_top_ref(sd, xa[pick], xb[pick], False, c_scan.selective_scan_c).float()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            lo = O.xfmamba_top_ref(sd, xa[pick], xb[pick], False, c_scan.selective_scan_c).float()
    scale = float(ref.abs().max())
    d_orc = float((lo - ref).abs().max()) / scale
    d_hip = float((logits[pick] - ref).abs().max()) / scale
    assert 1e-3 < d_orc < 3e-2, d_orc
    assert d_hip <= 1.5 * d_orc + 2e-3, (d_hip, d_orc)
    assert d_hip < 3e-2, d_hip
    # the other 27 samples went through the same launches: finite, and the batch is not a broadcast of one sample
    assert torch.isfinite(logits).all() and float(logits.std(0).min()) > 0


def _oracle_autocast_grads(ty_tag, sd, xa, xb, lab):
    """Gradients of the CPU oracle under bf16 autocast (same GEMM precision as the bench configuration, fp32 scan): the
    yardstick for what bf16 arithmetic itself does to the gradients."""
    from oracle import c_scan
    leaves = {k: v.clone().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in sd.items()}
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out = O.xfmamba_top_ref(leaves, xa, xb, True, c_scan.selective_scan_c)
    torch.nn.functional.cross_entropy(out.float(), lab).backward()
    return {k: v.grad for k, v in leaves.items() if v.requires_grad and v.grad is not None}


@pytest.mark.parametrize("cfg", ["tiny", "small", "base384"])
def test_model_bf16_autocast_gradients_track_the_oracle(cfg):
    """bf16 GRADIENT parity of XFMamba-T / -S (batch 2) and -B at 384^2 (batch 1) in the bench arithmetic
    (autocast GEMMs, bf16 scan I/O with fp32 state), with the same relative-yardstick rule as the logits test.  Every
    parameter gradient of the HIP path is compared with the fp32 reference record; its error must stay within 1.5x the
    error of the CPU oracle run under bf16 autocast (+ a floor), tensor by tensor for the gradient norms and in direction
    (cosine) for every tensor."""
    tag, ty, kw, size, batch = MODEL_CFGS[cfg]
    z = load_npz(f"{tag}_model.npz")
    names = load_json(f"{tag}_grad_names.json")
    m = _model_with_synth_weights(ty, kw).train()
    for mod in m.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    xa, xb, lab = g5_inputs(batch, size)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = m(xa.to(DEV), xb.to(DEV))
    torch.nn.functional.cross_entropy(out.float(), lab.to(DEV)).backward()
    hip = {k: p.grad.float().cpu() for k, p in m.named_parameters() if p.grad is not None}
    sd = O.synth_state_dict(load_json("g5_state_shapes.json")[ty], seed=0)
    orc = _oracle_autocast_grads(ty, sd, xa, xb, lab)
    ref_norm = {k: float(row[2]) for k, row in zip(names["grad_names"], z["grad_stats"])}
    worst = (0.0, None)
    tot_h = tot_o = tot_r = 0.0
    for k, rn in ref_norm.items():
        assert k in hip and k in orc, k
        gh, go = hip[k], orc[k].float()
        eh = abs(float(gh.double().norm()) - rn) / (rn + 1e-12)
        eo = abs(float(go.double().norm()) - rn) / (rn + 1e-12)
        # norm error of the HIP path within 1.5x the oracle's own bf16 error, with a floor of 5e-2 for small tensors
        # (1e-1 for the 27-block-deep stage of XFMamba-B at batch 1: the oracle's scan runs in fp32 on fp32 operands, the
        #  HIP path's on bf16 operands, so the yardstick underestimates the sums behind A_logs / Ds / dt_projs_bias)
        assert eh <= 1.5 * eo + (1e-1 if cfg == "base384" else 5e-2), (k, eh, eo)
        worst = max(worst, (eh, k))
        if rn > 1e-6 * max(ref_norm.values()):
            cos = float(torch.nn.functional.cosine_similarity(gh.flatten().double(), go.flatten().double(), dim=0))
            assert cos > 0.97, (k, cos)
        tot_h += float(gh.double().pow(2).sum()); tot_o += float(go.double().pow(2).sum()); tot_r += rn * rn
    # the global gradient norm: the oracle's bf16 run and the HIP run must sit equally close to the fp32 record
    eh, eo = abs(tot_h ** 0.5 - tot_r ** 0.5) / tot_r ** 0.5, abs(tot_o ** 0.5 - tot_r ** 0.5) / tot_r ** 0.5
    assert eh <= 1.5 * eo + 1e-2, (eh, eo, worst)
    # the sampled full tensors of the record: direction against the fp32 reference itself
    for k in z.files:
        if k.startswith("grad/") and k[5:] in hip:
            ref = torch.from_numpy(z[k]).flatten().double()
            if float(ref.norm()) > 0:
                cos = float(torch.nn.functional.cosine_similarity(hip[k[5:]].flatten().double(), ref, dim=0))
                assert cos > 0.97, (k, cos)


def test_model_tiny_bf16_batch32_training_gradients_match_the_oracle_on_four_samples():
    """BASELINE configs[1] at its REAL batch in TRAINING form: XFMamba-T, bf16 autocast, 32 two-view
    samples, forward + backward through the launch shapes the bench uses (batch 64 after the view merge: XCD-local sample
    maps, the wide-map tile plans, the deep block's 96 samples).  The loss is the cross entropy of FOUR of the 32 samples
    (its logit gradient taken at the fp32 oracle's logits and fed to all three backward passes as the same cotangent);
    with DropPath 0 and BatchNorm reading its running statistics the samples do not interact, so the parameter gradients
    equal those of the same four samples run as a batch of 4 -- which is what the CPU oracle computes (fp32 = truth, under
    bf16 autocast = yardstick).  Every parameter-gradient norm of the HIP path must sit within 1.5x the yardstick's own error
    (+ 5e-2, the floor of the batch-2 test), ten sampled tensors also in direction, and the global norm within 1.5x + 1e-2."""
    from oracle import c_scan
    m = _tiny_with_synth_weights().train()
    for mod in m.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eval()
    g = torch.Generator().manual_seed(4321)
    xa, xb = torch.randn(32, 1, 224, 224, generator=g), torch.randn(32, 1, 224, 224, generator=g)
    lab = torch.randint(0, 2, (32,), generator=g)
    pick = torch.tensor([1, 9, 18, 30])
    sd = O.synth_state_dict(load_json("g5_state_shapes.json")["tiny"], seed=0)
    cot = {}

    def oracle_grads(autocast):
        leaves = {k: v.clone().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in sd.items()}
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            o = O.xfmamba_top_ref(leaves, xa[pick], xb[pick], False, c_scan.selective_scan_c)     # (BatchNorm: running statistics)
        if "g" not in cot:
            # d CE / d logits at the fp32 oracle's logits: ONE cotangent for all three backward passes.  (With each run's own
            # softmax the synthetic weights' saturated logits turn a 2e-2 logit difference into a 20 % difference of
            # (p - onehot), i.e. of EVERY gradient: a property of the loss at these logits, not of any backward kernel.)
            lg = o.detach().float().requires_grad_()
            torch.nn.functional.cross_entropy(lg, lab[pick]).backward()
            cot["g"] = lg.grad.clone()
        o.float().backward(cot["g"])
        return {k: v.grad.float() for k, v in leaves.items() if v.requires_grad and v.grad is not None}, o.detach().float()

    ref, lo_ref = oracle_grads(False)
    orc, _ = oracle_grads(True)
    from xfmamba_amd import _lib
    timer = _lib.KernelTimer()
    _lib.set_timer(timer)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = m(xa.to(DEV), xb.to(DEV))
    out.float()[pick.to(DEV)].backward(cot["g"].to(DEV))
    _lib.set_timer(None)
    ran = set(timer.summary())
    # the shallow block's exchange ran as ONE kernel each way (xfm_ss2dc_fwd/_bwd, n_routes 1), not as the operator chain
    assert {"ss2dc16s_fwd", "ss2dc16s_bwd", "ss2dc16_fwd", "ss2dc16_bwd"} <= ran, sorted(ran)
    assert not ({"selective_scan_fwd", "selective_scan_bwd", "dt_proj_fwd" if False else "cross_scan"} & ran), sorted(ran)
    hip = {k: p.grad.float().cpu() for k, p in m.named_parameters() if p.grad is not None}
    assert torch.isfinite(out).all()
    d_log = float((out.float().cpu()[pick] - lo_ref).abs().max()) / float(lo_ref.abs().max())
    assert d_log < 3e-2, d_log
    assert set(ref) == set(hip), set(ref) ^ set(hip)
    sampled = [k for pat in ("patch_embed.0.weight", "layers.0.blocks.0.op.x_proj_weight", "layers.0.blocks.1.op.dt_projs_weight",
                             "layers.1.blocks.0.op.A_logs", "layers.1.downsample", "layers.2.blocks.3.mlp.fc1.weight",
                             "layers.2.blocks.7.op.out_proj.weight", "layers.3.blocks.1.op.in_proj.weight",
                             "shallowfuseSS2D.x_proj_weight", "dt_projs_weight", "final_conv")
               for k in sorted(ref) if pat in k][:14]
    assert len(sampled) >= 8, sampled
    tot_h = tot_o = tot_r = 0.0
    for k, gr in ref.items():
        rn = float(gr.double().norm())
        eh = abs(float(hip[k].double().norm()) - rn) / (rn + 1e-12)
        eo = abs(float(orc[k].double().norm()) - rn) / (rn + 1e-12)
        if gr.numel() < 16:
            # (the classifier's 2-element bias gradient is mean(softmax - onehot) over four samples: a cancelling sum of O(1)
            #  terms, so its NORM amplifies the logits' bf16 error; held element-wise on the scale of its terms instead)
            da, do = float((hip[k] - gr).abs().max()), float((orc[k] - gr).abs().max())
            assert da <= 1.5 * do + 2e-2, (k, da, do)
        else:
            assert eh <= 1.5 * eo + 5e-2, (k, eh, eo)
        tot_h += float(hip[k].double().pow(2).sum()); tot_o += float(orc[k].double().pow(2).sum()); tot_r += rn * rn
        if k in sampled and rn > 0:
            cos = float(torch.nn.functional.cosine_similarity(hip[k].flatten().double(), gr.flatten().double(), dim=0))
            cos_o = float(torch.nn.functional.cosine_similarity(orc[k].flatten().double(), gr.flatten().double(), dim=0))
            assert cos > min(0.97, cos_o - 0.02), (k, cos, cos_o)
    eh, eo = abs(tot_h ** 0.5 - tot_r ** 0.5) / tot_r ** 0.5, abs(tot_o ** 0.5 - tot_r ** 0.5) / tot_r ** 0.5
    assert eh <= 1.5 * eo + 1e-2, (eh, eo)


def test_model_tiny_fp32_every_gradient_tensor_matches_the_oracle():
    """the reference record holds ~10 full gradient tensors and the norm of every other one; here EVERY
    parameter gradient of the fp32 HIP path is compared element-wise with the CPU oracle (itself pinned to the record by
    tests/test_oracle_golden.py) at 2e-3 of the tensor's scale."""
    from oracle import c_scan
    m = _tiny_with_synth_weights().train()
    for mod in m.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    xa, xb, lab = g5_inputs()
    out = m(xa.to(DEV), xb.to(DEV))
    torch.nn.functional.cross_entropy(out, lab.to(DEV)).backward()
    sd = O.synth_state_dict(load_json("g5_state_shapes.json")["tiny"], seed=0)
    leaves = {k: v.clone().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in sd.items()}
    o = O.xfmamba_top_ref(leaves, xa, xb, True, c_scan.selective_scan_c)
    torch.nn.functional.cross_entropy(o, lab).backward()
    n = 0
    for k, p in m.named_parameters():
        if p.grad is None:
            continue
        ref = leaves[k].grad
        assert ref is not None, k
        assert_close(p.grad.cpu(), ref, 5e-3, 2e-3 * float(ref.abs().max()) + 1e-8, k)
        n += 1
    assert n > 300


@pytest.mark.parametrize("B,find,wstream", [
    (16, False, False), (32, True, False),
    (32, True, True),       # bench.py --wgrad-stream: weight-gradient kernels on a side stream = a parallel branch of the graph
    (32, True, "arena"),    # bench.py's default: weight gradients accumulate into one arena that is zeroed once per step,
                            # LayerNorm / bias column sums folded by one launch at the join (xfmamba_amd/deferred.py)
    # (until round 5 an expected failure: with a merged batch of 8 the convolution library's weight-gradient solver for the
    #  384 -> 768 stride-2 downsample convolution returned garbage from the SECOND replay on.  That layer and the 192 -> 384 one
    #  now run on this repository's kernels -- xfmamba_amd/conv_tokens.py, csrc/conv_tok.hip -- and the case replays like the rest.)
    (4, False, False),
])
def test_captured_training_step_replays_like_eager(B, find, wstream):
    """The bench path replays the whole step (fwd + bwd) from one hipGraph.  Every parameter gradient of replays 1..3
    must equal the eager gradient: guards the captured path against reductions that only work on their first run
    (seen with framework bias-gradient sums under replay -- the hot path keeps those inside its own kernels)."""
    from xfmamba_amd.net_fusionmamba import TwoViewXFMambaTop
    old_find = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = find             # bench.py runs MIOpen in find mode: other solvers, same contract
    torch.manual_seed(5)
    m = TwoViewXFMambaTop(in_channels=1, outputs=2, type="tiny").to(DEV).train()
    for mod in m.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0                       # deterministic step
    xa, xb = torch.randn(B, 1, 224, 224, device=DEV), torch.randn(B, 1, 224, 224, device=DEV)
    lab = torch.randint(0, 2, (B,), device=DEV)

    from xfmamba_amd.proj import WgradArena, join_wgrad_stream, set_wgrad_arena, wgrad_stream
    arena = [None]

    def step():
        for p in m.parameters():
            p.grad = None
        if arena[0] is not None:
            arena[0].zero()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(m(xa, xb).float(), lab)
        loss.backward()
        join_wgrad_stream()                           # (what FusedAdam.step / GradBuckets do before reading .grad)
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ref = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    ref = {k: v.clone() for k, v in ref.items()}      # (the arena run below must not alias the reference)
    from xfmamba_amd import deferred
    if wstream == "arena":
        arena[0] = WgradArena(m.parameters())
        set_wgrad_arena(arena[0])
        deferred.defer_partial_sums(True)             # bench.py's default pair: arena + one fold launch for the column sums
    wgrad_stream(wstream is True)
    try:
        step()                                        # one eager step with the side stream: same gradients
        torch.cuda.synchronize()
        for k, p in m.named_parameters():
            if p.grad is not None:
                scale = float(ref[k].abs().max()) + 1e-12
                assert float((p.grad - ref[k]).abs().max()) <= 1e-1 * scale + 1e-7, ("eager", k)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
    finally:
        wgrad_stream(False)
        set_wgrad_arena(None)
        deferred.defer_partial_sums(False)
    for i in range(3):
        g.replay()
        torch.cuda.synchronize()
        for k, p in m.named_parameters():
            if p.grad is None:
                continue
            assert bool(torch.isfinite(p.grad).all()), (i, k)
            scale = float(ref[k].abs().max()) + 1e-12
            # fp32 atomics feeding bf16 roundings make run-to-run differences of a few % of a (tiny) gradient's
            # scale legitimate; a reduction that breaks under replay is off by ~100 % (zeros / stale / NaN)
            assert float((p.grad - ref[k]).abs().max()) <= 1e-1 * scale + 1e-7, (i, k)
    torch.backends.cudnn.benchmark = old_find


@pytest.mark.parametrize("wire", [None, torch.bfloat16], ids=["fp32wire", "bf16wire"])
def test_captured_forward_backward_with_packed_gradient_buckets(monkeypatch, wire):
    """The N > 1 flow of bench.py on one GPU: the hipGraph holds forward + backward + the multi-tensor packing of the
    gradients into the flat buckets; the (here: stubbed) all-reduce and the optimizer run eagerly after each replay.
    After a replay every ``.grad`` must be a view of its bucket holding eager_gradient / world."""
    import torch.distributed as dist
    from xfmamba_amd.dp import GradBuckets
    from xfmamba_amd.net_fusionmamba import TwoViewXFMambaTop
    class _Done:
        def wait(self):
            return True

    monkeypatch.setattr(dist, "all_reduce", lambda t, **kw: _Done() if kw.get("async_op") else None)
    torch.manual_seed(9)
    m = TwoViewXFMambaTop(in_channels=1, outputs=2, type="tiny").to(DEV).train()
    for mod in m.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    B = 16
    xa, xb = torch.randn(B, 1, 224, 224, device=DEV), torch.randn(B, 1, 224, 224, device=DEV)
    lab = torch.randint(0, 2, (B,), device=DEV)
    gb = GradBuckets(m, bucket_mb=16.0, overlap=False, world=2, comm_dtype=wire)     # bf16 wire: bench.py's default at N > 1
    assert len(gb.buckets) >= 2

    def fwd_bwd():
        gb.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = torch.nn.functional.cross_entropy(m(xa, xb).float(), lab)
        loss.backward()

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fwd_bwd()
            gb.finish()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ref = {k: p.grad.clone() for k, p in m.named_parameters()}          # eager: packed, divided by world
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fwd_bwd()
        gb.pack_all()
    lo, hi = gb.buckets[0].data_ptr(), gb.buckets[0].data_ptr() + gb.buckets[0].numel() * 4
    for i in range(2):
        g.replay()
        gb.reduce_all()
        torch.cuda.synchronize()
        for k, p in m.named_parameters():
            assert p.grad is not None and p.grad.dtype == torch.float32 and bool(torch.isfinite(p.grad).all()), (i, k)
            scale = float(ref[k].abs().max()) + 1e-12
            assert float((p.grad - ref[k]).abs().max()) <= 1e-1 * scale + 1e-7, (i, k)
        assert any(lo <= p.grad.data_ptr() < hi for p in m.parameters())


def test_fp16_autocast_falls_back_to_planes():
    """The row kernels of the token-major stream emit fp32 / bf16 only: under fp16 autocast the trunk must fall back to
    the NCHW modules (no error, same result as STREAM_LAYOUT="planes")."""
    from xfmamba_amd import fusion_vmamba as fv
    m = _tiny_with_synth_weights().eval()
    xa, xb, _ = (t.to(DEV) for t in g5_inputs())
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        y16 = m(xa, xb)
        old = fv.STREAM_LAYOUT
        fv.STREAM_LAYOUT = "planes"
        try:
            yp = m(xa, xb)
        finally:
            fv.STREAM_LAYOUT = old
    assert_close(y16.float().cpu(), yp.float().cpu(), 5e-3, 5e-3 * float(yp.abs().max()), "fallback == planes")
    z = load_npz("g5_model.npz")
    ref = torch.from_numpy(z["logits_eval"])
    assert_close(y16.float().cpu(), ref, 2e-2, 2e-2 * float(ref.abs().max()), "fp16 autocast logits")


def test_fused_adam_matches_torch_adam_and_refreshes_shadows():
    """xfm_adam_multi (one multi-tensor launch: Adam with L2 weight decay as torch.optim.Adam, reference
    1_train_model.py:141, + bf16 shadow refresh) vs torch.optim.Adam over 3 steps on a mixed bag of tensor shapes."""
    from xfmamba_amd.amp import WeightCache, cast_weight
    from xfmamba_amd.optim import FusedAdam
    torch.manual_seed(3)
    shapes = [(96, 96), (4, 8, 96), (384,), (5,), (33, 7), (768, 3072)]
    net_a = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(*s, device=DEV)) for s in shapes])
    net_b = torch.nn.ParameterList([torch.nn.Parameter(p.detach().clone()) for p in net_a])
    wc = WeightCache(net_a)
    opt_a = FusedAdam(net_a.parameters(), lr=1e-2, weight_decay=1e-2, weight_cache=wc)
    opt_b = torch.optim.Adam(net_b.parameters(), lr=1e-2, weight_decay=1e-2)
    for step in range(3):
        for i, (pa, pb) in enumerate(zip(net_a, net_b)):
            g = torch.randn_like(pa) * (step + 1)
            pa.grad, pb.grad = (None, None) if i == 3 else (g.clone(), g.clone())   # one never-trained parameter
        opt_a.step()
        opt_b.step()
        for pa, pb in zip(net_a, net_b):
            assert_close(pa.detach().cpu(), pb.detach().cpu(), 1e-5, 1e-6, f"step {step}")
    for pa in net_a:
        sh = cast_weight(pa, torch.bfloat16)
        assert torch.equal(sh, pa.detach().to(torch.bfloat16))
    wc.close()


def test_fused_adam_reads_summed_bf16_wire_gradients_with_a_scale():
    """Data-parallel form of the same update (dp.PhasedGrads): the gradient is the SUM over the ranks as bf16 values in a
    flat wire bucket, read in place with grad_scale = 1 / world -- against torch.optim.Adam on fp32(bf16 sum) / world.
    Slots at 16-byte AND at odd offsets (the scalar path), plus one fp32 external gradient."""
    from xfmamba_amd.optim import FusedAdam
    torch.manual_seed(5)
    shapes = [(96, 96), (4, 8, 96), (383,), (33, 7), (768, 1024)]
    net_a = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(*s, device=DEV)) for s in shapes])
    net_b = torch.nn.ParameterList([torch.nn.Parameter(p.detach().clone()) for p in net_a])
    opt_a = FusedAdam(net_a.parameters(), lr=1e-2, weight_decay=1e-2)
    opt_b = torch.optim.Adam(net_b.parameters(), lr=1e-2, weight_decay=1e-2)
    world = 8
    flat = torch.zeros(sum(p.numel() for p in net_a) + 8, dtype=torch.bfloat16, device=DEV)
    views, off = {}, 0
    for p in net_a:
        views[p] = flat[off:off + p.numel()].view(p.shape)          # (383 elements make the later slots unaligned)
        off += p.numel()
    f32 = torch.zeros_like(net_a[1])
    views[net_a[1]] = f32                                           # a gradient handed over in fp32
    for step in range(3):
        for pa, pb in zip(net_a, net_b):
            g = torch.randn_like(pa) * (step + 1) * world
            views[pa].copy_(g)
            pb.grad = views[pa].float() / world
        opt_a.step(grads=views, grad_scale=1.0 / world)
        opt_b.step()
        for pa, pb in zip(net_a, net_b):
            assert_close(pa.detach().cpu(), pb.detach().cpu(), 1e-5, 1e-6, f"step {step}")


def test_checkpoint_round_trip_and_reference_format_load():
    """state_dict -> torch.save -> fresh model load_state_dict -> identical eval logits (what the reference's EarlyStopping
    writes every epoch, early_stop.py:43-51, and 2_inference_*.py read back); a checkpoint whose Linear2d weights are
    (out, in, 1, 1) convolution tensors (older VMamba files, models/fusion_vmamba.py:47-49) loads to the same logits."""
    import io
    from xfmamba_amd.net_fusionmamba import TwoViewXFMambaTop
    m = _tiny_with_synth_weights().eval()
    xa, xb, _ = (t.to(DEV) for t in g5_inputs())
    with torch.no_grad():
        ref = m(xa, xb)
    buf = io.BytesIO()
    torch.save(m.state_dict(), buf)
    buf.seek(0)
    sd = torch.load(buf, map_location="cpu")
    m2 = TwoViewXFMambaTop(in_channels=1, outputs=2, type="tiny")
    assert not m2.load_state_dict(sd, strict=True).missing_keys
    m2 = m2.to(DEV).eval()
    with torch.no_grad():
        assert_close(m2(xa, xb).cpu(), ref.cpu(), 1e-5, 1e-5, "reloaded logits")   # (library kernels are not bit-reproducible)
    conv_style = {k: (v[:, :, None, None] if (v.ndim == 2 and (".in_proj.weight" in k or ".out_proj.weight" in k or ".mlp.fc" in k)
                                         and k.startswith("mamba_feature_extrac.")) else v) for k, v in sd.items()}
    assert any(v.ndim == 4 and v.shape[-1] == 1 for v in conv_style.values())
    m3 = TwoViewXFMambaTop(in_channels=1, outputs=2, type="tiny")
    m3.load_state_dict(conv_style, strict=True)
    m3 = m3.to(DEV).eval()
    with torch.no_grad():
        assert_close(m3(xa, xb).cpu(), ref.cpu(), 1e-5, 1e-5, "conv-style checkpoint logits")


def test_drop_path_bank_samples_every_layer_with_its_own_rate():
    """fusion_vmamba._DropPathBank: one uniform draw serves all DropPath layers of a trunk pass.  Per layer the factor is
    Bernoulli(keep) / keep with that layer's rate (timm 0.4.12's drop_path, which the reference imports), independent
    across layers; a layer consumes its row exactly once and falls back to its own kernel afterwards."""
    from xfmamba_amd import fusion_vmamba as fv
    rates = [0.05, 0.2, 0.5]
    root = torch.nn.ModuleList([fv.DropPath(r) for r in rates] + [fv.DropPath(0.0)]).train()
    B = 20000
    torch.manual_seed(0)
    with fv._DropPathBank(root, B, torch.device(DEV)):
        rows = [m._preset for m in root]
        assert rows[3] is None and all(len(r) == 2 and r[0].shape == (B,) for r in rows[:3])     # two uses per module
        first = [m.sample_scale(B, torch.device(DEV)) for m in root]
        got = [m.sample_scale(B, torch.device(DEV)) for m in root]
        assert got[3] is None and first[3] is None
        assert all(not m._preset for m in root)                           # consumed
        c12 = float(torch.corrcoef(torch.stack([first[2], got[2]]))[0, 1])
        assert abs(c12) < 0.03                                            # the two uses of a module are independent
    for r, g in zip(rates, got[:3]):
        keep = 1.0 - r
        vals = torch.unique(g).cpu()
        assert len(vals) == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1.0 / keep) < 1e-6
        assert abs(float((g == 0).float().mean()) - r) < 4 * (r * keep / B) ** 0.5 + 1e-3
    c = float(torch.corrcoef(torch.stack([got[1], got[2]]))[0, 1])
    assert abs(c) < 0.03                                                   # independent rows
    again = root[1].sample_scale(B, torch.device(DEV))                    # no preset left: the per-layer kernel
    assert again is not got[1] and abs(float((again == 0).float().mean()) - 0.2) < 0.02


def test_deferred_column_sums_fill_the_same_gradients():
    """deferred.defer_partial_sums: the row-LayerNorm / bias+GELU / bias column-sum producers leave their partial rows in
    their workspaces and ONE xfm_partial_sums_multi launch folds them at the flush (proj.join_wgrad_stream, i.e. before the
    optimizer reads).  Every parameter gradient of a tiny-model step must equal the undeferred run's; a parameter used twice
    in one pass (merge_views=False: two trunk calls) is flushed before its second producer."""
    from xfmamba_amd import deferred
    from xfmamba_amd.proj import join_wgrad_stream
    xa, xb, lab = (t.to(DEV) for t in g5_inputs())
    for merge in (True, False):
        m = _tiny_with_synth_weights().train()
        m.merge_views = merge
        for mod in m.modules():
            if hasattr(mod, "drop_prob"):
                mod.drop_prob = 0.0
        grads = []
        for on in (False, True):
            for p in m.parameters():
                p.grad = None
            deferred.defer_partial_sums(on)
            try:
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    loss = torch.nn.functional.cross_entropy(m(xa, xb).float(), lab)
                loss.backward()
                if on and merge:
                    assert len(deferred._S["jobs"]) > 20          # something was actually deferred
                # (two trunk calls: the second gradient of every parameter flushes what was pending mid-pass)
                join_wgrad_stream()                               # what FusedAdam.step / the DP packers call first
                assert not deferred._S["jobs"]
            finally:
                deferred.defer_partial_sums(False)
            torch.cuda.synchronize()
            grads.append({k: p.grad.float().clone() for k, p in m.named_parameters() if p.grad is not None})
        assert grads[0].keys() == grads[1].keys()
        for k, ref in grads[0].items():
            scale = float(ref.abs().max()) + 1e-12
            # (bf16 step with fp32 atomics elsewhere: run-to-run noise of a few % of a tiny gradient's scale; an unfilled
            #  or mis-addressed sum is off by ~100 %)
            assert float((grads[1][k] - ref).abs().max()) <= 5e-2 * scale + 1e-7, (merge, k)


def test_padded_x_proj_copy_follows_the_fused_optimizer():
    """amp.padded_shadow: the channel-lane SS2D blocks adopt the row-padded bf16 copy of x_proj_weight they build in their
    first forward pass; FusedAdam.step (through WeightCache.mark_current -> amp.refresh_derived) rewrites it in place for all
    blocks with one multi-tensor copy.  After a step the registered copy must equal pad(bf16(new weight)) and the next
    forward must produce what a freshly padded weight produces; an in-place change of the parameter makes it stale."""
    from xfmamba_amd import amp
    from xfmamba_amd.amp import WeightCache
    from xfmamba_amd.optim import FusedAdam
    from xfmamba_amd.ss2d_chan import ss2d_chan_fn, _col_layout
    torch.manual_seed(1)
    B, D, H, R, N, K = 2, 64, 14, 8, 1, 4
    L = H * H
    xw = torch.nn.Parameter((torch.randn(K, R + 2 * N, D) * D ** -0.5).to(DEV))
    dtw = torch.nn.Parameter((torch.randn(K, D, R) * R ** -0.5).to(DEV))
    A = (-torch.rand(K * D, N) - 0.1).to(DEV)
    Dp, bias = torch.randn(K * D).to(DEV), (0.1 * torch.rand(K * D) - 4.0).to(DEV)
    x = torch.randn(B, D, L).bfloat16().to(DEV)
    holder = torch.nn.ParameterList([xw, dtw])
    wc = WeightCache(holder)
    opt = FusedAdam(holder.parameters(), lr=1e-2, weight_cache=wc)
    _, _, C2p, _ = _col_layout(R, N)
    try:
        assert amp.padded_shadow(xw, C2p, torch.bfloat16) is None
        y = ss2d_chan_fn(x, xw, dtw, A, Dp, bias, H, H)
        pad = amp.padded_shadow(xw, C2p, torch.bfloat16)
        assert pad is not None and pad.shape == (K, C2p, D)
        y.float().square().mean().backward()
        opt.step()
        torch.cuda.synchronize()
        assert amp.padded_shadow(xw, C2p, torch.bfloat16) is pad                      # same storage, rewritten in place
        want = torch.zeros(K, C2p, D, dtype=torch.bfloat16, device=DEV)
        want[:, :R + 2 * N] = xw.detach().to(torch.bfloat16)
        assert torch.equal(pad, want)
        y1 = ss2d_chan_fn(x, xw, dtw, A, Dp, bias, H, H)                              # served from the registered copy
        amp.invalidate_shadows(holder)
        y2 = ss2d_chan_fn(x, xw, dtw, A, Dp, bias, H, H)                              # padded afresh from the master
        assert torch.equal(y1, y2)
        wc.refresh()
        ss2d_chan_fn(x, xw, dtw, A, Dp, bias, H, H)
        assert amp.padded_shadow(xw, C2p, torch.bfloat16) is not None
        with torch.no_grad():
            xw.mul_(0.5)                                                              # in-place change: the copy is stale
        assert amp.padded_shadow(xw, C2p, torch.bfloat16) is None
    finally:
        wc.close()


def test_model_tiny_bf16_bench_mode_step_matches_the_oracle(monkeypatch):
    """BASELINE configs[1] in the MODE bench.py times: XFMamba-T, bf16 autocast, 32 two-view samples,
    ``train()`` -- DropPath at the rates the trunk is constructed with (linspace(0, 0.2) over its 14 blocks, reference
    models/fusion_vmamba.py:1390) and the shallow block's BatchNorm on batch statistics (reference :893, 906-907) -- forward and
    backward of ONE step against the fp32 CPU oracle run on the same 32 samples.  Stochastic depth is made comparable by
    handing the oracle the per-sample factors the implementation sampled (recorded at ``DropPath.sample_scale``: Bernoulli(1 - p)
    / (1 - p), one row per block branch and stacked view); the oracle trunk runs in chunks of four samples under activation
    checkpointing (memory of the CPU run), the two fusion blocks on the whole batch (BatchNorm couples the samples).
    Checked: all 32 logits, the BatchNorm running buffers after the step, every parameter-gradient norm, fourteen sampled
    gradient tensors in direction, the global gradient norm -- at the caps the eval-mode batch-32 and the gradient tests
    hold the bf16 path to (the oracle's own bf16 run sits 1.1e-2 ... 1.7e-2 from its fp32 run on these weights)."""
    from torch.utils.checkpoint import checkpoint
    from oracle import c_scan
    from xfmamba_amd import fusion_vmamba as fv
    m = _tiny_with_synth_weights().train()
    B, CH = 32, 4
    g = torch.Generator().manual_seed(20261)
    xa, xb = torch.randn(B, 1, 224, 224, generator=g), torch.randn(B, 1, 224, 224, generator=g)
    lab = torch.randint(0, 2, (B,), generator=g)
    names = {mod: name for name, mod in m.named_modules() if isinstance(mod, fv.DropPath)}
    rates = [mod.drop_prob for mod in names]
    assert max(rates) > 0.15 and sum(r > 0 for r in rates) >= 13, rates      # the constructed rates, not zeros
    rec = {}
    orig = fv.DropPath.sample_scale

    def spy(self, batch, device):
        r = orig(self, batch, device)
        rec.setdefault(names[self], []).append(None if r is None else r.detach().float().cpu().clone())
        return r

    monkeypatch.setattr(fv.DropPath, "sample_scale", spy)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = m(xa.to(DEV), xb.to(DEV))
    monkeypatch.setattr(fv.DropPath, "sample_scale", orig)
    assert torch.isfinite(out).all()
    # the factors: two per trunk block (SS2D branch, then Mlp branch), 2 B rows ([view 1 | view 2]); some rows dropped
    drop = {}
    n_dropped = 0
    for name, rows in rec.items():
        parts = name.split(".")
        if parts[0] != "mamba_feature_extrac":
            continue
        assert len(rows) == 2, (name, len(rows))
        for r in rows:
            if r is not None:
                assert r.shape == (2 * B,)
                n_dropped += int((r == 0).sum())
        drop[(int(parts[2]), int(parts[4]))] = rows
    assert len(drop) == 14 and n_dropped > 20, (len(drop), n_dropped)

    sd = O.synth_state_dict(load_json("g5_state_shapes.json")["tiny"], seed=0)
    leaves = {k: v.clone().requires_grad_(v.is_floating_point() and "running_" not in k) for k, v in sd.items()}
    scan = c_scan.selective_scan_c

    def trunk(x, view, c0):
        dd = {ij: tuple(None if r is None else r[view * B + c0:view * B + c0 + CH] for r in rows) for ij, rows in drop.items()}
        return checkpoint(lambda xx: O.backbone_ref(leaves, "mamba_feature_extrac.", xx.expand(-1, 3, -1, -1), scan, drop=dd),
                          x[c0:c0 + CH], use_reentrant=False)

    z_a = torch.cat([trunk(xa, 0, c0) for c0 in range(0, B, CH)])
    z_b = torch.cat([trunk(xb, 1, c0) for c0 in range(0, B, CH)])
    z_a, z_b = O.shallow_block_ref(leaves, "shallow_mamba_fusion.", z_a, z_b, True, scan)        # (batch statistics)
    zz = O.deep_block_ref(leaves, "fusemamba.blocks.0.", z_a, z_b, scan)
    zz = torch.nn.functional.conv2d(zz, leaves["final_conv.weight"], leaves["final_conv.bias"]).mean((2, 3))
    lo = torch.nn.functional.linear(zz, leaves["classifier.head.weight"], leaves["classifier.head.bias"])
    # d CE / d logits at the fp32 oracle's logits: ONE cotangent for both backward passes (saturated synthetic logits turn a
    # 2e-2 logit difference into a 20 % difference of softmax - onehot: a property of the loss there, not of a kernel)
    lg = lo.detach().clone().requires_grad_()
    torch.nn.functional.cross_entropy(lg, lab).backward()
    cot = lg.grad.clone()
    lo.backward(cot)
    ref = {k: v.grad.float() for k, v in leaves.items() if v.requires_grad and v.grad is not None}
    out.float().backward(cot.to(DEV))
    hip = {k: p.grad.float().cpu() for k, p in m.named_parameters() if p.grad is not None}

    lo_ref = lo.detach().float()
    d_log = float((out.detach().float().cpu() - lo_ref).abs().max()) / float(lo_ref.abs().max())
    assert d_log < 3e-2, d_log
    for k in ("running_mean", "running_var"):
        got, want = m.state_dict()["shallow_mamba_fusion.norm." + k].float().cpu(), leaves["shallow_mamba_fusion.norm." + k].float()
        assert_close(got, want, 3e-2, 3e-2 * float(want.abs().max()), k)
    assert set(ref) == set(hip), set(ref) ^ set(hip)
    tot_h = tot_r = 0.0
    worst = (0.0, None)
    for k, gr in ref.items():
        rn = float(gr.double().norm())
        eh = abs(float(hip[k].double().norm()) - rn) / (rn + 1e-12)
        if gr.numel() >= 16:
            assert eh <= 1e-1, (k, eh)
            worst = max(worst, (eh, k))
        tot_h += float(hip[k].double().pow(2).sum()); tot_r += rn * rn
    assert abs(tot_h ** 0.5 - tot_r ** 0.5) / tot_r ** 0.5 < 3e-2, (tot_h, tot_r, worst)
    sampled = [k for pat in ("patch_embed.0.weight", "layers.0.blocks.1.op.x_proj_weight", "layers.0.blocks.1.op.dt_projs_weight",
                             "layers.1.blocks.0.op.A_logs", "layers.1.downsample", "layers.2.blocks.3.mlp.fc1.weight",
                             "layers.2.blocks.7.op.out_proj.weight", "layers.3.blocks.1.op.in_proj.weight",
                             "shallowfuseSS2D.x_proj_weight", "dt_projs_weight", "final_conv")
               for k in sorted(ref) if pat in k][:14]
    assert len(sampled) >= 10, sampled
    for k in sampled:
        cos = float(torch.nn.functional.cosine_similarity(hip[k].flatten().double(), ref[k].flatten().double(), dim=0))
        assert cos > 0.97, (k, cos)


def test_model_wrapper_eval_entry_matches_the_oracle():
    """The inference entry (reference net_fusionmamba.py:10-26, 2_inference_chexpert.py:129-267):
    ``ModelWrapper`` takes the two views concatenated along the channel axis, splits them, and returns the logits (or the
    ``output_index``-th element of a tuple).  Eval mode, fp32, batch 2: against the CPU oracle on the same inputs."""
    from oracle import c_scan
    from xfmamba_amd.net_fusionmamba import ModelWrapper
    m = _tiny_with_synth_weights().eval()
    w = ModelWrapper(m).eval()
    xa, xb, _ = g5_inputs()
    with torch.no_grad():
        got = w(torch.cat([xa, xb], dim=1).to(DEV)).float().cpu()
        direct = m(xa.to(DEV), xb.to(DEV)).float().cpu()
    # (same kernels on channel slices of the concatenated tensor: equal up to the order of a few fp32 atomic sums)
    assert_close(got, direct, 1e-5, 1e-5 * float(direct.abs().max()), "wrapper vs direct call")
    sd = O.synth_state_dict(load_json("g5_state_shapes.json")["tiny"], seed=0)
    with torch.no_grad():
        ref = O.xfmamba_top_ref(sd, xa, xb, False, c_scan.selective_scan_c).float()
    assert_close(got, ref, 1e-3, 1e-3 * float(ref.abs().max()), "ModelWrapper logits")
