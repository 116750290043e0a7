"""CPU checks of the host-side module layer: state_dict compatibility with the reference
(key names and shapes recorded from the real reference in tests/golden/g5_state_shapes.json),
checkpoint-format tolerance, and that the modules refuse to run without the HIP path."""
import pytest
import torch

from tests.helpers import load_json
from xfmamba_amd.fusion_vmamba import (Backbone_VSSM, CSSFVSSLayer_v5, Linear2d, ShallowFusionBlock_v4, SS2Dv2,
                                       VSSBlock, LayerNorm2d)
from xfmamba_amd.net_fusionmamba import TwoViewXFMambaTop


@pytest.mark.parametrize("ty,kw", [("tiny", {}), ("small", {}), ("base", dict(hidden_dim=1024))])
def test_state_dict_keys_and_shapes_match_reference(ty, kw):
    shapes = load_json("g5_state_shapes.json")
    m = TwoViewXFMambaTop(in_channels=1, outputs=2, type=ty, **kw)
    sd = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert sd == shapes[ty]
    assert sum(p.numel() for p in m.parameters()) == shapes[ty + "_nparams"]


def test_reference_constructor_defaults():
    m = TwoViewXFMambaTop(in_channels=1, outputs=3)        # default type is 'small'
    assert len(m.mamba_feature_extrac.layers[2].blocks) == 15
    assert m.classifier.head.out_features == 3
    with pytest.raises(AssertionError):
        TwoViewXFMambaTop(in_channels=3, outputs=2)
    blk = ShallowFusionBlock_v4(hidden_dim=32)             # reference default d_state=4
    assert blk.shallowfuseSS2D.A_logs.shape == (2 * 64, 4)
    deep = CSSFVSSLayer_v5(hidden_dim=32, depth=1, drop_path=[0.0], attention_downsampling=4)
    assert deep.blocks[0].self_attention.in_proj.weight.shape == (128, 32)


def test_init_statistics_follow_mamba_init():
    torch.manual_seed(0)
    op = SS2Dv2(d_model=64, d_state=1, ssm_ratio=1.0, forward_type="v05_noz", channel_first=True, conv_bias=False)
    assert torch.all(op.A_logs == 0) and torch.all(op.Ds == 1)          # A = -1 for d_state 1
    dt = torch.nn.functional.softplus(op.dt_projs_bias)
    assert float(dt.min()) >= 1e-4 - 1e-7 and float(dt.max()) <= 0.1 + 1e-6
    assert float(op.dt_projs_weight.abs().max()) <= op.dt_rank ** -0.5 + 1e-7
    assert not hasattr(op.conv2d, "bias") or op.conv2d.bias is None
    op16 = SS2Dv2(d_model=32, d_state=16, forward_type="v05_noz", channel_first=True)
    assert torch.allclose(op16.A_logs[5], torch.log(torch.arange(1, 17.0)))


def test_linear2d_loads_conv_shaped_weights_and_legacy_names():
    lin = Linear2d(4, 6, bias=False)
    lin.load_state_dict({"weight": torch.ones(6, 4, 1, 1)})
    assert lin.weight.shape == (6, 4)
    bb = Backbone_VSSM(depths=[1, 1, 1, 1], dims=16, drop_path_rate=0.0, ssm_ratio=1.0)
    sd = bb.state_dict()
    legacy = {}
    for k, v in sd.items():
        k = k.replace("layers.0.blocks.0.norm.", "layers.0.blocks.0.ln_1.")
        k = k.replace("layers.0.blocks.0.op.", "layers.0.blocks.0.self_attention.")
        legacy[k] = v
    missing = bb.load_state_dict(legacy, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys


def test_unsupported_configurations_raise():
    with pytest.raises(NotImplementedError):
        SS2Dv2(d_model=16, forward_type="v052dc_noz", channel_first=True)
    with pytest.raises(NotImplementedError):
        VSSBlock(hidden_dim=16, norm_layer=LayerNorm2d, channel_first=True, forward_type="v05_noz", gmlp=True)


def test_modules_refuse_cpu_tensors():
    m = TwoViewXFMambaTop(in_channels=1, outputs=2, type="tiny").eval()
    with pytest.raises(RuntimeError, match="no CPU path"), torch.no_grad():
        m(torch.randn(1, 1, 64, 64), torch.randn(1, 1, 64, 64))


@pytest.mark.parametrize("in_tokens,out_tokens", [(False, False), (True, False), (False, True), (True, True)])
@pytest.mark.parametrize("bias", [False, True])
def test_batched_proj_layouts(in_tokens, out_tokens, bias):
    """1x1 projection with token- / plane-major operands on either side == einsum over channels (value and grads)."""
    from xfmamba_amd.proj import batched_proj
    g = torch.Generator().manual_seed(3)
    B, L, K, M = 3, 10, 6, 5
    xp = torch.randn(B, K, L, generator=g, dtype=torch.float64)
    w = torch.randn(M, K, generator=g, dtype=torch.float64, requires_grad=True)
    bb = torch.randn(M, generator=g, dtype=torch.float64, requires_grad=True) if bias else None
    x = (xp.transpose(1, 2).contiguous() if in_tokens else xp.clone()).requires_grad_()
    y = batched_proj(x, w, bb, in_tokens=in_tokens, out_tokens=out_tokens)
    xr = xp.clone().requires_grad_()
    wr = w.detach().clone().requires_grad_()
    br = bb.detach().clone().requires_grad_() if bias else None
    yr = torch.einsum("mk,bkl->bml", wr, xr) + (br[None, :, None] if bias else 0)
    gy = torch.randn(B, M, L, generator=g, dtype=torch.float64)
    yr.backward(gy)
    y.backward(gy.transpose(1, 2) if out_tokens else gy)
    assert y.shape == ((B, L, M) if out_tokens else (B, M, L)) and y.is_contiguous()
    torch.testing.assert_close(y.transpose(1, 2) if out_tokens else y, yr)
    torch.testing.assert_close(x.grad.transpose(1, 2) if in_tokens else x.grad, xr.grad)
    torch.testing.assert_close(w.grad, wr.grad)
    if bias:
        torch.testing.assert_close(bb.grad, br.grad)


def test_drop_path_sample_scale_matches_forward_distribution():
    from xfmamba_amd.fusion_vmamba import DropPath
    dp = DropPath(0.25).train()
    s = dp.sample_scale(4000, "cpu")
    assert all(min(abs(v), abs(v - 1 / 0.75)) < 1e-6 for v in s.unique().tolist()) and abs(float(s.mean()) - 1.0) < 0.05
    assert dp.eval().sample_scale(8, "cpu") is None and DropPath(0.0).train().sample_scale(8, "cpu") is None


def test_weight_cache_serves_current_shadows_only():
    """bf16 shadows are used while the parameter is unchanged since refresh(); any in-place update invalidates them."""
    from xfmamba_amd.amp import WeightCache, cast_weight
    lin = torch.nn.Linear(6, 4)
    assert cast_weight(lin.weight, torch.bfloat16).data_ptr() != cast_weight(lin.weight, torch.bfloat16).data_ptr()
    cache = WeightCache(lin)
    a = cast_weight(lin.weight, torch.bfloat16)
    assert a.data_ptr() == cast_weight(lin.weight, torch.bfloat16).data_ptr() and a.dtype == torch.bfloat16
    assert torch.equal(a, lin.weight.detach().to(torch.bfloat16))
    v = cast_weight(lin.weight.reshape(2, 12), torch.bfloat16)                   # reshaped view of the parameter
    assert v.shape == (2, 12) and v.data_ptr() == a.data_ptr()
    with torch.no_grad():
        lin.weight.add_(1.0)                                                    # optimizer-style in-place update
    stale = cast_weight(lin.weight, torch.bfloat16)
    assert stale.data_ptr() != a.data_ptr() and torch.equal(stale, lin.weight.detach().to(torch.bfloat16))
    cache.refresh()
    assert torch.equal(cast_weight(lin.weight, torch.bfloat16), lin.weight.detach().to(torch.bfloat16))
    assert cast_weight(lin.weight, torch.bfloat16).data_ptr() == a.data_ptr()
    cache.close()
    assert cast_weight(lin.weight, torch.bfloat16).data_ptr() != a.data_ptr()


def test_weight_cache_data_writes_need_explicit_invalidation():
    """amp.WeightCache: a shadow is served while the parameter is unchanged, dropped after an in-place update of the
    parameter, and -- the `.data` caveat -- must be invalidated explicitly after a write through ``p.data``."""
    from xfmamba_amd.amp import WeightCache, cast_weight, invalidate_shadows
    m = torch.nn.Linear(8, 4)
    wc = WeightCache(m, torch.bfloat16)
    s = cast_weight(m.weight, torch.bfloat16)
    assert s.data_ptr() == wc.shadows[0].data_ptr()                       # served from the cache
    assert cast_weight(m.weight.view(2, 16), torch.bfloat16).data_ptr() == s.data_ptr()     # reshaped view too
    with torch.no_grad():
        m.weight.fill_(3.0)                                               # version bump -> fresh cast
    assert float(cast_weight(m.weight, torch.bfloat16).float().mean()) == 3.0
    wc.refresh()
    assert cast_weight(m.weight, torch.bfloat16).data_ptr() == wc.shadows[0].data_ptr()
    m.weight.data.fill_(7.0)                                              # invisible to the version check ...
    invalidate_shadows(m)                                                 # ... so writers through .data must invalidate
    assert float(cast_weight(m.weight, torch.bfloat16).float().mean()) == 7.0
    wc.refresh()
    assert float(cast_weight(m.weight, torch.bfloat16).float().mean()) == 7.0
    wc.close()


def test_mark_current_leaves_unwritten_parameters_to_the_version_check():
    """``mark_current(written)`` re-registers only the shadows the optimizer kernel wrote; a parameter
    changed in place that the step did not touch keeps its stale-shadow protection."""
    from xfmamba_amd.amp import WeightCache, cast_weight
    m = torch.nn.Linear(8, 4)
    wc = WeightCache(m, torch.bfloat16)
    with torch.no_grad():
        m.bias.add_(1.0)                                                  # e.g. load_state_dict / broadcast
    wc.mark_current([m.weight])                                           # the step wrote the weight only
    assert torch.equal(cast_weight(m.bias, torch.bfloat16), m.bias.detach().to(torch.bfloat16))
    assert cast_weight(m.weight, torch.bfloat16).data_ptr() == wc.shadows[0].data_ptr()
    wc.close()


def test_wgrad_arena_scratch_region_is_sized_by_the_previous_step():
    """proj.zeros_f32: pieces of the arena's scratch region (one zero fill per step for every fp32 accumulator); the first
    step only records its demand and falls back to torch.zeros, the next one is served from the grown buffer."""
    from xfmamba_amd.proj import WgradArena, set_wgrad_arena, zeros_f32
    arena = WgradArena([])
    set_wgrad_arena(arena)
    try:
        dev = arena.device
        arena.zero()
        a = zeros_f32(100, dev)
        b = zeros_f32(30, dev)
        assert a.untyped_storage().data_ptr() != arena.buf.untyped_storage().data_ptr()     # fallback tensors
        assert arena.scratch_need == 128 + 64
        arena.zero()                                                                        # grows the region
        assert arena.scratch_cap >= 192
        a, b = zeros_f32(100, dev), zeros_f32(30, dev)
        base = arena.buf.data_ptr()
        assert a.data_ptr() == base + 4 * arena.n_slots and b.data_ptr() == a.data_ptr() + 4 * 128
        assert a.numel() == 100 and b.numel() == 30 and float(a.abs().sum() + b.abs().sum()) == 0.0
        a.fill_(3.0)
        big = zeros_f32(10_000, dev)                                                       # beyond the region: fresh tensor
        assert big.untyped_storage().data_ptr() != arena.buf.untyped_storage().data_ptr()
        arena.zero()
        assert float(zeros_f32(100, dev).abs().sum()) == 0.0 and arena.scratch_cap >= 10_000
    finally:
        set_wgrad_arena(None)
