"""Generate the golden vectors under ``tests/golden/`` by running the REAL reference.

TEST INFRASTRUCTURE.  Runs only in the build container (needs ``/root/reference``,
imported unmodified through ``oracle/_refshim.py``); the GPU box and the test-suite
only ever see the ``.npz`` / ``.json`` files this script writes -- plain arrays, no
pickled reference objects.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden [--only g1,g2,...]

Groups:
  g1  selective_scan_fn (CPU path = selective_scan_torch, models/csms6s.py:25-68,112-126):
      forward + all 7 grads through autograd
  g2  cross_scan_fn / cross_merge_fn (CrossScanF / CrossMergeF, models/csm_triton.py:182-273)
  g3  SwappingScan_multiview / SwappingMerge_multiview (models/fusion_vmamba.py:189-241)
  g4  SS2Dv2, VSSBlock, ShallowFusionBlock_v4 (train+eval), FusionBlock_v5 at tiny dims
  g5  TwoViewXFMambaTop(type='tiny') at 2x224^2, batch 2, synthetic weights from
      ``xfm_oracle.synth_state_dict``: logits, CE loss, per-parameter gradient statistics;
      plus the state_dict key->shape tables of the tiny/small/base models
  g5s TwoViewXFMambaTop(type='small') at 2x224^2, batch 2 (BASELINE configs[2], net_fusionmamba.py:151-153)
  g5b TwoViewXFMambaTop(type='base', hidden_dim=1024) at 2x384^2, batch 1 (BASELINE configs[3],
      net_fusionmamba.py:154-156, 1_train_model.py:127): same record as g5
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import _refshim  # noqa: E402
from oracle.xfm_oracle import synth_state_dict  # noqa: E402
from oracle.golden_inputs import G1_CASES, g1_inputs, g5_inputs  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _np(t):
    t = t.detach()
    if t.dtype == torch.bfloat16:
        return t.float().numpy()
    return t.numpy()


# ------------------------------------------------------------------------------------ g1
def gen_g1(cs):
    store = {}
    for case in G1_CASES:
        name = case[0]
        inp = g1_inputs(case)
        leaves = {k: (v.clone().requires_grad_() if isinstance(v, torch.Tensor) and k != "dout" else v)
                  for k, v in inp.items()}
        y = cs.selective_scan_fn(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"],
                                 leaves["D"], leaves["delta_bias"], inp["softplus"], True, None)
        assert y.dtype == torch.float32
        y.backward(inp["dout"])
        for k in ("u", "delta", "A", "B", "C", "D", "delta_bias", "dout"):
            if inp[k] is not None:
                store[f"{name}/in/{k}"] = _np(inp[k])
        store[f"{name}/y"] = _np(y)
        for k in ("u", "delta", "A", "B", "C", "D", "delta_bias"):
            if leaves[k] is not None:
                store[f"{name}/d{k}"] = _np(leaves[k].grad)
        print("g1", name, tuple(y.shape), float(y.abs().max()))
    np.savez_compressed(os.path.join(OUT, "g1_scan.npz"), **store)


# ------------------------------------------------------------------------------------ g2/g3
def gen_g2(ct):
    store = {}
    for name, shp in (("a", (2, 3, 5, 7)), ("b", (1, 4, 12, 12))):
        g = torch.Generator().manual_seed(1)
        x = torch.randn(*shp, generator=g).requires_grad_()
        ys = ct.cross_scan_fn(x)
        gy = torch.randn(*ys.shape, generator=g)
        ys.backward(gy)
        Bt, C, H, W = shp
        yin = torch.randn(Bt, 4, C, H, W, generator=g).requires_grad_()
        m = ct.cross_merge_fn(yin)
        gm = torch.randn(*m.shape, generator=g)
        m.backward(gm)
        for k, v in dict(x=x, scan=ys, gscan=gy, dx=x.grad, yin=yin, merge=m, gmerge=gm, dyin=yin.grad).items():
            store[f"{name}/{k}"] = _np(v)
    np.savez_compressed(os.path.join(OUT, "g2_cross.npz"), **store)
    print("g2 done")


def gen_g3(fv):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 6, 3, 3, generator=g).requires_grad_()
    x2 = torch.randn(2, 6, 3, 3, generator=g).requires_grad_()
    xs = fv.SwappingScan_multiview.apply(x, x2)
    gy = torch.randn(*xs.shape, generator=g)
    xs.backward(gy)
    ys = torch.randn(2, 2, 6, 9, generator=g).requires_grad_()
    o1, o2 = fv.SwappingMerge_multiview.apply(ys)
    g1, g2 = torch.randn(*o1.shape, generator=g), torch.randn(*o2.shape, generator=g)
    torch.autograd.backward([o1, o2], [g1, g2])
    store = dict(x=x, x2=x2, swap=xs, gswap=gy, dx=x.grad, dx2=x2.grad, ys=ys, o1=o1, o2=o2, g1=g1, g2=g2,
                 dys=ys.grad)
    np.savez_compressed(os.path.join(OUT, "g3_swap.npz"), **{k: _np(v) for k, v in store.items()})
    print("g3 done")


# ------------------------------------------------------------------------------------ g4
def _block_record(store, tag, mod, inputs, training):
    mod.train(training)
    ins = [t.clone().requires_grad_() for t in inputs]
    sd_before = {k: v.clone() for k, v in mod.state_dict().items()}
    out = mod(*ins)
    outs = out if isinstance(out, (tuple, list)) else (out,)
    g = torch.Generator().manual_seed(7)
    gos = [torch.randn(*o.shape, generator=g) for o in outs]
    mod.zero_grad()
    torch.autograd.backward(list(outs), gos)
    for k, v in sd_before.items():
        store[f"{tag}/sd/{k}"] = _np(v)
    for i, t in enumerate(inputs):
        store[f"{tag}/in{i}"] = _np(t)
        store[f"{tag}/din{i}"] = _np(ins[i].grad)
    for i, o in enumerate(outs):
        store[f"{tag}/out{i}"] = _np(o)
        store[f"{tag}/gout{i}"] = _np(gos[i])
    for k, p in mod.named_parameters():
        if p.grad is not None:
            store[f"{tag}/grad/{k}"] = _np(p.grad)
    for k, v in mod.state_dict().items():
        if "running" in k or "num_batches" in k:
            store[f"{tag}/sd_after/{k}"] = _np(v)
    print("g4", tag, [tuple(o.shape) for o in outs])


def gen_g4(fv):
    store = {}
    g = torch.Generator().manual_seed(3)
    torch.manual_seed(3)
    m = fv.SS2Dv2(d_model=16, d_state=1, ssm_ratio=1.0, dt_rank="auto", conv_bias=False,
                  forward_type="v05_noz", channel_first=True)
    _block_record(store, "ss2dv2", m, [torch.randn(2, 16, 7, 7, generator=g)], False)
    m = fv.SS2Dv2(d_model=16, d_state=1, ssm_ratio=2.0, dt_rank="auto", conv_bias=False,
                  forward_type="v05_noz", channel_first=True)
    _block_record(store, "ss2dv2_r2", m, [torch.randn(1, 16, 6, 10, generator=g)], False)
    m = fv.VSSBlock(hidden_dim=16, drop_path=0.0, norm_layer=fv.LayerNorm2d, channel_first=True,
                    ssm_d_state=1, ssm_ratio=1.0, ssm_dt_rank="auto", ssm_conv=3, ssm_conv_bias=False,
                    ssm_init="v0", forward_type="v05_noz", mlp_ratio=4.0)
    _block_record(store, "vssblock", m, [torch.randn(2, 16, 7, 7, generator=g)], True)
    m = fv.ShallowFusionBlock_v4(hidden_dim=32, d_state=16)
    xin = [torch.randn(2, 32, 5, 5, generator=g), torch.randn(2, 32, 5, 5, generator=g)]
    _block_record(store, "shallow_train", m, xin, True)
    _block_record(store, "shallow_eval", m, xin, False)
    m = fv.FusionBlock_v5(hidden_dim=32, drop_path=0.0, norm_layer=fv.LayerNorm2d, attn_drop_rate=0.0, d_state=16)
    _block_record(store, "deep", m, xin, True)
    np.savez_compressed(os.path.join(OUT, "g4_blocks.npz"), **store)


# ------------------------------------------------------------------------------------ g5
def gen_g5(net, fv):
    shapes = {}
    for ty, kw in (("tiny", {}), ("small", {}), ("base", dict(hidden_dim=1024))):
        m = net.TwoViewXFMambaTop(in_channels=1, outputs=2, type=ty, **kw)
        shapes[ty] = {k: list(v.shape) for k, v in m.state_dict().items()}
        shapes[ty + "_nparams"] = sum(p.numel() for p in m.parameters())
        if ty != "tiny":
            del m
        else:
            tiny = m
    json.dump(shapes, open(os.path.join(OUT, "g5_state_shapes.json"), "w"), indent=0, sort_keys=True)
    sd = synth_state_dict(shapes["tiny"], seed=0)
    tiny.load_state_dict(sd, strict=True)
    xa, xb, lab = g5_inputs()
    tiny.eval()
    with torch.no_grad():
        logits_eval = tiny(xa, xb)
        feats = tiny.mamba_feature_extrac(xa.expand(-1, 3, -1, -1))
    print("g5 eval logits", logits_eval)
    # training step semantics (1_train_model.py:134-141, libs/training.py:188-194) with DropPath p=0
    tiny.train()
    for mod in tiny.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    tiny.zero_grad()
    logits_tr = tiny(xa, xb)
    loss = torch.nn.functional.cross_entropy(logits_tr, lab)
    loss.backward()
    store = dict(logits_eval=_np(logits_eval), logits_train=_np(logits_tr), loss=_np(loss),
                 feat3_sample=_np(feats[3][:, :8]), feat0_sample=_np(feats[0][:, :4, :8, :8]))
    names, stats = [], []
    nograd = []
    for k, p in tiny.named_parameters():
        if p.grad is None:
            nograd.append(k)
            continue
        gk = p.grad.double()
        names.append(k)
        stats.append([float(gk.sum()), float(gk.abs().sum()), float(gk.norm()), float(gk.reshape(-1)[0])])
    store["grad_stats"] = np.asarray(stats, dtype=np.float64)
    for k in ("classifier.head.weight", "final_conv.bias",
              "fusemamba.blocks.0.self_attention.dt_projs_bias",
              "shallow_mamba_fusion.shallowfuseSS2D.fc1.0.weight",
              "mamba_feature_extrac.layers.0.blocks.0.op.x_proj_weight",
              "mamba_feature_extrac.layers.0.blocks.0.op.dt_projs_weight",
              "mamba_feature_extrac.layers.0.blocks.0.op.A_logs",
              "mamba_feature_extrac.layers.3.blocks.1.op.Ds",
              "mamba_feature_extrac.patch_embed.0.weight"):
        store["grad/" + k] = _np(dict(tiny.named_parameters())[k].grad)
    for k, v in tiny.state_dict().items():
        if "running" in k:
            store["bn_after/" + k] = _np(v)
    np.savez_compressed(os.path.join(OUT, "g5_model.npz"), **store)
    json.dump(dict(grad_names=names, no_grad=nograd), open(os.path.join(OUT, "g5_grad_names.json"), "w"), indent=0)
    print("g5 train logits", logits_tr, "loss", float(loss), "no-grad params", nograd)


_G5X_GRADS = ("classifier.head.weight", "final_conv.bias",
              "fusemamba.blocks.0.self_attention.dt_projs_bias",
              "fusemamba.blocks.0.self_attention.x_proj_weight",
              "shallow_mamba_fusion.shallowfuseSS2D.fc1.0.weight",
              "mamba_feature_extrac.layers.0.blocks.0.op.x_proj_weight",
              "mamba_feature_extrac.layers.0.blocks.0.op.dt_projs_weight",
              "mamba_feature_extrac.layers.1.blocks.1.op.A_logs",
              "mamba_feature_extrac.layers.2.blocks.7.op.dt_projs_bias",
              "mamba_feature_extrac.layers.2.blocks.14.op.out_norm.weight",
              "mamba_feature_extrac.layers.3.blocks.1.op.Ds",
              "mamba_feature_extrac.patch_embed.0.weight")


def gen_g5x(net, ty, kw, size, batch, tag):
    """Same record as g5 for the other BASELINE model sizes (weights from synth_state_dict, DropPath off)."""
    shapes = json.load(open(os.path.join(OUT, "g5_state_shapes.json")))[ty]
    m = net.TwoViewXFMambaTop(in_channels=1, outputs=2, type=ty, **kw)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == shapes
    m.load_state_dict(synth_state_dict(shapes, seed=0), strict=True)
    xa, xb, lab = g5_inputs(batch, size)
    m.eval()
    with torch.no_grad():
        logits_eval = m(xa, xb)
    print(tag, "eval logits", logits_eval)
    m.train()
    for mod in m.modules():
        if hasattr(mod, "drop_prob"):
            mod.drop_prob = 0.0
    m.zero_grad()
    logits_tr = m(xa, xb)
    loss = torch.nn.functional.cross_entropy(logits_tr, lab)
    loss.backward()
    store = dict(logits_eval=_np(logits_eval), logits_train=_np(logits_tr), loss=_np(loss))
    names, stats, nograd = [], [], []
    params = dict(m.named_parameters())
    for k, p in params.items():
        if p.grad is None:
            nograd.append(k)
            continue
        gk = p.grad.double()
        names.append(k)
        stats.append([float(gk.sum()), float(gk.abs().sum()), float(gk.norm()), float(gk.reshape(-1)[0])])
    store["grad_stats"] = np.asarray(stats, dtype=np.float64)
    for k in _G5X_GRADS:
        store["grad/" + k] = _np(params[k].grad)
    for k, v in m.state_dict().items():
        if "running" in k:
            store["bn_after/" + k] = _np(v)
    np.savez_compressed(os.path.join(OUT, f"{tag}_model.npz"), **store)
    json.dump(dict(grad_names=names, no_grad=nograd, type=ty, kwargs=kw, size=size, batch=batch),
              open(os.path.join(OUT, f"{tag}_grad_names.json"), "w"), indent=0)
    print(tag, "train logits", logits_tr, "loss", float(loss), "no-grad params", nograd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="g1,g2,g3,g4,g5,g5s,g5b")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    cs, ct, fv, net = _refshim.load()
    only = a.only.split(",")
    if "g1" in only:
        gen_g1(cs)
    if "g2" in only:
        gen_g2(ct)
    if "g3" in only:
        gen_g3(fv)
    if "g4" in only:
        gen_g4(fv)
    if "g5" in only:
        gen_g5(net, fv)
    if "g5s" in only:
        gen_g5x(net, "small", {}, 224, 2, "g5s")
    if "g5b" in only:
        gen_g5x(net, "base", dict(hidden_dim=1024), 384, 1, "g5b")


if __name__ == "__main__":
    main()
