"""Pins the CPU oracle (oracle/xfm_oracle.py, oracle/scan_oracle.c) to golden vectors that
were produced by importing the REAL reference (oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import c_scan
from oracle import xfm_oracle as O
from oracle.golden_inputs import G1_CASES, g5_inputs
from tests.helpers import assert_close, g1_case_tensors, load_json, load_npz

GRADS = ("u", "delta", "A", "B", "C", "D", "delta_bias")


@pytest.fixture(scope="module")
def g1():
    return load_npz("g1_scan.npz")


@pytest.mark.parametrize("case", G1_CASES, ids=[c[0] for c in G1_CASES])
def test_scan_torch_restatement_matches_reference(g1, case):
    name = case[0]
    if case[5] > 800:
        pytest.skip("long row: covered by the C oracle test")
    t = g1_case_tensors(g1, case)
    leaves = {k: (t[k].clone().requires_grad_() if t[k] is not None else None) for k in GRADS}
    y = O.selective_scan_ref(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], leaves["D"],
                             leaves["delta_bias"], case[7], True)
    assert_close(y, torch.from_numpy(g1[f"{name}/y"]), 1e-5, 1e-5, "y")
    y.backward(t["dout"])
    for k in GRADS:
        if leaves[k] is not None:
            ref = torch.from_numpy(g1[f"{name}/d{k}"])
            scale = float(ref.abs().max()) + 1e-6
            tol = 2e-2 if case[10] != "f32" else 2e-5       # 16-bit grads are rounded to the input dtype
            assert_close(leaves[k].grad.float(), ref, tol, tol * scale, "d" + k)


@pytest.mark.parametrize("case", G1_CASES, ids=[c[0] for c in G1_CASES])
def test_scan_c_oracle_matches_reference(g1, case):
    name = case[0]
    t = g1_case_tensors(g1, case)
    y = c_scan.scan_fwd_c(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["delta_bias"], case[7])
    ref = torch.from_numpy(g1[f"{name}/y"])
    assert_close(y, ref, 2e-5, 2e-5 * float(ref.abs().max()), "y")
    grads = c_scan.scan_bwd_c(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["delta_bias"], t["dout"], case[7])
    for k, gk in zip(GRADS, grads):
        if gk is None:
            continue
        ref = torch.from_numpy(g1[f"{name}/d{k}"])
        scale = float(ref.abs().max()) + 1e-6
        tol = 1e-2 if case[10] != "f32" else 5e-5          # reference rounds 16-bit grads to the input dtype
        assert_close(gk, ref, tol, tol * scale, "d" + k)


def test_scan_closed_form_bwd_matches_c(g1):
    case = G1_CASES[3]
    t = g1_case_tensors(g1, case)
    a = O.selective_scan_bwd_ref(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["delta_bias"], t["dout"], True)
    b = c_scan.scan_bwd_c(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], t["delta_bias"], t["dout"], True)
    for x, y, k in zip(a, b, GRADS):
        assert_close(y, x.float(), 1e-5, 1e-5 * float(x.abs().max()), k)


def test_cross_scan_merge():
    z = load_npz("g2_cross.npz")
    for n in ("a", "b"):
        x = torch.from_numpy(z[f"{n}/x"]).requires_grad_()
        ys = O.cross_scan_ref(x)
        assert torch.equal(ys, torch.from_numpy(z[f"{n}/scan"]))
        ys.backward(torch.from_numpy(z[f"{n}/gscan"]))
        assert_close(x.grad, torch.from_numpy(z[f"{n}/dx"]), 1e-6, 1e-6)
        yin = torch.from_numpy(z[f"{n}/yin"]).requires_grad_()
        m = O.cross_merge_ref(yin)
        assert_close(m, torch.from_numpy(z[f"{n}/merge"]), 1e-6, 1e-6)
        m.backward(torch.from_numpy(z[f"{n}/gmerge"]))
        assert torch.equal(yin.grad, torch.from_numpy(z[f"{n}/dyin"]))


def test_swap_passthrough_backward():
    z = load_npz("g3_swap.npz")
    x = torch.from_numpy(z["x"]).requires_grad_()
    x2 = torch.from_numpy(z["x2"]).requires_grad_()
    xs = O.swap_scan_ref(x, x2)
    assert torch.equal(xs, torch.from_numpy(z["swap"]))
    xs.backward(torch.from_numpy(z["gswap"]))
    assert torch.equal(x.grad, torch.from_numpy(z["dx"])) and torch.equal(x2.grad, torch.from_numpy(z["dx2"]))
    o1, o2 = O.swap_merge_ref(torch.from_numpy(z["ys"]))
    assert torch.equal(o1, torch.from_numpy(z["o1"])) and torch.equal(o2, torch.from_numpy(z["o2"]))


def _sd(z, tag):
    pre = f"{tag}/sd/"
    return {k[len(pre):]: torch.from_numpy(z[k]).clone() for k in z.files if k.startswith(pre)}


BLOCKS = [("ss2dv2", "ss2d", False), ("ss2dv2_r2", "ss2d", False), ("vssblock", "vss", True),
          ("shallow_train", "shallow", True), ("shallow_eval", "shallow", False), ("deep", "deep", True)]


@pytest.mark.parametrize("tag,kind,training", BLOCKS, ids=[b[0] for b in BLOCKS])
@pytest.mark.parametrize("scan", ["torch", "c"])
def test_blocks_match_reference(tag, kind, training, scan):
    z = load_npz("g4_blocks.npz")
    scan_fn = O.selective_scan_ref if scan == "torch" else c_scan.selective_scan_c
    sd = _sd(z, tag)
    params = {k: v.requires_grad_() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    sd.update(params)
    ins = [torch.from_numpy(z[f"{tag}/in{i}"]).requires_grad_() for i in range(2) if f"{tag}/in{i}" in z.files]
    if kind == "ss2d":
        outs = (O.ss2d_v2_ref(sd, "", ins[0], scan_fn),)
    elif kind == "vss":
        outs = (O.vss_block_ref(sd, "", ins[0], scan_fn),)
    elif kind == "shallow":
        outs = O.shallow_block_ref(sd, "", ins[0], ins[1], training, scan_fn)
    else:
        outs = (O.deep_block_ref(sd, "", ins[0], ins[1], scan_fn),)
    gos = [torch.from_numpy(z[f"{tag}/gout{i}"]) for i in range(len(outs))]
    for i, o in enumerate(outs):
        assert_close(o, torch.from_numpy(z[f"{tag}/out{i}"]), 1e-4, 1e-5, f"out{i}")
    torch.autograd.backward(list(outs), gos)
    for i, t in enumerate(ins):
        ref = torch.from_numpy(z[f"{tag}/din{i}"])
        assert_close(t.grad, ref, 1e-3, 1e-4 * float(ref.abs().max()), f"din{i}")
    pre = f"{tag}/grad/"
    n = 0
    for k in z.files:
        if k.startswith(pre):
            ref = torch.from_numpy(z[k])
            assert_close(params[k[len(pre):]].grad, ref, 1e-3, 1e-4 * float(ref.abs().max()) + 1e-7, k)
            n += 1
    assert n >= 8
    # parameters the reference leaves without a gradient must stay without one here too
    for k, v in params.items():
        assert (v.grad is not None) == ((pre + k) in z.files), k
    pre = f"{tag}/sd_after/"
    for k in z.files:
        if k.startswith(pre) and "running" in k:
            assert_close(sd[k[len(pre):]], torch.from_numpy(z[k]), 1e-5, 1e-6, k)


def test_model_tiny_matches_reference_eval_and_train():
    """G5: whole TwoViewXFMambaTop(type='tiny') at 2x224^2, batch 2 (BASELINE config 0)."""
    shapes = load_json("g5_state_shapes.json")["tiny"]
    z = load_npz("g5_model.npz")
    names = load_json("g5_grad_names.json")
    sd = O.synth_state_dict(shapes, seed=0)
    xa, xb, lab = g5_inputs()
    with torch.no_grad():
        logits = O.xfmamba_top_ref(dict(sd), xa, xb, False, c_scan.selective_scan_c)
    assert_close(logits, torch.from_numpy(z["logits_eval"]), 1e-3, 1e-4, "eval logits")
    params = {k: v.requires_grad_() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    sd.update(params)
    out = O.xfmamba_top_ref(sd, xa, xb, True, c_scan.selective_scan_c)
    assert_close(out, torch.from_numpy(z["logits_train"]), 1e-3, 1e-4, "train logits")
    loss = torch.nn.functional.cross_entropy(out, lab)
    assert abs(float(loss) - float(z["loss"])) < 1e-4
    loss.backward()
    assert sorted(k for k, v in params.items() if v.grad is None) == sorted(names["no_grad"])
    stats = z["grad_stats"]
    for k, row in zip(names["grad_names"], stats):
        g = params[k].grad.double()
        assert abs(float(g.norm()) - row[2]) <= 2e-3 * row[2] + 1e-7, (k, float(g.norm()), row[2])
    for k in z.files:
        if k.startswith("grad/"):
            ref = torch.from_numpy(z[k])
            assert_close(params[k[5:]].grad, ref, 2e-3, 2e-4 * float(ref.abs().max()) + 1e-8, k)
        if k.startswith("bn_after/"):
            assert_close(sd[k[9:]], torch.from_numpy(z[k]), 1e-4, 1e-6, k)


@pytest.mark.parametrize("tag,ty", [("g5s", "small"), ("g5b", "base")])
def test_model_small_base_match_reference(tag, ty):
    """G5s / G5b: TwoViewXFMambaTop(type='small') at 2x224^2 batch 2 and (type='base', hidden_dim=1024) at 2x384^2
    batch 1 (BASELINE configs[2], [3]; net_fusionmamba.py:150-156): the oracle vs the record of the real reference."""
    shapes = load_json("g5_state_shapes.json")[ty]
    z = load_npz(f"{tag}_model.npz")
    names = load_json(f"{tag}_grad_names.json")
    sd = O.synth_state_dict(shapes, seed=0)
    xa, xb, lab = g5_inputs(names["batch"], names["size"])
    with torch.no_grad():
        logits = O.xfmamba_top_ref(dict(sd), xa, xb, False, c_scan.selective_scan_c)
    ref = torch.from_numpy(z["logits_eval"])
    assert_close(logits, ref, 1e-3, 1e-4 * float(ref.abs().max()), "eval logits")
    params = {k: v.requires_grad_() for k, v in sd.items() if v.is_floating_point() and "running" not in k}
    sd.update(params)
    out = O.xfmamba_top_ref(sd, xa, xb, True, c_scan.selective_scan_c)
    ref = torch.from_numpy(z["logits_train"])
    assert_close(out, ref, 1e-3, 1e-4 * float(ref.abs().max()), "train logits")
    loss = torch.nn.functional.cross_entropy(out, lab)
    assert abs(float(loss) - float(z["loss"])) < 1e-3
    loss.backward()
    assert sorted(k for k, v in params.items() if v.grad is None) == sorted(names["no_grad"])
    for k, row in zip(names["grad_names"], z["grad_stats"]):
        g = params[k].grad.double()
        assert abs(float(g.norm()) - row[2]) <= 2e-3 * row[2] + 1e-7, (k, float(g.norm()), row[2])
    for k in z.files:
        if k.startswith("grad/"):
            ref = torch.from_numpy(z[k])
            assert_close(params[k[5:]].grad, ref, 2e-3, 2e-4 * float(ref.abs().max()) + 1e-8, k)
