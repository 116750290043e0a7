"""Shared helpers for the parity tests."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_DT = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def load_json(name):
    return json.load(open(os.path.join(GOLDEN, name)))


def g1_case_tensors(z, case):
    """Rebuild the typed input tensors of a g1 case from the stored fp32 arrays."""
    name, dt = case[0], _DT[case[10]]
    t = {}
    for k in ("u", "delta", "A", "B", "C", "D", "delta_bias", "dout"):
        key = f"{name}/in/{k}"
        t[k] = torch.from_numpy(z[key]) if key in z.files else None
    for k in ("u", "delta", "B", "C"):
        t[k] = t[k].to(dt)
    return t


def max_rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def assert_close(a, b, rtol, atol, what=""):
    a, b = a.double(), b.double()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bool(bad.any()), f"{what}: max abs err {float(err.max()):.3e} (ref max {float(b.abs().max()):.3e}), " \
                                f"{int(bad.sum())}/{bad.numel()} outside rtol={rtol} atol={atol}"
