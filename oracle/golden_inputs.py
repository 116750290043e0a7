"""Seeded input builders shared by ``oracle/make_golden.py`` and the tests.

TEST INFRASTRUCTURE.  Pure torch/numpy, no reference import.
"""
import numpy as np
import torch

G1_CASES = [
    # (name, B, K, Dg, N, L, dist, softplus, has_D, has_bias, dtype)
    ("n1_l49", 2, 4, 8, 1, 49, "test", True, True, True, "f32"),
    ("n1_l196", 2, 4, 8, 1, 196, "test", True, True, True, "f32"),
    ("n1_l784", 1, 4, 6, 1, 784, "model", True, True, True, "f32"),
    ("n16_k2_l49", 2, 2, 16, 16, 49, "test", True, True, True, "f32"),
    ("n16_k4_l49", 2, 4, 16, 16, 49, "model", True, True, True, "f32"),
    ("n16_l144", 1, 4, 8, 16, 144, "test", True, True, True, "f32"),
    ("n8_l130_ragged", 1, 4, 4, 8, 130, "test", True, True, True, "f32"),
    ("n8_nosoftplus", 2, 2, 4, 8, 64, "test", False, True, True, "f32"),
    ("n8_noD_nobias", 2, 2, 4, 8, 64, "test", True, False, False, "f32"),
    ("n4_bigdelta", 1, 2, 4, 4, 40, "big", True, True, True, "f32"),
    ("n1_l3136", 1, 4, 2, 1, 3136, "model", True, True, True, "f32"),
    ("n1_l196_bf16", 2, 4, 8, 1, 196, "test", True, True, True, "bf16"),
    ("n16_l49_bf16", 2, 4, 16, 16, 49, "model", True, True, True, "bf16"),
    ("n8_l372_f16", 2, 2, 12, 8, 372, "test", True, True, True, "f16"),
]


def g1_inputs(case, seed=0):
    """Input distributions follow the reference's own test (test_selective_scan.py:157-179):
    A=-0.5*rand, u,B,C~randn, delta=0.5*rand, delta_bias=0.5*rand, D~randn.  'model' draws
    A=-(1..N), delta pre-activations near softplus^-1([1e-3,1e-1]); 'big' pushes delta+bias
    past the softplus threshold 20."""
    name, Bt, K, Dg, N, L, dist, softplus, has_D, has_bias, dt = case
    g = torch.Generator().manual_seed(seed)
    KD = K * Dg
    r = lambda *s: torch.rand(*s, generator=g)
    n = lambda *s: torch.randn(*s, generator=g)
    u, Bm, Cm = n(Bt, KD, L), n(Bt, K, N, L), n(Bt, K, N, L)
    if dist == "test":
        A, delta, bias = -0.5 * r(KD, N), 0.5 * r(Bt, KD, L), 0.5 * r(KD)
    elif dist == "model":
        A = -torch.arange(1, N + 1, dtype=torch.float32).repeat(KD, 1) * (1 + 0.05 * n(KD, N))
        tgt = torch.exp(r(KD) * (np.log(0.1) - np.log(0.001)) + np.log(0.001))
        bias = tgt + torch.log(-torch.expm1(-tgt))
        delta = 0.5 * n(Bt, KD, L)
    else:  # big
        A, delta, bias = -0.05 * r(KD, N), 15.0 + 10.0 * r(Bt, KD, L), 2.0 * r(KD)
    D = n(KD)
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[dt]
    u, delta, Bm, Cm = (t.to(tdt) for t in (u, delta, Bm, Cm))
    dout = n(Bt, KD, L)
    return dict(u=u, delta=delta, A=A, B=Bm, C=Cm, D=D if has_D else None,
                delta_bias=bias if has_bias else None, dout=dout, softplus=softplus)



def g5_inputs(batch=2, size=224):
    rng = np.random.default_rng(1234)
    xa = torch.from_numpy(rng.standard_normal((batch, 1, size, size)).astype(np.float32))
    xb = torch.from_numpy(rng.standard_normal((batch, 1, size, size)).astype(np.float32))
    lab = torch.from_numpy(rng.integers(0, 2, (batch,)))
    return xa, xb, lab


