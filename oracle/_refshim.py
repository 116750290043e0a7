"""Import harness for the upstream XFMamba reference -- GOLDEN GENERATION ONLY.

TEST INFRASTRUCTURE.  Used by ``oracle/make_golden.py`` inside the build
container, where ``/root/reference`` is mounted read-only.  Nothing here is
imported by the product (``xfmamba_amd``), by ``bench.py`` or by the ``-m gpu``
tests: the reference never travels to the GPU box, only the ``.npz`` vectors
this harness helps to produce do.

The reference cannot be imported as-is in this image:
  * ``timm``, ``fvcore``, ``torchinfo``, ``torchvision`` are absent, so their
    names are pre-seeded in ``sys.modules`` (``DropPath`` = standard
    stochastic depth, ``trunc_normal_`` = ``torch.nn.init.trunc_normal_``);
  * ``cross_scan_fn``/``cross_merge_fn`` wrap their body in
    ``torch.cuda.device(x.device)`` which raises for CPU tensors
    (``models/csm_triton.py:506,516``) -- replaced by a no-op context.
No reference file is modified or copied.
"""
import contextlib
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("XFM_REFERENCE_ROOT", "/root/reference")


class _DropPath(nn.Module):
    """timm.models.layers.DropPath semantics (per-sample stochastic depth)."""

    def __init__(self, drop_prob: float = 0.0, scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = x.new_empty(shape).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True  # /root/reference is read-only
    _stub("timm")
    _stub("timm.models")
    _stub("timm.models.layers", DropPath=_DropPath,
          trunc_normal_=torch.nn.init.trunc_normal_)
    _noop = lambda *a, **k: None
    _stub("fvcore")
    _stub("fvcore.nn", FlopCountAnalysis=_noop, flop_count_str=_noop,
          flop_count=_noop, parameter_count=_noop)
    _stub("torchinfo", summary=_noop)
    _stub("torchvision")
    _stub("torchvision.models")
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]

    @contextlib.contextmanager
    def _nodev(*a, **k):
        yield

    torch.cuda.device = _nodev
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def load():
    """Returns (csms6s, csm_triton, fusion_vmamba, net_fusionmamba) reference modules."""
    install()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with contextlib.redirect_stdout(open(os.devnull, "w")):
            fv = importlib.import_module("models.fusion_vmamba")
            cs = importlib.import_module("models.csms6s")
            ct = importlib.import_module("models.csm_triton")
            net = importlib.import_module("net_fusionmamba")
    return cs, ct, fv, net
