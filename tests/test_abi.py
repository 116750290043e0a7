"""CPU checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol
include/xfm_hip.h declares; the host-side operators mirror the reference API and refuse to run
without the GPU path (no CPU fallback).  No compute calls here."""
import ctypes
import inspect
import os
import re

import pytest
import torch

import xfmamba_amd
from xfmamba_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "xfm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(xfm_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    l = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert sorted(_lib.SYMBOLS) == declared
    for s in declared:
        assert hasattr(l, s), s
    assert _lib.lib().xfm_abi_version() == 2
    assert _lib.lib().xfm_strerror(0) == b"ok" and b"dtype" in _lib.lib().xfm_strerror(-2)


def test_struct_layout_matches_header():
    # field order of the ctypes mirrors == field order in the header structs
    hdr = open(os.path.join(ROOT, "include", "xfm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for struct, cls in (("xfm_scan_params_t", _lib.ScanParams), ("xfm_ss2d_params_t", _lib.SS2DParams),
                        ("xfm_scan_plan_t", _lib.ScanPlan), ("xfm_ss2dc_params_t", _lib.SS2DCParams)):
        body = hdr[:hdr.index("} " + struct)].rsplit("typedef struct {", 1)[1]
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            parts = decl.split(",")
            first = parts[0].split()
            names.append(first[-1].lstrip("*"))
            names += [q.strip().lstrip("*") for q in parts[1:]]
        assert names == [f[0] for f in cls._fields_], struct


def test_scan_plan_is_host_side_and_deterministic():
    p1 = _lib.scan_plan(64, 384, 3136, 1, 4)
    p2 = _lib.scan_plan(64, 384, 3136, 1, 4)
    assert (p1.lanes_per_row, p1.items, p1.n_chunks) == (p2.lanes_per_row, p2.items, p2.n_chunks)
    assert p1.lanes_per_row * p1.items * p1.n_chunks >= 3136
    assert 64 % p1.lanes_per_row == 0 and 96 % (64 // p1.lanes_per_row) == 0
    for (b, d, l, n, k) in [(32, 6144, 49, 16, 4), (32, 3072, 49, 16, 2), (2, 24, 4096, 8, 1), (1, 4, 7, 256, 1)]:
        p = _lib.scan_plan(b, d, l, n, k)
        assert p.lanes_per_row * p.items * p.n_chunks >= l
        assert (d // k) % (64 // p.lanes_per_row) == 0
    with pytest.raises(RuntimeError):
        _lib.scan_plan(1, 6, 10, 4, 4)      # dim % n_groups != 0
    with pytest.raises(RuntimeError):
        _lib.scan_plan(1, 8, 10, 257, 1)    # dstate above the reference's own bound


def test_operator_signatures_mirror_reference():
    sig = inspect.signature(xfmamba_amd.selective_scan_fn)
    assert list(sig.parameters) == ["u", "delta", "A", "B", "C", "D", "delta_bias", "delta_softplus", "oflex", "backend"]
    assert sig.parameters["delta_softplus"].default is True and sig.parameters["oflex"].default is True
    for fn in (xfmamba_amd.cross_scan_fn, xfmamba_amd.cross_merge_fn):
        assert list(inspect.signature(fn).parameters)[1:] == ["in_channel_first", "out_channel_first", "one_by_one",
                                                              "scans", "force_torch"]


def test_no_cpu_fallback():
    u = torch.randn(1, 8, 16)
    Bm = torch.randn(1, 2, 4, 16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        xfmamba_amd.selective_scan_fn(u, u, -torch.rand(8, 4), Bm, Bm)
    with pytest.raises(RuntimeError, match="no CPU path"):
        xfmamba_amd.cross_scan_fn(torch.randn(1, 2, 3, 3))
    with pytest.raises(NotImplementedError):
        xfmamba_amd.selective_scan_fn(u, u, -torch.rand(8, 4), Bm, Bm, backend="torch")
    with pytest.raises(NotImplementedError):
        xfmamba_amd.cross_scan_fn(torch.randn(1, 2, 3, 3), scans=1)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "xfmamba_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
                assert "scan_oracle" not in src, f


def test_graft_entry_build_checks_the_headers_abi_version():
    """``__graft_entry__.build()`` (the driver's build check) compiles the library and the oracle's C restatement and compares the
    library's ABI version with the one ``include/xfm_hip.h`` declares -- not with a literal that a struct change leaves behind."""
    import __graft_entry__ as entry
    entry.build()
