"""xfmamba_amd -- MI355X-native (gfx950) implementation of the XFMamba hot path.

Operators (reference names): ``selective_scan_fn``, ``cross_scan_fn``, ``cross_merge_fn``,
``SwappingScan_multiview``, ``SwappingMerge_multiview``; fused ``ss2d_core_fn``.
Modules (reference names): see ``xfmamba_amd.fusion_vmamba`` and ``xfmamba_amd.net_fusionmamba``.
The compute path is ``libxfm_hip.so`` (C ABI in ``include/xfm_hip.h``); there is no CPU fallback.
"""
from . import _lib
from .csm import SwappingMerge_multiview, SwappingScan_multiview, cross_merge_fn, cross_scan_fn
from .csms6s import selective_scan_fn
from .ss2d import ss2d_core_fn

__all__ = ["selective_scan_fn", "cross_scan_fn", "cross_merge_fn", "SwappingScan_multiview",
           "SwappingMerge_multiview", "ss2d_core_fn", "build"]

build = _lib.build
