"""ctypes binding of ``libxfm_hip.so`` (C ABI declared in ``include/xfm_hip.h``).

There is NO fallback: if the shared library is missing or fails to load, every operator of
this package raises.  PyTorch is used only for device memory and the current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libxfm_hip.so")

XFM_F32, XFM_F16, XFM_BF16 = 0, 1, 2
_DT = {torch.float32: XFM_F32, torch.float16: XFM_F16, torch.bfloat16: XFM_BF16}

# every symbol include/xfm_hip.h declares (checked by tests/test_abi.py)
SYMBOLS = (
    "xfm_abi_version", "xfm_strerror", "xfm_last_hip_error", "xfm_prof_main_kernel", "xfm_scan_plan",
    "xfm_selective_scan_fwd", "xfm_selective_scan_bwd", "xfm_cross_scan", "xfm_cross_merge",
    "xfm_swap_scan", "xfm_ss2d_route_split", "xfm_ss2d_route_merge", "xfm_ss2d_dt_proj_supported", "xfm_ss2d_dt_proj_mfma_rp", "xfm_ss2d_dt_proj_fwd_mfma", "xfm_ss2d_dt_proj_bwd_mfma", "xfm_ss2d_dt_proj_fwd",
    "xfm_dwconv3x3_fwd", "xfm_dwconv3x3_bwd", "xfm_dwconv3x3_tokens_supported", "xfm_dwconv3x3_tokens_fwd", "xfm_dwconv3x3_tokens_bwd", "xfm_conv3x3s2_tokens_supported", "xfm_conv3x3s2_tokens_fwd", "xfm_conv3x3s2_tokens_bwd_data", "xfm_conv3x3s2_tokens_bwd_weight", "xfm_conv3x3s2_tokens_bwd_weight_x_supported", "xfm_conv3x3s2_tokens_bwd_weight_x", "xfm_conv3x3s2_gray_supported", "xfm_conv3x3s2_gray_ws_floats", "xfm_conv3x3s2_gray_fwd", "xfm_conv3x3s2_gray_bwd_weight", "xfm_layernorm2d_fwd", "xfm_layernorm2d_bwd", "xfm_layernorm2d_bwd_parts_blocks", "xfm_layernorm2d_bwd_parts", "xfm_layernorm2d_ws_floats", "xfm_layernorm2d_bwd_ws_floats", "xfm_layernorm2d_bwd_ws_blocks", "xfm_layernorm2d_fwd_ws", "xfm_layernorm2d_bwd_parts_ws",
    "xfm_add_layernorm_rows_supported", "xfm_add_layernorm_rows_bwd_blocks", "xfm_add_layernorm_rows_fwd",
    "xfm_add_layernorm_rows_bwd", "xfm_layernorm_rows_gelu_fwd", "xfm_layernorm_rows_gelu_bwd", "xfm_colsum_blocks", "xfm_bias_gelu_fwd", "xfm_bias_gelu_bwd", "xfm_colsum", "xfm_partial_sums_multi", "xfm_pooled_transpose_fwd", "xfm_pooled_transpose_bwd", "xfm_gated_transpose_fwd", "xfm_gated_transpose_bwd", "xfm_views_avg_stack_fwd", "xfm_views_avg_stack_bwd", "xfm_bn_tokens_supported", "xfm_bn_tokens_ws_floats", "xfm_bn_tokens_fwd", "xfm_bn_tokens_bwd", "xfm_transpose_short_supported", "xfm_transpose_short", "xfm_transpose_short_add_bf16", "xfm_residual_settle_fwd", "xfm_residual_settle_bwd", "xfm_tokens_gemm_supported", "xfm_tokens_gemm", "xfm_tokens_gemm2_supported", "xfm_tokens_gemm2", "xfm_tokens_gemm2_parts_blocks", "xfm_tokens_gemm2_parts", "xfm_proj_gemm_supported", "xfm_proj_gemm", "xfm_proj_gemm_accumulate", "xfm_planes_gemm_supported", "xfm_planes_gemm",
    "xfm_ss2d_plan", "xfm_ss2d_fwd", "xfm_ss2d_bwd", "xfm_ss2d_bwd_ws_bytes", "xfm_ss2d_bwd_ws", "xfm_ss2d_dtfused_rank", "xfm_ss2d_xr_rows", "xfm_ss2d_bc_f32", "xfm_ss2d_route_split_bc32",
    "xfm_ss2dc_supported", "xfm_ss2dc_ytokens_supported", "xfm_ss2dc_nsteps", "xfm_ss2dc_fwd", "xfm_ss2dc_bwd", "xfm_ss2dc_post",
    "xfm_fp8_planes_gemm_supported", "xfm_fp8_planes_gemm", "xfm_adam_multi", "xfm_adam_multi_scaled",
    "xfm_wgrad_supported", "xfm_wgrad",
)


class ScanPlan(C.Structure):
    _fields_ = [("lanes_per_row", C.c_int), ("items", C.c_int), ("n_chunks", C.c_int)]


class ScanParams(C.Structure):
    _fields_ = [
        ("batch", C.c_int), ("dim", C.c_int), ("seqlen", C.c_int), ("dstate", C.c_int), ("n_groups", C.c_int),
        ("delta_softplus", C.c_int), ("in_dtype", C.c_int), ("out_dtype", C.c_int),
        ("u", C.c_void_p), ("delta", C.c_void_p), ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("D", C.c_void_p), ("delta_bias", C.c_void_p),
        ("u_batch_stride", C.c_int64), ("u_d_stride", C.c_int64),
        ("delta_batch_stride", C.c_int64), ("delta_d_stride", C.c_int64),
        ("A_d_stride", C.c_int64),
        ("B_batch_stride", C.c_int64), ("B_group_stride", C.c_int64), ("B_dstate_stride", C.c_int64),
        ("C_batch_stride", C.c_int64), ("C_group_stride", C.c_int64), ("C_dstate_stride", C.c_int64),
        ("out", C.c_void_p), ("out_batch_stride", C.c_int64), ("out_d_stride", C.c_int64),
        ("x", C.c_void_p),
        ("dout", C.c_void_p), ("dout_batch_stride", C.c_int64), ("dout_d_stride", C.c_int64),
        ("du", C.c_void_p), ("ddelta", C.c_void_p),
        ("dA", C.c_void_p), ("dB", C.c_void_p), ("dC", C.c_void_p), ("dD", C.c_void_p), ("ddelta_bias", C.c_void_p),
    ]


class SS2DParams(C.Structure):
    _fields_ = [
        ("batch", C.c_int), ("d_inner", C.c_int), ("H", C.c_int), ("W", C.c_int), ("dstate", C.c_int),
        ("delta_softplus", C.c_int), ("in_dtype", C.c_int), ("out_dtype", C.c_int),
        ("x", C.c_void_p), ("dts", C.c_void_p), ("Bs", C.c_void_p), ("Cs", C.c_void_p),
        ("A", C.c_void_p), ("D", C.c_void_p), ("delta_bias", C.c_void_p),
        ("y", C.c_void_p), ("chk", C.c_void_p),
        ("dy", C.c_void_p), ("dx", C.c_void_p), ("ddts", C.c_void_p),
        ("dBs", C.c_void_p), ("dCs", C.c_void_p), ("dA", C.c_void_p), ("dD", C.c_void_p), ("ddelta_bias", C.c_void_p),
        ("xrt", C.c_void_p), ("dt_w", C.c_void_p), ("dt_rank_p", C.c_int), ("bc_f32", C.c_int),
        ("Bs32", C.c_void_p), ("Cs32", C.c_void_p),
    ]


class SS2DCParams(C.Structure):
    _fields_ = [
        ("batch", C.c_int), ("d_inner", C.c_int), ("H", C.c_int), ("W", C.c_int), ("dstate", C.c_int),
        ("dt_rank", C.c_int), ("n_routes", C.c_int), ("c_mod", C.c_int), ("c_off", C.c_int), ("wdiv", C.c_int),
        ("y_tokens", C.c_int), ("x_tokens", C.c_int),
        ("x", C.c_void_p), ("xdbl", C.c_void_p), ("wdt", C.c_void_p), ("zeros", C.c_void_p),
        ("A", C.c_void_p), ("D", C.c_void_p), ("delta_bias", C.c_void_p),
        ("y", C.c_void_p), ("chk", C.c_void_p), ("dy", C.c_void_p), ("dx", C.c_void_p), ("ddts", C.c_void_p),
        ("dBC", C.c_void_p), ("dA", C.c_void_p), ("dD", C.c_void_p), ("ddelta_bias", C.c_void_p),
    ]


_lib = None


def build(verbose: bool = False) -> None:
    """Compile libxfm_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"] + ([] if verbose else ["-s"]))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"xfmamba_amd: HIP extension {LIB_PATH} not built (run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` or `make -C xfmamba_amd/csrc`). There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        l.xfm_abi_version.restype = C.c_int
        l.xfm_strerror.restype = C.c_char_p
        l.xfm_strerror.argtypes = [C.c_int]
        l.xfm_last_hip_error.restype = C.c_char_p
        l.xfm_prof_main_kernel.restype = C.c_int
        l.xfm_prof_main_kernel.argtypes = [C.c_void_p, C.c_void_p]
        l.xfm_scan_plan.argtypes = [C.c_int] * 5 + [C.POINTER(ScanPlan)]
        l.xfm_ss2d_plan.argtypes = [C.c_int] * 6 + [C.POINTER(ScanPlan)]
        for fn in (l.xfm_selective_scan_fwd, l.xfm_selective_scan_bwd):
            fn.argtypes = [C.POINTER(ScanParams), C.c_void_p]
            fn.restype = C.c_int
        for fn in (l.xfm_ss2d_fwd, l.xfm_ss2d_bwd):
            fn.argtypes = [C.POINTER(SS2DParams), C.c_void_p]
            fn.restype = C.c_int
        l.xfm_ss2d_bwd_ws_bytes.argtypes = [C.POINTER(SS2DParams)]
        l.xfm_ss2d_bwd_ws_bytes.restype = C.c_size_t
        l.xfm_ss2d_bwd_ws.argtypes = [C.POINTER(SS2DParams), C.c_void_p, C.c_size_t, C.c_void_p]
        l.xfm_ss2d_bwd_ws.restype = C.c_int
        for fn in (l.xfm_ss2dc_fwd, l.xfm_ss2dc_bwd):
            fn.argtypes = [C.POINTER(SS2DCParams), C.c_void_p]
            fn.restype = C.c_int
        l.xfm_ss2dc_supported.argtypes = [C.c_int] * 6
        l.xfm_ss2dc_nsteps.argtypes = [C.c_int] * 3
        l.xfm_ss2dc_ytokens_supported.argtypes = [C.c_int] * 4
        l.xfm_fp8_planes_gemm_supported.argtypes = [C.c_int] * 2
        l.xfm_fp8_planes_gemm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
        l.xfm_partial_sums_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        l.xfm_transpose_short_supported.argtypes = [C.c_int, C.c_int]
        l.xfm_bn_tokens_supported.argtypes = [C.c_int] * 3
        l.xfm_pooled_transpose_fwd.argtypes = [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]
        l.xfm_pooled_transpose_bwd.argtypes = [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]
        l.xfm_gated_transpose_fwd.argtypes = [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]
        l.xfm_gated_transpose_bwd.argtypes = [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p]
        l.xfm_views_avg_stack_fwd.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
        l.xfm_views_avg_stack_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_void_p]
        l.xfm_bn_tokens_ws_floats.argtypes = [C.c_int] * 3
        l.xfm_bn_tokens_fwd.argtypes = [C.c_void_p] * 5 + [C.c_float, C.c_float] + [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
        l.xfm_bn_tokens_bwd.argtypes = [C.c_void_p] * 9 + [C.c_int] * 4 + [C.c_void_p]
        l.xfm_transpose_short.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        l.xfm_transpose_short_add_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        l.xfm_residual_settle_fwd.argtypes = [C.c_void_p] * 5 + [C.c_int] * 5 + [C.c_void_p]
        l.xfm_residual_settle_bwd.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]
        l.xfm_adam_multi.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_void_p] + [C.c_float] * 5 + [C.c_void_p]
        l.xfm_adam_multi_scaled.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_void_p] + [C.c_float] * 6 + [C.c_void_p]
        l.xfm_ss2dc_post.argtypes = [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_void_p]
        l.xfm_wgrad_supported.argtypes = [C.c_int] * 5
        l.xfm_wgrad.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_int64] * 2 + [C.c_int] * 2 + [C.c_void_p]
        l.xfm_cross_scan.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
        l.xfm_cross_merge.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]
        l.xfm_swap_scan.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]
        l.xfm_dwconv3x3_fwd.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]
        l.xfm_dwconv3x3_bwd.argtypes = [C.c_void_p] * 7 + [C.c_int] * 6 + [C.c_void_p]
        l.xfm_layernorm2d_fwd.argtypes = [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_float] + [C.c_int] * 2 + [C.c_void_p]
        l.xfm_layernorm2d_bwd_parts_blocks.argtypes = [C.c_int] * 5
        l.xfm_layernorm2d_bwd_parts.argtypes = [C.c_void_p] * 7 + [C.c_int] * 5 + [C.c_void_p]
        l.xfm_layernorm2d_ws_floats.argtypes = [C.c_int] * 3
        l.xfm_layernorm2d_bwd_ws_floats.argtypes = [C.c_int] * 3
        l.xfm_layernorm2d_bwd_ws_blocks.argtypes = [C.c_int] * 3
        l.xfm_layernorm2d_fwd_ws.argtypes = [C.c_void_p] * 7 + [C.c_int] * 3 + [C.c_float, C.c_int, C.c_int, C.c_void_p]
        l.xfm_layernorm2d_bwd_parts_ws.argtypes = [C.c_void_p] * 8 + [C.c_int] * 5 + [C.c_void_p]
        l.xfm_layernorm2d_bwd.argtypes = [C.c_void_p] * 8 + [C.c_int] * 5 + [C.c_void_p]
        l.xfm_add_layernorm_rows_supported.argtypes = [C.c_int]
        l.xfm_add_layernorm_rows_bwd_blocks.argtypes = [C.c_int, C.c_int]
        l.xfm_add_layernorm_rows_fwd.argtypes = [C.c_void_p] * 10 + [C.c_int] * 3 + [C.c_float, C.c_int, C.c_int, C.c_void_p]
        l.xfm_add_layernorm_rows_bwd.argtypes = [C.c_void_p] * 14 + [C.c_int] * 5 + [C.c_void_p]
        l.xfm_dwconv3x3_tokens_supported.argtypes = [C.c_int] * 3
        l.xfm_dwconv3x3_tokens_fwd.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
        l.xfm_dwconv3x3_tokens_bwd.argtypes = [C.c_void_p] * 9 + [C.c_int] * 4 + [C.c_void_p]
        l.xfm_ss2d_route_split.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]
        l.xfm_ss2d_route_merge.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]
        l.xfm_ss2d_dtfused_rank.argtypes = [C.c_int] * 7
        l.xfm_ss2d_bc_f32.argtypes = [C.c_int] * 6
        l.xfm_ss2d_route_split_bc32.argtypes = [C.c_void_p] * 6 + [C.c_int] * 6 + [C.c_void_p]
        l.xfm_ss2d_xr_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong] + [C.c_int] * 4 + [C.c_void_p]
        l.xfm_ss2d_dt_proj_supported.argtypes = [C.c_int] * 3
        l.xfm_ss2d_dt_proj_mfma_rp.argtypes = [C.c_int] * 3
        l.xfm_ss2d_dt_proj_fwd_mfma.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
        l.xfm_ss2d_dt_proj_bwd_mfma.argtypes = [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]
        l.xfm_ss2d_dt_proj_fwd.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]
        l.xfm_conv3x3s2_tokens_supported.argtypes = [C.c_int] * 4
        l.xfm_conv3x3s2_tokens_fwd.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]
        l.xfm_conv3x3s2_tokens_bwd_data.argtypes = [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]
        l.xfm_conv3x3s2_tokens_bwd_weight.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p]
        l.xfm_conv3x3s2_tokens_bwd_weight_x_supported.argtypes = [C.c_int] * 5
        l.xfm_conv3x3s2_tokens_bwd_weight_x.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p]
        l.xfm_conv3x3s2_gray_supported.argtypes = [C.c_int] * 3
        l.xfm_conv3x3s2_gray_fwd.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p]
        l.xfm_conv3x3s2_gray_ws_floats.argtypes = [C.c_int]
        l.xfm_conv3x3s2_gray_bwd_weight.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
        l.xfm_layernorm_rows_gelu_fwd.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p]
        l.xfm_layernorm_rows_gelu_bwd.argtypes = [C.c_void_p] * 12 + [C.c_int] * 4 + [C.c_void_p]
        l.xfm_colsum_blocks.argtypes = [C.c_longlong, C.c_int, C.c_int]
        l.xfm_bias_gelu_fwd.argtypes = [C.c_void_p] * 3 + [C.c_longlong, C.c_int, C.c_int, C.c_void_p]
        l.xfm_bias_gelu_bwd.argtypes = [C.c_void_p] * 6 + [C.c_longlong, C.c_int, C.c_int, C.c_void_p]
        l.xfm_colsum.argtypes = [C.c_void_p] * 3 + [C.c_longlong, C.c_int, C.c_int, C.c_void_p]
        l.xfm_tokens_gemm_supported.argtypes = [C.c_int, C.c_int]
        l.xfm_tokens_gemm.argtypes = [C.c_void_p] * 4 + [C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]
        l.xfm_tokens_gemm2_parts_blocks.argtypes = [C.c_longlong, C.c_int, C.c_int]
        l.xfm_tokens_gemm2_parts.argtypes = [C.c_void_p] * 6 + [C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p]
        l.xfm_tokens_gemm2_supported.argtypes = [C.c_int, C.c_int]
        l.xfm_tokens_gemm2.argtypes = [C.c_void_p] * 6 + [C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        l.xfm_proj_gemm_supported.argtypes = [C.c_int] * 3
        l.xfm_proj_gemm_accumulate.argtypes = [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p]
        l.xfm_proj_gemm.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]
        l.xfm_planes_gemm_supported.argtypes = [C.c_int] * 3
        l.xfm_planes_gemm.argtypes = [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]
        if l.xfm_abi_version() != 2:
            raise RuntimeError("xfmamba_amd: libxfm_hip.so ABI version mismatch")
        _lib = l
    return _lib


def dtype_code(dt: torch.dtype) -> int:
    try:
        return _DT[dt]
    except KeyError:
        raise RuntimeError(f"xfmamba_amd: unsupported dtype {dt} (fp32, fp16, bf16 only)") from None


def check(rc: int, what: str) -> None:
    if rc != 0:
        l = lib()
        msg = l.xfm_strerror(rc).decode()
        if rc == -4:
            msg += ": " + l.xfm_last_hip_error().decode()
        raise RuntimeError(f"xfmamba_amd.{what}: {msg}")


def require_cuda(*tensors) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("xfmamba_amd: operands must live on an MI355X device (no CPU path; "
                               "the CPU oracle under oracle/ is test infrastructure only)")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


# ---- optional per-kernel timing (bench.py): HIP events on the stream the kernel is launched on ----
class KernelTimer:
    """Collects (start, end) HIP events and algorithmic bytes per launch, keyed by kernel family."""

    def __init__(self):
        self.records = {}

    def add(self, name, start, end, nbytes, nbytes_alt=None):
        self.records.setdefault(name, []).append((start, end, nbytes, nbytes_alt))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [r[0].elapsed_time(r[1]) for r in recs]
            out[name] = dict(launches=len(recs), total_ms=sum(ms), avg_us=1e3 * sum(ms) / len(recs),
                             bytes=sum(r[2] for r in recs))
            if all(r[3] is not None for r in recs):       # a second algorithmic boundary (dt_proj fused as well)
                out[name]["bytes_alt"] = sum(r[3] for r in recs)
        return out


_TIMER = None


def set_timer(t):
    global _TIMER
    _TIMER = t


class timed:
    """``with timed(name, nbytes): launch`` -- free when no timer is installed.  ``main_kernel=True``: the entry point launches its
    main kernel and a small finishing kernel behind it; the event pair goes to the library (``xfm_prof_main_kernel``), which records
    it around the MAIN kernel only, and the finishing kernel is kept as ``name + "_finish"`` (falls back to the whole call if the
    entry point took a path without the hook)."""
    __slots__ = ("name", "nbytes", "nbytes_alt", "s", "e", "main")

    def __init__(self, name, nbytes, nbytes_alt=None, main_kernel=False):
        self.name, self.nbytes, self.nbytes_alt, self.s, self.e, self.main = name, nbytes, nbytes_alt, None, None, main_kernel

    def __enter__(self):
        if _TIMER is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.s.record()
            if self.main:
                self.e = torch.cuda.Event(enable_timing=True)
                self.e.record()                      # (creates the handle; the library records both again)
                lib().xfm_prof_main_kernel(self.s.cuda_event, self.e.cuda_event)

    def __exit__(self, *exc):
        if self.s is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            if self.main and not lib().xfm_prof_main_kernel(None, None):     # consumed: (s, self.e) bracket the main kernel
                _TIMER.add(self.name, self.s, self.e, self.nbytes, self.nbytes_alt)
                _TIMER.add(self.name + "_finish", self.e, e, 0)
            else:
                _TIMER.add(self.name, self.s, e, self.nbytes, self.nbytes_alt)
        return False


def scan_plan(batch: int, dim: int, seqlen: int, dstate: int, n_groups: int) -> ScanPlan:
    plan = ScanPlan()
    check(lib().xfm_scan_plan(batch, dim, seqlen, dstate, n_groups, C.byref(plan)), "scan_plan")
    return plan
