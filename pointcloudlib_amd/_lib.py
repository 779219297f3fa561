"""ctypes loader of ``libpcl_hip.so`` (the C ABI declared in ``include/pcl_hip.h``).

There is NO fallback: if the HIP library is missing or a call fails, we raise.  The CPU oracle under
``oracle/`` is test infrastructure and is never imported from here.
"""
import ctypes
import functools
import os
import re
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
# PCL_HIP_SO: developer knob for timing experiments (a variant build of the same sources, csrc/Makefile EXP=n)
_SO = os.environ.get("PCL_HIP_SO") or os.path.join(_PKG, "libpcl_hip.so")
_HEADER = os.path.join(os.path.dirname(_PKG), "include", "pcl_hip.h")
_lib = None

c_void_p, c_int, c_float, c_double, c_size_t = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                                ctypes.c_double, ctypes.c_size_t)


class PclError(RuntimeError):
    pass


def so_path():
    return _SO


def build(verbose=False):
    """Compile every HIP source for gfx950 into ``pointcloudlib_amd/libpcl_hip.so`` (in-tree)."""
    cmd = ["make", "-C", os.path.join(_PKG, "csrc"), "-j8"]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return _SO


def declared_symbols():
    """Every function name ``include/pcl_hip.h`` declares (used by the symbol-export test)."""
    txt = open(_HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pcl_[a-z0-9_]+)\s*\(", txt)))


_P = c_void_p
_SIGS = {
    "pcl_version": (c_int, []),
    "pcl_time_next_launch": (None, [_P, _P]),
    "pcl_time_tagged_launch": (None, [_P, _P, ctypes.c_char_p]),
    "pcl_last_launch_kernel": (ctypes.c_char_p, []),
    "pcl_mlp_stack_sizes": (c_int, [_P, _P, _P, _P]),
    "pcl_mlp_stack_last": (c_int, [_P, _P, _P, _P]),
    "pcl_pointconv_contract_bn_f32": (c_int, [_P, _P, _P, c_float, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "pcl_pointconv_contract_bn_stat_rows": (c_int, [c_int]),
    "pcl_pointconv_contract_bn_bwd_f32": (c_int, [_P, _P, _P, _P, c_float, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "pcl_mlp_stack_fwd_f32": (c_int, [_P]),
    "pcl_mlp_stack_bwd_f32": (c_int, [_P]),
    "pcl_bn_rows_stats_f32": (c_int, [_P, _P, c_int, c_int, _P, ctypes.POINTER(c_int), _P]),
    "pcl_bn_rows_bwd_apply_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "pcl_fc_head_sizes": (c_int, [_P, _P, _P]),
    "pcl_fc_head_fwd_f32": (c_int, [_P]),
    "pcl_fc_head_bwd_f32": (c_int, [_P]),
    "pcl_xconv_core_supported": (c_int, [c_int, c_int, c_int]),
    "pcl_xconv_core_partials": (c_int, [c_int, c_int]),
    "pcl_xconv_core_fwd_f32": (c_int, [_P, _P, c_int, _P, c_int, _P, _P, c_int, c_int, c_int, _P, _P]),
    "pcl_xconv_core_bwd_f32": (c_int, [_P, _P, c_int, _P, c_int, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    "pcl_last_error": (ctypes.c_char_p, []),
    "pcl_optimal_block": (c_int, [c_int]),
    "pcl_fps_f32": (c_int, [_P, c_int, c_int, c_int, c_int, c_double, _P, _P, _P, _P]),
    "pcl_ball_query_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_float, c_int, _P, _P, _P]),
    "pcl_ball_query_multi_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "pcl_group_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "pcl_group_bwd_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "pcl_group_all_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "pcl_group_all_bwd_f32": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "pcl_gather_rows_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "pcl_gather_rows_bwd_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "pcl_edge_feature_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "pcl_edge_feature_bwd_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "pcl_knn_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "pcl_knn_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_size_t, _P]),
    "pcl_knn_fma_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_size_t, _P]),
    "pcl_knn_point_matmul_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "pcl_three_nn_f32": (c_int, [_P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "pcl_three_interp_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "pcl_three_interp_bwd_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "pcl_density_f32": (c_int, [_P, c_int, c_int, c_float, _P, _P]),
    "pcl_group_linear_stat_rows": (c_int, [c_int, c_int]),
    "pcl_group_linear_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "pcl_group_linear_bwd_f32": (c_int, [_P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int, _P]),
    "pcl_group_rows_transpose_supported": (c_int, [c_int, c_int, c_int]),
    "pcl_group_rows_transpose_i32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "pcl_group_linear_bwd_gather_supported": (c_int, [c_int]),
    "pcl_set_scatter_form": (None, [c_int]),
    "pcl_set_pointconv_paths": (None, [c_int]),
    "pcl_group_linear_bwd_gather_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, c_int, _P]),
    "pcl_head_layer_fwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, _P, _P, _P, _P, _P, c_size_t, _P]),
    "pcl_head_layer_fwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "pcl_head_layer_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "pcl_soft_ce_f32": (c_int, [_P, _P, c_float, c_int, c_int, _P, _P, _P]),
    "pcl_soft_ce_rows_blocks": (c_int, [c_int]),
    "pcl_soft_ce_rows_f32": (c_int, [_P, _P, c_float, c_int, c_int, _P, _P, _P, _P]),
    "pcl_edgeconv_stat_rows": (c_int, [c_int, c_int]),
    "pcl_edgeconv_wcat_f32": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "pcl_edgeconv_gather_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "pcl_edgeconv_gather_hilo_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "pcl_edgeconv_scatter_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "pcl_knn_transpose_i32": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P]),
    "pcl_pointconv_contract_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "pcl_pointconv_contract_bwd_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "pcl_mlp_stat_rows": (c_int, [c_int, c_int, c_int]),
    "pcl_linear_fwd_f32": (c_int, [_P, _P, _P, _P, _P, c_float, c_int, c_int, c_int, _P, _P, _P]),
    "pcl_linear_fwd_gmax_f32": (c_int, [_P, _P, _P, _P, _P, c_float, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "pcl_group_minmax_finalize_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_float, c_int, c_int, _P, _P, _P, _P]),
    "pcl_group_minmax_finalize2_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_float, c_int, c_int, _P, _P, _P, _P, c_int, _P]),
    "pcl_group_minmax_finalize_t_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_float, c_int, c_int, c_int, _P, _P, _P, _P, c_int, _P, _P]),
    "pcl_bn_finalize_f32": (c_int, [_P, c_int, _P, _P, c_int, c_int, c_float, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "pcl_bn_act_max_f32": (c_int, [_P, _P, _P, c_float, c_int, c_int, c_int, _P, _P, _P, _P]),
    "pcl_bn_act_f32": (c_int, [_P, _P, _P, c_float, c_int, c_int, _P, _P]),
    "pcl_bn_act_bwd_f32": (c_int, [_P, _P, _P, _P, c_float, c_int, c_int, _P, _P, ctypes.POINTER(c_int), _P]),
    "pcl_bn_act_max_mean_f32": (c_int, [_P, _P, _P, c_float, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "pcl_bn_act_max_mean_bwd_f32": (c_int, [_P, _P, c_int, _P, _P, _P, _P, c_float, c_int, c_int, c_int, _P, _P, ctypes.POINTER(c_int), _P]),
    "pcl_maxgrad_prep_f32": (c_int, [_P, _P, _P, c_float, c_int, c_int, _P, _P, ctypes.POINTER(c_int), _P]),
    "pcl_bn_bwd_consts_f32": (c_int, [_P, c_int, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "pcl_linear_bwd_dx_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, _P, _P, _P, c_float,
                                      _P, _P, _P]),
    "pcl_group_offsets_i32": (c_int, [_P, c_int, _P, _P]),
    "pcl_group_offsets_multi_i32": (c_int, [c_int, _P, c_int, _P, _P]),
    "pcl_group_compact_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "pcl_linear_fwd_rows_f32": (c_int, [_P, _P, _P, _P, _P, c_float, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "pcl_bn_act_max_rows_f32": (c_int, [_P, _P, _P, _P, c_float, c_int, c_int, _P, _P, _P, _P]),
    "pcl_linear_bwd_dx_rows_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, _P, _P, _P, c_float,
                                           _P, _P, _P, _P, c_int, c_int, _P]),
    "pcl_linear_bwd_dw_rows_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_float, c_int, c_int, c_int, _P,
                                           _P, c_size_t, _P, _P, c_int, _P]),
    "pcl_scatter_rows_add_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "pcl_linear_bwd_dw_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "pcl_frag_stat_rows": (c_int, [c_int]),
    "pcl_frag_max_rows": (c_int, []),
    "pcl_frag_set_tuning": (None, [c_int] * 7),
    "pcl_frag_linear_fwd_f32": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, c_float, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P]),
    "pcl_frag_dy_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, c_int, _P]),
    "pcl_frag_linear_bwd_dx_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_float, _P, c_int, _P, c_int, _P]),
    "pcl_frag_dw_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "pcl_frag_dw_counter_words": (c_int, [c_int, c_int, c_int]),
    "pcl_frag_linear_bwd_dw_f32": (c_int, [_P, _P, c_int, _P, _P, c_float, c_int, c_int, c_int, _P, c_int, _P, c_size_t, c_int, _P]),
    "pcl_set_fb_max_blocks": (None, [c_int]),
    "pcl_set_fb_two_images": (None, [c_int]),
    "pcl_get_fb_two_images": (c_int, []),
    "pcl_set_kernel_paths": (None, [c_int, c_int, c_int]),
    "pcl_sgd_momentum_f32": (c_int, [_P, _P, _P, _P, c_int, c_double, c_double, c_double, c_double, _P]),
    "pcl_set_stack_overlap": (None, [c_int, c_int]),
    "pcl_get_stack_overlap": (c_int, []),
    "pcl_bn_bwd_dy_supported": (c_int, [c_int, c_int]),
    "pcl_bn_bwd_dy_f32": (c_int, [_P, c_int, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "pcl_linear_bwd_dw_plain_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "pcl_linear_bwd_dw_plain_f32": (c_int, [_P, _P, _P, _P, c_float, c_int, c_int, c_int, _P, _P, c_size_t, c_int, _P]),
    "pcl_mlp_fewrow_layer": (c_int, [c_int, c_int, c_int, c_int]),
    "pcl_set_fewrow_backward": (None, [c_int]),
    "pcl_get_fewrow_backward": (c_int, []),
    "pcl_set_dw_tuning": (None, [c_int]),
    "pcl_knn_nk_supported": (c_int, [c_int]),
    "pcl_knn_nk_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "pcl_linear_bwd_pair_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "pcl_linear_bwd_pair_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, c_float, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, c_size_t, _P]),
    "pcl_linear_bwd_pair_finish_f32": (c_int, [_P, c_size_t, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "pcl_set_bwd_pair": (None, [c_int]),
    "pcl_get_bwd_pair": (c_int, []),
    "pcl_set_matrix_form": (None, [c_int]),
    "pcl_get_matrix_form": (c_int, []),
    "pcl_set_fps_tuning": (None, [c_int, c_int]),
    "pcl_linear_bwd_fused_supported": (c_int, [c_int, c_int]),
    "pcl_linear_bwd_fused_stat_rows": (c_int, [c_int, c_int]),
    "pcl_linear_bwd_fused_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "pcl_linear_bwd_fused_rows_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, _P, _P, _P, c_float,
                                               _P, _P, _P, c_size_t, _P, _P, _P]),
    "pcl_linear_bwd_fused_finish_f32": (c_int, [_P, c_size_t, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "pcl_linear_bwd_dw_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_float, c_int, c_int, c_int, _P,
                                      _P, c_size_t, _P]),
}


def lib():
    """The loaded library; raises PclError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise PclError(f"{_SO} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch-ROCm bundles its own libamdhip64 (same SONAME).  Kernels launched here must run in the SAME HIP
        # runtime instance that owns torch's device buffers and streams, so torch's copy has to be the one the
        # dynamic linker resolves for us: load torch first, then dlopen (otherwise /opt/rocm's copy is pulled in as
        # a second runtime and every launch fails with hipErrorNoDevice).
        import torch  # noqa: F401
        L = ctypes.CDLL(_SO)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        # lab switches of the kernel selection: read HERE (host side) and handed over as a C call -- the library itself reads no environment
        sw = [-1 if os.environ.get(k) is None else int(os.environ[k] != "0") for k in ("PCL_FWD_RES", "PCL_NARROW", "PCL_FUSED_BWD")]
        if any(v >= 0 for v in sw):
            L.pcl_set_kernel_paths(*sw)
        if os.environ.get("PCL_SIDE_DW") is not None:
            L.pcl_set_stack_overlap(int(os.environ["PCL_SIDE_DW"] != "0"), -1)
        if os.environ.get("PCL_BWD_W_ROWS") is not None:
            L.pcl_set_pointconv_paths(int(os.environ["PCL_BWD_W_ROWS"] != "0"))
        if os.environ.get("PCL_SCATTER") is not None:
            L.pcl_set_scatter_form(int(os.environ["PCL_SCATTER"] != "0"))
        if os.environ.get("PCL_BWD_PAIR") is not None:
            L.pcl_set_bwd_pair(int(os.environ["PCL_BWD_PAIR"] != "0"))
        if os.environ.get("PCL_FB_TWO") is not None:
            L.pcl_set_fb_two_images(int(os.environ["PCL_FB_TWO"] != "0"))
        if os.environ.get("PCL_FEWROW") is not None:
            L.pcl_set_fewrow_backward(int(os.environ["PCL_FEWROW"] != "0"))
        if os.environ.get("PCL_DW_GX") is not None:
            L.pcl_set_dw_tuning(int(os.environ["PCL_DW_GX"]))
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().pcl_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise PclError(f"{what}: rc={rc}: {msg}")


# ---------------------------------------------------------------- per-kernel event timing (bench.py)
PROFILER = None   # set to a KernelTimer by bench.py; None in normal operation (zero overhead)


# entry points that launch exactly one GEMM-family kernel (plus, for dW, small reductions that are not the kernel of interest)
KERNEL_TIMED = {"pcl_linear_fwd_rows_f32", "pcl_linear_fwd_f32", "pcl_linear_bwd_dx_rows_f32", "pcl_linear_bwd_dx_f32",
                "pcl_linear_bwd_dw_rows_f32", "pcl_linear_bwd_dw_f32", "pcl_linear_bwd_fused_rows_f32",
                "pcl_linear_bwd_dw_plain_f32", "pcl_frag_linear_bwd_dx_f32", "pcl_frag_linear_fwd_f32",
                "pcl_knn_f32", "pcl_knn_fma_f32", "pcl_knn_nk_f32"}      # (k-NN: the fused kernel; the two-pass form arms nothing and is not recorded)
_hip = None


def _hiprt():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        _hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        _hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
    return _hip


class _HipEventPair:
    """Two timing-enabled hipEvent_t owned by this object."""

    def __init__(self):
        h = _hiprt()
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        if h.hipEventCreate(ctypes.byref(a)) or h.hipEventCreate(ctypes.byref(b)):
            raise PclError("hipEventCreate failed")
        self.start, self.stop = a, b

    def elapsed_ms(self):
        ms = ctypes.c_float()
        rc = _hiprt().hipEventElapsedTime(ctypes.byref(ms), self.start, self.stop)
        if rc:                                       # (this call launched no kernel of the timed families: the events were never recorded)
            _hiprt().hipGetLastError()               # the runtime keeps the code as its "last error": torch's next check would raise it
            return None
        return ms.value

    def __del__(self):
        try:
            _hiprt().hipEventDestroy(self.start); _hiprt().hipEventDestroy(self.stop)
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass


class KernelTimer:
    """Brackets selected C-ABI calls with HIP events on the stream they are launched on (torch's current
    stream) and accumulates per-entry-point time plus the algorithmic bytes/flops the caller states."""

    def __init__(self, names=None, tags=None, max_records=None, inner=None):
        """``inner`` = (entry point, launch tag, algo_bytes, algo_flops): time ONE kernel inside the per-stack entry points
        (csrc/stack.hip) -- the launch whose tag matches gets the armed events (pcl_time_tagged_launch); the record is filed
        under (entry point, launch tag) with the stated algorithmic bytes / flops per launch."""
        self.inner = inner
        if inner is not None:
            names = ["pcl_mlp_stack_fwd_f32", "pcl_mlp_stack_bwd_f32"]
            tags = None
        self.names = None if names is None else set(names)
        self.tags = None if tags is None else set(tags)          # restrict to these launch shapes
        self.max_records = max_records                           # per (name, tag): timing events perturb the stream
        self.records = {}                                        # (each record is a marker packet), so bound them
        self.order = []                                          # (name, tag) in launch order
        self.calls = 0                                           # every C-ABI call seen, recorded or not

    def want(self, name, tag=None):
        self.calls += 1
        if self.names is not None and name not in self.names:
            return False
        if self.tags is not None and tag is not None and tag not in self.tags:
            return False
        if self.inner is not None:
            return self.max_records is None or len(self.records.get(self.inner[:2], ())) < 3 * self.max_records
        if self.max_records is not None and tag is not None:
            return len(self.records.get((name, tag), ())) < self.max_records
        return True

    def begin(self, name=None):
        if self.inner is not None:
            pair = _HipEventPair()
            lib().pcl_time_tagged_launch(pair.start, pair.stop, self.inner[1].encode())
            return pair
        if name in KERNEL_TIMED:
            # the GEMM-family kernel this entry point launches reports its own begin / end timestamps into two events
            # (pcl_time_next_launch): the same interval rocprofv3's kernel trace shows, without the two marker packets'
            # dispatch gaps (~20 us around a 0.2 ms kernel)
            pair = _HipEventPair()
            lib().pcl_time_next_launch(pair.start, pair.stop)
            return pair
        import torch
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def end(self, name, tag, start, algo_bytes, algo_flops):
        if self.inner is not None:
            name, tag, algo_bytes, algo_flops = self.inner
        if isinstance(start, _HipEventPair):
            lib().pcl_time_next_launch(None, None)                # (disarm: a call that launched no such kernel)
            ev = None
        else:
            import torch
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
        self.records.setdefault((name, tag), []).append((start, ev, algo_bytes, algo_flops))
        fam = ""
        if isinstance(start, _HipEventPair) and self.inner is None:      # which kernel the entry point chose (tools/pmc_traffic.py)
            fam = (lib().pcl_last_launch_kernel() or b"").decode().strip("()").split("<")[0]
        self.order.append((name, tag, fam))

    def summary(self):
        """{(name, tag): dict(launches, avg_ms, algo_bytes, algo_flops)} -- call after a device sync."""
        out = {}
        def val(v):            # algorithmic bytes/flops may depend on a device-resident row count: resolved here,
            return v() if callable(v) else v          # after the timed region, never inside it

        for key, recs in self.records.items():
            ms = [s.elapsed_ms() if e is None else s.elapsed_time(e) for s, e, _, _ in recs]
            recs = [r for r, m in zip(recs, ms) if m is not None]
            ms = [m for m in ms if m is not None]
            if not ms:
                continue
            ab = [val(r[2]) for r in recs]
            af = [val(r[3]) for r in recs]
            out[key] = {"launches": len(recs), "avg_ms": sum(ms) / len(ms), "total_ms": sum(ms),
                        "algo_bytes": sum(ab) / len(ab), "algo_flops": sum(af) / len(af)}
        return out


@functools.lru_cache(maxsize=None)
def size_query(name, *ints):
    """Pure size functions of the ABI (``pcl_mlp_stat_rows``, ``pcl_linear_bwd_dw_workspace_bytes``, ...): asked once per shape."""
    return getattr(lib(), name)(*ints)


_FN = {}            # entry point name -> bound ctypes function


def call(name, *args, algo_bytes=0, algo_flops=0, tag=""):
    """Invoke one C-ABI entry point; raises on a non-zero return code."""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(lib(), name)
    prof = PROFILER
    if prof is None:                               # normal operation: straight through
        rc = fn(*args)
    else:
        tag = tag or (name if callable(algo_bytes) else f"{algo_bytes}")
        if prof.want(name, tag):
            start = prof.begin(name)
            rc = fn(*args)
            prof.end(name, tag, start, algo_bytes, algo_flops)
        else:
            rc = fn(*args)
    if rc != 0:
        check(rc, name)
