"""ctypes binding of libistnet_pn2.so (include/istnet_pn2.h).

There is NO fallback: if the library is missing or a launch fails, the caller gets a RuntimeError.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libistnet_pn2.so")
INCLUDE_DIR = os.path.join(_HERE, "..", "include")
HEADER_PATH = os.path.join(INCLUDE_DIR, "istnet_pn2.h")
HEADER_PATHS = [HEADER_PATH, os.path.join(INCLUDE_DIR, "istnet_pw.h"), os.path.join(INCLUDE_DIR, "istnet_preproc.h"),
                os.path.join(INCLUDE_DIR, "istnet_optim.h"), os.path.join(INCLUDE_DIR, "istnet_rgb.h"),
                os.path.join(INCLUDE_DIR, "istnet_heads.h"), os.path.join(INCLUDE_DIR, "istnet_conv.h")]
ABI_VERSION = 2

_i, _f, _p, _d, _l = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_double, ctypes.c_longlong
# name -> argtypes (restype is always int); mirrors include/istnet_pn2.h
SIGNATURES = {
    "istnet_pn2_csr_build": [_i, _i, _i, _p, _p, _p, _p],
    "istnet_pn2_csr_build_multi": [_i, _i, _p, _p, _p, _p, _p, _p],
    "istnet_pn2_csr_build_segmented": [_i, _i, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_pw_scatter_csr_chunks": [_i],
    "istnet_pw_scatter_dy_csr": [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _l, _p, _p, _i, _p, _p],
    "istnet_pw_scatter_dy_csr_fin": [_i, _i, _i, _i, _p, _p, _p, _i, _d, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _l, _p, _p, _i, _p, _p],
    "istnet_adam_step": [_l, _p, _p, _p, _p, _p, _p, _d, _d, _d, _d, _d, _d, _p],
    "istnet_adam_step_counting": [_l, _p, _p, _p, _p, _p, _p, _p, _d, _d, _d, _d, _d, _d, _p],
    "istnet_conv_supported": [_i] * 6,
    "istnet_conv_workspace_floats": [_i] * 10,
    "istnet_conv_forward": [_i] * 9 + [_p] * 5,
    "istnet_conv_backward_data": [_i] * 9 + [_p] * 5,
    "istnet_conv_wrw_splits": [_i] * 9,
    "istnet_conv_backward_weights": [_i] * 9 + [_p] * 5,
    "istnet_nhwc_gram64_parts": [_l],
    "istnet_nhwc_gram64": [_l, _p, _p, _p, _p, _p, _p],
    "istnet_nhwc_rowmix64": [_l, _p, _p, _p, _p, _p],
    "istnet_final_chosen_workgroups": [_l],
    "istnet_final_chosen_forward": [_i, _l, _i, _i] + [_p] * 10 + [_d] + [_p] * 8,
    "istnet_final_chosen_backward": [_i, _l, _i, _i] + [_p] * 24,
    "istnet_nhwc_stat_parts": [_l],
    "istnet_nhwc_channel_stats": [_l, _i, _p, _p, _p, _p],
    "istnet_nhwc_bn_prelu_apply": [_i, _l, _i, _p, _p, _p, _p, _p, _p],
    "istnet_nhwc_bn_prelu_bwd_stats": [_i, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_nhwc_bn_prelu_bwd_finalize": [_i, _i, _d] + [_p] * 11,
    "istnet_nhwc_bn_prelu_bwd_apply": [_i, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_nhwc_bn_act_res_apply": [_i, _l, _i, _p, _p, _p, _p, _p, _p],
    "istnet_nhwc_bn_act_res_bwd_stats": [_i, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_prelu_bwd_parts": [_l],
    "istnet_prelu_bwd": [_l, _p, _p, _p, _p, _p, _p],
    "istnet_upsample_bilinear_ac_bwd_nhwc": [_i, _i, _i, _i, _i, _i, _p, _p, _p],
    "istnet_upsample_bilinear_ac_fwd_nhwc": [_i, _i, _i, _i, _i, _i, _p, _p, _p],
    "istnet_upconv3_fwd_nhwc": [_i, _i, _i, _i, _i, _i, _p, _p, _p, _p],
    "istnet_upconv3_bwd_nhwc": [_i, _i, _i, _i, _i, _i, _p, _p, _p],
    "istnet_backproject_choose": [_i, _i, _i, _i, _p, _i, _l, _p, _p, _d, _d, _d, _d, _d, _i, _p, _p, _p],
    "istnet_crop_resize_normalize": [_i, _i, _i, _p, _l, _i, _p, _i, _p, _p, _p, _p, _p],
    "istnet_depth_fill_scratch_floats": [_i, _i, _i],
    "istnet_depth_fill_multiscale": [_i, _i, _i, _p, _f, _p, _p, _p],
    "istnet_conv_set_tuning": [_i, _i],
    "istnet_conv_get_tuning": [_i],
    "istnet_depth_fill_missing": [_i, _i, _i, _p, _i, _d, _d, _f, _p, _p, _p],
    "istnet_pn2_set_tuning": [_i, _i],
    "istnet_pn2_furthest_point_sampling": [_i, _i, _i, _p, _p, _p, _p],
    "istnet_debug_marker": [_p, _p],
    "istnet_pn2_fps_gather": [_i, _i, _i, _p, _p, _p, _p],
    "istnet_pn2_fps_gather_chain": [_i, _i, _i, _p, _p, _p, _p, _p, _i, _p],
    "istnet_pn2_gather_points": [_i, _i, _i, _i, _p, _p, _p, _p],
    "istnet_pn2_gather_points_grad": [_i, _i, _i, _i, _p, _p, _p, _p],
    "istnet_pn2_query_ball_point": [_i, _i, _i, _f, _i, _p, _p, _p, _p],
    "istnet_pn2_query_ball_point_pair": [_i, _i, _i, _f, _i, _f, _i, _p, _p, _p, _p, _p, _p, _p],
    "istnet_pn2_group_points": [_i, _i, _i, _i, _i, _p, _p, _p, _p],
    "istnet_pn2_group_points_grad": [_i, _i, _i, _i, _i, _p, _p, _p, _p],
    "istnet_pn2_three_nn": [_i, _i, _i, _p, _p, _p, _p, _p],
    "istnet_pn2_three_nn_weights": [_i, _i, _i, _p, _p, _p, _p, _p],
    "istnet_pn2_three_nn_weights_multi": [_i, _i, _p, _p, _p, _p, _p, _p, _p],
    "istnet_pn2_three_interpolate": [_i, _i, _i, _i, _p, _p, _p, _p, _p],
    "istnet_pn2_three_interpolate_grad": [_i, _i, _i, _i, _p, _p, _p, _p, _p],
    "istnet_pn2_interp_csr_build": [_i, _i, _i, _p, _p, _p, _p],
    "istnet_pn2_three_interpolate_grad_csr": [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p],
    "istnet_pn2_group_points_grad_csr": [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p],
    # include/istnet_pw.h
    "istnet_pw_tile_cfg": [_i, _i, _i],
    "istnet_pw_wgrad_tile_cfg": [_i, _i, _i, _i],
    "istnet_pw_dgrad_tile_cfg": [_i, _i, _i],
    "istnet_pw_set_tuning": [_i, _i],
    "istnet_pw_get_tuning": [_i],
    "istnet_pw_stat_tiles": [_i, _i, _i],
    "istnet_pw_forward_tiles": [_i, _i, _i, _i],
    "istnet_pw_forward_cfg": [_i, _i, _i, _i],
    "istnet_pw_forward_ld_tiles": [_i, _i, _i, _i],
    "istnet_pw_forward_acc_interp": [_i, _i, _i, _i, _p, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p],
    "istnet_pw_dgrad_sk": [_i, _i, _i, _i],
    "istnet_pw_sk_tm": [_i, _i, _i],
    "istnet_pw_dgrad_tiles": [_i, _i, _i, _i, _i],
    "istnet_pw_dgrad_rs": [_i, _i, _i, _i, _i],
    "istnet_pw_forward": [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_pw_forward_ld": [_i, _i, _i, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p],
    "istnet_pw_forward_acc": [_i, _i, _i, _i, _p, _p, _i, _p, _p, _p, _p, _p],
    "istnet_pw_forward_multi": [_i, _i, _p, _p, _i, _i, _p, _i, _p, _p, _p],
    "istnet_pw_channel_stats": [_i, _i, _i, _p, _p, _p, _p],
    "istnet_pw_interp_stats": [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p],
    "istnet_pw_interp_stats_tiles": [_i, _i],
    "istnet_pw_dy": [_i, _i, _i, _p, _p, _p, _p, _p, _p],
    "istnet_pw_gather_add_tiles": [_i, _i],
    "istnet_pw_gather_add": [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p],
    "istnet_pw_wgrad_gather": [_i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p],
    "istnet_bn_finalize_fwd": [_i, _i, _d, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p],
    "istnet_bn_finalize_fwd_nbt": [_i, _i, _d, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p],
    "istnet_bn_relu_pool": [_i, _i, _i, _i, _p, _p, _p, _l, _p, _p, _p],
    "istnet_pw_bwd_stats_pooled": [_i, _i, _i, _p, _l, _p, _p, _p, _p, _p],
    "istnet_bn_bwd_dense_finalize": [_i, _i, _i, _d, _i, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_bn_bwd_pooled_finalize": [_i, _i, _i, _d, _i, _p, _l, _p, _p, _p, _p, _p, _p, _p],
    "istnet_affine_consts": [_i, _p, _p, _p, _p, _f, _p, _p],
    "istnet_affine_consts_multi": [_i, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_affine_apply": [_i, _i, _i, _i, _p, _p, _p, _p],
    "istnet_pw_bwd_stat_tiles": [_i, _i],
    "istnet_pw_bwd_stats": [_i, _i, _i, _i, _p, _p, _p, _l, _p, _p, _p, _p, _p],
    "istnet_bn_finalize_bwd": [_i, _i, _d, _i, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_pw_dgrad_stat_tiles": [_i, _i, _i],
    "istnet_pw_dgrad": [_i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_pw_dwx_chunks": [_i, _i, _i],
    "istnet_pw_scatter_dy": [_i, _i, _i, _i, _i, _p, _p, _p, _l, _p, _p, _p, _p, _p, _l, _p, _p, _i, _p, _p],
    "istnet_pw_bwd_small_ok": [_i, _i, _i],
    "istnet_pw_bwd_small_splits": [_i, _i],
    "istnet_pw_bwd_small": [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_pw_wgrad_splits": [_i, _i, _i, _i],
    "istnet_pw_wgrad": [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p],
    "istnet_pw_wgrad_reduce": [_i, _i, _p, _p, _p],
    "istnet_pw_wgrad_reduce_multi": [_i, _p, _p, _p, _p, _p],
    "istnet_pw_wgrad_reduce_multi_ld": [_i, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_pack_words": [_i, _p, _p, _p, _p],
    "istnet_bn_relu_mean": [_i, _i, _i, _p, _p, _p, _p],
    "istnet_bn_fin_relu_pool": [_i, _i, _i, _i, _i, _d, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _l, _p, _p, _p],
    "istnet_bn_fin_relu_pool_nbt": [_i, _i, _i, _i, _i, _d, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _l, _p, _p, _p, _p],
    "istnet_interp_grad_csr_dy": [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_expand_rows": [_i, _i, _p, _p, _p],
    # include/istnet_heads.h (csrc/pose_tail.hip)
    "istnet_fc_forward": [_i, _i, _i, _p, _p, _p, _p, _p, _i, _p],
    "istnet_fc_backward": [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p],
    "istnet_ortho6d_forward": [_i, _p, _p, _p],
    "istnet_ortho6d_backward": [_i, _p, _p, _p, _p],
    "istnet_pose_dis_forward": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_pose_dis_backward": [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_smooth_l1_parts": [_l],
    "istnet_smooth_l1_forward": [_l, _f, _p, _p, _p, _p, _p],
    "istnet_smooth_l1_backward": [_l, _f, _p, _p, _p, _p, _p],
    "istnet_mse_parts": [_l],
    "istnet_mse_value_grad": [_l, _p, _p, _p, _p, _p, _p],
    # compact-column form of a set-abstraction scale (csrc/sa_compact.hip)
    "istnet_sa_compact": [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p],
    "istnet_sa_compact_pair": [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _p],
    "istnet_pw_gather_add_cols": [_i, _i, _i, _l, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p],
    "istnet_pw_forward_cols": [_i, _i, _l, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_bn_relu_pool_cols": [_i, _i, _i, _l, _p, _p, _p, _p, _l, _p, _p, _p],
    "istnet_pw_pooled_grad_cols": [_i, _i, _i, _l, _p, _l, _p, _p, _p, _p, _p],
    "istnet_pw_bwd_mid_ok": [_i, _i, _i],
    "istnet_pw_bwd_mid_splits": [_i, _i, _i, _i],
    "istnet_pw_bwd_mid": [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _l, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_pw_bwd_small_cols_splits": [],
    "istnet_pw_bwd_small_cols": [_i, _i, _l, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_pw_dwx_cols_chunks": [_i],
    "istnet_pw_dgrad_cols": [_i, _i, _i, _i, _l, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_pw_wgrad_cols_splits": [_i, _i],
    "istnet_pw_wgrad_cols": [_i, _i, _l, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "istnet_pw_scatter_dy_csr_cols": [_i, _i, _i, _i, _l, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _l, _p, _p, _p, _p],
    "istnet_pw_dwx_cols": [_i, _l, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
}

_lib = None


def declared_symbols():
    """Every function name declared in include/*.h."""
    names = set()
    for path in HEADER_PATHS:
        with open(path) as fh:
            names.update(re.findall(r"ISTNET_PN2_API\s+[a-z ]+\*?\s*\*?(istnet_[a-z0-9_]+)\s*\(", fh.read()))
    return sorted(names)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python __graft_entry__.py build`). There is no CPU or PyTorch fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        handle.istnet_pn2_abi_version.restype = ctypes.c_int
        handle.istnet_pn2_target.restype = ctypes.c_char_p
        got = handle.istnet_pn2_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"libistnet_pn2.so ABI {got} != expected {ABI_VERSION}; rebuild")
        for name, argtypes in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        raise RuntimeError(f"{what} failed with status {status} (hipError_t / ISTNET_PN2_EINVAL=100001)")


# ---- optional per-launch timing (bench.py roofline leg) --------------------------------------
# When TIMING is a list, call sites bracket their kernel launches with HIP events recorded on the
# stream the kernel is launched on (torch's current stream) and append
# (kernel_name, flops, algorithmic_bytes, start_event, end_event).
# TIMING_IN_GRAPH: the step is being CAPTURED for timing.  HIP refuses to read events recorded in a capturing stream
# (hipErrorCapturedEvent, also for external event nodes on this stack), so inside a capture a launch is bracketed by two
# one-thread marker kernels on the launch stream (istnet_debug_marker: each stores the 100 MHz wall clock when the
# stream reaches it) writing to TIMING_BUF; after a replay the buffer holds that replay's times, taken with the
# step's real stream concurrency (side streams and deferred weight gradients stay on: fused_mlp._scale_streams /
# _can_defer).  A bracket includes the two launch gaps (~1.5 us each), so it never flatters the kernel.
TIMING = None
TIMING_IN_GRAPH = False
TIMING_BUF = None      # int64 CUDA tensor of wall-clock slots, two per timed launch
TIMING_ONLY = None     # in-graph timing: bracket only launches of this kernel -- a name, or a predicate of the name (fewer markers = less perturbation)


def timed(name, flops, nbytes, launch):
    if TIMING is None:
        return launch()
    import torch
    if TIMING_IN_GRAPH:
        i = len(TIMING)
        skip = TIMING_ONLY is not None and not (TIMING_ONLY(name) if callable(TIMING_ONLY) else name == TIMING_ONLY)
        if TIMING_BUF is None or 2 * i + 1 >= TIMING_BUF.numel() or skip:
            return launch()
        st = torch.cuda.current_stream(TIMING_BUF.device).cuda_stream
        check(lib().istnet_debug_marker(TIMING_BUF.data_ptr() + 16 * i, st), "debug_marker")
        status = launch()
        check(lib().istnet_debug_marker(TIMING_BUF.data_ptr() + 16 * i + 8, st), "debug_marker")
        TIMING.append((name, flops, nbytes, i, None))
        if i % 4 == 0 and 2 * i + 3 < TIMING_BUF.numel():
            # calibration: an EMPTY bracket at the same place of the step measures what a bracket costs by itself
            # (marker execution + launch gap under the step's load); roofline.measure_replayed subtracts its median
            check(lib().istnet_debug_marker(TIMING_BUF.data_ptr() + 16 * (i + 1), st), "debug_marker")
            check(lib().istnet_debug_marker(TIMING_BUF.data_ptr() + 16 * (i + 1) + 8, st), "debug_marker")
            TIMING.append(("", 0.0, 0.0, i + 1, None))
        return status
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    status = launch()
    end.record()
    TIMING.append((name, flops, nbytes, start, end))
    return status


MARKERS = None   # debug: {"buf": int64 cuda tensor, "names": [...]} set by tools/step_timeline.py


def mark(name):
    """Debug: record when the CURRENT stream reaches this point (no-op unless MARKERS is set)."""
    if MARKERS is None:
        return
    import torch
    i = len(MARKERS["names"])
    if i >= MARKERS["buf"].numel():
        return
    MARKERS["names"].append(name)
    check(lib().istnet_debug_marker(MARKERS["buf"].data_ptr() + 8 * i, torch.cuda.current_stream().cuda_stream),
          "debug_marker")


def affine_consts_multi(items, stream):
    """items: list of (c, gamma_ptr|None, beta_ptr, mean_ptr|None, var_ptr|None, eps, bn_ptr); one launch per <= 8."""
    handle = lib()
    for i in range(0, len(items), 8):
        chunk = items[i:i + 8]
        n = len(chunk)
        arr = lambda k: (ctypes.c_void_p * n)(*[it[k] for it in chunk])
        check(handle.istnet_affine_consts_multi(
            n, (ctypes.c_int * n)(*[it[0] for it in chunk]), arr(1), arr(2), arr(3), arr(4),
            (ctypes.c_float * n)(*[it[5] for it in chunk]), arr(6), stream), "affine_consts_multi")


def reduce_multi(items, stream):
    """items: list of (count, splits, part_ptr, dw_ptr[, cols, ld, pstride]); one launch per <= 8 items.  The long form
    writes a column block of a wider destination from a row block of wider partials (istnet_pw_wgrad_reduce_multi_ld)."""
    handle = lib()
    for i in range(0, len(items), 8):
        chunk = items[i:i + 8]
        n = len(chunk)
        counts = (ctypes.c_int * n)(*[c[0] for c in chunk])
        splits = (ctypes.c_int * n)(*[c[1] for c in chunk])
        parts = (ctypes.c_void_p * n)(*[c[2] for c in chunk])
        dws = (ctypes.c_void_p * n)(*[c[3] for c in chunk])
        if any(len(c) > 4 for c in chunk):
            full = [c if len(c) > 4 else (*c, c[0], c[0], c[0]) for c in chunk]
            cols = (ctypes.c_int * n)(*[c[4] for c in full])
            lds = (ctypes.c_int * n)(*[c[5] for c in full])
            pst = (ctypes.c_longlong * n)(*[c[6] for c in full])
            check(handle.istnet_pw_wgrad_reduce_multi_ld(n, counts, splits, parts, dws, cols, lds, pst, stream),
                  "pw_wgrad_reduce_multi_ld")
        else:
            check(handle.istnet_pw_wgrad_reduce_multi(n, counts, splits, parts, dws, stream), "pw_wgrad_reduce_multi")


def pack_words(srcs, dst, stream):
    """Copy the tensors ``srcs`` (4-byte element types, contiguous) back to back into ``dst``; one launch per <= 64."""
    handle = lib()
    off = 0
    for i in range(0, len(srcs), 64):
        chunk = srcs[i:i + 64]
        n = len(chunk)
        words = [t.numel() * t.element_size() // 4 for t in chunk]
        check(handle.istnet_pack_words(n, (ctypes.c_void_p * n)(*[t.data_ptr() for t in chunk]),
                                       (ctypes.c_longlong * n)(*words), dst.data_ptr() + 4 * off, stream), "pack_words")
        off += sum(words)
