"""ctypes binding of libmtg_hip.so (include/mtg_hip.h).  Fails loudly if the HIP library is missing."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MTG_HIP_LIB selects an alternative build of the same HIP library (A/B kernel experiments); still HIP-only.
LIB_PATH = os.environ.get("MTG_HIP_LIB") or os.path.join(_HERE, "csrc", "libmtg_hip.so")

c_double_p = ctypes.c_void_p  # raw device or host addresses


class PlanDesc(ctypes.Structure):
    _fields_ = [("n_coeffs", ctypes.c_int32), ("dimension", ctypes.c_int32), ("n_segments", ctypes.c_int32),
                ("derivative_to_optimize", ctypes.c_int32), ("fixed_mask", ctypes.POINTER(ctypes.c_uint32))]


class PlanInfo(ctypes.Structure):
    _fields_ = [("n_all", ctypes.c_int32), ("n_fixed", ctypes.c_int32), ("n_free", ctypes.c_int32),
                ("kernel_variant", ctypes.c_int32), ("algorithmic_bytes_per_trajectory", ctypes.c_int64)]


class Layout(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in (
        "times_stride_b", "times_stride_k", "fixed_stride_b", "fixed_stride_d", "fixed_stride_c",
        "free_stride_b", "free_stride_d", "free_stride_c")]


class MultiItem(ctypes.Structure):
    _fields_ = [("plan", ctypes.c_void_p), ("batch", ctypes.c_int64), ("layout", Layout), ("times", ctypes.c_void_p),
                ("d_fixed", ctypes.c_void_p), ("coeffs", ctypes.c_void_p), ("d_free", ctypes.c_void_p),
                ("cost", ctypes.c_void_p)]


EXPORTS = {
    "mtg_context_create": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    "mtg_context_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "mtg_context_sync": (ctypes.c_int, [ctypes.c_void_p]),
    "mtg_context_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int]),   # include/mtg_hip_lab.h
    "mtg_lab_segment_cost_matrices": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]),   # include/mtg_hip_lab.h
    "mtg_lab_refine_residual": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p]),   # include/mtg_hip_lab.h
    "mtg_lab_clock_probe_start": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.POINTER(ctypes.c_void_p)]),   # include/mtg_hip_lab.h
    "mtg_lab_clock_probe_finish": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "mtg_last_error_string": (ctypes.c_char_p, [ctypes.c_void_p]),
    "mtg_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "mtg_plan_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(PlanDesc), ctypes.POINTER(ctypes.c_void_p)]),
    "mtg_plan_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "mtg_plan_get_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(PlanInfo)]),
    "mtg_plan_context": (ctypes.c_void_p, [ctypes.c_void_p]),
    "mtg_plan_rank_deficiency": (ctypes.c_int, [ctypes.c_void_p]),
    "mtg_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "mtg_comm_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    "mtg_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "mtg_comm_rank": (ctypes.c_int, [ctypes.c_void_p]),
    "mtg_comm_world": (ctypes.c_int, [ctypes.c_void_p]),
    "mtg_comm_last_error": (ctypes.c_char_p, [ctypes.c_void_p]),
    "mtg_comm_all_gather": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
    "mtg_comm_solve_all_gather": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(Layout), ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32]),
    "mtg_comm_sync": (ctypes.c_int, [ctypes.c_void_p]),
    "mtg_plan_get_shape": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.POINTER(ctypes.c_int32)] * 4),
    "mtg_structural_rank_deficiency": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_uint32)]),
    "mtg_plan_launch_form": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(Layout), ctypes.c_uint32]),
    "mtg_layout_aos": (None, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(Layout)]),
    "mtg_layout_soa": (None, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(Layout)]),
    "mtg_layout_soa_padded": (None, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(Layout)]),
    "mtg_plan_set_workspace": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "mtg_solve_linear": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(Layout), c_double_p, c_double_p,
                                        c_double_p, c_double_p, c_double_p, ctypes.c_uint32]),
    "mtg_solve_linear_status": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(Layout), c_double_p, c_double_p,
                                               c_double_p, c_double_p, c_double_p, ctypes.c_void_p, ctypes.c_uint32]),
    "mtg_basic_solution_host": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, c_double_p, ctypes.POINTER(ctypes.c_int32)]),
    "mtg_solve_linear_sequence": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.POINTER(Layout),
                                                 ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]),
    "mtg_solve_linear_sequence_events": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.POINTER(Layout),
                                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32,
                                                        ctypes.c_void_p, ctypes.c_void_p]),
    "mtg_mellinger_cost_gradient": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(Layout), c_double_p,
                                                   c_double_p, ctypes.c_double, ctypes.c_double, c_double_p, c_double_p]),
    "mtg_generate_waypoints": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(Layout), ctypes.c_uint64,
                                              ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int32, c_double_p,
                                              c_double_p]),
    "mtg_compare_coefficients": (ctypes.c_int, [ctypes.c_void_p, c_double_p, c_double_p, ctypes.c_int64, ctypes.c_int32,
                                                ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "mtg_shard_range": (None, [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]),
    "mtg_device_group_create": (ctypes.c_int, [ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(PlanDesc), ctypes.POINTER(ctypes.c_void_p)]),
    "mtg_device_group_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "mtg_device_group_size": (ctypes.c_int, [ctypes.c_void_p]),
    "mtg_device_group_context": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int32]),
    "mtg_device_group_plan": (ctypes.c_void_p, [ctypes.c_void_p, ctypes.c_int32]),
    "mtg_device_group_solve_linear": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]),
    "mtg_device_group_sync": (ctypes.c_int, [ctypes.c_void_p]),
    "mtg_device_group_gather_coeffs": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]),
    "mtg_update_segments_from_free": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.POINTER(Layout), c_double_p,
                                                     c_double_p, c_double_p, c_double_p, c_double_p, ctypes.c_uint32]),
    "mtg_device_malloc": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    "mtg_device_free": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "mtg_copy_to_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "mtg_copy_to_host": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "mtg_sample_range": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                        c_double_p, c_double_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_double,
                                        ctypes.c_double, ctypes.c_int32, ctypes.c_int32, c_double_p, ctypes.c_void_p]),
    "mtg_minmax_magnitude": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                            c_double_p, c_double_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                            ctypes.c_uint32, c_double_p, c_double_p, ctypes.c_void_p]),
    "mtg_scale_segment_times_to_meet_constraints": (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64, c_double_p, c_double_p,
        ctypes.c_int64, ctypes.c_int64, ctypes.c_double, ctypes.c_double, ctypes.c_int32, c_double_p, c_double_p,
        ctypes.c_void_p]),
    "mtg_multi_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(MultiItem), ctypes.c_uint32,
                                        ctypes.POINTER(ctypes.c_void_p)]),
    "mtg_multi_solve": (ctypes.c_int, [ctypes.c_void_p]),
    "mtg_multi_launch_count": (ctypes.c_int, [ctypes.c_void_p]),
    "mtg_multi_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "mtg_time_last_solve": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
    "mtg_selftest_rcp": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]),
}

# Measurement knobs (include/mtg_hip_lab.h): the library never reads the environment; Context forwards these variables of
# the process it is created in -- environment variable -> (option name, how its text becomes the option's value)
_flag = lambda e: 1
ENV_OPTIONS = {
    "MTG_FORCE_DG": ("force_dg", int), "MTG_PREFER_ROLLED": ("prefer_rolled", _flag), "MTG_NO_DIMLANE": ("no_dimlane", _flag),
    "MTG_NO_SLAB": ("no_slab", _flag), "MTG_NO_QUEUE": ("no_queue", _flag),
    "MTG_NO_SLAB_EXTRA": ("no_slab_extra", _flag), "MTG_NO_DL_EXTRA": ("no_dl_extra", _flag),
    "MTG_NO_BALANCE": ("no_balance", _flag), "MTG_DL_RT": ("dl_rt", int),
    "MTG_DL_GRID_PER_CU": ("dl_grid_per_cu", int), "MTG_DL_ANY_SCHED": ("dl_any_sched_rr", lambda e: int(e == "rr")),
    "MTG_SLAB_POLICY": ("slab_policy", lambda e: 1 if int(e) else 0), "MTG_ROLLED_WG_PER_CU": ("rolled_wg_per_cu", int),
    "MTG_DL_MAX_UNITS": ("dl_max_units", int), "MTG_SAMPLE_GENERIC": ("sample_generic", _flag),
    "MTG_COOP": ("coop", int),
    "MTG_EXTREMA_SPLIT": ("extrema_split", int),
}

FLAG_HOST_POINTERS = 1
FLAG_GENERIC_KERNEL = 2
FLAG_FUSED_DIMS = 4
FLAG_SPLIT_DIMS = 8
FLAG_COST_ONLY = 16
FLAG_DIMLANE = 32
FLAG_HOST_BACKEND = 64
FLAG_CONCURRENT_ITEMS = 128
FLAG_SEQUENCE_ONE_LAUNCH_PER_BATCH = 256
FLAG_QUERY_EXTRA_OUTPUTS = 512
FLAG_BASIC_SOLUTION = 1024
FLAG_COOPERATIVE = 2048
FLAG_REFINE = 4096

_lib = None


def load():
    """Load libmtg_hip.so and attach prototypes.  Raises if it was not built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  The batched solver has no CPU fallback (only single-trajectory host calls can opt "
            "into the library's own host build of the kernel code, MTG_FLAG_HOST_BACKEND -- which lives in the same library).")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
