"""mav_trajectory_generation_amd -- MI355X-native batched PolynomialOptimization<N>::solveLinear().

Host-side plumbing only: the compute path is the hand-written HIP library
csrc/libmtg_hip.so (C ABI in include/mtg_hip.h).  There is NO CPU fallback: importing the
solver entry points without the built library, or calling them without a GPU, raises.
"""
from .core import Context, Plan, MtgError, MultiSolve, solve_linear_batch, library_path, sample_range  # noqa: F401
from .core import minmax_magnitude, scale_segment_times_to_meet_constraints, pack_multi_items, PackedMultiSolve  # noqa: F401
from .workload import ends_full_masks, random_waypoint_batch  # noqa: F401
from .buckets import MergedRequest, MixedBatchSolver  # noqa: F401
from .time_gradient import mellinger_cost_and_gradient  # noqa: F401
