"""ctypes binding of include/nlmc.h (libnlmc_hip.so).  No torch types cross this boundary.

The library is built in-tree by `__graft_entry__.build()` / `build_library()` (hipcc --offload-arch=gfx950).
There is no CPU fallback: if the shared object is missing, import of the engine fails loudly.
"""
import ctypes
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NLMC_LIB") or os.path.join(_PKG, "lib", "libnlmc_hip.so")   # NLMC_LIB: diagnostic builds
SRC_DIR = os.path.join(_PKG, "csrc")

OK, ERR_ARG, ERR_HIP, ERR_UNSUPPORTED, ERR_STATE = 0, -1, -2, -3, -4
F32, F64 = 0, 1
ORDER_SHARED, ORDER_PER_CHAIN = 0, 1
SPIN_NORMAL, SPIN_SCALED, SPIN_FROZEN_UP, SPIN_FROZEN_DOWN = 0, 1, 2, 3
LDS_N = 24576          # chains up to this length live in LDS (include/nlmc.h: NLMC_LDS_N)
MAX_N = 16777216
ABI_VERSION = 3                      # include/nlmc.h: NLMC_ABI_VERSION
CHAINS_ALL, CHAINS_UNMARKED, CHAINS_MARKED = 0, 1, 2
PHASE_ALL, PHASE_BACKBONE_HOT, PHASE_BACKBONE_FROZEN = 0, 1, 2

EXPORTS = [
    "nlmc_abi_version", "nlmc_device_count", "nlmc_create", "nlmc_destroy", "nlmc_last_error", "nlmc_set_spins",
    "nlmc_get_spins", "nlmc_set_flags", "nlmc_energy", "nlmc_energy_tracked", "nlmc_energy_dev", "nlmc_set_energy_sink", "nlmc_energy_scale", "nlmc_field_scale", "nlmc_energy_of", "nlmc_sweep_stream",
    "nlmc_sweep_philox", "nlmc_plan_philox", "nlmc_plan_philox_fused", "nlmc_fused_modes", "nlmc_plan_reserve_fused", "nlmc_pt_init", "nlmc_pt_get_slots", "nlmc_pt_set_slots",
    "nlmc_pt_apply_swap", "nlmc_pt_swap_philox", "nlmc_pt_plan", "nlmc_pt_rounds_fused", "nlmc_pt_rounds_deferred", "nlmc_pt_check", "nlmc_pt_swap_philox_host", "nlmc_pt_log_begin", "nlmc_pt_log_read", "nlmc_icm_components", "nlmc_icm_move", "nlmc_icm_get_labels", "nlmc_icm_round_philox", "nlmc_icm_round_ladders",
    "nlmc_lbp_convexified", "nlmc_find_clusters", "nlmc_trace_layout", "nlmc_energy_of_recorded",
    "nlmc_last_timing", "nlmc_timing_reset", "nlmc_timing_total", "nlmc_last_schedule_stats",
    "nlmc_pt_mark_slots", "nlmc_select_chains", "nlmc_subset_count", "nlmc_get_subset", "nlmc_track_minimum", "nlmc_backbone_seed", "nlmc_adopt_best",
    "nlmc_backbone_clusters", "nlmc_backbone_check", "nlmc_get_cluster_mask", "nlmc_set_phase", "nlmc_plan_slot", "nlmc_overlap_subsets", "nlmc_own_stream", "nlmc_plan_get_levels", "nlmc_probe_level_round", "nlmc_comm_unique_id", "nlmc_comm_init", "nlmc_comm_probe", "nlmc_comm_check", "nlmc_apt_shard", "nlmc_apt_pack", "nlmc_apt_swap_host", "nlmc_apt_swap_collective", "nlmc_apt_selftest_exchange", "nlmc_pt_swap_philox_collective", "nlmc_set_cluster_mask", "nlmc_host_prefault",
]


def build_library(force=False, verbose=False):
    """hipcc cross-compiles for gfx950 without a GPU present."""
    srcs = [os.path.join(SRC_DIR, f) for f in sorted(os.listdir(SRC_DIR))]
    hdr = os.path.join(os.path.dirname(_PKG), "include", "nlmc.h")
    newest = max(os.path.getmtime(p) for p in srcs + [hdr])
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= newest:
        return LIB_PATH
    os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-o", LIB_PATH, os.path.join(SRC_DIR, "nlmc.hip")]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None
_vp = ctypes.c_void_p
_i = ctypes.c_int
_i64 = ctypes.c_int64
_u32 = ctypes.c_uint32
_u64 = ctypes.c_uint64
_dbl = ctypes.c_double


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the sweep path)")
    L = ctypes.CDLL(LIB_PATH)
    L.nlmc_abi_version.restype = _i
    if L.nlmc_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} implements ABI version {L.nlmc_abi_version()}, this binding needs {ABI_VERSION}: "
                          "rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
    L.nlmc_device_count.restype = _i
    L.nlmc_last_error.restype = ctypes.c_char_p
    L.nlmc_last_error.argtypes = [_vp]
    L.nlmc_create.restype = _i
    L.nlmc_create.argtypes = [ctypes.POINTER(_vp), _i, _vp, _i, _i64, _vp, _vp, _vp, _vp, _i, _i, _i]
    L.nlmc_destroy.restype = None
    L.nlmc_destroy.argtypes = [_vp]
    for name in ("nlmc_set_spins", "nlmc_get_spins", "nlmc_energy", "nlmc_energy_tracked", "nlmc_energy_dev", "nlmc_set_energy_sink", "nlmc_pt_get_slots",
                 "nlmc_pt_set_slots"):
        f = getattr(L, name)
        f.restype = _i
        f.argtypes = [_vp, _vp]
    L.nlmc_set_flags.restype = _i
    L.nlmc_set_flags.argtypes = [_vp, _vp, _dbl]
    L.nlmc_energy_scale.restype = _i
    L.nlmc_energy_scale.argtypes = [_vp]
    L.nlmc_field_scale.restype = _i
    L.nlmc_field_scale.argtypes = [_vp]
    L.nlmc_energy_of.restype = _i
    L.nlmc_energy_of.argtypes = [_vp, _vp, _i64, _vp]
    L.nlmc_sweep_stream.restype = _i
    L.nlmc_sweep_stream.argtypes = [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]
    L.nlmc_sweep_philox.restype = _i
    L.nlmc_sweep_philox.argtypes = [_vp, _i, _i, _i, _u32, _u64, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]
    L.nlmc_plan_philox.restype = _i
    L.nlmc_plan_philox.argtypes = [_vp, _i, _i, _u32, _i, _u64]
    L.nlmc_plan_philox_fused.restype = _i
    L.nlmc_plan_philox_fused.argtypes = [_vp, _u32, _i, _i, _u64, _vp]
    L.nlmc_probe_level_round.restype = _i
    L.nlmc_probe_level_round.argtypes = [_vp, _i, _i, _i, _i, _vp]
    L.nlmc_fused_modes.restype = _i
    L.nlmc_fused_modes.argtypes = [_vp, _i]
    L.nlmc_plan_reserve_fused.restype = _i
    L.nlmc_plan_reserve_fused.argtypes = [_vp, _i, _i]
    L.nlmc_pt_init.restype = _i
    L.nlmc_pt_init.argtypes = [_vp, _i, _vp]
    L.nlmc_pt_apply_swap.restype = _i
    L.nlmc_pt_apply_swap.argtypes = [_vp, _i, _i, _i]
    L.nlmc_pt_swap_philox.restype = _i
    L.nlmc_pt_swap_philox.argtypes = [_vp, _u32, _u64, _i, _vp, _vp, _vp]
    L.nlmc_pt_plan.restype = _i
    L.nlmc_pt_plan.argtypes = [_vp, _u32, _i, _u64, _i]
    L.nlmc_pt_swap_philox_host.restype = _i
    L.nlmc_pt_swap_philox_host.argtypes = [_vp, _u32, _u64, _i, _vp, _vp, _vp]
    L.nlmc_pt_log_begin.restype = _i
    L.nlmc_pt_log_begin.argtypes = [_vp, _u32, _i, _i]
    L.nlmc_pt_log_read.restype = _i
    L.nlmc_pt_log_read.argtypes = [_vp, _vp, _vp]
    L.nlmc_pt_check.restype = _i
    L.nlmc_pt_check.argtypes = [_vp]
    L.nlmc_icm_components.restype = _i
    L.nlmc_icm_components.argtypes = [_vp, _i, _i, _vp]
    L.nlmc_icm_move.restype = _i
    L.nlmc_icm_move.argtypes = [_vp, _i, _i, _i64, _i, _vp]
    L.nlmc_icm_get_labels.restype = _i
    L.nlmc_icm_get_labels.argtypes = [_vp, _vp]
    L.nlmc_icm_round_philox.restype = _i
    L.nlmc_icm_round_philox.argtypes = [_vp, _vp, _i, _u32, _u64, _i, _vp]
    L.nlmc_icm_round_ladders.restype = _i
    L.nlmc_icm_round_ladders.argtypes = [_vp, _u32, _u64, _i, _vp, _vp]
    L.nlmc_lbp_convexified.restype = _i
    L.nlmc_lbp_convexified.argtypes = [_vp, _i, _vp, _vp, _vp, _i, _dbl, _dbl, _i, _dbl, _vp, _vp, _vp, _vp, _vp]
    L.nlmc_find_clusters.restype = _i
    L.nlmc_find_clusters.argtypes = [_i, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _vp, _i64, _vp, _vp]
    L.nlmc_energy_of_recorded.restype = _i
    L.nlmc_energy_of_recorded.argtypes = [_vp, _i, _i, _vp]
    L.nlmc_trace_layout.restype = _i
    L.nlmc_trace_layout.argtypes = [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _i64, _vp, _i, _i]
    L.nlmc_host_prefault.restype = _i
    L.nlmc_host_prefault.argtypes = [_vp, _i64, _i]
    L.nlmc_last_timing.restype = _i
    L.nlmc_last_timing.argtypes = [_vp, _vp, _vp, _vp]
    L.nlmc_timing_reset.restype = _i
    L.nlmc_timing_reset.argtypes = [_vp, _i]
    L.nlmc_timing_total.restype = _i
    L.nlmc_timing_total.argtypes = [_vp, _vp, _vp, _vp, _vp]
    L.nlmc_last_schedule_stats.restype = _i
    L.nlmc_last_schedule_stats.argtypes = [_vp, _vp, _vp]
    for name, args in (("nlmc_pt_mark_slots", [_vp, _vp]), ("nlmc_select_chains", [_vp, _i]), ("nlmc_subset_count", [_vp]),
                       ("nlmc_get_subset", [_vp, _vp]), ("nlmc_track_minimum", [_vp, _i]), ("nlmc_backbone_seed", [_vp, _i]), ("nlmc_adopt_best", [_vp]),
                       ("nlmc_backbone_clusters", [_vp, _vp, _vp, _i, _dbl, _dbl, _i, _dbl, _vp, _i]),
                       ("nlmc_backbone_check", [_vp]), ("nlmc_get_cluster_mask", [_vp, _vp]), ("nlmc_set_cluster_mask", [_vp, _vp]), ("nlmc_set_phase", [_vp, _i, _dbl]),
                       ("nlmc_plan_slot", [_vp, _i]), ("nlmc_overlap_subsets", [_vp, _i]), ("nlmc_own_stream", [_vp]),
                       ("nlmc_plan_get_levels", [_vp, _i, _vp, _i, _vp]), ("nlmc_comm_unique_id", [_vp]), ("nlmc_comm_init", [_vp, _vp, _i, _i]),
                       ("nlmc_pt_swap_philox_collective", [_vp, _u32, _u64, _i, _i, _vp, _vp]), ("nlmc_comm_probe", []), ("nlmc_comm_check", [_vp, _i]),
                       ("nlmc_pt_rounds_fused", [_vp, _i, _i, _i, _u32, _u32, _u64, _i]),
                       ("nlmc_pt_rounds_deferred", [_vp, _i, _i, _i, _u32, _u32, _u64, _i]),
                       ("nlmc_apt_shard", [_vp, _i, _i, _i, _vp]), ("nlmc_apt_pack", [_vp, _vp, _vp, _vp]),
                       ("nlmc_apt_swap_host", [_vp, _u32, _u64, _i, _vp, _vp, _vp, _vp, _vp]),
                       ("nlmc_apt_swap_collective", [_vp, _u32, _u64, _i, _vp, _vp]), ("nlmc_apt_selftest_exchange", [_vp, _vp, _vp])):
        f = getattr(L, name)
        f.restype = _i
        f.argtypes = args
    _lib = L
    return L


def ptr(a):
    """numpy array (C-contiguous) -> void*; None -> NULL."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_vp)


def check(rc, ctx=None):
    """Map C-ABI codes to the exception types the reference raises (ValueError) or RuntimeError."""
    if rc == OK:
        return
    msg = lib().nlmc_last_error(ctx)
    msg = msg.decode() if msg else f"nlmc error {rc}"
    if rc == ERR_ARG:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(msg)


def as_c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)
