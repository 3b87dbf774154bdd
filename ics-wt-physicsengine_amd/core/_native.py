"""ctypes binding of libwtphys.so (include/wtphys.h).

There is deliberately no fallback: if the shared library is missing, or no HIP
device is present, every compute entry point raises.  The CPU oracle under
``oracle/`` is test infrastructure and is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(_PKG, "csrc")
LIB_PATH = os.environ.get("WTPHYS_LIB", os.path.join(CSRC, "libwtphys.so"))  # override: diagnostic builds

WT_OK, WT_E_ARG, WT_E_HIP, WT_E_NOGPU, WT_E_STATE = 0, 1, 2, 3, 4


class WtError(RuntimeError):
    """A libwtphys call returned a non-zero code."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libwtphys error {code}: {message}")
        self.code = code
        self.message = message


class SolverStats(C.Structure):
    _fields_ = [("nfev", C.c_int32), ("njev", C.c_int32), ("nlu", C.c_int32),
                ("nsteps", C.c_int32), ("nrej", C.c_int32)]


# what libwtphys.so is built from: wtphys.hip and exactly the headers it includes
# (tests/test_host_api.py::test_build_staleness_list_matches_the_includes)
BUILD_SOURCES = ("wtphys.hip", "wt_device.hpp", "wt_sensors.hpp", "wt_plc.hpp", "wt_diag.hpp", "wt_place.hpp")


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/wtphys.hip for gfx950 into csrc/libwtphys.so (hipcc)."""
    srcs = [os.path.join(CSRC, f) for f in BUILD_SOURCES]
    srcs.append(os.path.join(os.path.dirname(_PKG), "include", "wtphys.h"))
    stale = (not os.path.exists(LIB_PATH)
             or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs))
    if force or stale:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-o", LIB_PATH, os.path.join(CSRC, "wtphys.hip")]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    return LIB_PATH


_lib = None


def lib():
    """Load libwtphys.so; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc, gfx950). "
            "There is no CPU fallback for the physics step.")
    L = C.CDLL(LIB_PATH)
    dp, u32p, i32p, vp = C.POINTER(C.c_double), C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.c_void_p
    L.wt_abi_version.restype = C.c_int
    L.wt_wave_diag_slots.restype = C.c_int
    L.wt_last_error.restype = C.c_char_p
    L.wt_device_count.argtypes = [C.POINTER(C.c_int)]
    L.wt_ensemble_create.argtypes = [C.c_int64, C.c_int, C.c_int, dp, C.POINTER(vp)]
    L.wt_ensemble_destroy.argtypes = [vp]
    L.wt_ensemble_set_state.argtypes = [vp, dp, dp, dp, dp]
    L.wt_ensemble_set_boundary.argtypes = [vp, dp]
    L.wt_ensemble_step.argtypes = [vp, C.c_double, C.c_int, C.c_int]
    L.wt_ensemble_set_schedule.argtypes = [vp, C.c_int, C.c_int]
    ip = C.POINTER(C.c_int)
    L.wt_ensemble_get_schedule.argtypes = [vp, ip, ip, ip, ip]
    L.wt_ensemble_get_schedule.restype = C.c_int
    L.wt_ensemble_set_sync.argtypes = [vp, C.c_int]
    L.wt_ensemble_set_step_limit.argtypes = [vp, C.c_int]
    L.wt_ensemble_set_placement.argtypes = [vp, C.c_int]
    L.wt_ensemble_get_placement.argtypes = [vp, ip, i32p]
    L.wt_ensemble_placement_info.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.wt_ensemble_placement_info.restype = C.c_int
    L.wt_ensemble_launch_timing.argtypes = [vp, C.c_int]
    L.wt_ensemble_launch_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.wt_ensemble_synchronize.argtypes = [vp]
    L.wt_ensemble_get_state.argtypes = [vp, dp, dp, dp, dp, dp]
    L.wt_ensemble_get_derived.argtypes = [vp, dp, dp, dp]
    L.wt_ensemble_get_snapshot.argtypes = [vp, dp, dp, dp, dp, dp, dp, dp, dp, u32p]
    L.wt_ensemble_get_snapshot.restype = C.c_int
    L.wt_ensemble_get_status.argtypes = [vp, u32p]
    L.wt_ensemble_get_bad_temperature.argtypes = [vp, dp]
    L.wt_ensemble_get_bad_temperature.restype = C.c_int
    L.wt_ensemble_item_trace.argtypes = [vp, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int)]
    L.wt_ensemble_item_trace.restype = C.c_int
    L.wt_ensemble_item_steps.argtypes = [vp, C.c_int]
    L.wt_ensemble_item_steps.restype = C.c_int
    L.wt_ensemble_queue_error.argtypes = [vp, C.POINTER(C.c_int)]
    L.wt_ensemble_queue_error.restype = C.c_int
    L.wt_ensemble_clear_status.argtypes = [vp]
    L.wt_ensemble_get_stats.argtypes = [vp, C.POINTER(SolverStats)]
    L.wt_ensemble_rhs.argtypes = [vp, dp, dp, dp, dp, dp, dp, u32p]
    L.wt_ensemble_export_state_device.argtypes = [vp, vp]
    L.wt_ensemble_set_stream.argtypes = [vp, vp]
    L.wt_ensemble_timer_start.argtypes = [vp]
    L.wt_ensemble_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    L.wt_selftest_shuffles.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    u8p, fp = C.POINTER(C.c_uint8), C.POINTER(C.c_float)
    L.wt_ensemble_sensors_enable.argtypes = [vp, C.c_uint64, C.c_int64, dp, dp, dp, C.c_int]
    L.wt_ensemble_sensors_get.argtypes = [vp, fp, u8p, u8p]
    L.wt_ensemble_sensors_history.argtypes = [vp, fp, u8p, u8p, i32p]
    u16p = C.POINTER(C.c_uint16)
    L.wt_ensemble_plc_enable.argtypes = [vp]
    L.wt_ensemble_plc_write_holding.argtypes = [vp, u16p, C.c_int64, C.c_int64]
    L.wt_ensemble_plc_read_inputs.argtypes = [vp, u16p, u8p]
    L.wt_ensemble_plc_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    L.wt_ensemble_get_boundary.argtypes = [vp, dp]
    L.wt_ensemble_diagnostics.argtypes = [vp, dp]
    L.wt_ensemble_wave_diag.argtypes = [vp, C.POINTER(C.c_int64), C.c_int64, C.POINTER(C.c_int64)]
    L.wt_ensemble_size.argtypes = [vp]
    L.wt_ensemble_size.restype = C.c_int64
    L.wt_ensemble_zones.argtypes = [vp]
    L.wt_ph_solve.argtypes = [C.c_int, C.c_int64, dp, dp, dp, dp, dp, dp, C.c_double, C.c_int, dp, i32p, i32p]
    for name in ("wt_device_count", "wt_ensemble_create", "wt_ensemble_destroy", "wt_ensemble_set_state",
                 "wt_ensemble_set_boundary", "wt_ensemble_step", "wt_ensemble_synchronize",
                 "wt_ensemble_get_state", "wt_ensemble_get_derived", "wt_ensemble_get_status",
                 "wt_ensemble_clear_status", "wt_ensemble_get_stats", "wt_ensemble_rhs",
                 "wt_ensemble_export_state_device", "wt_ensemble_set_stream", "wt_ensemble_timer_start",
                 "wt_ensemble_diagnostics", "wt_ensemble_plc_enable", "wt_ensemble_plc_write_holding", "wt_ensemble_plc_read_inputs", "wt_ensemble_plc_device", "wt_ensemble_get_boundary",
                 "wt_ensemble_timer_stop", "wt_ensemble_zones", "wt_ph_solve", "wt_selftest_shuffles", "wt_ensemble_wave_diag", "wt_ensemble_set_schedule", "wt_ensemble_set_sync", "wt_ensemble_set_step_limit", "wt_ensemble_set_placement", "wt_ensemble_get_placement", "wt_ensemble_sensors_enable", "wt_ensemble_sensors_get",
                 "wt_ensemble_sensors_history", "wt_ensemble_launch_timing", "wt_ensemble_launch_stats"):
        getattr(L, name).restype = C.c_int
    if L.wt_abi_version() != 1:
        raise ImportError("libwtphys.so ABI version mismatch; rebuild it")
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != WT_OK:
        raise WtError(rc, lib().wt_last_error().decode("utf-8", "replace"))


def dptr(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().wt_device_count(C.byref(n))
    return n.value if rc == WT_OK else 0
