"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package never does (see
oracle/wt_oracle.h).  It wraps ``oracle/libwtoracle.so`` built by
``oracle/Makefile``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libwtoracle.so")

NP = 16
NB = 10
ST_T_RANGE, ST_SOLVER_FAILED, ST_CLAMP_PH, ST_CLAMP_CL, ST_CLAMP_T, ST_T_RANGE_POST, ST_NONFINITE = 1, 2, 4, 8, 16, 32, 64
ST_STEP_LIMIT = 128


class Stats(C.Structure):
    _fields_ = [("nfev", C.c_int), ("njev", C.c_int), ("nlu", C.c_int), ("nsteps", C.c_int),
                ("nrej", C.c_int), ("nrhs_total", C.c_int), ("t_internal", C.c_double * 64)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc if the shared object is missing/stale."""
    src = os.path.join(_HERE, "wt_oracle.c")
    hdr = os.path.join(_HERE, "wt_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libwtoracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp = C.POINTER(C.c_double)
        L.wto_rhs.argtypes = [C.c_int, dp, dp, dp, dp]
        L.wto_rhs.restype = C.c_int
        L.wto_step.argtypes = [C.c_int, dp, dp, C.c_double, dp, dp, dp, C.POINTER(Stats)]
        L.wto_step.restype = C.c_int
        L.wto_ensemble_step.argtypes = [C.c_int, C.c_int, dp, dp, C.c_double, C.c_int, dp, dp, dp,
                                        C.POINTER(C.c_int), C.c_int]
        L.wto_ensemble_step.restype = None
        L.wto_calculate_pH.argtypes = [C.c_double] * 7 + [C.c_int, dp, C.POINTER(C.c_int)]
        L.wto_calculate_pH.restype = C.c_int
        L.wto_set_linsolve.argtypes = [C.c_int]
        L.wto_set_step_limit.argtypes = [C.c_long]
        _lib = L
    return _lib


def _dp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def set_linsolve(mode: int) -> None:
    """0 = dense partial-pivot LU (scipy's), 1 = block-triangular tridiagonal (the HIP kernel's)."""
    lib().wto_set_linsolve(int(mode))


def set_step_limit(max_attempts: int) -> None:
    """0 = unlimited (the reference); mirrors ReactorEnsemble.set_step_limit."""
    lib().wto_set_step_limit(int(max_attempts))


def rhs(n: int, par: np.ndarray, bc: np.ndarray, y: np.ndarray) -> Tuple[np.ndarray, int]:
    """derivatives(t, y, boundary) for one reactor. par (NP,), bc (NB,), y (3n,) = [pH.., Cl.., T..]."""
    par = np.ascontiguousarray(par, dtype=np.float64)
    bc = np.ascontiguousarray(bc, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty(3 * n)
    st = lib().wto_rhs(n, _dp(par), _dp(bc), _dp(y), _dp(out))
    return out, st


def step(n: int, par: np.ndarray, bc: np.ndarray, dt: float, y: np.ndarray, t: float,
         want_stats: bool = False):
    """One IntegratedCSTR.step. Returns (y_new, t_new, derived(3n), status[, Stats])."""
    par = np.ascontiguousarray(par, dtype=np.float64)
    bc = np.ascontiguousarray(bc, dtype=np.float64)
    y = np.array(y, dtype=np.float64, copy=True)
    tt = np.array([t], dtype=np.float64)
    der = np.full(3 * n, np.nan)
    st = Stats()
    status = lib().wto_step(n, _dp(par), _dp(bc), float(dt), _dp(y), _dp(tt), _dp(der), C.byref(st))
    if want_stats:
        return y, float(tt[0]), der, status, st
    return y, float(tt[0]), der, status


def ensemble_step(n: int, par_soa: np.ndarray, bc_soa: np.ndarray, dt: float, nsteps: int,
                  pH: np.ndarray, Cl: np.ndarray, T: np.ndarray, t: np.ndarray,
                  nthreads: int = 0, want_derived: bool = False):
    """Advance N reactors ``nsteps`` outer steps on the CPU.

    par_soa (NP, N), bc_soa (NB, N) are the same SoA blocks the HIP library
    takes; pH/Cl/T are (N, n).  Returns (pH, Cl, T, t, status[, derived (N,3,n)]).
    """
    N = pH.shape[0]
    par = np.ascontiguousarray(np.asarray(par_soa, dtype=np.float64).T)
    bc = np.ascontiguousarray(np.asarray(bc_soa, dtype=np.float64).T)
    y = np.ascontiguousarray(np.concatenate([pH, Cl, T], axis=1), dtype=np.float64)
    tt = np.array(t, dtype=np.float64, copy=True)
    status = np.zeros(N, dtype=np.int32)
    der = np.full((N, 3 * n), np.nan) if want_derived else None
    lib().wto_ensemble_step(N, n, _dp(par), _dp(bc), float(dt), int(nsteps), _dp(y), _dp(tt),
                            _dp(der) if want_derived else None,
                            status.ctypes.data_as(C.POINTER(C.c_int)), int(nthreads))
    out = (y[:, :n].copy(), y[:, n:2 * n].copy(), y[:, 2 * n:].copy(), tt, status)
    if want_derived:
        out = out + (der.reshape(N, 3, n),)
    return out


def calculate_pH(Kw: float, Ka1: float, Ka2: float, CT_mol: float, alk: float, guess: float = 7.0,
                 tol: float = 1e-6, max_iter: int = 100):
    """AqueousChemistry.calculate_pH (chemistry.py:271-330). Returns (pH, iters, rc)."""
    out = C.c_double(0.0)
    it = C.c_int(0)
    rc = lib().wto_calculate_pH(Kw, Ka1, Ka2, CT_mol, alk, guess, tol, max_iter, C.byref(out), C.byref(it))
    return out.value, it.value, rc
