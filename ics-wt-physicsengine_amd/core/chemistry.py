"""Batched carbonate-system pH solver (the reference's AqueousChemistry.calculate_pH,
chemistry.py:193-398), one Newton-Raphson solve per element on the GPU."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Tuple

import numpy as np

from . import _native, params


@dataclass
class BufferSystem:
    """chemistry.py:54-80."""
    alkalinity: float
    total_carbonate: float
    temperature: float = 20.0

    def validate(self) -> None:
        if self.alkalinity < 0:
            raise ValueError(f"Alkalinity cannot be negative: {self.alkalinity}")
        if self.total_carbonate < 0:
            raise ValueError(f"Total carbonate cannot be negative: {self.total_carbonate}")


def solve_pH(alkalinity, total_carbonate, temperature, initial_guess=7.0, tolerance: float = 1e-6,
             max_iter: int = 100, device: int = 0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Equilibrium pH of each (alkalinity [mg/L CaCO3], C_T [mmol/L], T [degC]) triple.

    Returns (pH, iterations, rc) with rc 0 = converged, 1 = derivative too small,
    2 = not converged (the reference raises RuntimeError for 1 and 2).
    """
    alk, ct, T, g = np.broadcast_arrays(*[np.asarray(x, dtype=np.float64) for x in
                                          (alkalinity, total_carbonate, temperature, initial_guess)])
    shape = alk.shape
    alk, ct, T, g = [np.ascontiguousarray(x.ravel()) for x in (alk, ct, T, g)]
    n = alk.size
    Kw = np.ascontiguousarray(params.water_ionization_constant(T))
    Ka1 = np.ascontiguousarray(params._pow10_neg(params.carbonate_pKa(T, 1)))
    Ka2 = np.ascontiguousarray(params._pow10_neg(params.carbonate_pKa(T, 2)))
    ct_mol = np.ascontiguousarray(ct / 1000.0)
    pH = np.empty(n)
    it = np.zeros(n, dtype=np.int32)
    rc = np.zeros(n, dtype=np.int32)
    i32 = C.POINTER(C.c_int32)
    _native.check(_native.lib().wt_ph_solve(int(device), n, _native.dptr(Kw), _native.dptr(Ka1), _native.dptr(Ka2),
                                            _native.dptr(ct_mol), _native.dptr(alk), _native.dptr(g),
                                            float(tolerance), int(max_iter), _native.dptr(pH),
                                            it.ctypes.data_as(i32), rc.ctypes.data_as(i32)))
    return pH.reshape(shape), it.reshape(shape), rc.reshape(shape)


class AqueousChemistry:
    """Scalar convenience wrapper with the reference's method names and errors."""

    PH_TOLERANCE = 1e-6
    MAX_ITERATIONS = 100

    def __init__(self, buffer_system: BufferSystem, device: int = 0):
        buffer_system.validate()
        self.buffer = buffer_system
        self.device = device

    def calculate_pH(self, initial_guess: float = 7.0, tolerance: float = PH_TOLERANCE,
                     max_iter: int = MAX_ITERATIONS) -> float:
        pH, it, rc = solve_pH(self.buffer.alkalinity, self.buffer.total_carbonate, self.buffer.temperature,
                              initial_guess, tolerance, max_iter, self.device)
        if int(rc) == 1:
            raise RuntimeError(f"Derivative too small at pH={float(pH):.3f}, cannot continue")
        if int(rc) == 2:
            raise RuntimeError(f"pH calculation did not converge after {max_iter} iterations. "
                               f"Final pH={float(pH):.3f}")
        return float(pH)

    def add_acid(self, volume_L: float, acid_mol: float, current_pH: float) -> float:
        delta_alk = -(acid_mol / volume_L) * 50000.0            # chemistry.py:355-358
        nb = BufferSystem(self.buffer.alkalinity + delta_alk, self.buffer.total_carbonate, self.buffer.temperature)
        return AqueousChemistry(nb, self.device).calculate_pH(initial_guess=current_pH)

    def add_base(self, volume_L: float, base_mol: float, current_pH: float) -> float:
        delta_alk = (base_mol / volume_L) * 50000.0             # chemistry.py:386-387
        nb = BufferSystem(self.buffer.alkalinity + delta_alk, self.buffer.total_carbonate, self.buffer.temperature)
        return AqueousChemistry(nb, self.device).calculate_pH(initial_guess=current_pH)
