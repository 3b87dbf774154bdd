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
        # chemistry.py:116-132: equilibrium constants frozen at the buffer's temperature (the same
        # values params.derive_constants uploads as the reactor's constants)
        from .physics import TemperatureDependentKinetics
        self.thermo = TemperatureDependentKinetics()
        T = buffer_system.temperature
        self.Kw = self.thermo.water_ionization_constant(T)
        self.pKw = -np.log10(self.Kw)
        self.pKa1 = self.thermo.carbonate_pKa(T, dissociation=1)
        self.Ka1 = 10 ** (-self.pKa1)
        self.pKa2 = self.thermo.carbonate_pKa(T, dissociation=2)
        self.Ka2 = 10 ** (-self.pKa2)
        self.pKa_HOCl = 7.5 + 0.01 * (T - 25.0)
        self.Ka_HOCl = 10 ** (-self.pKa_HOCl)

    # scalar closed forms of one reactor's chemistry (diagnostics, validators); inside step() the same
    # expressions run per zone per RHS evaluation in the kernel (wt_device.hpp prop_pH)
    def H_from_pH(self, pH: float) -> float:
        return 10 ** (-pH)

    def pH_from_H(self, H: float) -> float:
        return -np.log10(H)

    def alpha_carbonate(self, pH: float):
        """chemistry.py:158-191."""
        H = self.H_from_pH(pH)
        D = H ** 2 + self.Ka1 * H + self.Ka1 * self.Ka2
        return H ** 2 / D, (self.Ka1 * H) / D, (self.Ka1 * self.Ka2) / D

    def buffering_capacity(self, pH: float) -> float:
        """chemistry.py:400-437."""
        H = self.H_from_pH(pH)
        beta_water = 2.303 * (H + self.Kw / H)
        C_T_mol = self.buffer.total_carbonate / 1000.0
        a0, a1, a2 = self.alpha_carbonate(pH)
        return beta_water + 2.303 * C_T_mol * (a0 * a1 + 4 * a1 * a2 + a0 * a2)

    def chlorine_speciation(self, total_chlorine_mg_L: float, pH: float):
        """chemistry.py:439-481."""
        H = self.H_from_pH(pH)
        a_HOCl = H / (H + self.Ka_HOCl)
        a_OCl = self.Ka_HOCl / (H + self.Ka_HOCl)
        return {"HOCl": a_HOCl * total_chlorine_mg_L, "OCl": a_OCl * total_chlorine_mg_L, "HOCl_fraction": a_HOCl,
                "OCl_fraction": a_OCl, "effective_disinfection": a_HOCl}

    def pH_dependent_chlorine_decay_factor(self, pH: float) -> float:
        """chemistry.py:483-523."""
        H = self.H_from_pH(pH)
        return H / (H + self.Ka_HOCl) * 1.0 + self.Ka_HOCl / (H + self.Ka_HOCl) * 0.02

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
