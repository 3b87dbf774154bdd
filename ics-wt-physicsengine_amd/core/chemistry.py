"""Carbonate-buffered water chemistry: the batched equilibrium-pH solve (the reference's
``AqueousChemistry.calculate_pH``, chemistry.py:193-398) runs on the GPU, one Newton-Raphson iteration chain per
element (``ph_solve_kernel`` in csrc/wt_device.hpp); the closed forms around it are ``params`` routines."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np

from . import _native, params


@dataclass
class BufferSystem:
    alkalinity: float           # [mg/L as CaCO3]
    total_carbonate: float      # [mmol/L]
    temperature: float = 20.0   # [degC]

    def validate(self) -> None:
        for name in ("alkalinity", "total_carbonate"):
            if getattr(self, name) < 0:
                raise ValueError(f"{name.replace('_', ' ').capitalize()} cannot be negative: {getattr(self, name)}")


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
    eq = params.equilibrium_constants(T)
    Kw, Ka1, Ka2 = (np.ascontiguousarray(eq[k]) for k in ("Kw", "Ka1", "Ka2"))
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
    """One buffer's equilibrium constants (frozen at the buffer temperature: the same numbers the ensemble uploads as
    that reactor's constants) under the reference's attribute and method names.  Methods take a pH (or an array of
    them) and evaluate the column-wise routines of ``params``."""

    PH_TOLERANCE = 1e-6
    MAX_ITERATIONS = 100
    CONSTANTS = ("Kw", "pKw", "pKa1", "Ka1", "pKa2", "Ka2", "pKa_HOCl", "Ka_HOCl")

    def __init__(self, buffer_system: BufferSystem, device: int = 0):
        from .physics import TemperatureDependentKinetics
        buffer_system.validate()
        self.buffer, self.device = buffer_system, device
        self.thermo = TemperatureDependentKinetics()
        self.thermo.celsius_to_kelvin(buffer_system.temperature)
        frozen = params.equilibrium_constants(np.array([float(buffer_system.temperature)]))
        for name in self.CONSTANTS:
            setattr(self, name, float(frozen[name][0]))

    @staticmethod
    def _out(value):
        value = np.asarray(value)
        return float(value) if value.ndim == 0 else value

    def H_from_pH(self, pH):
        return self._out(params._pow10_neg(pH))

    def pH_from_H(self, H):
        return self._out(-np.log10(H))

    def alpha_carbonate(self, pH):
        return tuple(self._out(a) for a in params.carbonate_fractions(params._pow10_neg(pH), self.Ka1, self.Ka2))

    def buffering_capacity(self, pH):
        return self._out(params.buffer_capacity(params._pow10_neg(pH), self.Kw, self.Ka1, self.Ka2,
                                                self.buffer.total_carbonate / 1000.0))

    def chlorine_speciation(self, total_chlorine_mg_L, pH) -> Dict[str, float]:
        hocl, ocl = (self._out(a) for a in params.hypochlorous_fraction(params._pow10_neg(pH), self.Ka_HOCl))
        return {"HOCl": hocl * total_chlorine_mg_L, "OCl": ocl * total_chlorine_mg_L, "HOCl_fraction": hocl,
                "OCl_fraction": ocl, "effective_disinfection": hocl}

    def pH_dependent_chlorine_decay_factor(self, pH):
        return self._out(params.chlorine_decay_factor(params._pow10_neg(pH), self.Ka_HOCl))

    def calculate_pH(self, initial_guess: float = 7.0, tolerance: float = PH_TOLERANCE, max_iter: int = MAX_ITERATIONS) -> float:
        b = self.buffer
        pH, _, rc = solve_pH(b.alkalinity, b.total_carbonate, b.temperature, initial_guess, tolerance, max_iter, self.device)
        failure = {1: f"Derivative too small at pH={float(pH):.3f}, cannot continue",
                   2: f"pH calculation did not converge after {max_iter} iterations. Final pH={float(pH):.3f}"}.get(int(rc))
        if failure:
            raise RuntimeError(failure)
        return float(pH)

    def _dosed(self, alkalinity_shift: float, current_pH: float) -> float:
        """Equilibrium pH after the alkalinity moved by ``alkalinity_shift`` mg/L as CaCO3 (50 g per equivalent)."""
        b = self.buffer
        return AqueousChemistry(BufferSystem(b.alkalinity + alkalinity_shift, b.total_carbonate, b.temperature),
                                self.device).calculate_pH(initial_guess=current_pH)

    def add_acid(self, volume_L: float, acid_mol: float, current_pH: float) -> float:
        return self._dosed(-(acid_mol / volume_L) * 50000.0, current_pH)       # chemistry.py:355-358

    def add_base(self, volume_L: float, base_mol: float, current_pH: float) -> float:
        return self._dosed((base_mol / volume_L) * 50000.0, current_pH)        # chemistry.py:386-387
