"""Init-time constants of the CSTR hot path, derived per reactor on the host.

The reference computes these once in ``IntegratedCSTR.__init__`` through its
physics sub-objects (reactor.py:229-270).  They never change during a run, so
the ensemble uploads them once as a structure-of-arrays block ``par[k][r]``.
Every expression below is evaluated in the same order, and with the same
scalar math routine, as the reference line it cites, so that the values are
bit-identical (checked against tests/golden/g1_constants.json).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

# index of each per-reactor constant in the SoA block (mirrors include/wtphys.h)
P_VOLUME, P_HEIGHT, P_DIAMETER = 0, 1, 2
P_KW, P_KA1, P_KA2, P_KA_HOCL, P_CT_MOL = 3, 4, 5, 6, 7
P_KEX, P_USUP, P_STRAT, P_RI_CRIT, P_SUPP = 8, 9, 10, 11, 12
NP = 16

# BoundaryConditions field order (reactor.py:169-186) == boundary SoA row order
BOUNDARY_FIELDS = (
    "inlet_flow_rate", "inlet_pH", "inlet_chlorine", "inlet_temperature",
    "acid_flow_rate", "acid_concentration",
    "chlorine_flow_rate", "chlorine_concentration",
    "ambient_temperature", "heat_loss_coefficient",
)
NB = len(BOUNDARY_FIELDS)

R_GAS = 8.314            # thermodynamics.py:54
T_REFERENCE_K = 293.15   # thermodynamics.py:55


def _pow10_neg(x: np.ndarray) -> np.ndarray:
    """``10 ** (-x)`` with Python-float semantics (libm pow), element by element.

    chemistry.py:123,126,132 evaluate ``10 ** (-pKa)`` on Python floats; numpy's
    vectorised power differs from libm in the last bit for ~5 % of inputs, so
    the scalar routine is used on the unique values.
    """
    x = np.asarray(x, dtype=np.float64)
    uniq, inv = np.unique(x, return_inverse=True)
    vals = np.array([10 ** (-float(v)) for v in uniq], dtype=np.float64)
    return vals[inv].reshape(x.shape)


def water_ionization_constant(temp_c: np.ndarray) -> np.ndarray:
    """Kw(T), van't Hoff form (thermodynamics.py:195-226)."""
    T_K = np.asarray(temp_c, dtype=np.float64) + 273.15
    exponent = (55900.0 / R_GAS) * (1.0 / 298.15 - 1.0 / T_K)
    return 1.0e-14 * np.exp(exponent)


def carbonate_pKa(temp_c: np.ndarray, dissociation: int) -> np.ndarray:
    """Linear pKa(T) (thermodynamics.py:254-290)."""
    if dissociation not in (1, 2):
        raise ValueError(f"Dissociation must be 1 or 2, got {dissociation}")
    ref = 6.35 if dissociation == 1 else 10.33
    return ref + (-0.008) * (np.asarray(temp_c, dtype=np.float64) - 25.0)


def diffusion_coefficient(temp_c: np.ndarray) -> np.ndarray:
    """Stokes-Einstein D(T) (thermodynamics.py:292-331)."""
    T_K = np.asarray(temp_c, dtype=np.float64) + 273.15
    exponent = 1800.0 * (1.0 / T_K - 1.0 / T_REFERENCE_K)
    viscosity_ratio = np.exp(-exponent)
    return 1.0e-9 * (T_K / T_REFERENCE_K) * viscosity_ratio


def derive_constants(cfg: Dict[str, np.ndarray], n_zones: int) -> np.ndarray:
    """Per-reactor constants ``par[NP][N]`` from configuration columns.

    ``cfg`` maps ReactorConfiguration field names (reactor.py:61-89) to arrays
    of length N.  Temperature-dependent equilibrium constants are frozen at the
    *configuration* temperature exactly as chemistry.py:116-132 does.
    """
    f64 = lambda k: np.ascontiguousarray(cfg[k], dtype=np.float64)
    volume, height, diameter = f64("volume"), f64("height"), f64("diameter")
    T = f64("temperature")
    N = volume.shape[0]
    par = np.zeros((NP, N), dtype=np.float64)
    par[P_VOLUME], par[P_HEIGHT], par[P_DIAMETER] = volume, height, diameter

    # chemistry.py:116-132
    par[P_KW] = water_ionization_constant(T)
    par[P_KA1] = _pow10_neg(carbonate_pKa(T, 1))
    par[P_KA2] = _pow10_neg(carbonate_pKa(T, 2))
    par[P_KA_HOCL] = _pow10_neg(7.5 + 0.01 * (T - 25.0))
    par[P_CT_MOL] = f64("total_carbonate") / 1000.0            # chemistry.py:428

    # transport.py:202-254, 256-290
    area = np.pi * (diameter / 2) ** 2                          # transport.py:101-104
    zone_height = height / n_zones
    zone_volume = volume / n_zones
    q_m3_s = f64("flow_rate") / 60000.0
    par[P_USUP] = q_m3_s / area
    n_rps = f64("impeller_speed") / 60.0
    d_imp = f64("impeller_diameter")
    d_turb = 0.1 * n_rps * d_imp ** 2
    d_eff = d_turb + diffusion_coefficient(T)
    k_exchange = d_eff * area / zone_height
    par[P_KEX] = k_exchange / (zone_volume / 1000.0)

    # spatial.py:57-72 via reactor.py:259-262
    par[P_STRAT] = np.asarray(cfg["enable_thermal_stratification"], dtype=bool).astype(np.float64)
    par[P_RI_CRIT] = 0.25
    par[P_SUPP] = 0.5
    return par


def residence_time_min(volume: float, flow_rate: float):
    """transport.py:216-219 (None in batch mode)."""
    return volume / flow_rate if flow_rate > 0 else None
