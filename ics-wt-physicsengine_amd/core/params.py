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

    chemistry.py:123,126,132,144 evaluate ``10 ** (-x)`` on Python floats; numpy's
    vectorised power differs from libm in the last bit for ~5 % of inputs, so
    the scalar routine is used on the unique values.
    """
    x = np.asarray(x, dtype=np.float64)
    uniq, inv = np.unique(x, return_inverse=True)
    vals = np.array([10 ** (-float(v)) for v in uniq], dtype=np.float64)
    return vals[inv].reshape(x.shape)


# --------------------------------------------------------------------------- temperature relations (thermodynamics.py)
T_MIN_C, T_MAX_C = 0.0, 100.0          # thermodynamics.py:117-118
LIQUID_RANGE_TEXT = ("Temperature {value}°C outside liquid water range [{lo}, {hi}]°C. This indicates either:\n"
                     "  1. Invalid input data\n"
                     "  2. Numerical instability in ODE integration (reduce tolerances)\n"
                     "  3. System requires pressurized/supercooled water model")


def kelvin(temp_c) -> np.ndarray:
    """Absolute temperature of every element; the first element outside the liquid range raises the reference's
    ValueError (thermodynamics.py:129-158) naming that element."""
    t = np.asarray(temp_c, dtype=np.float64)
    outside = ~((t >= T_MIN_C) & (t <= T_MAX_C))
    if outside.any():
        first = t[outside].flat[0]
        shown = float(first) if t.ndim == 0 else np.float64(first)
        raise ValueError(LIQUID_RANGE_TEXT.format(value=shown, lo=T_MIN_C, hi=T_MAX_C))
    return t + 273.15


def arrhenius(temp_c, k_ref: float, activation_energy: float, t_ref_k: float = T_REFERENCE_K) -> np.ndarray:
    """k(T) = k_ref exp(-(E_a / R)(1/T - 1/T_ref))  (thermodynamics.py:160-193)."""
    slope = -(activation_energy / R_GAS)
    return k_ref * np.exp(slope * (1.0 / kelvin(temp_c) - 1.0 / t_ref_k))


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


def equilibrium_constants(temp_c) -> Dict[str, np.ndarray]:
    """Kw, Ka1, Ka2, Ka(HOCl) and their pK values frozen at ``temp_c`` (chemistry.py:116-132): the four chemistry
    rows of the constant block, and what an ``AqueousChemistry`` object holds."""
    t = np.asarray(temp_c, dtype=np.float64)
    out = {"Kw": water_ionization_constant(t), "pKa1": carbonate_pKa(t, 1), "pKa2": carbonate_pKa(t, 2),
           "pKa_HOCl": 7.5 + 0.01 * (t - 25.0)}
    out["pKw"] = -np.log10(out["Kw"])
    for name in ("Ka1", "Ka2", "Ka_HOCl"):
        out[name] = _pow10_neg(out["p" + name])
    return out


# --------------------------------------------------------------------------- carbonate / chlorine closed forms (chemistry.py)
def carbonate_fractions(H, Ka1, Ka2):
    """(alpha0, alpha1, alpha2) of H2CO3 / HCO3- / CO3-- at hydrogen-ion activity H (chemistry.py:158-191)."""
    first, both = Ka1 * H, Ka1 * Ka2
    total = H ** 2 + first + both
    return H ** 2 / total, first / total, both / total


def buffer_capacity(H, Kw, Ka1, Ka2, carbonate_mol):
    """beta = 2.303 (H + Kw/H) + 2.303 C_T (a0 a1 + 4 a1 a2 + a0 a2)  (chemistry.py:400-437)."""
    a0, a1, a2 = carbonate_fractions(H, Ka1, Ka2)
    water = 2.303 * (H + Kw / H)
    return water + 2.303 * carbonate_mol * (a0 * a1 + 4 * a1 * a2 + a0 * a2)


def hypochlorous_fraction(H, Ka_HOCl):
    """(HOCl fraction, OCl- fraction) of free chlorine (chemistry.py:439-481)."""
    pool = H + Ka_HOCl
    return H / pool, Ka_HOCl / pool


def chlorine_decay_factor(H, Ka_HOCl, ocl_relative_rate: float = 0.02):
    """HOCl decays at the full rate, OCl- at 2 % of it (chemistry.py:483-523)."""
    hocl, ocl = hypochlorous_fraction(H, Ka_HOCl)
    return hocl * 1.0 + ocl * ocl_relative_rate


# --------------------------------------------------------------------------- water column (spatial.py)
G_GRAVITY = 9.81


def water_density(temp_c, salinity_g_L=0.0):
    """rho(T): parabola around the 4 degC maximum up to 8 degC, linear expansion from the 20 degC value above
    (the two branches do not meet at 8 degC: spatial.py:142-197), plus 0.7 kg/m3 per g/L of dissolved solids."""
    t = np.asarray(temp_c, dtype=np.float64)
    cold = 999.97 + (-0.008 * (t - 4.0) ** 2)
    warm = 998.2 + (-2.1e-4 * 998.2 * (t - 20.0))
    return np.where(t <= 8.0, cold, warm) + 0.7 * np.asarray(salinity_g_L, dtype=np.float64)


def interface_richardson(density, zone_height, velocity_scale):
    """Gradient Richardson number of every interface between zone i and i+1 along the last axis
    (spatial.py:239-277); +inf where the velocity scale is not above 1e-6 m/s."""
    rho = np.asarray(density, dtype=np.float64)
    below, above = rho[..., :-1], rho[..., 1:]
    if not velocity_scale > 1e-6:
        return np.full(below.shape, np.inf)
    return (G_GRAVITY * (above - below) * zone_height) / ((0.5 * (below + above)) * velocity_scale ** 2)


def suppression_factors(density, zone_height, velocity_scale, critical_richardson=0.25, factor=0.5):
    """Interface mixing factors: ``factor`` where Ri exceeds the critical value, 1 elsewhere (spatial.py:295-320)."""
    ri = interface_richardson(density, zone_height, velocity_scale)
    return np.where(ri > critical_richardson, factor, 1.0)


def profile_statistics(profile, zone_height) -> Dict[str, np.ndarray]:
    """The eight numbers of spatial.py:440-477 for every profile along the last axis."""
    x = np.asarray(profile, dtype=np.float64)
    steep = np.abs(np.diff(x, axis=-1) / zone_height)
    top, bottom = x.max(axis=-1), x.min(axis=-1)
    return {"mean_value": x.mean(axis=-1), "std_value": x.std(axis=-1), "max_value": top, "min_value": bottom,
            "range": top - bottom, "max_gradient": steep.max(axis=-1), "mean_gradient": steep.mean(axis=-1),
            "gradient_location": steep.argmax(axis=-1)}


def mixing_quality(profile):
    """(coefficient of variation, Danckwerts-style segregation index clipped to [0, 1]) of every profile along the
    last axis (transport.py:338-384); both 0 where the mean is not positive."""
    x = np.asarray(profile, dtype=np.float64)
    mean, spread = x.mean(axis=-1), x.std(axis=-1)
    live = mean > 0
    safe = np.where(live, mean, 1.0)
    cv = np.where(live, spread / safe, 0.0)
    seg = np.where(live, np.clip(spread ** 2 / safe ** 2, 0.0, 1.0), 0.0)
    return cv, seg


# --------------------------------------------------------------------------- transport (transport.py)
WATER_VISCOSITY = 1e-6      # transport.py:162
C_MIXING = 12.0             # transport.py:168


def transport_columns(cfg: Dict[str, np.ndarray], n_zones: int) -> Dict[str, np.ndarray]:
    """Everything ``TransportModel.__init__`` derives (transport.py:202-254, 256-290), column-wise for N reactors.
    ``cfg`` needs volume, height, diameter, flow_rate, impeller_speed, impeller_diameter, power_number, temperature."""
    col = lambda k: np.ascontiguousarray(cfg[k], dtype=np.float64)
    volume, height, diameter, flow = col("volume"), col("height"), col("diameter"), col("flow_rate")
    d_imp, rpm = col("impeller_diameter"), col("impeller_speed")
    n_p = col("power_number") if "power_number" in cfg else np.full_like(volume, 5.0)   # only the mixing-time estimate uses it
    t = {"cross_sectional_area": np.pi * (diameter / 2) ** 2, "zone_height": height / n_zones, "zone_volume": volume / n_zones}
    t["superficial_velocity"] = (flow / 60000.0) / t["cross_sectional_area"]
    n_rps = rpm / 60.0
    t["impeller_tip_speed"] = np.pi * d_imp * rpm / 60.0
    t["Re"] = (rpm / 60.0) * d_imp ** 2 / WATER_VISCOSITY
    t["D_turbulent"] = 0.1 * n_rps * d_imp ** 2
    t["D_molecular"] = diffusion_coefficient(col("temperature"))
    t["D_effective"] = t["D_turbulent"] + t["D_molecular"]
    with np.errstate(divide="ignore", invalid="ignore"):
        t["mixing_time_seconds"] = C_MIXING * (height / d_imp) / (n_rps * n_p ** (1.0 / 3.0))
        t["residence_time"] = np.where(flow > 0, volume / np.where(flow > 0, flow, 1.0), np.nan)
    t["mixing_time"] = t["mixing_time_seconds"] / 60.0
    t["Pe"] = height * t["superficial_velocity"] / t["D_effective"]
    t["K_exchange_per_s"] = (t["D_effective"] * t["cross_sectional_area"] / t["zone_height"]) / (t["zone_volume"] / 1000.0)
    t["Q_per_V"] = (flow / 60.0) / volume
    return t


def exchange_matrix(k_exchange_per_s: float, q_per_v: float, n_zones: int) -> np.ndarray:
    """The constant inter-zone exchange operator [1/s] (transport.py:256-336): nearest-neighbour exchange with a
    zero row sum, and the outflow sink on the top zone.  The kernel never forms it: it is a three-point stencil
    (csrc/wt_device.hpp, mix3)."""
    K = k_exchange_per_s * (np.eye(n_zones, k=1) + np.eye(n_zones, k=-1))
    K -= np.diag(K.sum(axis=1))
    K[-1, -1] -= q_per_v
    return K


def derive_constants(cfg: Dict[str, np.ndarray], n_zones: int) -> np.ndarray:
    """Per-reactor constants ``par[NP][N]`` from configuration columns.

    ``cfg`` maps ReactorConfiguration field names (reactor.py:61-89) to arrays
    of length N.  Temperature-dependent equilibrium constants are frozen at the
    *configuration* temperature exactly as chemistry.py:116-132 does.
    """
    f64 = lambda k: np.ascontiguousarray(cfg[k], dtype=np.float64)
    N = f64("volume").shape[0]
    par = np.zeros((NP, N), dtype=np.float64)
    par[P_VOLUME], par[P_HEIGHT], par[P_DIAMETER] = f64("volume"), f64("height"), f64("diameter")
    eq = equilibrium_constants(f64("temperature"))
    par[P_KW], par[P_KA1], par[P_KA2], par[P_KA_HOCL] = eq["Kw"], eq["Ka1"], eq["Ka2"], eq["Ka_HOCl"]
    par[P_CT_MOL] = f64("total_carbonate") / 1000.0            # chemistry.py:428
    tr = transport_columns(cfg, n_zones)
    par[P_USUP], par[P_KEX] = tr["superficial_velocity"], tr["K_exchange_per_s"]
    # spatial.py:57-72 via reactor.py:259-262
    par[P_STRAT] = np.asarray(cfg["enable_thermal_stratification"], dtype=bool).astype(np.float64)
    par[P_RI_CRIT] = 0.25
    par[P_SUPP] = 0.5
    return par


def residence_time_min(volume: float, flow_rate: float):
    """transport.py:216-219 (None in batch mode)."""
    return volume / flow_rate if flow_rate > 0 else None
