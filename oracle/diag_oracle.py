"""CPU oracle of the reactor diagnostics (TEST INFRASTRUCTURE ONLY) -- SURVEY.md section 8(f) NEXT-4.

Restates, vectorised over reactors but with numpy's own per-array summation order over zones,

    IntegratedCSTR.validate_conservation          core/reactor.py:570-611
    TransportModel.calculate_mixing_quality       core/transport.py:338-384   (pH and chlorine, reactor.py:638-639)
    SpatialModel.identify_thermocline             core/spatial.py:352-379
    SpatialModel.calculate_spatial_gradients      core/spatial.py:440-477

np.sum / np.mean / np.std of a contiguous float64 vector use numpy's pairwise summation
(numpy/core/src/umath/loops_utils.h.src, numpy 2.2): fewer than 8 elements are added left to
right; up to 128 elements go through eight interleaved accumulators that are combined as
((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and the tail is added one by one.  Pinned against
tests/golden/g9_diag_n*.npz (outputs of the reference itself).
"""
from __future__ import annotations

import numpy as np

N_DIAG = 34
FIELDS = ("total_chlorine_mg", "total_H_mol", "total_OH_mol", "charge_balance_mol", "thermal_energy_kJ",
          "pH_CV", "pH_segregation", "chlorine_CV", "chlorine_segregation", "thermocline_depth_m") + tuple(
    f"{p}_{k}" for p in ("pH", "chlorine", "temperature")
    for k in ("mean_value", "std_value", "max_value", "min_value", "range", "max_gradient", "mean_gradient", "gradient_location"))
KW_25C, DELTA_H_WATER, R_GAS = 1.0e-14, 55900.0, 8.314     # thermodynamics.py:54,103-104 (pinned by g1_constants.json through Kw(T))


def np_sum(x: np.ndarray) -> np.ndarray:
    """numpy's pairwise sum along the last axis (length <= 128), for every row."""
    n = x.shape[-1]
    if n < 8:
        res = np.zeros(x.shape[:-1])
        for i in range(n):
            res = res + x[..., i]
        return res
    assert n <= 128
    r = [x[..., j].copy() for j in range(8)]
    i = 8
    while i < n - (n % 8):
        for j in range(8):
            r[j] = r[j] + x[..., i + j]
        i += 8
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
    while i < n:
        res = res + x[..., i]
        i += 1
    return res


def np_mean(x):
    return np_sum(x) / x.shape[-1]


def np_std(x):
    d = x - np_mean(x)[..., None]
    return np.sqrt(np_sum(d * d) / x.shape[-1])


def mixing_quality(c):
    mean, std = np_mean(c), np_std(c)
    with np.errstate(divide="ignore", invalid="ignore"):
        cv = np.where(mean > 0, std / mean, 0.0)
        var, seg = std * std, mean * mean
        s = np.where(seg > 0.0, np.clip(var / seg, 0.0, 1.0), 0.0)
    return cv, s


def gradients(p, zone_height):
    g = np.diff(p, axis=-1) / zone_height[..., None]
    ag = np.abs(g)
    return [np_mean(p), np_std(p), p.max(-1), p.min(-1), p.max(-1) - p.min(-1), ag.max(-1), np_mean(ag), ag.argmax(-1).astype(np.float64)]


def diagnostics(pH, Cl, T, H, volume, height, strat):
    """(N, n) state arrays, (N,) volume [L], height [m], stratification flag -> (34, N) in FIELDS order."""
    pH, Cl, T, H = (np.asarray(a, dtype=np.float64) for a in (pH, Cl, T, H))
    N, n = pH.shape
    volume = np.broadcast_to(np.asarray(volume, dtype=np.float64), (N,)); height = np.broadcast_to(np.asarray(height, dtype=np.float64), (N,))
    zone_volume = volume / n
    out = np.empty((N_DIAG, N))
    out[0] = np_sum(Cl) * zone_volume
    out[1] = np_sum(H) * zone_volume / 1000
    exponent = (DELTA_H_WATER / R_GAS) * (1.0 / 298.15 - 1.0 / (T[:, 0] + 273.15))
    Kw = KW_25C * np.exp(exponent)
    out[2] = np_sum(Kw[:, None] / H) * zone_volume / 1000
    out[3] = out[1] - out[2]
    out[4] = 998.2 * 4184 * (volume / 1000) * np_mean(T - 20.0) / 1000
    out[5], out[6] = mixing_quality(pH)
    out[7], out[8] = mixing_quality(Cl)
    zh = height / n
    tg = np.abs(np.diff(T, axis=-1)) / zh[:, None]
    idx = tg.argmax(-1)
    depth = height - (idx + 0.5) * zh
    out[9] = np.where(np.asarray(strat, dtype=bool) & (tg.max(-1) > 0.5), depth, np.nan)
    for k, p in enumerate((pH, Cl, T)):
        out[10 + 8 * k: 18 + 8 * k] = gradients(p, zh)
    return out
