"""NEXT-1: the sensor-suite oracle against vectors produced by the
reference's own sensor classes run with the same injected random stream
(oracle/gen_golden_sensors.py)."""
import numpy as np
import pytest

from conftest import golden_npz

CASES = ("main5", "dose8", "quiet4")


def _run_oracle(g):
    import sensor_oracle as SO
    n = int(g["n_zones"])
    suite = SO.SensorSuite(float(g["cfg_flow_rate"]), float(g["cfg_initial_chlorine"]), float(g["cfg_temperature"]),
                           float(g["t0"]), int(g["seed"]), int(g["reactor"]))
    taps = g["taps"]
    steps = taps.shape[0]
    vals = np.empty((steps, 7)); stat = np.empty((steps, 7), dtype=np.int8); flt = np.empty((steps, 7), dtype=np.int8)
    for k in range(steps):
        pH0, pHN, Cl0, ClN, T0, TN, flow = taps[k]
        # zone 0 and zone -1 are all the suite looks at (sensors/__init__.py:79,92,113)
        v, s, f = suite.read_all({0: pH0, -1: pHN}, {0: Cl0, -1: ClN}, {0: T0, -1: TN}, flow,
                                 float(g["t0"]) + (k + 1) * float(g["dt"]))
        vals[k], stat[k], flt[k] = v, s, f
    return vals, stat, flt


@pytest.mark.parametrize("case", CASES)
def test_sensor_oracle_reproduces_reference(case):
    g = golden_npz(f"g7_sensors_{case}.npz")
    vals, stat, flt = _run_oracle(g)
    ref = g["values"]
    assert np.array_equal(np.isnan(vals), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.max(np.abs(vals[ok] - ref[ok])) < 1e-12
    assert np.array_equal(stat, g["status"])
    assert np.array_equal(flt, g["fault"])


def test_golden_covers_the_interesting_paths():
    """The vectors exercise warm-up gating, the shared sample line quirk, random faults and dead sensors."""
    import sensor_oracle as SO
    g = golden_npz("g7_sensors_main5.npz")
    st, fl, v, taps = g["status"], g["fault"], g["values"], g["taps"]
    assert (st[:1799, SO.S_PH_IN] == SO.ST_WARMING_UP).all() and st[1800, SO.S_PH_IN] != SO.ST_WARMING_UP
    assert (st[:9, SO.S_FLOW] == SO.ST_WARMING_UP).all() and np.isfinite(v[20, SO.S_FLOW])
    # before the pH sensor warms up the inlet RTD reads T + 5 degC (2.6 degC lead-resistance error fed back
    # through the 0.5 lag, temperature_sensor.py:149-171 / base_sensor.py:626-630) ...
    before = slice(1700, 1790)
    assert abs(np.nanmean(v[before, SO.S_T_IN] - taps[before, 4]) - 5.05) < 0.3
    # ... afterwards it starts pulling pH samples out of the delay line it shares with pH_inlet
    # (sensors/__init__.py:62-64,74,108) and settles near 12 degC while the water is at 19.4 degC;
    # the pH sensor in turn first reads temperature samples (saturating at 14)
    late = slice(1850, 1890)
    assert np.nanmean(taps[late, 4] - v[late, SO.S_T_IN]) > 6.0
    assert np.nanmean(v[1795:1810, SO.S_PH_IN]) > 13.0 and abs(np.nanmean(v[late, SO.S_PH_IN]) - 7.2) < 0.1
    allf = np.concatenate([golden_npz(f"g7_sensors_{c}.npz")["fault"].ravel() for c in CASES])
    assert (allf != 0).any()


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors of the Random123 distribution."""
    import sensor_oracle as SO
    assert SO.philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert SO.philox4x32_10((0xffffffff,) * 4, (0xffffffff, 0xffffffff)) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert SO.philox4x32_10((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)
