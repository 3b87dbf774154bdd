"""NEXT-1: the fused sensor suite on the GPU against the sensor oracle / the reference's golden
vectors (same injected Philox stream).  The signal path is fp32 (BASELINE config 5)."""
import numpy as np
import pytest

from conftest import golden_npz

pytestmark = pytest.mark.gpu
CASES = ("main5", "dose8", "quiet4")


@pytest.mark.parametrize("case", CASES)
def test_suite_with_physics_vs_oracle(gpu, wt, case):
    """Physics + sensors on the device, 2 200 steps of 1 s; the oracle suite is fed the device's own
    per-step state (downloaded every step), so this isolates the sensor pipeline."""
    import sensor_oracle as SO
    g = golden_npz(f"g7_sensors_{case}.npz")
    n = int(g["n_zones"])
    seed, rid = int(g["seed"]), int(g["reactor"])
    cfg = wt.ReactorConfiguration(n_zones=n, flow_rate=float(g["cfg_flow_rate"]), initial_chlorine=float(g["cfg_initial_chlorine"]),
                                  temperature=float(g["cfg_temperature"]))
    steps = 2200
    ens = wt.ReactorEnsemble([cfg])
    ens.set_boundary(wt.BoundaryConditions())
    ens.set_schedule(1, 1)
    ens.enable_sensors(seed=seed, reactor_base=rid, history=steps)
    suite = SO.SensorSuite(cfg.flow_rate, cfg.initial_chlorine, cfg.temperature, 0.0, seed, rid)
    ov = np.empty((steps, 7)); os_ = np.empty((steps, 7), dtype=np.uint8); of = np.empty((steps, 7), dtype=np.uint8)
    for k in range(steps):
        es = ens.step(1.0, n_steps=1)
        f32 = lambda x: float(np.float32(x))                      # the device taps are fp32
        v, s, f = suite.read_all({0: f32(es.pH[0, 0]), -1: f32(es.pH[0, -1])}, {0: f32(es.chlorine[0, 0]), -1: f32(es.chlorine[0, -1])},
                                 {0: f32(es.temperature[0, 0]), -1: f32(es.temperature[0, -1])}, f32(es.flow_rate[0]), float(es.time[0]))
        ov[k], os_[k], of[k] = v, s, f
    hv, hs, hf, nf = ens.sensor_history()
    assert nf[0] == steps
    hv, hs, hf = hv[:, :, 0], hs[:, :, 0], hf[:, :, 0]
    assert np.array_equal(np.isnan(hv), np.isnan(ov))
    ok = ~np.isnan(ov)
    assert np.max(np.abs(hv[ok] - ov[ok]) / (1.0 + np.abs(ov[ok]))) < 2e-5      # fp32 signal path
    assert np.array_equal(hs, os_) and np.array_equal(hf, of)
    ens.close()


def test_suite_statistics_and_independence(gpu, wt):
    """4 096 reactors: noise has the configured sigma, reactors have independent streams, the result
    does not depend on the launch schedule, and the sharded reactor_base reproduces a slice."""
    N, n, steps = 4096, 8, 400
    cols, bc = wt.make_ensemble(N)
    def run(streams, chunk, base=0, sl=slice(None)):
        c = {k: v[sl] for k, v in cols.items()}
        ens = wt.ReactorEnsemble(c, n_zones=n); ens.set_boundary(np.ascontiguousarray(bc[:, sl]))
        ens.set_schedule(streams, chunk)
        ens.enable_sensors(seed=77, reactor_base=base)
        ens.step(1.0, n_steps=steps)
        out = ens.sensor_readings(); st = ens.state
        ens.close()
        return out, st
    (v, s, f), st = run(0, 50)            # work-queue schedule (default)
    for sched in ((1, 0), (4, 7), (0, 1)):
        (v2, s2, f2), _ = run(*sched)
        assert np.array_equal(v, v2, equal_nan=True) and np.array_equal(s, s2) and np.array_equal(f, f2), sched
    (v3, s3, f3), _ = run(2, 25, base=1000, sl=slice(1000, 1500))
    assert np.array_equal(v[:, 1000:1500], v3, equal_nan=True) and np.array_equal(s[:, 1000:1500], s3)
    # at t = 400 s: flow (10 s), RTD (30 s), DPD (60 s), amperometric (300 s) are warm, pH (1800 s) is not
    # (a few sensors sit in the reference's sticky POWER_FAULT: a 4-sigma supply-voltage draw, base_sensor.py:549-572)
    assert np.isnan(v[0]).all() and np.isnan(v[1]).all() and np.isin(s[0], (2, 10)).all() and (s[0] == 2).mean() > 0.95
    flow_true = st.flow_rate
    ok = np.isfinite(v[4])
    # the reference never revives a sensor after a random open/short-circuit fault (1e-4 per read, NaN feeds
    # back through the lag) or a power fault, so ~6 % of the flowmeters are dead after 400 reads
    assert 0.9 < ok.mean() < 0.98
    # magnetic flowmeter: reading = lagged(true + calibration offset) + electrical noise; the start-up
    # calibration bakes in +flow_rate (reference quirk, __main__.py:96-105), so the reading sits near
    # true + cfg flow and saturates at the full scale 2 * cfg flow above it; the status is DRIFT_WARNING (5: the baked-in offset
    # is half the span, base_sensor.py:677-682) or OUT_OF_RANGE (9: beyond 110 % of the span)
    fs_all = 2 * cols["flow_rate"]
    expect = flow_true + cols["flow_rate"]
    sat = ok & (expect > 1.03 * fs_all)
    # (the 0.1 % electrode noise is added after the clamp and clamped again: reading in (fs - 0.6 %, fs])
    rs = v[4][sat] / fs_all[sat]
    assert sat.sum() > 50 and np.all(rs <= 1 + 1e-6) and np.all(rs > 0.994) and np.isin(s[4][sat], (5, 9)).all()
    lin = ok & (expect < 0.97 * fs_all)
    assert lin.sum() > 200
    resid = v[4][lin] - expect[lin]
    z = resid / fs_all[lin]
    assert abs(np.mean(z)) < 5e-4 and 0.002 < np.std(z) < 0.006
    # distinct reactors draw distinct noise
    assert np.unique(np.round(resid, 6)).size > 0.95 * resid.size


@pytest.mark.parametrize("N,n", [(12500, 8), (4000, 4), (4100, 2), (3000, 5)])
def test_suite_at_config5_share_vs_oracle(gpu, wt, N, n):
    """BASELINE config 5's per-GPU share (12 500 reactors x 8 zones, sensors on): the readings of a sample of
    reactors over 45 steps against the sensor oracle fed the device's own per-step state; every wavefront position
    (first / last reactor of a wavefront, last wavefront of the ensemble) is in the sample.  Likewise with 16, 32 and
    12 reactors per wavefront, where the suite takes two / four passes of nine reactors (seven lanes each)."""
    import sensor_oracle as SO
    steps, seed = 45, 0xC0FFEE
    cols, bc = wt.make_ensemble(N)
    ens = wt.ReactorEnsemble(cols, n_zones=n); ens.set_boundary(bc)
    ens.enable_sensors(seed=seed, reactor_base=0, history=steps)
    R = 64 // n
    sample = sorted({0, R - 1, R, 8, 9, 10, 17, 18, 26, 27, 63, 64, N // 3, N // 3 + 9, N - R - 1, N - 5, N - 4, N - 1} & set(range(N)))
    suites = {r: SO.SensorSuite(float(cols["flow_rate"][r]), float(cols["initial_chlorine"][r]), float(cols["temperature"][r]),
                                0.0, seed, r) for r in sample}
    ov = np.empty((steps, 7, len(sample))); os_ = np.empty((steps, 7, len(sample)), dtype=np.uint8); of = np.empty_like(os_)
    f32 = lambda x: float(np.float32(x))
    for k in range(steps):
        es = ens.step(1.0, n_steps=1)
        for j, r in enumerate(sample):
            v, s, f = suites[r].read_all({0: f32(es.pH[r, 0]), -1: f32(es.pH[r, -1])}, {0: f32(es.chlorine[r, 0]), -1: f32(es.chlorine[r, -1])},
                                         {0: f32(es.temperature[r, 0]), -1: f32(es.temperature[r, -1])}, f32(es.flow_rate[r]), float(es.time[r]))
            ov[k, :, j], os_[k, :, j], of[k, :, j] = v, s, f
    hv, hs, hf, nf = ens.sensor_history()
    assert np.all(nf == steps)
    hv, hs, hf = hv[:, :, sample], hs[:, :, sample], hf[:, :, sample]
    assert np.array_equal(np.isnan(hv), np.isnan(ov))
    ok = ~np.isnan(ov)
    assert ok.sum() > 500 and np.max(np.abs(hv[ok] - ov[ok]) / (1.0 + np.abs(ov[ok]))) < 2e-5      # fp32 signal path
    assert np.array_equal(hs, os_) and np.array_equal(hf, of)
    v, s, f = ens.sensor_readings()
    assert np.array_equal(v[:, sample], hv[-1], equal_nan=True)
    ens.close()


def test_history_counts_only_the_reads_taken(gpu, wt):
    """A reactor whose step raises takes no reading from then on (the reference's loop stops at the raise): its history
    length is the number of completed steps, what lies beyond is zero, not uninitialised memory; its neighbour in the
    same ensemble keeps reading.  (Cold-run fixture g4: the reference raises at step index 34.)"""
    from conftest import golden_json
    g = golden_json("g4_faults.json")["cold_run"]
    cfg = wt.ReactorConfiguration(**g["config"])
    b = wt.BoundaryConditions(**dict(zip(wt.params.BOUNDARY_FIELDS, g["bc"])))
    ens = wt.ReactorEnsemble([cfg, wt.ReactorConfiguration(n_zones=cfg.n_zones)])
    ens.set_boundary([b, wt.BoundaryConditions()])
    ens.enable_sensors(seed=3, history=64)
    for k in (10, 30, 20):
        ens.step(1.0, n_steps=k)
    v, st, fl, n = ens.sensor_history()
    assert n[0] == g["raise_step_index"] and n[1] == 60
    assert not v[n[0]:, :, 0].any() and not st[n[0]:, :, 0].any() and not fl[n[0]:, :, 0].any()
    assert st[:n[1], :, 1].any()                      # the neighbour's sixty reads are there (status codes of warming-up sensors)
    ens.close()
