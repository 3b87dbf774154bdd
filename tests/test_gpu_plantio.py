"""NEXT-2 / NEXT-3 on the GPU: the per-reactor Modbus register image and the command path against
oracle/plc_oracle.py (pinned by the reference's encoder / register map, tests/golden/g8_modbus.json)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ensemble(wt, N, n=8, seed=99, base=0):
    cols, bc = wt.make_ensemble(N)
    ens = wt.ReactorEnsemble(cols, n_zones=n); ens.set_boundary(bc)
    ens.enable_sensors(seed=seed, reactor_base=base)
    ens.enable_plant_io()
    return ens, cols, bc


def test_input_image_vs_oracle(gpu, wt):
    """Image after 400 scans of one step (pH still warming up -> 0.0, ~6 % of sensors faulted) equals the
    oracle's update_modbus_inputs applied to the device's own readings, word for word."""
    import plc_oracle as PO
    N = 2048
    ens, cols, bc = _ensemble(wt, N)
    ens.set_schedule(2, 1)
    ens.step(1.0, n_steps=400)
    v, s, f = ens.sensor_readings()
    img, ok = ens.input_image()
    assert ok.all() and f.any() and np.isnan(v).any()
    for r in list(range(0, N, 37)) + list(np.nonzero(f.any(axis=0))[0][:40]):
        io = PO.PlantIO()
        assert io.update_inputs([float(x) for x in v[:, r]], [int(x) for x in f[:, r]], 399.0)   # sim_time lags one dt
        ir, di = ens.input_blocks(r, img)
        assert ir == io.ir[:200] and di == io.di[:100], r
    ens.close()


def test_scan_interval_and_time_register(gpu, wt):
    """chunked scans publish the last step of the launch; simulation_time = (steps - 1) * dt accumulated."""
    import plc_oracle as PO
    N = 512
    ens, cols, bc = _ensemble(wt, N)
    ens.set_schedule(1, 25)
    ens.step(0.5, n_steps=60)                     # launches of 25, 25, 10 steps
    img, ok = ens.input_image()
    t = 0.0
    for _ in range(59):
        t += 0.5
    assert np.all(wt.ReactorEnsemble.decode_float32(img[:, 16:18]) == np.float32(t))
    v, s, f = ens.sensor_readings()
    io = PO.PlantIO(); io.update_inputs([float(x) for x in v[:, 5]], [int(x) for x in f[:, 5]], t)
    assert ens.input_blocks(5, img)[0] == io.ir[:200]
    ens.close()


def test_command_path_vs_oracle(gpu, wt):
    """Holding image -> boundary conditions: validate_flow_rate clamps, NaN -> 0, inlet only if > 0.1;
    acts from the scan after the write; untouched reactors keep their boundary."""
    import plc_oracle as PO
    N = 1024
    ens, cols, bc = _ensemble(wt, N)
    ens.set_schedule(1, 1)
    ens.step(1.0, n_steps=2)
    assert np.array_equal(ens.boundary()[[0, 4, 6]], np.stack([bc[0], np.zeros(N), np.zeros(N)]))   # zeros in the registers: dosing off
    rng = np.random.default_rng(5)
    special = np.array([np.nan, np.inf, -np.inf, -1.0, 0.0, 0.1, 0.100001, 0.05, 2.0, 2.5, 1.0, 1.5, 20.0, 25.0, 1e30, -1e30, 1e-40])
    acid = np.concatenate([special, rng.uniform(-0.5, 3.0, N - special.size)])
    chl = np.concatenate([special[::-1], rng.uniform(-0.5, 1.5, N - special.size)])
    inlet = np.concatenate([np.roll(special, 5), rng.uniform(-1.0, 25.0, N - special.size)])
    half = N // 2
    ens.write_commands(acid[:half], chl[:half], inlet[:half], first_reactor=0)
    before = ens.boundary()
    assert np.array_equal(before[[0, 4, 6]], np.stack([bc[0], np.zeros(N), np.zeros(N)]))           # not yet: no scan since the write
    ens.step(1.0, n_steps=1)
    after = ens.boundary()
    exp = np.array(bc, copy=True); exp[4] = 0.0; exp[6] = 0.0
    for r in range(half):
        io = PO.PlantIO()
        io.write_holding("acid_flow_rate", float(acid[r])); io.write_holding("chlorine_flow_rate", float(chl[r])); io.write_holding("inlet_flow_rate", float(inlet[r]))
        col = list(exp[:, r]); PO.apply_boundary_conditions(col, io.read_commands()); exp[:, r] = col
    assert np.array_equal(after, exp)
    assert (after[0, :half] != bc[0, :half]).sum() > 300 and np.array_equal(after[:, half:], exp[:, half:])
    # and the physics sees it: dosing acid lowers the inlet-zone pH relative to an undosed twin
    ens.step(1.0, n_steps=120)
    twin, _, _ = _ensemble(wt, N)
    twin.set_schedule(1, 1); twin.step(1.0, n_steps=123)
    dosed = np.nonzero(after[4, :half] > 0.5)[0]
    assert dosed.size > 50 and np.all(ens.state.pH[dosed, 0] < twin.state.pH[dosed, 0] - 1e-3)
    ens.close(); twin.close()


def test_closed_loop_vs_reference_order(gpu, wt):
    """A host-side proportional controller closes the loop through the images every step; the same
    loop run with the oracle's PlantIO on the device's readings produces the same boundary history."""
    import plc_oracle as PO
    N = 64
    ens, cols, bc = _ensemble(wt, N, n=4)
    ens.set_schedule(1, 1)
    ios = [PO.PlantIO() for _ in range(N)]
    exp = np.array(bc, copy=True)
    for k in range(90):
        ens.step(1.0, n_steps=1, fused=False)
        v, s, f = ens.sensor_readings()
        img, _ = ens.input_image()
        for r in range(N):                                   # reference order: update inputs, then read commands
            ios[r].update_inputs([float(x) for x in v[:, r]], [int(x) for x in f[:, r]], float(k))
            col = list(exp[:, r]); PO.apply_boundary_conditions(col, ios[r].read_commands()); exp[:, r] = col
        assert np.array_equal(ens.boundary(), exp), k
        assert all(ens.input_blocks(r, img)[0] == ios[r].ir[:200] for r in (0, 17, 63))
        # the "PLC": chlorine dosing proportional to the shortfall of the outlet DPD reading (register 8-9) from 1.5 mg/L
        cl_out = wt.ReactorEnsemble.decode_float32(img[:, 8:10]).astype(np.float64)
        cmd_cl = np.clip(0.4 * (1.5 - cl_out), -0.2, 1.3)
        cmd_in = np.where(np.arange(N) % 3 == 0, 6.0 + 0.01 * k, 0.0)
        ens.write_commands(np.full(N, 0.05), cmd_cl, cmd_in)
        for r in range(N):
            ios[r].write_holding("acid_flow_rate", 0.05); ios[r].write_holding("chlorine_flow_rate", float(cmd_cl[r])); ios[r].write_holding("inlet_flow_rate", float(cmd_in[r]))
    assert ens.boundary()[6].max() > 0.1
    ens.close()


def test_set_boundary_wins_over_earlier_commands(gpu, wt):
    """With plant I/O on, every PLC scan rewrites boundary rows 0 / 4 / 6 on the device; a later ``set_boundary`` with
    the very block the host sent before must still be uploaded (the host's copy is no longer what the device holds)."""
    cols, bc = wt.make_ensemble(64)
    ens = wt.ReactorEnsemble(cols, n_zones=4)
    ens.set_boundary(bc)
    ens.enable_sensors(seed=5); ens.enable_plant_io()
    ens.write_commands(1.5, 0.75, 9.0)
    ens.step(1.0, n_steps=3)
    assert np.allclose(ens.boundary()[[4, 6, 0]], np.array([[1.5], [0.75], [9.0]]))
    for _ in range(2):                       # the first and every later identical call restore the host's rows
        ens.write_commands(0.0, 0.0, 0.0)    # (0 inlet command = "leave the inlet alone", so nothing overrides it again)
        ens.set_boundary(bc)
        assert np.array_equal(ens.boundary(), bc)
        ens.write_commands(1.5, 0.75, 9.0)
        ens.step(1.0, n_steps=2)
        assert np.allclose(ens.boundary()[[4, 6, 0]], np.array([[1.5], [0.75], [9.0]]))
    ens.close()
