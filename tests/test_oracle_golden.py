"""The CPU oracle (oracle/wt_oracle.c) against vectors produced by the Python
reference (oracle/gen_golden.py).  This is what pins the oracle; the GPU tests
then compare the HIP path with the oracle and with the same vectors."""
import numpy as np
import pytest

from conftest import SCENARIOS, cfg_columns, golden_json, golden_npz, relerr


def test_constants_bit_identical(wt):
    g = golden_json("g1_constants.json")
    P = wt.params
    for entry in g["configs"]:
        cfg = entry["config"]
        cols = {k: np.array([v]) for k, v in cfg.items() if k != "n_zones"}
        par = P.derive_constants(cols, cfg["n_zones"])[:, 0]
        assert par[P.P_KW] == entry["Kw"]
        assert par[P.P_KA1] == entry["Ka1"]
        assert par[P.P_KA2] == entry["Ka2"]
        assert par[P.P_KA_HOCL] == entry["Ka_HOCl"]
        assert par[P.P_KEX] == entry["K_exchange_per_s"]
        assert par[P.P_USUP] == entry["superficial_velocity"]


def test_known_answers_of_reference_validators(wt, oracle):
    """validate_thermodynamics / validate_transport known answers (thermodynamics.py:403-417,
    transport.py:536-554), re-expressed on the build's own code."""
    P = wt.params
    assert abs(P.water_ionization_constant(np.array([25.0]))[0] - 1e-14) < 1e-20
    assert abs(P.carbonate_pKa(np.array([25.0]), 1)[0] - 6.35) < 1e-12
    # k(20 degC) = 1e-4 via the RHS: uniform state, no inlet -> dCl = -k*phi*Cl in interior zones
    g = golden_json("g1_constants.json")
    cols = {k: np.array([v]) for k, v in g["configs"][5]["config"].items() if k != "n_zones"}
    n = g["configs"][5]["config"]["n_zones"]
    par = P.derive_constants(cols, n)[:, 0]
    # K-matrix invariants: interior row sums vanish, outlet row sum = -Q/V: probe with x = 1
    bc = np.array([6.0, 7.0, 0.0, 20.0, 0, 0.1, 0, 50.0, 20.0, 0.0])
    y = np.concatenate([np.full(n, 7.0), np.full(n, 0.0), np.full(n, 1.0)])
    f, st = oracle.rhs(n, par, bc, y)
    dT = f[2 * n:]
    Qv = (6.0 / 60) / cols["volume"][0]
    assert np.all(np.abs(dT[1:-1]) < 1e-12)
    assert abs(dT[-1] - (-Qv)) < 1e-12
    assert abs(dT[0] - Qv * (20.0 - 1.0)) < 1e-12


@pytest.mark.parametrize("n", [4, 8, 20])
def test_rhs_matches_reference(wt, oracle, n):
    g = golden_npz(f"g2_rhs_n{n}.npz")
    cols = cfg_columns(g["cfg"], g["cfg_fields"])
    par = wt.params.derive_constants(cols, n)
    nbit = 0
    for c in range(g["y"].shape[0]):
        f, st = oracle.rhs(n, par[:, c], g["bc"][c], g["y"][c])
        ref = g["f"][c]
        assert st == 0
        # temperature rows are pure arithmetic: bit-identical (pins the K@x order)
        assert np.array_equal(f[2 * n:], ref[2 * n:]), f"case {c}"
        # chlorine rows add exp()/pow(): numpy's SIMD routines differ from libm by 1 ulp
        # on ~5 % of inputs; compare against the magnitude of the row's terms
        Cl = g["y"][c][n:2 * n]
        magc = np.abs(ref[n:2 * n]) + 4 * par[wt.params.P_KEX, c] * np.abs(Cl).max() + 1e-300
        assert np.all(np.abs(f[n:2 * n] - ref[n:2 * n]) <= 4e-16 * magc), f"case {c}"
        # pH rows go through numpy's vectorised pow in the reference (1 ulp vs libm);
        # the mixing term cancels, so compare against the magnitude of its terms
        H = 10.0 ** (-g["y"][c][:n])
        mag = np.abs(ref[:n]) + 4 * par[wt.params.P_KEX, c] * H.max() / 1e-4 * 1e-0
        assert np.all(np.abs(f[:n] - ref[:n]) <= 1e-12 * np.maximum(mag, 1e-30)), f"case {c}"
        nbit += int(np.count_nonzero(f[:n] != ref[:n]))
    assert nbit < 0.2 * g["y"].shape[0] * n


@pytest.mark.parametrize("n", [4, 8, 20])
@pytest.mark.parametrize("scen", SCENARIOS)
@pytest.mark.parametrize("linsolve", [0, 1])
def test_trajectory_and_decision_sequence(wt, oracle, n, scen, linsolve):
    """Per-step state within 1e-9 of the reference (observed <= 3e-11) and the scipy counters
    (nfev, njev, nlu, accepted steps) identical on every step.  linsolve=1 swaps
    the dense LU for the block-triangular tridiagonal solve the HIP kernel uses."""
    g = golden_npz(f"g3_traj_{scen}_n{n}.npz")
    cols = cfg_columns(g["cfg"], g["cfg_fields"])
    par = wt.params.derive_constants(cols, n)[:, 0]
    bc, dt = g["bc"], float(g["dt"])
    traj, stats = g["traj"], g["stats"]
    nst = min(traj.shape[0] - 1, 200 if linsolve else 10_000)
    y = traj[0].reshape(-1).copy()
    t = 0.0
    oracle.set_linsolve(linsolve)
    try:
        worst = 0.0
        for k in range(nst):
            y, t, der, status, st = oracle.step(n, par, bc, dt, y, t, want_stats=True)
            assert status == 0
            worst = max(worst, relerr(y, traj[k + 1].reshape(-1)))
            assert (st.nfev, st.njev, st.nlu, st.nsteps) == tuple(stats[k][:4]), f"step {k}"
            assert relerr(der.reshape(3, n), g["derived"][k]) < 1e-9
        assert worst < 1e-9
        assert abs(t - g["time"][nst - 1]) < 1e-9
    finally:
        oracle.set_linsolve(0)


def test_cold_run_raises_at_reference_step(wt, oracle):
    g = golden_json("g4_faults.json")["cold_run"]
    cfg = g["config"]
    n = cfg["n_zones"]
    cols = {k: np.array([v]) for k, v in cfg.items() if k != "n_zones"}
    par = wt.params.derive_constants(cols, n)[:, 0]
    bc = np.array(g["bc"])
    y = np.concatenate([np.full(n, cfg["initial_pH"]), np.full(n, cfg["initial_chlorine"]), np.full(n, cfg["temperature"])])
    t = 0.0
    for k in range(200):
        y2, t2, der, status = oracle.step(n, par, bc, 1.0, y, t)
        if status & oracle.ST_T_RANGE:
            assert k == g["raise_step_index"]
            assert np.array_equal(y2, y) and t2 == t          # state not advanced
            assert relerr(y, np.concatenate(g["state_before_raise"])) < 1e-11
            return
        y, t = y2, t2
    pytest.fail("oracle never flagged the temperature range")


def test_ph_solver_known_answers(wt, oracle):
    g = golden_json("g5_ph_solver.json")
    P = wt.params
    for c in g["cases"]:
        T = np.array([c["temperature"]])
        Kw = P.water_ionization_constant(T)[0]
        Ka1 = P._pow10_neg(P.carbonate_pKa(T, 1))[0]
        Ka2 = P._pow10_neg(P.carbonate_pKa(T, 2))[0]
        pH, it, rc = oracle.calculate_pH(Kw, Ka1, Ka2, c["total_carbonate"] / 1000.0, c["alkalinity"], c["guess"])
        if c["rc"] == 0:
            assert rc == 0 and abs(pH - c["pH"]) < 1e-12
        else:
            assert rc != 0
    # SURVEY.md section 8 A9 spot values
    assert abs(g["pH_eq_default"] - 8.398396410366111) < 1e-12


@pytest.mark.parametrize("n", [4, 8, 20])
def test_synthetic_ensemble_sample(wt, oracle, n):
    """First 64 reactors of the bench ensemble, 50 steps: oracle == reference."""
    g = golden_npz(f"g6_ensemble_n{n}.npz")
    S, steps, every = int(g["n_reactors"]), int(g["steps"]), int(g["every"])
    cols, bc = wt.make_ensemble(S)
    d = wt.ReactorConfiguration()
    full = {k: np.broadcast_to(np.asarray(cols.get(k, getattr(d, k))), (S,)).copy()
            for k in ("volume", "height", "diameter", "flow_rate", "impeller_speed", "impeller_diameter",
                      "total_carbonate", "temperature", "enable_thermal_stratification")}
    par = wt.params.derive_constants(full, n)
    shape = (S, n)
    pH = np.broadcast_to(cols["initial_pH"][:, None], shape).copy()
    Cl = np.broadcast_to(cols["initial_chlorine"][:, None], shape).copy()
    T = np.broadcast_to(cols["temperature"][:, None], shape).copy()
    t = np.zeros(S)
    for k in range(steps // every):
        pH, Cl, T, t, st = oracle.ensemble_step(n, par, bc, 1.0, every, pH, Cl, T, t, nthreads=4)
        assert not st.any()
        snap = g["snaps"][k]
        # observed: <= 6e-9 (n=20 chlorine; 1-ulp pow/exp differences between numpy's SIMD
        # routines and libm, amplified through the finite-difference Jacobian), <= 1e-10 otherwise
        assert relerr(pH, snap[:, 0]) < 1e-7
        assert relerr(Cl, snap[:, 1]) < 1e-7
        assert relerr(T, snap[:, 2]) < 1e-7
