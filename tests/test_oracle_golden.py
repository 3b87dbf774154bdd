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


# ---------------------------------------------------------------- g11: branches no other family reaches
BRANCH_CASES = ("clamp_cl", "clamp_ph_hi", "clamp_ph_lo", "host_edit", "low_u_n4", "low_u_n8")


def branch_case(wt, name):
    """(n, par, bc, dt, pre-state (steps, 3n), pre-time, post-state, derived, time, flow, stats) of a g11 case:
    `pre` is self.state as the reference's step() found it (host edits included), `traj[k + 1]` what it left."""
    g = golden_npz("g11_branches.npz")
    meta = golden_json("g11_branches.json")[name]
    n = meta["n_zones"]
    cols = cfg_columns(g[f"{name}__cfg"], g["cfg_fields"])
    par = wt.params.derive_constants(cols, n)[:, 0]
    return dict(n=n, par=par, cols=cols, bc=g[f"{name}__bc"], dt=meta["dt"], pre=g[f"{name}__pre"], pre_time=g[f"{name}__pre_time"],
                traj=g[f"{name}__traj"], derived=g[f"{name}__derived"], time=g[f"{name}__time"], flow=g[f"{name}__flow"],
                stats=g[f"{name}__stats"])


@pytest.mark.parametrize("name", BRANCH_CASES)
def test_branch_fixtures_oracle_vs_reference(wt, oracle, name):
    """Clamps (reactor.py:526-541), host-edited state and clock between steps (:467-472), velocity scale
    <= 1e-6 -> Ri = inf (spatial.py:270-275): every step started from the reference's own pre-step state."""
    c = branch_case(wt, name)
    n = c["n"]
    expect_flags = {"clamp_cl": oracle.ST_CLAMP_CL, "clamp_ph_hi": oracle.ST_CLAMP_PH, "clamp_ph_lo": oracle.ST_CLAMP_PH}
    for k in range(c["pre"].shape[0]):
        y, t, der, status, st = oracle.step(n, c["par"], c["bc"], c["dt"], c["pre"][k].reshape(-1), float(c["pre_time"][k]),
                                            want_stats=True)
        post = c["traj"][k + 1].reshape(-1)
        tol = 1e-7 if name == "clamp_ph_lo" and k == 0 else 1e-9     # 80 internal steps through pH 0..2
        assert np.all(np.abs(y - post) <= tol * np.abs(post) + 1e-300), f"{name} step {k}"
        assert t == c["time"][k]
        assert (st.nfev, st.njev, st.nlu, st.nsteps) == tuple(c["stats"][k][:4]), f"{name} step {k}"
        assert relerr(der.reshape(3, n), c["derived"][k]) < 10 * tol
        if k == 0 and name in expect_flags:
            assert status == expect_flags[name]
            # the clip really happened in the reference: the post-state sits on the bound
            assert (post[n:2 * n] == 0).all() if name == "clamp_cl" else ((post[:n] == 14.0).any() or (post[:n] == 0.0).any())
        elif name not in expect_flags:
            assert status == 0


def test_low_velocity_rhs_matches_reference(wt, oracle):
    """u <= 1e-6 m/s: calculate_richardson_number returns +inf, every interface is suppressed."""
    g = golden_npz("g11_branches.npz")
    n = 8
    cols = cfg_columns(g["rhs_low_u__cfg"], g["cfg_fields"])
    par = wt.params.derive_constants(cols, n)
    assert np.all(par[wt.params.P_USUP] <= 1e-6)
    for c in range(g["rhs_low_u__y"].shape[0]):
        f, st = oracle.rhs(n, par[:, c], g["rhs_low_u__bc"][c], g["rhs_low_u__y"][c])
        ref = g["rhs_low_u__f"][c]
        assert st == 0
        assert np.array_equal(f[2 * n:], ref[2 * n:]), f"case {c}"      # temperature rows: pure arithmetic
        assert np.allclose(f[:2 * n], ref[:2 * n], rtol=1e-9, atol=1e-18), f"case {c}"


def test_nonfinite_state_raises_like_reference(wt, oracle):
    """scipy refuses a non-finite y0: ValueError out of step(), state and clock untouched."""
    nf = golden_json("g11_branches.json")["nonfinite"]
    assert nf["nan"]["raised"] == nf["inf"]["raised"] == "ValueError"
    assert nf["nan"]["message"] == "All components of the initial state `y0` must be finite."
    cols = {k: np.array([getattr(wt.ReactorConfiguration(), k)]) for k in
            ("volume", "height", "diameter", "flow_rate", "impeller_speed", "impeller_diameter", "total_carbonate",
             "temperature", "enable_thermal_stratification", "alkalinity")}
    par = wt.params.derive_constants(cols, 4)[:, 0]
    bc = np.array([getattr(wt.BoundaryConditions(), k) for k in wt.params.BOUNDARY_FIELDS])
    for bad in (np.nan, np.inf, -np.inf):
        y0 = np.concatenate([[7.0, bad, 7.0, 7.0], np.full(4, 2.0), np.full(4, 20.0)])
        y, t, der, status = oracle.step(4, par, bc, 1.0, y0, 1.0)
        assert status == oracle.ST_NONFINITE and t == 1.0 == nf["nan"]["time_after"]
        assert np.array_equal(y, y0, equal_nan=True)


def outlier_errors(wt, oracle, n, linsolve=0):
    """Worst relative error per g10 reactor of the CPU oracle against 100 steps of the reference."""
    g = golden_npz(f"g10_outliers_n{n}.npz")
    R, every, steps = g["reactors"], int(g["every"]), int(g["steps"])
    cols, bc = wt.make_ensemble(int(R.max()) + 1)
    d = wt.ReactorConfiguration()
    full = {k: np.broadcast_to(np.asarray(cols.get(k, getattr(d, k))), (int(R.max()) + 1,)).copy()
            for k in ("volume", "height", "diameter", "flow_rate", "impeller_speed", "impeller_diameter",
                      "total_carbonate", "temperature", "enable_thermal_stratification", "alkalinity")}
    par = np.ascontiguousarray(wt.params.derive_constants(full, n)[:, R]); b = np.ascontiguousarray(bc[:, R])
    S = len(R)
    A = [np.broadcast_to(cols[k][R][:, None], (S, n)).copy() for k in ("initial_pH", "initial_chlorine", "temperature")] + [np.zeros(S)]
    worst = np.zeros(S)
    oracle.set_linsolve(linsolve)
    try:
        for k in range(steps // every):
            A = list(oracle.ensemble_step(n, par, b, 1.0, every, *A, nthreads=8))
            assert not A[4].any()
            A = A[:4]
            snap = g["snaps"][:, k]
            worst = np.maximum(worst, np.stack([np.abs(A[i] - snap[:, i]) / np.abs(snap[:, i]) for i in range(3)]).max(axis=(0, 2)))
    finally:
        oracle.set_linsolve(0)
    return worst


@pytest.mark.parametrize("n", [4, 8, 20])
def test_outlier_reactors_oracle_vs_reference(wt, oracle, n):
    """g10: the reactors of the bench ensembles on which GPU and oracle differ by more than 1e-7, 100 steps of the
    reference itself.  Even the bit-faithful oracle (same algorithm, dense LU as scipy; it differs from the
    reference by libm-vs-numpy-SIMD last bits of pow/exp only) leaves the 1e-6 band on some of them, and so does
    its tridiagonal variant: on these reactors the solve crosses stratification switches of the RHS with repeated
    rejections, and two correct executions agree to the solver's own tolerance (rtol 1e-6 per internal step) only.
    Observed here: n = 4 max 7.7e-9; n = 8 max 6.0e-7 / 1.4e-6 (dense / tridiagonal); n = 20 max 4.0e-6 / 1.7e-6."""
    e0, e1 = outlier_errors(wt, oracle, n, 0), outlier_errors(wt, oracle, n, 1)
    bound = {4: 1e-7, 8: 5e-6, 20: 2e-5}[n]
    assert e0.max() < bound and e1.max() < bound
    if n == 20:      # the phenomenon is a property of the algorithm on these inputs, not of one implementation
        assert (e0 > 1e-6).sum() >= 3 and (e1 > 1e-6).sum() >= 3
