"""NEXT-4 on the GPU: per-reactor diagnostics (conservation, mixing quality, thermocline, spatial gradients)
against the reference's own outputs (tests/golden/g9_diag_n*.npz) and the numpy-order oracle."""
import numpy as np
import pytest

from conftest import golden_npz

pytestmark = pytest.mark.gpu
H_COLS = (1, 2, 3)       # built on 10**-pH / exp(): the last bit depends on the libm


def _check(out, ref, n_exact_min=0.999):
    assert out.shape == ref.shape and np.array_equal(np.isnan(out), np.isnan(ref))
    same = (out == ref) | (np.isnan(out) & np.isnan(ref))
    exact_cols = [c for c in range(ref.shape[1]) if c not in H_COLS]
    assert same[:, exact_cols].all(), [(c, int((~same[:, c]).sum())) for c in exact_cols if not same[:, c].all()]
    scale = np.maximum(np.abs(ref[:, 1]), np.abs(ref[:, 2]))
    for c in H_COLS:
        assert np.all(np.abs(out[:, c] - ref[:, c]) <= 1e-15 * scale)


@pytest.mark.parametrize("n", (4, 8, 20))
def test_diagnostics_vs_reference_golden(gpu, wt, n):
    g = golden_npz(f"g9_diag_n{n}.npz")
    cols = {k[4:]: g[k] for k in g.files if k.startswith("cfg_")}
    cols["enable_thermal_stratification"] = g["strat"]
    ens = wt.ReactorEnsemble(cols, n_zones=n)
    st = g["state"]
    ens.set_state(st[:, 0], st[:, 1], st[:, 2])
    out = ens.diagnostics(as_dict=False).T
    _check(out, g["diag"])
    d = ens.diagnostics()
    assert list(d) == list(wt.ReactorEnsemble.DIAGNOSTIC_FIELDS) and np.array_equal(d["chlorine_range"], out[:, 22])
    ens.close()


@pytest.mark.parametrize("n", (2, 5, 8, 9, 16, 20))
def test_diagnostics_after_steps_vs_oracle(gpu, wt, n):
    """state after 30 steps of the synthetic ensemble (zone counts on both sides of numpy's 8-element switch)"""
    import diag_oracle as DO
    N = 3000
    cols, bc = wt.make_ensemble(N, seed=31 + n)
    ens = wt.ReactorEnsemble(cols, n_zones=n); ens.set_boundary(bc)
    es = ens.step(1.0, n_steps=30)
    out = ens.diagnostics(as_dict=False).T
    ref = DO.diagnostics(es.pH, es.chlorine, es.temperature, es.H_concentration, 1000.0, 2.0, np.ones(N, bool)).T
    _check(out, ref)
    # conservation sanity: chlorine mass = mean concentration x volume
    assert np.allclose(out[:, 0], es.chlorine.mean(1) * 1000.0, rtol=1e-13)
    ens.close()


def test_dropin_validate_conservation_vs_reference(gpu, wt):
    """IntegratedCSTR.validate_conservation / mixing_quality of the single-reactor drop-in (same keys as the
    reference's dict) on golden states"""
    g = golden_npz("g9_diag_n8.npz")
    for i in (0, 3, 6, 11):
        kw = {k[4:]: float(g[k][i]) for k in g.files if k.startswith("cfg_")}
        cfg = wt.ReactorConfiguration(n_zones=8, enable_thermal_stratification=bool(g["strat"][i]), **kw)
        r = wt.IntegratedCSTR(cfg)
        st = g["state"][i]
        r.state.pH, r.state.chlorine, r.state.temperature = st[0].copy(), st[1].copy(), st[2].copy()
        c = r.validate_conservation()
        ref = g["diag"][i]
        assert list(c) == ["total_chlorine_mg", "total_H_mol", "total_OH_mol", "charge_balance_mol", "thermal_energy_kJ", "zones", "timestamp"]
        assert c["total_chlorine_mg"] == ref[0] and c["thermal_energy_kJ"] == ref[4] and c["zones"] == 8
        assert abs(c["total_H_mol"] - ref[1]) <= 1e-15 * abs(ref[1]) and abs(c["total_OH_mol"] - ref[2]) <= 1e-15 * abs(ref[2])
        m = r.mixing_quality()
        assert (m["pH_CV"], m["pH_segregation"], m["chlorine_CV"], m["chlorine_segregation"]) == tuple(ref[5:9])
