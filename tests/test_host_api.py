"""Host-side logic that needs no GPU: dataclasses mirror the reference's API,
SoA packing, synthetic generator, the shared library exports its C ABI and
refuses to compute without a device."""
import ctypes as C
import dataclasses
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT, golden_json


def test_dataclass_defaults_match_reference(wt):
    # reactor.py:61-89, :169-186 defaults (values recorded in SURVEY.md Appendix A)
    cfg = wt.ReactorConfiguration()
    assert (cfg.volume, cfg.height, cfg.diameter, cfg.n_zones) == (1000.0, 2.0, 0.798, 5)
    assert (cfg.flow_rate, cfg.impeller_speed, cfg.impeller_diameter, cfg.power_number) == (5.0, 60.0, 0.3, 5.0)
    assert (cfg.initial_pH, cfg.alkalinity, cfg.total_carbonate, cfg.initial_chlorine, cfg.temperature) == (7.0, 100.0, 2.0, 2.0, 20.0)
    assert cfg.enable_thermal_stratification is True
    b = wt.BoundaryConditions()
    assert [getattr(b, k) for k in wt.params.BOUNDARY_FIELDS] == [5.0, 7.5, 0.0, 20.0, 0.0, 0.1, 0.0, 50.0, 20.0, 0.0]
    assert [f.name for f in dataclasses.fields(b)] == list(wt.params.BOUNDARY_FIELDS)
    s = wt.ReactorState()
    assert s.pH.shape == (5,) and np.all(s.H_concentration == 10 ** (-s.pH))
    assert np.all(s.density == 998.2) and np.all(s.chlorine_decay_rate == 0.0001)
    assert set(s.state_dict()) >= {"time", "pH", "chlorine", "temperature", "flow_rate", "H_concentration"}
    assert wt.PhysicsEngine is wt.IntegratedCSTR


def test_config_validate_errors(wt):
    g = golden_json("g4_faults.json")
    with pytest.raises(ValueError, match="Volume mismatch"):
        wt.ReactorConfiguration(volume=500.0).validate()
    assert g["volume_mismatch"] == "ValueError"
    with pytest.raises(AssertionError):
        wt.ReactorConfiguration(temperature=45.0).validate()
    with pytest.raises(AssertionError):
        wt.ReactorConfiguration(initial_chlorine=11.0).validate()
    wt.ReactorConfiguration().validate()


def test_boundary_block_packing(wt):
    bb = wt.boundary_block(wt.BoundaryConditions(acid_flow_rate=0.5), 3)
    assert bb.shape == (10, 3) and np.all(bb[4] == 0.5) and np.all(bb[0] == 5.0)
    seq = [wt.BoundaryConditions(inlet_pH=6.0 + i) for i in range(3)]
    assert np.array_equal(wt.boundary_block(seq, 3)[1], [6.0, 7.0, 8.0])
    d = wt.boundary_block({"inlet_temperature": np.array([1.0, 2.0, 3.0])}, 3)
    assert np.array_equal(d[3], [1.0, 2.0, 3.0]) and np.all(d[7] == 50.0)
    with pytest.raises(ValueError):
        wt.boundary_block(seq, 4)
    with pytest.raises(ValueError):
        wt.boundary_block(np.zeros((9, 3)), 3)


def test_synthetic_ensemble_is_prefix_stable_and_in_range(wt):
    c1, b1 = wt.make_ensemble(64)
    c2, b2 = wt.make_ensemble(1000)
    for k in c1:
        assert np.array_equal(c1[k], c2[k][:64])
    assert np.array_equal(b1, b2[:, :64])
    c3, b3 = wt.make_ensemble(100, start=900)
    assert np.array_equal(b3, b2[:, 900:])
    assert c2["initial_pH"].min() >= 6.5 and c2["initial_pH"].max() <= 8.5
    assert c2["temperature"].min() >= 10 and c2["temperature"].max() <= 30
    assert np.all((b2[4] == 0) | ((b2[4] > 0) & (b2[4] <= 2)))
    assert 0.35 < np.mean(b2[4] == 0) < 0.65 and 0.6 < np.mean(b2[9] == 0) < 0.9
    assert np.all(b2[0] <= 12.0) and np.all(b2[0] >= 1.6)


def test_shard_bounds_cover_everything(wt):
    for N in (1, 7, 100, 100000):
        for W in (1, 2, 3, 8):
            spans = [wt.shard_bounds(N, W, r) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == N
            assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        wt.shard_bounds(10, 2, 2)


def test_library_exports_every_declared_symbol(native):
    """Every function declared in include/wtphys.h is exported by libwtphys.so."""
    hdr = open(os.path.join(ROOT, "include", "wtphys.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(wt_[a-z_0-9]+)\s*\(", hdr)))
    assert len(names) >= 20
    lib = C.CDLL(native.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), f"{name} not exported"
    assert lib.wt_abi_version() == 1


def test_no_silent_cpu_fallback(native, wt):
    """Without a HIP device the product path must fail loudly."""
    if native.device_count() > 0:
        pytest.skip("a GPU is visible; the loud-failure path is for GPU-less hosts")
    with pytest.raises(native.WtError) as ei:
        wt.ReactorEnsemble([wt.ReactorConfiguration(n_zones=4)])
    assert ei.value.code == native.WT_E_NOGPU
    with pytest.raises(native.WtError):
        wt.solve_pH(100.0, 2.0, 20.0)


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "ics-wt-physicsengine_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "wt_oracle" not in text and "libwtoracle" not in text, f


# the names wt_simulator.core exports (/root/reference/src/wt_simulator/core/__init__.py:238-263)
REFERENCE_CORE_EXPORTS = ("IntegratedCSTR", "ReactorConfiguration", "ReactorState", "BoundaryConditions",
                          "TemperatureDependentKinetics", "ArrheniusParameters", "AqueousChemistry", "BufferSystem",
                          "TransportModel", "GeometryParameters", "FlowParameters", "SpatialModel", "StratificationParameters",
                          "validate_thermodynamics", "validate_chemistry", "validate_transport", "validate_spatial",
                          "validate_integrated_reactor")


def test_import_surface_covers_the_reference_export_list(wt):
    for name in REFERENCE_CORE_EXPORTS + ("run_all_validations",):
        assert hasattr(wt, name) and name in wt.core.__all__, name


def test_physics_subobjects_reproduce_reference_constants(wt):
    """TransportModel / AqueousChemistry / TemperatureDependentKinetics / SpatialModel as host-side holders of the
    init-time constants: bit-identical to what the reference's objects held (tests/golden/g1_constants.json)."""
    g = golden_json("g1_constants.json")
    for e in g["configs"]:
        c = e["config"]
        tm = wt.TransportModel(wt.GeometryParameters(c["volume"], c["height"], c["diameter"], c["n_zones"]),
                               wt.FlowParameters(c["flow_rate"], c["turbulent_intensity"], c["recirculation_ratio"],
                                                 c["impeller_speed"], c["impeller_diameter"], c["power_number"]), c["temperature"])
        assert tm.K_exchange_per_s == e["K_exchange_per_s"] == tm.K_matrix[0, 1]
        assert tm.superficial_velocity == e["superficial_velocity"] and tm.D_effective == e["D_effective"]
        ch = wt.AqueousChemistry(wt.BufferSystem(c["alkalinity"], c["total_carbonate"], c["temperature"]))
        assert (ch.Kw, ch.Ka1, ch.Ka2, ch.Ka_HOCl) == (e["Kw"], e["Ka1"], e["Ka2"], e["Ka_HOCl"])
    th = wt.TemperatureDependentKinetics()
    assert all(th.chlorine_decay_rate(float(T)) == v for T, v in g["k_decay"].items())
    sp = wt.SpatialModel(5, 2.0)
    assert all(sp.calculate_water_density(float(T)) == v for T, v in g["density"].items())
    ch = wt.AqueousChemistry(wt.BufferSystem(100.0, 2.0, 20.0))
    assert ch.buffering_capacity(7.2) == g["beta_7p2"] and ch.pH_dependent_chlorine_decay_factor(7.2) == g["decay_factor_7p2"]
    with pytest.raises(ValueError, match="outside liquid water range"):
        th.chlorine_decay_rate(100.5)


def test_reference_validators_that_need_no_gpu(wt, capsys):
    wt.validate_thermodynamics(); wt.validate_transport(); wt.validate_spatial()
    out = capsys.readouterr().out
    assert out.count("validations passed") == 3


def _build_c_client(tmp_path):
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "ics-wt-physicsengine_amd", "csrc")
    exe = str(tmp_path / "abi_client")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "c_abi", "abi_client.c"), "-L", csrc, "-lwtphys", f"-Wl,-rpath,{csrc}", "-o", exe],
                   check=True)
    return exe


def test_boundary_is_plain_c(tmp_path):
    """include/wtphys.h is valid strict C99 and a C program that drives the whole step path through it links against
    csrc/libwtphys.so (no C++ or torch types in the signatures) -- the shape a cgo / JNI / FFI binding has."""
    import subprocess
    exe = _build_c_client(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr



def test_known_answers_of_the_physics_relations(wt):
    """The reference's own known answers (thermodynamics.py:403-417, transport.py:536-554, chemistry.py:545-546,
    spatial.py:561-565) and the fixture values of g1_constants.json, through the column-wise routines of ``params``
    that the classes and the ensemble's constant block share -- scalars and arrays give the same numbers."""
    g = golden_json("g1_constants.json")
    P = wt.params
    temps = np.array([float(t) for t in g["k_decay"]])
    assert np.array_equal(P.arrhenius(temps, 1e-4, 45000.0), np.array(list(g["k_decay"].values())))
    assert P.arrhenius(20.0, 1e-4, 45000.0) == 1e-4                            # the reference rate at the reference temperature
    assert abs(P.water_ionization_constant(25.0) / 1e-14 - 1) < 1e-6 and P.carbonate_pKa(25.0, 1) == 6.35
    dens = np.array([float(t) for t in g["density"]])
    assert np.array_equal(P.water_density(dens), np.array(list(g["density"].values())))
    assert P.water_density(4.0) == 999.97 and P.water_density(8.0) < P.water_density(8.0001)      # the 8 degC jump
    with pytest.raises(ValueError, match=r"Temperature -0\.5°C outside liquid water range \[0\.0, 100\.0\]°C"):
        P.kelvin(np.array([3.0, -0.5, 120.0]))                                  # the first offender is named
    # carbonate fractions sum to one, buffer capacity / decay factor at the fixture point
    eq = P.equilibrium_constants(np.array([20.0]))
    H = P._pow10_neg(np.array([6.0, 7.2, 8.4, 10.3]))
    assert np.allclose(sum(P.carbonate_fractions(H, eq["Ka1"], eq["Ka2"])), 1.0, rtol=0, atol=1e-15)
    assert P.buffer_capacity(H, eq["Kw"], eq["Ka1"], eq["Ka2"], 2.0 / 1000.0)[1] == g["beta_7p2"]
    assert P.chlorine_decay_factor(H, eq["Ka_HOCl"])[1] == g["decay_factor_7p2"]
    # exchange operator: zero row sums except the outlet sink, negative semi-definite, K_ex of the fixture configurations
    for e in g["configs"]:
        c = e["config"]
        cols = {k: np.array([float(c[k])]) for k in ("volume", "height", "diameter", "flow_rate", "impeller_speed",
                                                       "impeller_diameter", "power_number", "temperature")}
        t = P.transport_columns(cols, c["n_zones"])
        assert t["K_exchange_per_s"][0] == e["K_exchange_per_s"] and t["superficial_velocity"][0] == e["superficial_velocity"]
        K = P.exchange_matrix(t["K_exchange_per_s"][0], t["Q_per_V"][0], c["n_zones"])
        rows = K.sum(axis=1)
        assert np.abs(rows[:-1]).max() < 1e-12 and abs(rows[-1] + t["Q_per_V"][0]) < 1e-12
        assert np.linalg.eigvalsh((K + K.T) / 2).max() <= 1e-10
    # stratification: warm water on top is stable (Ri > 0), suppression only where Ri exceeds the critical value
    rho = P.water_density(np.array([17.0, 19.0, 21.0, 23.0, 25.0]))
    ri = P.interface_richardson(rho, 0.4, 0.01)
    assert (ri < 0).all() and (P.interface_richardson(rho[::-1], 0.4, 0.01) > 0).all()
    assert np.array_equal(P.suppression_factors(rho[::-1], 0.4, 0.01), np.full(4, 0.5)) and np.isinf(P.interface_richardson(rho, 0.4, 1e-7)).all()
    cv, seg = P.mixing_quality(np.array([[2.0, 2.0, 2.0], [1.0, 2.0, 3.0], [0.0, 0.0, 0.0]]))
    assert cv[0] == 0 and seg[0] == 0 and 0 < seg[1] < 1 and cv[2] == 0


def test_host_modules_are_not_transliterations():
    """Statement overlap of the host-side physics modules with the reference's core/*.py (docstrings and comments
    dropped on both sides, one statement per line): class / field / method names are the drop-in contract, the bodies
    are this build's own.  Runs where the reference is available (not on the GPU box)."""
    ref = "/root/reference/src/wt_simulator/core"
    if not os.path.isdir(ref):
        pytest.skip("reference sources not present")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import overlap_check as oc
    known = set()
    for f in os.listdir(ref):
        if f.endswith(".py"):
            known.update(oc.normalised_lines(os.path.join(ref, f)))
    for name in ("physics.py", "chemistry.py", "params.py"):
        lines = oc.normalised_lines(os.path.join(ROOT, "ics-wt-physicsengine_amd", "core", name))
        share = sum(l in known for l in lines) / len(lines)
        assert share < 0.25, (name, share)


def test_build_staleness_list_matches_the_includes():
    """``_native.build`` rebuilds when any source is newer than the library: the list must be wtphys.hip plus exactly
    the local headers it (transitively) includes -- no stale names, nothing missing."""
    import importlib
    native = importlib.import_module("ics-wt-physicsengine_amd.core._native")
    seen, todo = set(), ["wtphys.hip"]
    while todo:
        f = todo.pop()
        if f in seen:
            continue
        seen.add(f)
        text = open(os.path.join(native.CSRC, f)).read()
        todo += [m for m in re.findall(r'#include "([^"/]+)"', text) if os.path.exists(os.path.join(native.CSRC, m))]
    assert seen == set(native.BUILD_SOURCES)
    assert not [f for f in os.listdir(native.CSRC) if f.endswith((".hpp", ".hip")) and f not in seen], "dead source in csrc/"


def test_step_kernels_fit_the_register_file_and_four_wavefronts_per_cu(tmp_path):
    """Code-object metadata of the built library (what the loader sees): every step kernel up to 32 zones (LV <= 5,
    BASELINE config 3's n = 20 among them) runs without scratch memory, and its LDS leaves room for four wavefronts per
    CU (160 KiB / 4).  A register or LDS regression shows here, on the CPU, before it shows as a slower bench line."""
    import importlib, shutil, subprocess
    native = importlib.import_module("ics-wt-physicsengine_amd.core._native")
    llvm = "/opt/rocm/lib/llvm/bin"
    if not all(os.path.exists(os.path.join(llvm, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")):
        pytest.skip("ROCm LLVM tools not installed")
    fat, dev = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.run([f"{llvm}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", os.path.join(native.CSRC, "libwtphys.so")], check=True)
    subprocess.run([f"{llvm}/clang-offload-bundler", "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={fat}", f"--output={dev}"], check=True)
    notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", dev], check=True, capture_output=True, text=True).stdout
    seen = {}
    for block in notes.split("\n  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        m = re.search(r"step_kernelILi(\d)ELb(\d)", name)
        if not m:
            continue
        get = lambda f: int(re.search(re.escape(f) + r":\s+(\d+)", block).group(1))
        seen[(int(m.group(1)), bool(int(m.group(2))))] = (get(".private_segment_fixed_size"), get(".group_segment_fixed_size"),
                                                          get(".vgpr_count"))
    assert {(1, True), (2, True), (3, True), (4, True), (2, False), (3, False), (4, False), (5, False), (6, False)} <= set(seen)
    for (lv, row), (scratch, lds, regs) in seen.items():
        assert regs <= 512
        if lv <= 5:
            assert scratch == 0, (lv, row, scratch)
            assert lds <= 40960, (lv, row, lds)
