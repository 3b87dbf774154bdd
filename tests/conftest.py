"""Shared fixtures.  GPU tests are marked ``@pytest.mark.gpu`` and call the HIP
path through the C ABI; everything else runs on the CPU (oracle vs golden
vectors, host logic, symbol export of the shared library)."""
import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (runs the HIP kernels through libwtphys.so)")


@pytest.fixture(scope="session")
def wt():
    return importlib.import_module("ics-wt-physicsengine_amd")


@pytest.fixture(scope="session")
def oracle():
    import wt_oracle
    wt_oracle.lib()
    wt_oracle.set_linsolve(0)
    return wt_oracle


@pytest.fixture(scope="session")
def native(wt):
    """Built libwtphys.so (compiles it with hipcc if missing; cross-compiles without a GPU)."""
    from importlib import import_module
    nat = import_module("ics-wt-physicsengine_amd.core._native")
    nat.build()
    nat.lib()
    return nat


@pytest.fixture(scope="session")
def gpu(native):
    if native.device_count() < 1:
        pytest.fail("gpu-marked test started without a visible HIP device")
    return native


def golden_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def cfg_columns(cfg_rows, cfg_fields):
    """(cases, n_fields) array + field names -> dict of columns (without n_zones)."""
    cfg_rows = np.atleast_2d(cfg_rows)
    cols = {}
    for j, name in enumerate([str(x) for x in cfg_fields]):
        if name == "n_zones":
            continue
        col = cfg_rows[:, j]
        cols[name] = col.astype(bool) if name == "enable_thermal_stratification" else col.astype(np.float64)
    return cols


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))


SCENARIOS = ("quiet", "main", "dose", "heat", "nostrat", "dt30")
