"""Synthetic reactor ensemble used by bench.py and the parity tests.

Build-owned generator; needs nothing from the
reference.  Draws are laid out per reactor row so that the first k reactors of
an N-reactor ensemble do not depend on N.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from . import params

SEED = 20260204


def make_ensemble(n_reactors: int, seed: int = SEED, start: int = 0) -> Tuple[Dict[str, np.ndarray], np.ndarray]:
    """Returns (configuration columns, boundary block (NB, N)) for reactors
    ``start .. start+n_reactors`` of the infinite synthetic population."""
    total = start + n_reactors
    rng = np.random.default_rng(seed)
    u = rng.random((total, 20))[start:]
    U = lambda j, lo, hi: lo + (hi - lo) * u[:, j]
    cols = {
        "initial_pH": U(0, 6.5, 8.5),
        "initial_chlorine": U(1, 0.5, 4.0),
        "temperature": U(2, 10.0, 30.0),
        "flow_rate": U(3, 2.0, 10.0),
        "alkalinity": U(4, 50.0, 200.0),
        "total_carbonate": U(5, 1.0, 4.0),
    }
    bc = np.empty((params.NB, n_reactors))
    bc[0] = cols["flow_rate"] * U(6, 0.8, 1.2)          # inlet_flow_rate
    bc[1] = U(7, 6.5, 8.5)                               # inlet_pH
    bc[2] = U(8, 0.0, 1.0)                               # inlet_chlorine
    bc[3] = cols["temperature"] + U(9, -5.0, 5.0)        # inlet_temperature
    bc[4] = np.where(u[:, 10] < 0.5, 0.0, U(11, 0.0, 2.0))   # acid_flow_rate
    bc[5] = 0.1                                          # acid_concentration
    bc[6] = np.where(u[:, 12] < 0.5, 0.0, U(13, 0.0, 1.0))   # chlorine_flow_rate
    bc[7] = 50.0                                         # chlorine_concentration
    bc[8] = U(14, 5.0, 25.0)                             # ambient_temperature
    bc[9] = np.where(u[:, 15] < 0.75, 0.0, U(16, 0.0, 10.0)) # heat_loss_coefficient
    return cols, np.ascontiguousarray(bc)
