#!/usr/bin/env python3
"""Golden vectors for the sensor suite (NEXT-1) from the Python reference.

Build container only.  Imports the reference's sensors and physics from /root/reference/src, replaces
each sensor's private numpy Generator by the counter-based stream of oracle/sensor_oracle.py (the
reference seeds from ``secrets`` and is otherwise not reproducible), runs the orchestrator's
per-step sequence (``reactor.step`` -> ``read_all_sensors``, __main__.py:398-410) and records the
inputs the sensors saw and everything they returned.  Writes tests/golden/g7_sensors_*.npz.
"""
from __future__ import annotations

import importlib
import logging
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference/src"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
logging.disable(logging.CRITICAL)

from wt_simulator.core import BoundaryConditions, IntegratedCSTR, ReactorConfiguration  # noqa: E402
from wt_simulator.sensors import create_realistic_sensor_suite  # noqa: E402
from wt_simulator.sensors.base_sensor import SensorFault, SensorStatus  # noqa: E402
import sensor_oracle as SO  # noqa: E402

STATUS = {s: i for i, s in enumerate(SensorStatus)}
FAULT = {f: i for i, f in enumerate(SensorFault)}
SEED = 0x5EED5EED1234


def run(case, cfg_kw, bc_kw, steps, reactor_id, t0=1000.0, dt=1.0, bc_switch=None):
    cfg = ReactorConfiguration(**cfg_kw)
    r = IntegratedCSTR(cfg)
    b = BoundaryConditions(**bc_kw)
    sensors = create_realistic_sensor_suite(cfg)
    assert tuple(sensors) == SO.SENSOR_NAMES
    for i, (name, s) in enumerate(sensors.items()):
        s._rng = SO.SuiteRng(SEED, reactor_id, i)
        # initialize_sensors (__main__.py:96-105)
        ref = 7.0 if "pH" in name else cfg.initial_chlorine if "chlorine" in name else cfg.temperature if "temp" in name else cfg.flow_rate
        s.calibrate(ref, t0, "system_init")
    n = cfg.n_zones
    taps = np.empty((steps, 7)); vals = np.empty((steps, 7)); stat = np.empty((steps, 7), dtype=np.int8); flt = np.empty((steps, 7), dtype=np.int8)
    for k in range(steps):
        if bc_switch and k in bc_switch:
            for kk, vv in bc_switch[k].items():
                setattr(b, kk, vv)
        st = r.step(dt, b)
        t = t0 + (k + 1) * dt
        taps[k] = [st.pH[0], st.pH[-1], st.chlorine[0], st.chlorine[-1], st.temperature[0], st.temperature[-1], st.flow_rate]
        for i, (name, s) in enumerate(sensors.items()):
            rd = s.read(st, current_time=t)
            vals[k, i] = rd.value; stat[k, i] = STATUS[rd.status]; flt[k, i] = FAULT[rd.fault]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"g7_sensors_{case}.npz"), taps=taps, values=vals, status=stat,
                        fault=flt, t0=t0, dt=dt, seed=SEED, reactor=reactor_id, cfg_flow_rate=cfg.flow_rate,
                        cfg_initial_chlorine=cfg.initial_chlorine, cfg_temperature=cfg.temperature, n_zones=n)
    print(case, "done; faults:", int((flt != 0).sum()), "nan readings:", int(np.isnan(vals).sum()), flush=True)


if __name__ == "__main__":
    run("main5", dict(n_zones=5, initial_pH=7.2), dict(inlet_flow_rate=5.0, inlet_pH=7.5, inlet_chlorine=0.0, inlet_temperature=20.0),
        2600, reactor_id=0, bc_switch={1900: dict(acid_flow_rate=1.5), 2100: dict(acid_flow_rate=0.0, chlorine_flow_rate=0.8),
                                       2300: dict(inlet_flow_rate=12.0, inlet_temperature=27.0)})
    run("dose8", dict(n_zones=8, temperature=14.0, flow_rate=8.0, initial_chlorine=3.0),
        dict(inlet_flow_rate=8.0, acid_flow_rate=0.5, chlorine_flow_rate=0.2, inlet_temperature=25.0, heat_loss_coefficient=5.0,
             ambient_temperature=10.0), 2300, reactor_id=7)
    run("quiet4", dict(n_zones=4), dict(), 2000, reactor_id=123456)
