"""Bitwise fingerprint of what a build computes (developer tool): sha256 over state, status, solver counters, sensor
readings and register images after a fixed schedule of calls.  An instruction-stream change must not move any of them.
   [WTPHYS_LIB=tools/scratch/x.so] python tools/bits_check.py [quick]"""
import hashlib, importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
wt = importlib.import_module("ics-wt-physicsengine_amd")


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def run(n, N, calls, sensors=False, plc=False):
    cols, bc = wt.make_ensemble(N)
    ens = wt.ReactorEnsemble(cols, n_zones=n)
    ens.set_boundary(bc)
    if sensors:
        ens.enable_sensors(seed=11)
    if plc:
        ens.enable_plant_io(); ens.write_commands(0.4, 0.2, 5.5)
    t0 = time.time()
    for k in calls:
        es = ens.step(1.0, n_steps=k)
    dt = time.time() - t0
    st = ens.solver_stats()
    parts = [es.pH, es.chlorine, es.temperature, es.time, es.status, st]
    if sensors:
        parts += list(ens.sensor_readings())
    if plc:
        parts += list(ens.input_image())
    d = digest(*parts)
    ens.close()
    return d, dt


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
cases = [("n8  10000 [5,20,100]", 8, 10000, [5, 20, 100], False, False),
         ("n8  2000 sensors+plc [3,40]", 8, 2000, [3, 40], True, True)]
if not quick:
    cases += [("n4  3000 [30]", 4, 3000, [30], False, False), ("n5  1000 [30]", 5, 1000, [30], False, False),
              ("n16 1000 [30]", 16, 1000, [30], False, False), ("n20 2000 [30]", 20, 2000, [30], False, False),
              ("n40 300 [10]", 40, 300, [10], False, False), ("n2  3000 [20]", 2, 3000, [20], False, False)]
for name, n, N, calls, s, p in cases:
    d, dt = run(n, N, calls, s, p)
    print(f"{name:32s} {d}  ({dt:.2f} s)")
