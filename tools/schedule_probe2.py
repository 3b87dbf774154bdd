import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode = sys.argv[1]
if mode in ("torch_first", "torch_first_q8"):
    import torch
    torch.cuda.set_device(0); x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
import numpy as np
wt = importlib.import_module("ics-wt-physicsengine_amd")
n, N, STEPS = 8, 10000, 500
cols, bc = wt.make_ensemble(N)
ens = wt.ReactorEnsemble(cols, n_zones=n); ens.set_boundary(bc)
if mode == "torch_after":
    import torch
    torch.cuda.set_device(0); x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
ens.step(1.0, n_steps=100, download=False); ens.synchronize()
ens.launch_timing(True)
t0 = time.perf_counter()
ens.step(1.0, n_steps=STEPS, download=False); ens.synchronize()
dt = time.perf_counter() - t0
nl, sm, mx = ens.launch_stats()
print(f"{mode}: {dt/STEPS*1e6:7.1f} us/step {N*n*STEPS/dt:.3e} zs/s in-flight {sm/(dt*1e3):.2f}", flush=True)
