"""Run by tests/test_gpu_parity.py::test_device_export_and_rccl_gather in its own process (GPU box)."""
import importlib
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist

assert torch.cuda.is_available()
torch.zeros(1, device="cuda:0")                     # torch's HIP runtime first
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wt = importlib.import_module("ics-wt-physicsengine_amd")

N, n = 999, 8
cols, bc = wt.make_ensemble(N, seed=5)
ens = wt.ReactorEnsemble(cols, n_zones=n)
ens.set_boundary(bc)
es = ens.step(1.0, n_steps=3)
local = torch.empty((3, N, n), dtype=torch.float64, device="cuda:0")
ens.export_state_device(local.data_ptr())
ens.synchronize()
torch.cuda.synchronize()
host = local.cpu().numpy()
assert np.array_equal(host[0], es.pH) and np.array_equal(host[1], es.chlorine) and np.array_equal(host[2], es.temperature)

with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = str(port)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
try:
    full = wt.gather_state(local, 1, force_collective=True, sizes=[N])
    torch.cuda.synchronize()
    assert full.shape == (3, N, n) and torch.equal(full, local)
finally:
    dist.destroy_process_group()
ens.close()
print("export and gather ok")
