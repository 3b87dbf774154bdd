"""world_size-2 gloo rehearsal of the multi-GPU path: contiguous sharding of the
synthetic ensemble by reactor index, independent stepping per rank (the CPU
oracle stands in for the kernel here, as the checker), one final all_gather."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

N_TOTAL, NZ, STEPS = 48, 4, 3


def _columns(wt, cols, S):
    d = wt.ReactorConfiguration()
    return {k: np.broadcast_to(np.asarray(cols.get(k, getattr(d, k))), (S,)).copy()
            for k in ("volume", "height", "diameter", "flow_rate", "impeller_speed", "impeller_diameter",
                      "total_carbonate", "temperature", "enable_thermal_stratification")}


def _step_block(wt, O, lo, hi):
    S = hi - lo
    cols, bc = wt.make_ensemble(S, start=lo)
    par = wt.params.derive_constants(_columns(wt, cols, S), NZ)
    shape = (S, NZ)
    pH = np.broadcast_to(cols["initial_pH"][:, None], shape).copy()
    Cl = np.broadcast_to(cols["initial_chlorine"][:, None], shape).copy()
    T = np.broadcast_to(cols["temperature"][:, None], shape).copy()
    pH, Cl, T, t, st = O.ensemble_step(NZ, par, bc, 1.0, STEPS, pH, Cl, T, np.zeros(S), nthreads=1)
    return np.stack([pH, Cl, T])


def _worker(rank, world, port, q, n_total=N_TOTAL):
    import importlib
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wt = importlib.import_module("ics-wt-physicsengine_amd")
    import wt_oracle as O
    lo, hi = wt.shard_bounds(n_total, world, rank)
    local = torch.from_numpy(_step_block(wt, O, lo, hi))
    sizes = [wt.shard_bounds(n_total, world, r)[1] - wt.shard_bounds(n_total, world, r)[0] for r in range(world)]
    full = wt.gather_state(local, world, sizes=sizes)
    if rank == 0:
        q.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [N_TOTAL, 37])   # 37: unequal blocks (19 + 18), padded for the collective
def test_two_rank_sharding_and_gather(wt, oracle, n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (1 if n_total != N_TOTAL else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, n_total)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    single = _step_block(wt, oracle, 0, n_total)
    assert gathered.shape == (3, n_total, NZ)
    assert np.array_equal(gathered, single)
