#!/usr/bin/env python3
"""Benchmark of the multi-zone CSTR physics step on MI355X.

One "step" = one outer dt = 1 s advance (IntegratedCSTR.step) of every reactor
of the synthetic ensemble resident on this rank's GPU.  Metric:
reactor-zone-steps/s, whole job.

    python bench.py                                  # 1 GPU, 10k x 8, 500 steps
    python bench.py --gpus 8                         # spawns 8 ranks itself (torch.distributed.run)
    python bench.py --gpus 8 --total-reactors 100000 # BASELINE config 4: ONE ensemble cut by reactor index
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...   # what the driver runs

Sharding: reactors are independent, so every rank owns a
contiguous block of reactor indices and stepping needs no collective.  Default is
weak scaling (--reactors per GPU fixed); --total-reactors cuts one ensemble with
core.sharding.shard_bounds (strong scaling).  The final state is gathered once with
RCCL (all_gather) outside the timed region and timed separately.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

# before anything initialises HIP (torch included): one hardware queue per reactor-range stream
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEFAULT_CHUNK = 50            # outer steps per work item (the library's default schedule, WT_DEFAULT_CHUNK)
HBM_PEAK_GBS = 8000.0         # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CUs x 4 SIMDs x 16 fp64 FMA lanes/clk x 2 flop x 2.4 GHz
PROFILE_DIR = os.path.join(ROOT, "profiles", "r3")
# BASELINE.md section 2 (survey container, 1 core of an 8-core Xeon @ 2.1 GHz): the reference cannot
# travel to the GPU box, so its measured rate is carried as a labelled constant
REFERENCE_PYTHON = {"value": 1.63e3, "unit": "reactor-zone-steps/s", "cores": 1,
                    "label": "reference Python baseline (survey container)",
                    "source": "BASELINE.md section 2: 8 zones, default boundary, 4.92 ms per outer step"}


def algorithmic_bytes_per_zone_step(n_zones: int, steps_per_item: int) -> float:
    """3 state doubles read + 3 written per zone-step (48 B).  The derived
    arrays (24 B) and the 10 boundary scalars (80/n B) move once per work item of
    ``steps_per_item`` outer steps, not once per step, so they are amortised over it."""
    return 48.0 + (24.0 + 80.0 / n_zones) / max(1, steps_per_item)


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--reactors", type=int, default=10000, help="reactors per GPU (weak scaling)")
    ap.add_argument("--total-reactors", type=int, default=0,
                    help="strong scaling: ONE ensemble of this many reactors cut into contiguous blocks by "
                         "shard_bounds (BASELINE config 4: 100000 over 8 GPUs = 12500 each); overrides --reactors")
    ap.add_argument("--zones", type=int, default=8)
    ap.add_argument("--chunk", type=int, default=0,
                    help="outer steps per work item / PLC scan interval (1 = one scan per outer step; 0 = library default)")
    ap.add_argument("--streams", type=int, default=0,
                    help="legacy multi-stream schedule: reactor ranges / HIP streams per GPU (0 = library default)")
    ap.add_argument("--sensors", action="store_true",
                    help="BASELINE config 5: also run the fused fp32 sensor suite (7 readings per reactor per step)")
    ap.add_argument("--plant-io", action="store_true",
                    help="also keep the per-reactor Modbus register images and run the command path once per scan "
                         "(--chunk 1 = one PLC scan per outer step, as the reference loop); implies --sensors")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-reactors", type=int, default=8192)
    ap.add_argument("--cpu-sample-steps", type=int, default=600)
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl")
    ap.add_argument("--placement", choices=("adaptive", "identity"), default="adaptive",
                    help="adaptive (library default): reactors of similar solver cost share a wavefront, re-dealt from the "
                         "solver counters of the previous calls; identity: reactor r in slot r")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the launcher / sharding / timing / gather logic (gloo, no HIP call, no "
                         "stepping): what tests/test_bench_launcher.py runs")
    return ap


# --------------------------------------------------------------------------- launcher
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n_ranks: int, argv) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as CHILD processes.
    Nothing in this process has touched HIP or torch.cuda yet (the parent must never re-exec after it has),
    and it never does: it only waits for the children and passes rank 0's JSON line through."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


# --------------------------------------------------------------------------- CPU baseline leg
def cpu_baseline(ens_constants, cols, bc, n_zones: int, sample_reactors: int, sample_steps: int, warm_steps: int):
    """Times the CPU oracle (a C port of the reference algorithm, dense LU as scipy does) on a bounded
    sample of the same synthetic workload: on all host cores, and on one core (smaller sample)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import wt_oracle as O  # noqa: E402  (cpu_baseline leg only)

    def timed(S, steps, threads):
        par = np.ascontiguousarray(ens_constants[:, :S])
        bcs = np.ascontiguousarray(bc[:, :S])
        shape = (S, n_zones)
        pH = np.broadcast_to(cols["initial_pH"][:S, None], shape).copy()
        Cl = np.broadcast_to(cols["initial_chlorine"][:S, None], shape).copy()
        T = np.broadcast_to(cols["temperature"][:S, None], shape).copy()
        t = np.zeros(S)
        if warm_steps:
            pH, Cl, T, t, _ = O.ensemble_step(n_zones, par, bcs, 1.0, warm_steps, pH, Cl, T, t, nthreads=threads)
        t0 = time.perf_counter()
        O.ensemble_step(n_zones, par, bcs, 1.0, steps, pH, Cl, T, t, nthreads=threads)
        dt = time.perf_counter() - t0
        return S * n_zones * steps / dt, dt

    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    S = sample_reactors
    v_all, dt_all = timed(S, sample_steps, cores)
    S1 = max(64, min(S, 1024))                              # ~10 s on one core
    steps1 = max(20, sample_steps // 2)
    v_one, dt_one = timed(S1, steps1, 1)
    return {
        "value": v_all,
        "unit": "reactor-zone-steps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {S} reactors x {n_zones} zones of the same synthetic ensemble, "
                  f"{sample_steps} steps after {warm_steps} warm-up steps, {dt_all:.2f} s wall, OpenMP over reactors",
        "one_core": {"value": v_one, "cores": 1,
                     "sample": f"first {S1} reactors, {steps1} steps after {warm_steps} warm-up steps, {dt_one:.2f} s wall"},
        "reference_python": REFERENCE_PYTHON,
    }


def _dropin_loop(wt, device: int, steps: int) -> float:
    r = wt.IntegratedCSTR(wt.ReactorConfiguration(n_zones=4), device=device)
    b = wt.BoundaryConditions()
    for _ in range(20):
        r.step(1.0, b)
    t0 = time.perf_counter()
    for _ in range(steps):
        r.step(1.0, b)
    return (time.perf_counter() - t0) * 1e3 / steps


def dropin_latency(wt, device: int, steps: int = 600):
    """BASELINE config 1's shape on the GPU drop-in: one 4-zone IntegratedCSTR, dt = 1 s, `steps` calls of
    step() with the default boundary, host round trip included (state download every step, as the reference's
    callers see it).  The reference needs 2.64 ms per step on one CPU core.
    Measured where the drop-in lives -- a process of its own that loads libwtphys only, like the reference's loop:
    with torch's HIP runtime in the same process (this one) every launch + synchronisation takes 2.4x as long; that
    figure is reported beside it."""
    here = _dropin_loop(wt, device, steps)
    alone = None
    try:
        code = ("import importlib, json, sys; sys.path.insert(0, %r); import bench; "
                "wt = importlib.import_module('ics-wt-physicsengine_amd'); print(json.dumps(bench._dropin_loop(wt, %d, %d)))" % (ROOT, device, steps))
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
        if p.returncode == 0:
            alone = float(json.loads(p.stdout.strip().splitlines()[-1]))
    except Exception:
        alone = None
    return {"ms_per_step": alone if alone is not None else here, "ms_per_step_in_this_process_with_torch": here, "steps": steps, "zones": 4,
            "reference_python_ms_per_step": 2.64,
            "note": "IntegratedCSTR.step() of one reactor incl. launch, synchronisation and state download, in a process that loads "
                    "libwtphys only (the drop-in's habitat); reference figure: BASELINE.md section 2 (survey container, 1 core)"}


def _profile_json(name: str):
    p = os.path.join(PROFILE_DIR, name)
    if os.path.exists(p):
        with open(p) as fh:
            return json.load(fh)
    return None


# --------------------------------------------------------------------------- one rank
def run_rank(args) -> int:
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2

    import torch
    import torch.distributed as dist

    dry = args.dry_run
    if not dry and not torch.cuda.is_available():
        print("bench.py: no GPU visible; the physics step has no CPU path", file=sys.stderr)
        return 3
    backend = "gloo" if dry else args.backend
    dev = torch.device("cpu") if dry else torch.device("cuda", local_rank)
    if not dry:
        torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ     # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    wt = importlib.import_module("ics-wt-physicsengine_amd")
    n = args.zones
    strong = args.total_reactors > 0
    if strong:       # one ensemble, contiguous blocks of reactor indices (core/sharding.py)
        total = args.total_reactors
        lo, hi = wt.shard_bounds(total, world, rank)
        sizes = [wt.shard_bounds(total, world, r)[1] - wt.shard_bounds(total, world, r)[0] for r in range(world)]
    else:            # weak scaling: every rank owns --reactors reactors, a distinct slice of the population
        total = world * args.reactors
        lo, hi = rank * args.reactors, (rank + 1) * args.reactors
        sizes = [args.reactors] * world
    N = hi - lo
    if N <= 0:
        print(f"bench.py: rank {rank} owns no reactors ({total} over {world} ranks)", file=sys.stderr)
        return 2
    cols, bc = wt.make_ensemble(N, start=lo)
    if args.plant_io:
        args.sensors = True

    waves = -(-N // (64 // n))
    if args.chunk <= 0:
        if args.streams > 0:  # stream schedule: a launch lasts as long as its slowest wavefront, keep >= 4 launches per range
            args.chunk = min(DEFAULT_CHUNK, max(1, -(-args.steps // (4 if args.steps > 50 else 2))))
        else:                 # queue schedule: the chunk is only the PLC scan interval
            args.chunk = DEFAULT_CHUNK
    ens = None
    if not dry:
        ens = wt.ReactorEnsemble(cols, n_zones=n, device=local_rank)
        ens.set_boundary(bc)
        if args.sensors:
            ens.enable_sensors(seed=0x5EED5EED1234, reactor_base=lo)
        if args.plant_io:
            ens.enable_plant_io()
            # the masters' setpoints = the synthetic boundary (as float32 registers), so the physics workload stays the same
            ens.write_commands(bc[4], bc[6], bc[0])
        ens.set_schedule(args.streams, max(1, args.chunk))
        ens.set_placement(args.placement == "adaptive")
    sched = ens.schedule() if ens is not None else {"mode": "dry-run", "streams": 0, "chunk": args.chunk, "workers": 0}

    def barrier():
        if ens is not None:
            ens.synchronize()
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    def run(k: int):
        if ens is not None and k > 0:
            ens.step(1.0, n_steps=k, fused=True, download=False)

    run(args.warmup)
    barrier()
    if ens is not None:
        ens.launch_timing(True)      # HIP events around every launch, on the stream it runs on
        ens.timer_start()
    t0 = time.perf_counter()
    run(args.steps)
    kernel_ms = ens.timer_stop() if ens is not None else 0.0   # HIP events on the handle's stream around the region
    barrier()
    elapsed = time.perf_counter() - t0
    n_launch, launch_sum_ms, launch_max_ms = ens.launch_stats() if ens is not None else (0, 0.0, 0.0)
    if ens is not None:
        ens.launch_timing(False)
        sched = ens.schedule()       # as it stands after the run: "redeals" says whether the cost-aware placement ever acted

    el = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms = float(el[0]), float(el[1])

    # final state gather (RCCL over xGMI), outside the timed region
    gather_ms = None
    if ens is not None:
        local = torch.empty((3, N, n), dtype=torch.float64, device=dev)
        ens.export_state_device(local.data_ptr())
        ens.synchronize()
    else:
        shape = (N, n)
        local = torch.from_numpy(np.stack([np.broadcast_to(cols[k][:, None], shape)
                                           for k in ("initial_pH", "initial_chlorine", "temperature")]).copy())
    if use_dist:
        if not dry:
            torch.cuda.synchronize()
        dist.barrier()
        g0 = time.perf_counter()
        final = wt.gather_state(local, world, force_collective=True, sizes=sizes)
        if not dry:
            torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
    else:
        final = local
    st = ens.status() if ens is not None else np.zeros(N, dtype=np.uint32)
    flagged = torch.tensor([int(np.count_nonzero(st))], device=dev)
    if use_dist:
        dist.all_reduce(flagged)
    checksum = float(final.sum())
    assert final.shape == (3, total, n), final.shape

    if rank == 0:
        zone_steps = total * n * args.steps
        value = zone_steps / elapsed if not dry else None
        out = {
            "metric": "reactor-zone-steps/sec",
            "value": value,
            "unit": "reactor-zone-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(1, args.steps) * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"{total}-reactor x {n}-zone ensemble cut over {world} GPU(s) by reactor index "
                             f"({max(sizes)} per GPU)" if strong else f"{N}-reactor x {n}-zone ensemble per GPU")
                            + f", dt=1 s, fp64, {sched['mode']} schedule"
                            + (" + fused fp32 sensor suite" if args.sensors else "")
                            + (" + Modbus register image / command path per scan" if args.plant_io else ""),
                "reactors_total": total, "reactors_per_gpu": max(sizes), "zones": n, "dt_s": 1.0,
                "scan_interval_steps": sched["chunk"], "schedule": sched,
                "sharding": f"instance-parallel x{world} (contiguous reactor blocks, shard_bounds), final RCCL all_gather only",
            },
            "final_gather_ms": gather_ms,
            "flagged_reactors": int(flagged.item()),
            "state_checksum": checksum,
        }
        if dry:
            out["dry_run"] = True
            out["shard_sizes"] = sizes
        else:
            # roofline of the dominant kernel (wt::step_kernel / wt::step_worker_kernel).  One launch
            # advances this rank's reactors by `steps` outer steps (persistent schedule) or one reactor
            # range by <= chunk steps (stream schedule); algorithmic bytes = per-unit figure x zone-steps.
            item_steps = ens.item_steps(args.steps)     # steps a reactor's state stays in registers between memory round trips
            per_unit = algorithmic_bytes_per_zone_step(n, item_steps)
            avg_launch_s = launch_sum_ms * 1e-3 / max(n_launch, 1)
            in_flight = (launch_sum_ms / kernel_ms) if kernel_ms > 0 else 1.0   # launches overlapping on the GPU
            zs_per_launch = N * n * args.steps / max(n_launch, 1)
            bytes_per_launch = per_unit * zs_per_launch
            achieved = bytes_per_launch / avg_launch_s * in_flight / 1e9       # == rank-0 bytes / region time
            # measured HBM traffic: rocprofv3 PMC passes (tools/collect_r3.sh), one entry per profiled work-item length --
            # how often a group's state returns to memory depends on it, so a figure is only quoted for a run of the
            # same zone count and item length as the profiled one, never scaled from another
            tr = _profile_json("traffic.json")
            entry = ((tr or {}).get("by_steps_per_item", {}).get(str(item_steps))
                     if tr and tr.get("zones") == n and not args.sensors else None)      # (the sensor suite adds traffic of its own)
            traffic = entry["hbm_bytes_per_zone_step"] * zs_per_launch if entry else None
            fl = _profile_json("pmc_fp64.json")
            compute = None
            if fl and n == fl.get("zones"):
                fpz = fl["fp64_flop_per_zone_step"]
                ach = value * fpz / 1e12
                compute = {"bound": "fp64 vector ALU", "achieved": ach, "peak": FP64_VECTOR_PEAK_TFLOPS * world,
                           "unit": "TFLOP/s", "frac": ach / (FP64_VECTOR_PEAK_TFLOPS * world),
                           "fp64_flop_per_zone_step": fpz,
                           "source": "profiles/r3/pmc_fp64.json: (SQ_INSTS_VALU_ADD_F64 + MUL_F64 + 2 FMA_F64 + TRANS_F64) "
                                     "x 64 lanes / zone-steps of the profiled run; peak = 1024 SIMDs x 16 FMA lanes x 2 x 2.4 GHz"}
            out["roofline"] = {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": (f"profiles/r3/traffic.json[steps_per_item={item_steps}]: rocprofv3 (2 x FETCH_SIZE + WRITE_SIZE) per "
                                   f"zone-step of the profiled run ({entry['workload']}) x zone-steps per launch") if entry else None,
                "kernel": sched.get("kernel", "wt::step_kernel"),
                "avg_launch_us": avg_launch_s * 1e6,
                "max_launch_us": launch_max_ms * 1e3,
                "launches": n_launch,
                "launches_in_flight": in_flight,
                "algorithmic_bytes_per_zone_step": per_unit,
                "steps_per_item": item_steps,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "formula": "achieved = (48 + (24 + 80/n)/steps_per_item) B x zone-steps per launch / avg_launch_us "
                           "x launches_in_flight (= all algorithmic bytes / HIP-event time of the timed region)",
                "note": "path is fp64-VALU/latency bound (adaptive implicit solve per reactor), not HBM bound; "
                        "see DESIGN.md roofline section",
            }
            out["roofline_compute"] = compute
            out["sensors"] = ({"suite": "7 sensors/reactor (pH in/out, Cl amperometric/DPD, magnetic flow, RTD in/out), fp32, "
                                        "one read per outer step, Philox4x32-10 streams",
                               "readings_per_s": total * 7 * args.steps / elapsed} if args.sensors else None)
            if not args.no_cpu_baseline and world == 1:
                out["dropin_n1"] = dropin_latency(wt, local_rank)
                out["cpu_baseline"] = cpu_baseline(ens.constants, cols, bc, n, min(args.cpu_sample_reactors, N),
                                                   args.cpu_sample_steps, warm_steps=5)
            elif not args.no_cpu_baseline:
                out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if ens is not None:
        ens.close()
    return 0


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
