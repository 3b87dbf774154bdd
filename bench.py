#!/usr/bin/env python3
"""Benchmark of the multi-zone CSTR physics step on MI355X.

One "step" = one outer dt = 1 s advance (IntegratedCSTR.step) of every reactor
of the synthetic ensemble resident on this rank's GPU.  The library advances the
ensemble as a few contiguous reactor ranges on their own HIP streams, --chunk
outer steps per kernel launch (state stays in registers inside a launch).
Metric: reactor-zone-steps/s, whole job.

    python bench.py --gpus 1 --steps 500 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Weak scaling: every rank owns --reactors reactors (instance-parallel, no
collective on the data path); the final state is gathered once with RCCL
(all_gather) outside the timed region and timed separately.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

# before anything initialises HIP (torch included): one hardware queue per reactor-range stream
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DEFAULT_CHUNK = 50     # outer steps per launch (the library's default schedule, WT_DEFAULT_CHUNK)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def algorithmic_bytes_per_zone_step(n_zones: int) -> float:
    """3 state doubles read + 3 written (48 B), the three
    derived arrays written (24 B), the 10 boundary scalars re-read per launch
    (80/n B)."""
    return 48.0 + 24.0 + 80.0 / n_zones


def cpu_baseline(ens, cols, bc, n_zones: int, sample_reactors: int, sample_steps: int, warm_steps: int):
    """Times the CPU oracle (a C port of the reference algorithm, dense LU as
    scipy does) on a bounded sample of the same synthetic workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import wt_oracle as O  # noqa: E402  (cpu_baseline leg only)

    S = sample_reactors
    par = np.ascontiguousarray(ens.constants[:, :S])
    bcs = np.ascontiguousarray(bc[:, :S])
    shape = (S, n_zones)
    pH = np.broadcast_to(cols["initial_pH"][:S, None], shape).copy()
    Cl = np.broadcast_to(cols["initial_chlorine"][:S, None], shape).copy()
    T = np.broadcast_to(cols["temperature"][:S, None], shape).copy()
    t = np.zeros(S)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    if warm_steps:
        pH, Cl, T, t, _ = O.ensemble_step(n_zones, par, bcs, 1.0, warm_steps, pH, Cl, T, t, nthreads=cores)
    t0 = time.perf_counter()
    O.ensemble_step(n_zones, par, bcs, 1.0, sample_steps, pH, Cl, T, t, nthreads=cores)
    dt = time.perf_counter() - t0
    return {
        "value": S * n_zones * sample_steps / dt,
        "unit": "reactor-zone-steps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {S} reactors x {n_zones} zones of the same synthetic ensemble, "
                  f"{sample_steps} steps after {warm_steps} warm-up steps, {dt:.2f} s wall, OpenMP over reactors",
    }


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--reactors", type=int, default=10000, help="reactors per GPU")
    ap.add_argument("--zones", type=int, default=8)
    ap.add_argument("--chunk", type=int, default=0,
                    help="outer steps per kernel launch (1 = one launch per outer step; 0 = the library default of "
                         "50, shortened for short runs so that every reactor range still gets several launches)")
    ap.add_argument("--streams", type=int, default=0,
                    help="reactor ranges / HIP streams per GPU (0 = library default: min(4, wavefronts/64))")
    ap.add_argument("--sensors", action="store_true",
                    help="BASELINE config 5: also run the fused fp32 sensor suite (7 readings per reactor per step)")
    ap.add_argument("--plant-io", action="store_true",
                    help="also keep the per-reactor Modbus register images and run the command path once per launch "
                         "(--chunk 1 = one PLC scan per outer step, as the reference loop); implies --sensors")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-reactors", type=int, default=8192)
    ap.add_argument("--cpu-sample-steps", type=int, default=600)
    args = ap.parse_args()

    if args.chunk <= 0:      # a launch lasts as long as its slowest wavefront: keep >= 4 launches per range (2 for tiny runs)
        args.chunk = min(DEFAULT_CHUNK, max(1, -(-args.steps // (4 if args.steps > 50 else 2))))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            return 2

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the physics step has no CPU path", file=sys.stderr)
        return 3
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ     # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    wt = importlib.import_module("ics-wt-physicsengine_amd")
    n, N = args.zones, args.reactors
    cols, bc = wt.make_ensemble(N, start=rank * N)  # every rank owns a distinct slice
    ens = wt.ReactorEnsemble(cols, n_zones=n, device=local_rank)
    ens.set_boundary(bc)
    if args.plant_io:
        args.sensors = True
    if args.sensors:
        ens.enable_sensors(seed=0x5EED5EED1234, reactor_base=rank * N)
    if args.plant_io:
        ens.enable_plant_io()
        # the masters' setpoints = the synthetic boundary (as float32 registers), so the physics workload stays the same
        ens.write_commands(bc[4], bc[6], bc[0])

    def barrier():
        ens.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()

    if args.streams > 0 or args.chunk != DEFAULT_CHUNK:
        waves = -(-N // (64 // n))
        ens.set_schedule(args.streams if args.streams > 0 else max(1, min(4, waves // 64)), max(1, args.chunk))
    n_streams = args.streams if args.streams > 0 else max(1, min(4, (-(-N // (64 // n))) // 64))

    def run(k: int):
        ens.step(1.0, n_steps=k, fused=True, download=False)

    run(args.warmup)
    barrier()
    ens.launch_timing(True)          # HIP events around every launch, on the stream it runs on
    ens.timer_start()
    t0 = time.perf_counter()
    run(args.steps)
    kernel_ms = ens.timer_stop()     # HIP events on the handle's stream around the whole region
    barrier()
    elapsed = time.perf_counter() - t0
    n_launch, launch_sum_ms, launch_max_ms = ens.launch_stats()
    ens.launch_timing(False)

    el = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed, kernel_ms = float(el[0]), float(el[1])

    # final state gather (RCCL over xGMI), outside the timed region
    gather_ms = None
    local = torch.empty((3, N, n), dtype=torch.float64, device="cuda")
    ens.export_state_device(local.data_ptr())
    ens.synchronize()
    if use_dist:
        torch.cuda.synchronize(); dist.barrier()
        g0 = time.perf_counter()
        final = wt.gather_state(local, world, force_collective=True)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
    else:
        final = local
    st = ens.status()
    flagged = torch.tensor([int(np.count_nonzero(st))], device="cuda")
    if use_dist:
        dist.all_reduce(flagged)
    checksum = float(final.sum())

    if rank == 0:
        zone_steps = world * N * n * args.steps
        value = zone_steps / elapsed
        # roofline of the dominant kernel (wt::step_kernel).  One launch advances one reactor
        # range by <= chunk steps; algorithmic bytes = per-unit figure x zone-steps in the launch.
        avg_launch_s = launch_sum_ms * 1e-3 / max(n_launch, 1)
        in_flight = (launch_sum_ms / kernel_ms) if kernel_ms > 0 else 1.0   # launches overlapping on the GPU
        bytes_per_launch = algorithmic_bytes_per_zone_step(n) * N * n * args.steps / max(n_launch, 1)
        achieved = bytes_per_launch / avg_launch_s * in_flight / 1e9       # == total bytes / region time
        # HBM traffic per launch measured offline with rocprofv3 PMC passes for this exact workload
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r1", "traffic.json")
        if os.path.exists(tpath) and (N, n, args.chunk, n_streams) == (10000, 8, DEFAULT_CHUNK, 4):
            with open(tpath) as fh:
                traffic = json.load(fh).get("traffic_bytes_per_launch")
        # secondary, compute-side roofline: the kernel is bound by fp64 VALU issue, not by HBM.  Peak = what the
        # chip's SIMDs can issue (one fp64 VALU instruction per 4 cycles per SIMD) divided by the measured VALU
        # instruction count per wavefront-step of this kernel (rocprofv3 SQ_INSTS_VALU / SQ_WAVES / steps).
        compute = None
        ppath = os.path.join(ROOT, "profiles", "r1", "pmc_summary.json")
        if os.path.exists(ppath) and n == 8:
            with open(ppath) as fh:
                valu = json.load(fh).get("per_wavefront_step", {}).get("valu_insts")
            if valu:
                peak = 256 * 4 * 2.4e9 / 4.0 / valu * 64.0 * world
                compute = {"bound": "fp64 VALU issue", "achieved": value, "peak": peak, "unit": "reactor-zone-steps/s",
                           "frac": value / peak, "valu_insts_per_wavefront_step": valu,
                           "source": "profiles/r1/pmc_summary.json (SQ_INSTS_VALU / SQ_WAVES / steps per launch); "
                                     "peak = 1024 SIMDs x 2.4 GHz / 4 cycles per fp64 VALU op / insts x 64 zones"}
        out = {
            "metric": "reactor-zone-steps/sec",
            "value": value,
            "unit": "reactor-zone-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{N}-reactor x {n}-zone ensemble per GPU, dt=1 s, fp64, "
                            f"{args.chunk} outer step(s) per launch, {n_streams} reactor range(s)/stream(s)"
                            + (" + fused fp32 sensor suite" if args.sensors else "")
                            + (" + Modbus register image / command path per launch" if args.plant_io else ""),
                "reactors_per_gpu": N, "zones": n, "dt_s": 1.0, "steps_per_launch": args.chunk,
                "sharding": f"instance-parallel x{world}, final RCCL all_gather only",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": "profiles/r1/traffic.json (rocprofv3 FETCH_SIZE + WRITE_SIZE, bytes per launch)" if traffic else None,
                "kernel": "wt::step_kernel",
                "avg_launch_us": avg_launch_s * 1e6,
                "max_launch_us": launch_max_ms * 1e3,
                "launches": n_launch,
                "launches_in_flight": in_flight,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "formula": "achieved = algorithmic_bytes_per_launch / avg_launch_us * launches_in_flight "
                           "(= all algorithmic bytes / HIP-event time of the timed region)",
                "note": "path is fp64-VALU/latency bound (adaptive implicit solve per reactor), not HBM bound; "
                        "see DESIGN.md roofline section",
            },
            "roofline_compute": compute,
            "sensors": ({"suite": "7 sensors/reactor (pH in/out, Cl amperometric/DPD, magnetic flow, RTD in/out), fp32, "
                                  "one read per outer step, Philox4x32-10 streams",
                         "readings_per_s": world * N * 7 * args.steps / elapsed} if args.sensors else None),
            "final_gather_ms": gather_ms,
            "flagged_reactors": int(flagged.item()),
            "state_checksum": checksum,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(ens, cols, bc, n, min(args.cpu_sample_reactors, N),
                                               args.cpu_sample_steps, warm_steps=5)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ens.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
