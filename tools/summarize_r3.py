"""Fold gpurun_out/r3 (tools/collect_r3.sh) into profiles/r3/: kernel_stats.csv, launch_agreement.json, pmc_summary.json
(n = 8) and pmc_n20.json, traffic.json (HBM bytes per zone-step, keyed by work-item length), pmc_fp64.json, bench_*.json."""
import collections, csv, glob, json, os, shutil, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(R, "gpurun_out", "r3"); P = os.path.join(R, "profiles", "r3")
os.makedirs(P, exist_ok=True)
K = "step_kernel"
shutil.copy(os.path.join(O, "trace", "plain_kernel_stats.csv"), os.path.join(P, "kernel_stats.csv"))
COUNTERS_ONLY = "--counters-only" in sys.argv      # first call of collect_r3.sh: the bench rows do not exist yet
for f in ([] if COUNTERS_ONLY else glob.glob(os.path.join(O, "bench_*.json"))):
    if os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(P, os.path.basename(f)))
rows = [r for r in csv.DictReader(open(os.path.join(O, "trace", "plain_kernel_trace.csv"))) if K in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
bt = json.load(open(os.path.join(O, "bench_under_rocprof.json")))
nl = bt["roofline"]["launches"]
agree = {"source": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 500 --warmup 100 (profiles/r3/kernel_stats.csv)",
         "step_kernel_dispatches": len(d), "durations_us": d, "timed_dispatch_us_rocprof": sum(d[-nl:]) / nl,
         "bench_avg_launch_us_same_run": bt["roofline"]["avg_launch_us"], "steps_per_timed_dispatch": bt["steps"] / nl}
json.dump(agree, open(os.path.join(P, "launch_agreement.json"), "w"), indent=1)


def counters(*passes):
    """timed-launch value of every counter of the named passes, and the dispatch metadata"""
    out, meta = {}, None
    for p in passes:
        for f in sorted(glob.glob(os.path.join(O, p, "**", "p_counter_collection.csv"), recursive=True)):
            acc = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                if K in row["Kernel_Name"]:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                    meta = {k: row[k] for k in ("Kernel_Name", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                                "Workgroup_Size", "Grid_Size") if k in row}
            for k, v in acc.items():
                out[k] = {"dispatches": len(v), "timed_launch": v[-1], "warmup_launch": v[0]}
    out["dispatch_meta"] = meta
    return out


def traffic_entry(c, N, n, steps, item, workload):
    fetch, write = c["FETCH_SIZE"]["timed_launch"] * 1024.0, c["WRITE_SIZE"]["timed_launch"] * 1024.0
    # MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts 64 B per 128-B request on gfx950 -> doubled; WRITE_SIZE is exact
    zs = N * n * steps
    return {"hbm_bytes_per_zone_step": (2.0 * fetch + write) / zs, "fetch_size_bytes_raw": fetch, "write_size_bytes": write,
            "zone_steps_profiled": zs, "steps_per_item": item, "workload": workload}


def per_group_step(c, N, n, steps):
    T = lambda k: c[k]["timed_launch"]
    gs = -(-N // (64 // n)) * steps
    f64 = T("SQ_INSTS_VALU_ADD_F64") + T("SQ_INSTS_VALU_MUL_F64") + T("SQ_INSTS_VALU_FMA_F64") + T("SQ_INSTS_VALU_TRANS_F64")
    return {"valu_insts": T("SQ_INSTS_VALU") / gs, "salu_insts": T("SQ_INSTS_SALU") / gs, "lds_insts": T("SQ_INSTS_LDS") / gs,
            "smem_insts": T("SQ_INSTS_SMEM") / gs, "fp64_valu_insts": f64 / gs, "wave_cycles": 4.0 * T("SQ_WAVE_CYCLES") / gs,
            "valu_busy_frac": T("SQ_ACTIVE_INST_VALU") / T("SQ_WAVE_CYCLES"), "wait_any_frac": T("SQ_WAIT_ANY") / T("SQ_WAVE_CYCLES"),
            "lane_utilisation": T("SQ_THREAD_CYCLES_VALU") / (64.0 * T("SQ_ACTIVE_INST_VALU"))}


def flop_per_zone_step(c, N, n, steps):
    T = lambda k: c[k]["timed_launch"]
    return (T("SQ_INSTS_VALU_ADD_F64") + T("SQ_INSTS_VALU_MUL_F64") + 2.0 * T("SQ_INSTS_VALU_FMA_F64") + T("SQ_INSTS_VALU_TRANS_F64")) * 64.0 / (N * n * steps)


N, n, steps = bt["config"]["reactors_per_gpu"], bt["config"]["zones"], bt["steps"]
long_c = counters("long_fetch", "long_write", "long_mix", "long_f64", "long_act")
short_c = counters("short_fetch", "short_write")
item_long, item_short = bt["roofline"]["steps_per_item"], 3      # queue schedule: 20 steps / 6 -> 3-step items (wt_ensemble_item_steps)
src = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, KiB x 1024), timed launch; read side doubled per the gfx950 "
       "correction of MI355X_MICROARCH.md")
json.dump({"zones": n, "source": src, "by_steps_per_item": {
    str(item_long): traffic_entry(long_c, N, n, steps, item_long, f"{N} x {n}, one queue-schedule launch of {steps} outer steps"),
    str(item_short): traffic_entry(short_c, N, n, 20, item_short, f"{N} x {n}, one queue-schedule launch of 20 outer steps after 5 warm-up steps")}},
    open(os.path.join(P, "traffic.json"), "w"), indent=1)
T = lambda k: long_c[k]["timed_launch"]
json.dump({"zones": n, "fp64_flop_per_zone_step": flop_per_zone_step(long_c, N, n, steps), "zone_steps_profiled": N * n * steps,
           "wave_instructions": {k: T(k) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")},
           "source": "rocprofv3 --pmc SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 on the timed launch; flop = (ADD + MUL + 2 FMA + TRANS) x 64 lanes (inactive lanes included)"},
          open(os.path.join(P, "pmc_fp64.json"), "w"), indent=1)
long_c["per_group_step"] = per_group_step(long_c, N, n, steps)
json.dump(long_c, open(os.path.join(P, "pmc_summary.json"), "w"), indent=1)
n20 = counters("n20_mix", "n20_f64", "n20_fetch", "n20_write")
n20["per_group_step"] = per_group_step(n20, N, 20, steps)
n20["fp64_flop_per_zone_step"] = flop_per_zone_step(n20, N, 20, steps)
n20["traffic"] = traffic_entry(n20, N, 20, steps, 32, f"{N} x 20, one queue-schedule launch of {steps} outer steps")
json.dump(n20, open(os.path.join(P, "pmc_n20.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in agree.items() if k != "durations_us"}, indent=1))
print("n = 8 per group-step:", json.dumps(long_c["per_group_step"], indent=1))
print("n = 20 per group-step:", json.dumps(n20["per_group_step"], indent=1))
print("traffic:", {k: v["hbm_bytes_per_zone_step"] for k, v in json.load(open(os.path.join(P, "traffic.json")))["by_steps_per_item"].items()})
