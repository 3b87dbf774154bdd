"""Fold gpurun_out/r1pmc (tools/collect_r1_pmc.sh) into profiles/r1/{kernel_stats.csv,launch_agreement.json,pmc_summary.json,traffic.json,bench_*.json}."""
import collections, csv, glob, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(R, "gpurun_out", "r1pmc"); P = os.path.join(R, "profiles", "r1")
K = "step_kernel"
shutil.copy(os.path.join(O, "trace", "plain_kernel_stats.csv"), os.path.join(P, "kernel_stats.csv"))
shutil.copy(os.path.join(O, "bench_under_rocprof.json"), os.path.join(P, "bench_trace.json"))
shutil.copy(os.path.join(O, "bench_default.json"), os.path.join(P, "bench_default.json"))
rows = [r for r in csv.DictReader(open(os.path.join(O, "trace", "plain_kernel_trace.csv"))) if K in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
bt = json.load(open(os.path.join(O, "bench_under_rocprof.json")))
warm = len(d) - bt["roofline"]["launches"]
agree = {"source": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline (profiles/r1/kernel_stats.csv)",
         "step_kernel_dispatches": len(d), "avg_us_all": sum(d) / len(d), f"avg_us_warmup_{warm}": sum(d[:warm]) / max(warm, 1),
         f"avg_us_timed_{len(d) - warm}": sum(d[warm:]) / (len(d) - warm), "bench_avg_launch_us_same_run": bt["roofline"]["avg_launch_us"]}
json.dump(agree, open(os.path.join(P, "launch_agreement.json"), "w"), indent=1)
summ = {}
meta = None
for f in sorted(glob.glob(os.path.join(O, "pmc*", "p_counter_collection.csv"))):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if K in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta = {k: row[k] for k in ("Kernel_Name", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size")}
    for k, v in acc.items():
        t = v[warm:] if len(v) > warm else v       # timed dispatches only
        summ[k] = {"dispatches_timed": len(t), "mean_per_launch": sum(t) / len(t), "min": min(t), "max": max(t)}
summ["dispatch_meta"] = meta
fetch, write = summ["FETCH_SIZE"]["mean_per_launch"] * 1024.0, summ["WRITE_SIZE"]["mean_per_launch"] * 1024.0
summ["traffic_bytes_per_launch"] = {"fetch": fetch, "write": write, "total": fetch + write,
    "note": "FETCH_SIZE/WRITE_SIZE (KiB) x 1024, separate --pmc passes. 8 B/lane accesses: the gfx950 half-count correction of MI355X_MICROARCH.md is calibrated for 16 B/lane streams only, so the read side may be under-counted by up to 2x."}
waves = summ["SQ_WAVES"]["mean_per_launch"]; steps = bt["config"]["steps_per_launch"]
summ["per_wavefront_step"] = {"valu_insts": summ["SQ_INSTS_VALU"]["mean_per_launch"] / waves / steps, "salu_insts": summ["SQ_INSTS_SALU"]["mean_per_launch"] / waves / steps,
    "lds_insts": summ["SQ_INSTS_LDS"]["mean_per_launch"] / waves / steps,
    "valu_busy_frac": summ["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / summ["SQ_WAVE_CYCLES"]["mean_per_launch"],
    "wait_any_frac": summ["SQ_WAIT_ANY"]["mean_per_launch"] / summ["SQ_WAVE_CYCLES"]["mean_per_launch"],
    "lane_utilisation": summ["SQ_THREAD_CYCLES_VALU"]["mean_per_launch"] / (64.0 * summ["SQ_ACTIVE_INST_VALU"]["mean_per_launch"])}
json.dump(summ, open(os.path.join(P, "pmc_summary.json"), "w"), indent=1)
json.dump({"traffic_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "workload": "10000 x 8, %d steps/launch, 4 ranges" % bt["config"]["steps_per_launch"],
           "source": "profiles/r1/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"}, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print(json.dumps(agree, indent=1)); print(json.dumps(summ["per_wavefront_step"], indent=1)); print(fetch, write)
