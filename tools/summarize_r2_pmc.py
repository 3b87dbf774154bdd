"""Fold gpurun_out/r2pmc (tools/collect_r2_pmc.sh) into profiles/r2/: kernel_stats.csv, launch_agreement.json,
pmc_summary.json, traffic.json (HBM bytes per zone-step), pmc_fp64.json (fp64 flop per zone-step), bench_*.json."""
import collections, csv, glob, json, os, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(R, "gpurun_out", "r2pmc"); P = os.path.join(R, "profiles", "r2")
os.makedirs(P, exist_ok=True)
K = "step_kernel"
shutil.copy(os.path.join(O, "trace", "plain_kernel_stats.csv"), os.path.join(P, "kernel_stats.csv"))
if os.path.exists(os.path.join(O, "trace_io", "plain_kernel_stats.csv")):
    shutil.copy(os.path.join(O, "trace_io", "plain_kernel_stats.csv"), os.path.join(P, "kernel_stats_plantio_scan1.csv"))
for f in glob.glob(os.path.join(O, "bench_*.json")):
    if os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(P, os.path.basename(f)))
rows = [r for r in csv.DictReader(open(os.path.join(O, "trace", "plain_kernel_trace.csv"))) if K in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
bt = json.load(open(os.path.join(O, "bench_under_rocprof.json")))
nl = bt["roofline"]["launches"]
agree = {"source": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --steps 500 --warmup 100 (profiles/r2/kernel_stats.csv)",
         "step_kernel_dispatches": len(d), "durations_us": d, "timed_dispatch_us_rocprof": sum(d[-nl:]) / nl,
         "bench_avg_launch_us_same_run": bt["roofline"]["avg_launch_us"], "steps_per_timed_dispatch": bt["steps"] / nl}
json.dump(agree, open(os.path.join(P, "launch_agreement.json"), "w"), indent=1)
summ = {}; meta = None
for f in sorted(glob.glob(os.path.join(O, "pmc*", "p_counter_collection.csv"))):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if K in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
            meta = {k: row[k] for k in ("Kernel_Name", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size") if k in row}
    for k, v in acc.items():
        summ[k] = {"dispatches": len(v), "timed_launch": v[-1], "warmup_launch": v[0]}
summ["dispatch_meta"] = meta
N, n, steps = bt["config"]["reactors_per_gpu"], bt["config"]["zones"], bt["steps"]
zs = N * n * steps                     # zone-steps of the timed launch
groups = -(-N // (64 // n))
T = lambda k: summ[k]["timed_launch"]
fetch, write = T("FETCH_SIZE") * 1024.0, T("WRITE_SIZE") * 1024.0
# MI355X_MICROARCH.md, HBM section: FETCH_SIZE counts 64 B per 128-B request on gfx950 -> doubled; WRITE_SIZE is exact
hbm = 2.0 * fetch + write
json.dump({"hbm_bytes_per_zone_step": hbm / zs, "fetch_size_bytes_raw": fetch, "write_size_bytes": write, "zone_steps_profiled": zs,
           "workload": f"{N} x {n}, one queue-schedule launch of {steps} outer steps",
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, KiB x 1024), timed launch; read side doubled per the gfx950 correction of MI355X_MICROARCH.md"},
          open(os.path.join(P, "traffic.json"), "w"), indent=1)
flop = (T("SQ_INSTS_VALU_ADD_F64") + T("SQ_INSTS_VALU_MUL_F64") + 2.0 * T("SQ_INSTS_VALU_FMA_F64") + T("SQ_INSTS_VALU_TRANS_F64")) * 64.0
json.dump({"zones": n, "fp64_flop_per_zone_step": flop / zs, "zone_steps_profiled": zs,
           "wave_instructions": {k: T(k) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")},
           "source": "rocprofv3 --pmc SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 on the timed launch; flop = (ADD + MUL + 2 FMA + TRANS) x 64 lanes (inactive lanes included)"},
          open(os.path.join(P, "pmc_fp64.json"), "w"), indent=1)
gs = groups * steps
summ["per_group_step"] = {
    "valu_insts": T("SQ_INSTS_VALU") / gs, "salu_insts": T("SQ_INSTS_SALU") / gs, "lds_insts": T("SQ_INSTS_LDS") / gs,
    "fp64_valu_insts": (T("SQ_INSTS_VALU_ADD_F64") + T("SQ_INSTS_VALU_MUL_F64") + T("SQ_INSTS_VALU_FMA_F64") + T("SQ_INSTS_VALU_TRANS_F64")) / gs,
    "wave_cycles": 4.0 * T("SQ_WAVE_CYCLES") / gs,
    "valu_busy_frac": T("SQ_ACTIVE_INST_VALU") / T("SQ_WAVE_CYCLES"), "wait_any_frac": T("SQ_WAIT_ANY") / T("SQ_WAVE_CYCLES"),
    "lane_utilisation": T("SQ_THREAD_CYCLES_VALU") / (64.0 * T("SQ_ACTIVE_INST_VALU")) if "SQ_THREAD_CYCLES_VALU" in summ else None}
json.dump(summ, open(os.path.join(P, "pmc_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in agree.items() if k != "durations_us"}, indent=1)); print(json.dumps(summ["per_group_step"], indent=1)); print("hbm B/zone-step", hbm / zs, "flop/zone-step", flop / zs)
