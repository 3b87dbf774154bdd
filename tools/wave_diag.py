"""Per-wavefront trip / clock distribution of one launch (developer tool)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
wt = importlib.import_module("ics-wt-physicsengine_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
cols, bc = wt.make_ensemble(N)
ens = wt.ReactorEnsemble(cols, n_zones=n); ens.set_boundary(bc)
sens = len(sys.argv) > 3 and sys.argv[3] == "sensors"
if sens:
    ens.enable_sensors(seed=1)
ens.step(1.0, n_steps=100, fused=True, download=False); ens.synchronize()
ens.wave_diag()
for label, k, fused in (("queue 20 steps", 20, True), ("queue 100 steps", 100, True), ("stream(1) 50 steps", 50, True)):
    ens.set_schedule(1 if label.startswith("stream") else 0, 0 if label.startswith("stream") else 50)
    ens.timer_start(); ens.step(1.0, n_steps=k, fused=fused, download=False); ms = ens.timer_stop()
    d = ens.wave_diag()
    st = ens.solver_stats()
    clk = d[:, 2] / k; wall = d[:, 3] / 100.0 / k  # us
    print(f"{label}: launch {ms*1e3/k:.1f} us/step | trips/step mean {d[:,0].mean()/k:.1f} max {d[:,0].max()/k:.1f} | newton trips mean {d[:,1].mean()/k:.1f}"
          f" | wave us/step mean {wall.mean():.1f} p50 {np.median(wall):.1f} p90 {np.percentile(wall,90):.1f} p99 {np.percentile(wall,99):.1f} max {wall.max():.1f}"
          f" | fact {d[:,4].mean()/k:.2f} jac {d[:,5].mean()/k:.2f} f3 {d[:,6].mean()/k:.2f}"
          f" | clk/wall GHz {np.mean(clk/wall)/1e3:.2f} | reactor nfev mean {st[:,0].mean():.1f} max {st[:,0].max()}")
    if d.shape[1] > 8:
        names = ["setup", "prologue", "factorize", "rhs", "epilogue", "num_jac", "post-step", "sensors/io"]
        tot = d[:, 8:16].sum()
        print("   section shares: " + "  ".join(f"{nm} {d[:, 8 + i].sum() / tot * 100:.1f}%" for i, nm in enumerate(names) if nm != "-")
              + f" | stamped clocks/step {d[:, 8:16].sum(1).mean() / k:.0f}")
    print(f"   items per group {d[:,7].mean():.2f}")
    if k == 1:
        srt = np.sort(wall)[::-1][:8]; print("   slowest waves us:", np.round(srt, 1), "trips:", np.sort(d[:, 0])[::-1][:8])
