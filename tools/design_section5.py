"""Regenerate the figures of DESIGN.md section 5 from profiles/r3/ (developer tool): prints the table rows and the key
numbers quoted in the text, so that the section can be refreshed after a new collection pass.
   python tools/design_section5.py"""
import json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(R, "profiles", "r3")
B = lambda n: json.load(open(os.path.join(P, f"bench_{n}.json")))
J = lambda n: json.load(open(os.path.join(P, n)))
v = lambda b: b["value"] / 1e9
us = lambda b: b["ms_per_step"] * 1e3
d, ds = B("default"), B("driver_style")
for name in ("default", "identity_placement", "driver_style", "sensors", "plantio_scan50", "plantio_scan1", "12500", "sensors_12500",
             "n4", "n16", "n20", "n20_driver_style", "100k"):
    b = B(name)
    print(f"{name:22s} {v(b):.3f}e9  {us(b):7.1f} us/step  hbm frac {100 * b['roofline']['frac']:.2f} %  traffic {b['roofline']['traffic']}  "
          f"redeals {b['config']['schedule'].get('redeals')}")
g = J("pmc_summary.json")["per_group_step"]; g20 = J("pmc_n20.json")["per_group_step"]
print("n = 8 per group-step:", {k: round(x, 3) for k, x in g.items()})
print("n = 20 per group-step:", {k: round(x, 3) for k, x in g20.items()})
print("traffic B/zone-step:", {k: round(e["hbm_bytes_per_zone_step"], 2) for k, e in J("traffic.json")["by_steps_per_item"].items()},
      "n20", round(J("pmc_n20.json")["traffic"]["hbm_bytes_per_zone_step"], 2))
print("fp64 flop/zone-step n8", round(J("pmc_fp64.json")["fp64_flop_per_zone_step"]), "n20", round(J("pmc_n20.json")["fp64_flop_per_zone_step"]))
la = J("launch_agreement.json")
print("launch us: rocprof", la["timed_dispatch_us_rocprof"], "bench", la["bench_avg_launch_us_same_run"])
print("compute roofline: default", d["roofline_compute"]["achieved"], d["roofline_compute"]["frac"], "driver", ds["roofline_compute"]["frac"])
print("cpu baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["one_core"]["value"], "dropin ms", d["dropin_n1"]["ms_per_step"], ds["dropin_n1"]["ms_per_step"])
