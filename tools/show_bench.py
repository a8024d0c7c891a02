#!/usr/bin/env python3
"""Condensed view of a bench.py JSON line: python tools/show_bench.py FILE"""
import json, sys
r = json.load(open(sys.argv[1]))
rf = r["roofline"]
print("value %.1f %s  ms/step %.1f  operator %s  frac %.3f  apply %.1f us  algorithmic MB %.1f" % (
    r["value"], r["unit"], r["ms_per_step"], r["config"]["operator"], rf["frac"] or 0, rf["avg_launch_us"] or 0, (rf.get("algorithmic_bytes_per_launch") or 0) / 1e6))
print("breakdown", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in r["breakdown_ms_per_step"].items()})
print("config", {k: r["config"][k] for k in ("mesh_T", "n_free", "nnz", "max_pcg_iterations", "nan_points") if k in r["config"]}, "oracle diff", r["config"].get("gpu_vs_oracle_max_rel_diff_batch0_rhs0"))
for k in ("value_two_contexts", "value_h2d_inclusive"):
    if k in r:
        print(k, round(r[k]["value"], 1))
cb = r.get("cpu_baseline")
if cb:
    print("cpu_baseline", cb.get("value"), cb.get("cores"), cb.get("seconds"), cb.get("pcg_iterations"))
for s in r.get("sizes", []):
    print("  %-18s %-8s %-7s %7.1f points/s  steps/batch %6.1f  apply %7.1f us  solve %7.1f ms/batch  frac %s" % (
        s["workload"], s["precision"], s["operator"], s["value"], s["pcg_steps_per_batch"], s["apply_avg_launch_us"] or 0, s["solve_ms_per_batch"],
        ("%.3f" % s["operator_frac_of_hbm_peak"]) if s.get("operator_frac_of_hbm_peak") else "-"))
