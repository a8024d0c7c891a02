#!/bin/bash
# A/B of one remo_debug_tune key inside one gpurun call (same box, alternating): usage tools/ab_bench.sh OUT_PREFIX "KEY=VALUE" [bench args...]
# prints points/s, SpMM average launch time and roofline fraction of both arms
OUT=$1; TUNE=$2; shift 2
for rep in 1 2; do
  python bench.py --no-cpu --no-extras "$@" --tune "$TUNE" > ${OUT}_tuned_$rep.json 2>> ${OUT}.err
  python bench.py --no-cpu --no-extras "$@" > ${OUT}_default_$rep.json 2>> ${OUT}.err
done
python - "$OUT" <<'PY'
import json, sys, glob
for arm in ("tuned", "default"):
    for f in sorted(glob.glob(sys.argv[1] + "_%s_*.json" % arm)):
        d = json.load(open(f)); r = d["roofline"]
        print("%-8s %s  %.1f points/s  solve %.1f ms  SpMM %.2f us  frac %.4f  steps %d" % (arm, f.split("_")[-1][:1], d["value"], d["breakdown_ms_per_step"]["solve"], r["avg_launch_us"], r["frac"], d["breakdown_ms_per_step"]["pcg_steps"]))
PY
