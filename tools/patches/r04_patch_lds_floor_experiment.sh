# A/B (experiment build only: REMO_EXPERIMENT_PATCH_LDS_FLOOR in patch.hip's launcher): fewer workgroups of the patch kernel per CU
# (LDS floor 41000 -> 3 per CU, 54000 -> 2) so that waves of the streaming launches of other contexts can sit beside them
run() { f=$1; c=$2; tag=$3; name=floor${f}_ctx${c}_$tag; REMO_EXPERIMENT_PATCH_LDS_FLOOR=$f timeout -k 10 300 python bench.py --streams $c --gpus 1 --steps 5 --warmup 2 --no-cpu --no-extras > gpurun_out/r04_bo_$name.json 2> gpurun_out/r04_bo_$name.err || { echo "FAILED $name"; tail -5 gpurun_out/r04_bo_$name.err; return 1; }; python -c "
import json,sys
d=json.loads(open('gpurun_out/r04_bo_$name.json').read().strip().splitlines()[-1])
r=d.get('roofline',{}).get('in_timed_region') or {}
print('$name', round(d['value'],1), 'points/s; apply us in the timed region', r.get('avg_launch_us'))"; }
for rep in a b; do
run 0 5 $rep && run 41000 5 $rep && run 54000 5 $rep && run 0 1 $rep && run 41000 1 $rep || exit 1
done
