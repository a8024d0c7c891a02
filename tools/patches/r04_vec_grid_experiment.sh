# A/B (experiment build only: REMO_EXPERIMENT_VEC_GRID in kernels.hip's vec_grid): workgroups of the PCG's vector launches (product: 512 = two per CU)
run() { g=$1; c=$2; tag=$3; name=grid${g}_ctx${c}_$tag; REMO_EXPERIMENT_VEC_GRID=$g timeout -k 10 300 python bench.py --streams $c --gpus 1 --steps 5 --warmup 2 --no-cpu --no-extras > gpurun_out/r04_bs_$name.json 2> gpurun_out/r04_bs_$name.err || { echo "FAILED $name"; tail -5 gpurun_out/r04_bs_$name.err; return 1; }; python -c "
import json,sys
d=json.loads(open('gpurun_out/r04_bs_$name.json').read().strip().splitlines()[-1])
print('$name', round(d['value'],1), 'points/s', d['config'].get('pcg_steps_per_batch'))"; }
for rep in a b; do
run 512 5 $rep && run 256 5 $rep && run 384 5 $rep && run 768 5 $rep && run 1024 5 $rep && run 512 1 $rep && run 1024 1 $rep && run 256 1 $rep || exit 1
done
