# A/B: the SAME 40 size-L batches (--total-depths 100) on ONE GPU as P processes (gloo ranks, REMO_DEVICE=0) x C contexts each (name pPxC);
# every process has its own HIP runtime (eight hardware queues each: remo3d_amd/__init__.py)
export REMO_DIST_BACKEND=gloo REMO_DEVICE=0
run() { p=$1; c=$2; tag=$3; name=p${p}x${c}_$tag; timeout -k 10 300 python bench.py --gpus $p --total-depths 100 --streams $c --steps 5 --warmup 2 --no-cpu --no-extras --vertex-solver amg_or_chebyshev > gpurun_out/r04_bq_$name.json 2> gpurun_out/r04_bq_$name.err || { echo "FAILED $name"; tail -5 gpurun_out/r04_bq_$name.err; return 1; }; python -c "
import json,sys
d=json.loads(open('gpurun_out/r04_bq_$name.json').read().strip().splitlines()[-1])
print('$name', round(d['value'],1), 'points/s', d['config'].get('vertex_block_solver'), d['config'].get('pcg_steps_per_batch'))"; }
for rep in a b; do
run 1 5 $rep && run 2 3 $rep && run 2 4 $rep && run 2 5 $rep && run 3 2 $rep || exit 1
done
