# A/B: vertex-block solver (multigrid cycle / Chebyshev polynomial) against contexts per GPU on eight hardware queues
run() { v=$1; c=$2; tag=$3; name=${v}_ctx${c}_$tag; timeout -k 10 300 python bench.py --vertex-solver $v --streams $c --gpus 1 --steps 5 --warmup 2 --no-cpu --no-extras > gpurun_out/r04_bp_$name.json 2> gpurun_out/r04_bp_$name.err || { echo "FAILED $name"; tail -5 gpurun_out/r04_bp_$name.err; return 1; }; python -c "
import json,sys
d=json.loads(open('gpurun_out/r04_bp_$name.json').read().strip().splitlines()[-1])
print('$name', round(d['value'],1), 'points/s', d['config'].get('vertex_block_solver'), d['config'].get('pcg_steps_per_batch'))"; }
for rep in a b; do
run amg_or_chebyshev 5 $rep && run chebyshev 5 $rep && run chebyshev 6 $rep && run chebyshev 8 $rep && run amg_or_chebyshev 7 $rep || exit 1
done
