# A/B: contexts per GPU (threads of one process) against the number of hardware queues the HIP runtime may use (GPU_MAX_HW_QUEUES, default 4)
# usage: bash tools/ab_hw_queues.sh   (on the GPU box; writes gpurun_out/r04_bl_*.json, prints one line per run)
run() { q=$1; c=$2; tag=$3; name=q${q}_ctx${c}_$tag; GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --streams $c --gpus 1 --steps 5 --warmup 2 --no-cpu --no-extras > gpurun_out/r04_bl_$name.json 2> gpurun_out/r04_bl_$name.err || { echo "FAILED $name"; tail -5 gpurun_out/r04_bl_$name.err; return 1; }; python -c "
import json,sys
d=json.loads(open('gpurun_out/r04_bl_$name.json').read().strip().splitlines()[-1])
print('$name', round(d['value'],1), 'points/s', d['config'].get('vertex_block_solver'), d['config'].get('pcg_steps_per_batch'))"; }
for rep in a b; do
run 4 3 $rep && run 8 5 $rep && run 4 5 $rep && run 8 6 $rep && run 12 6 $rep && run 12 8 $rep && run 8 4 $rep || exit 1
done
