export REMO_DIST_BACKEND=gloo REMO_DEVICE=0
run() { name=$1; shift; timeout -k 10 300 python bench.py "$@" --no-cpu --no-extras > gpurun_out/r04_bk_$name.json 2> gpurun_out/r04_bk_$name.err || { echo "FAILED $name"; tail -5 gpurun_out/r04_bk_$name.err; return 1; }; python -c "
import json,sys
d=json.loads(open('gpurun_out/r04_bk_$name.json').read().strip().splitlines()[-1])
print('$name', round(d['value'],1), 'points/s', d['config'].get('vertex_block_solver'), d['config'].get('pcg_steps_per_batch'))"; }
run p3x1_cycle --gpus 3 --total-depths 100 --streams 1 --steps 5 --warmup 2 --vertex-solver amg_or_chebyshev &&
run p3x1_cheb --gpus 3 --total-depths 100 --streams 1 --steps 5 --warmup 2 &&
run p2x2_cycle --gpus 2 --total-depths 100 --streams 2 --steps 5 --warmup 2 &&
run p4x1_cycle --gpus 4 --total-depths 100 --streams 1 --steps 5 --warmup 2 --vertex-solver amg_or_chebyshev &&
run p1x3_dyn --gpus 1 --streams 3 --steps 5 --warmup 2 &&
REMO_BENCH_CTX_DRAW=static run p1x3_static --gpus 1 --streams 3 --steps 5 --warmup 2
