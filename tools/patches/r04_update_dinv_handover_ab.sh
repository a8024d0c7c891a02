run() { tag=$1; lib=$2; c=$3; name=${tag}_ctx$c; REMO_LIB=$lib timeout -k 10 300 python bench.py --gpus 1 --streams $c --steps 5 --warmup 2 --no-cpu --no-extras > gpurun_out/r04_bv_$name.json 2> gpurun_out/r04_bv_$name.err || { echo "FAILED $name"; tail -5 gpurun_out/r04_bv_$name.err; return 1; }; python -c "
import json,sys
d=json.loads(open('gpurun_out/r04_bv_$name.json').read().strip().splitlines()[-1])
print('$name', round(d['value'],2), 'points/s', d['breakdown_ms_per_step']['solve'] if 'breakdown_ms_per_step' in d else '')"; }
OLD=$PWD/remo3d_amd/libremo3d_hip.so; NEW=$PWD/remo3d_amd/libremo3d_hip_exp.so
REMO_LIB=$NEW timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "headline or full_size or patch_operator_solves or assembly_and_solve" > gpurun_out/r04_bv_pytest_new_lib.log 2>&1; tail -2 gpurun_out/r04_bv_pytest_new_lib.log
for rep in a b c; do
run old_$rep $OLD 5 && run new_$rep $NEW 5 && run old_$rep $OLD 1 && run new_$rep $NEW 1 || exit 1
done
