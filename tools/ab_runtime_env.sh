# A/B of HIP / ROCr runtime environment settings on the default headline leg (five contexts): name=VALUE pairs, alternating with the default
run() { tag=$1; shift; name=$tag; env "$@" timeout -k 10 300 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu --no-extras > gpurun_out/r04_bu_$name.json 2> gpurun_out/r04_bu_$name.err || { echo "FAILED $name"; tail -5 gpurun_out/r04_bu_$name.err; return 1; }; python -c "
import json,sys
d=json.loads(open('gpurun_out/r04_bu_$name.json').read().strip().splitlines()[-1])
print('$name', round(d['value'],1), 'points/s')"; }
for rep in a b; do
run default_$rep REMO_NOP=1 && run no_interrupt_$rep HSA_ENABLE_INTERRUPT=0 && run dev_kernarg1_$rep HIP_FORCE_DEV_KERNARG=1 && run dev_kernarg0_$rep HIP_FORCE_DEV_KERNARG=0 && run queues6_$rep GPU_MAX_HW_QUEUES=6 || exit 1
done
