#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; tail -5 gpurun_out/pytest_gpu.txt
run() { name=$1; shift; timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > gpurun_out/b3_$name.json 2>gpurun_out/b3_$name.err; }
run default
run if1 --in-flight 1
run if3 --in-flight 3
run b32_if2 --batch 32 --in-flight 2
run b32_if4 --batch 32 --in-flight 4
run if1_again --in-flight 1
run default_again
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b3_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],2), round(d['ms_per_step'],1), d['config'].get('batches_in_flight'), d['config']['per_gpu_batch'], d.get('roofline_block',{}).get('block_us'), d.get('roofline_block',{}).get('frac'), d['roofline']['avg_launch_us'], d['clock_mhz_under_mfma_load'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-600:])
PY
