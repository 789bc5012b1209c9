#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest "tests/test_gpu_dit.py::test_folded_ln_epilogues_match_separate_launches_and_the_oracle" tests/test_gpu_cosched.py tests/test_gpu_unet.py tests/test_gpu_edm.py -q > gpurun_out/pytest_call5.txt 2>&1; tail -6 gpurun_out/pytest_call5.txt
timeout 200 python tools/splitk256_probe.py 20 > gpurun_out/splitk256_probe.txt 2>&1; cat gpurun_out/splitk256_probe.txt | tail -12
run() { name=$1; shift; timeout 400 python bench.py --no-cpu-baseline "$@" > gpurun_out/b5_$name.json 2>gpurun_out/b5_$name.err; }
run c6 --config 6 --steps 4 --warmup 1
run c5 --config 5 --steps 4 --warmup 1
run c2 --steps 6 --warmup 2
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b5_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],2), round(d['ms_per_step'],1), d['config'].get('batches_in_flight'), round(d['mfma_frac_whole_path'],4), d['split_ms'], d.get('roofline_block'))
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
