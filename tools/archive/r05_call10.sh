#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/cosched_dump.py tools/variants/fixed_pk/liblfm_hip.so 100 > gpurun_out/cosched_fixed_pk.txt 2>&1; tail -2 gpurun_out/cosched_fixed_pk.txt
timeout 600 python -m pytest tests/test_gpu_cosched.py "tests/test_gpu_dit.py::test_folded_ln_epilogues_match_separate_launches_and_the_oracle" tests/test_gpu_configs.py::test_dit_l2_batch64_auto_dispatch_vs_oracle -q > gpurun_out/pytest_call10.txt 2>&1; tail -3 gpurun_out/pytest_call10.txt
for i in 1 2; do
  timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --in-flight 1 > gpurun_out/b10_pk_$i.json 2>/dev/null
  LFM_HIP_LIBRARY=tools/variants/scalar/liblfm_hip.so timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --in-flight 1 > gpurun_out/b10_scalar_$i.json 2>/dev/null
done
timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/b10_pk_if2.json 2>/dev/null
LFM_HIP_LIBRARY=tools/variants/scalar/liblfm_hip.so timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/b10_scalar_if2.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b10_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],2), round(d['ms_per_step'],1), d['config'].get('batches_in_flight'), 'fc1', round(d['roofline']['avg_launch_us'],2), 'block', round(d['roofline_block']['block_us'],1), 'two', round(d['roofline_block']['two_lanes']['block_us_upper_bound'],1), d['clock_mhz_under_mfma_load'])
    except Exception as e: print(f, 'ERR', e)
PY
