#!/bin/bash
mkdir -p gpurun_out
timeout 240 python tools/pkfma_opsel_probe.py 30 > gpurun_out/pkfma_opsel_probe.txt 2>&1; tail -25 gpurun_out/pkfma_opsel_probe.txt
timeout 300 python tools/cosched_dump.py tools/variants/fixed/liblfm_hip.so 80 > gpurun_out/cosched_fixed.txt 2>&1; tail -2 gpurun_out/cosched_fixed.txt
for i in 1 2; do
  timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_fixed_$i.json 2>gpurun_out/bench_fixed_$i.err
  LFM_HIP_LIBRARY=tools/variants/noslp/liblfm_hip.so timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/bench_noslp_$i.json 2>gpurun_out/bench_noslp_$i.err
done
timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --in-flight 2 > gpurun_out/bench_fixed_if2.json 2>gpurun_out/bench_fixed_if2.err
LFM_HIP_LIBRARY=tools/variants/noslp/liblfm_hip.so timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --in-flight 2 > gpurun_out/bench_noslp_if2.json 2>gpurun_out/bench_noslp_if2.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],2), round(d['ms_per_step'],1), d.get('split_ms'), d['roofline']['avg_launch_us'] if 'roofline' in d else None)
    except Exception as e: print(f, 'ERR', e)
PY
