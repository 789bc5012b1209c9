#!/bin/bash
mkdir -p gpurun_out
T='tests/test_gpu_dit.py::test_folded_ln_epilogues_match_separate_launches_and_the_oracle'
for v in product slp_fma noslp_packed; do
  for i in 1 2; do
    if [ $v = product ]; then timeout 300 python -m pytest "$T" -q 2>&1 | tail -3 > gpurun_out/v6rep_${v}_$i.txt
    else LFM_HIP_LIBRARY=tools/variants/$v/liblfm_hip.so timeout 300 python -m pytest "$T" -q 2>&1 | tail -3 > gpurun_out/v6rep_${v}_$i.txt; fi
    echo "$v $i: $(tail -1 gpurun_out/v6rep_${v}_$i.txt)"
  done
done
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_all.txt 2>&1; tail -8 gpurun_out/pytest_gpu_all.txt
run() { name=$1; shift; timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/b4_$name.json 2>gpurun_out/b4_$name.err; }
run c6_if1 --config 6 --in-flight 1
run c6_if2 --config 6 --in-flight 2
run c5_if1 --config 5 --in-flight 1
run c5_if2 --config 5 --in-flight 2
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b4_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d['value'],2), round(d['ms_per_step'],1), d['config'].get('batches_in_flight'), round(d['mfma_frac_whole_path'],4), d['split_ms'])
    except Exception as e: print(f, 'ERR', e, open(f.replace('.json','.err')).read()[-800:])
PY
