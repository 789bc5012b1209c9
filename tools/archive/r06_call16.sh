#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c16; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_edm.py tests/test_gpu_cosched.py -q -m gpu -x -k "edm or unet" 2>&1 | tail -n 4 | tee gpurun_out/c16/pytest.txt
timeout 300 python bench.py --config 6 --steps 4 --warmup 2 --in-flight 1 --no-roofline 2>/dev/null | grep '^{' > gpurun_out/c16/config6_one_lane.json; python -c "
import json; d=json.load(open('gpurun_out/c16/config6_one_lane.json')); print('config 6 one lane', d['value'], d['mfma_frac_whole_path'], d['clock_mhz_under_mfma_load'])"
timeout 300 python bench.py --config 6 --steps 4 --warmup 2 2>/dev/null | grep '^{' > gpurun_out/c16/config6.json; python -c "
import json; d=json.load(open('gpurun_out/c16/config6.json')); print('config 6 two lanes', d['value'], d['mfma_frac_whole_path'], d['clock_mhz_under_mfma_load'])"
