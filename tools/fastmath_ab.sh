#!/bin/bash
# A/B of -ffast-math on the parity library: bench line + parity suite with the shipped flags (no -ffast-math since round 4), then the library rebuilt WITH
# -ffast-math on the same box (LFM_FAST_MATH=1), same commands.  usage (on the GPU box): tools/fastmath_ab.sh <tag>
TAG=${1:-r4fm}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
run() {  # label
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/bench_$1.json
  timeout 900 python -m pytest tests/test_gpu_dit.py tests/test_gpu_vae.py tests/test_gpu_unet.py tests/test_gpu_sampler.py -q -m gpu -x 2>&1 | tail -3 > $O/tests_$1.txt
  timeout 200 python tools/fastmath_err.py > $O/err_$1.txt 2>&1
}
run nofast_default
export LFM_FAST_MATH=1
python -m lfm_amd._build --force > $O/build_fast.log 2>&1
run fast
for l in nofast_default fast; do echo "== $l"; python -c "import json;j=json.load(open('$O/bench_$l.json'));print(j['value'], j['split_ms'], j['roofline']['avg_launch_us'], j.get('clock_mhz_under_mfma_load'))"; cat $O/tests_$l.txt; cat $O/err_$l.txt; done
