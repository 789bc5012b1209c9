cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2m
timeout 900 python -m pytest tests/test_gpu_dit.py tests/test_gpu_configs.py -x -q -m gpu -k "gemm256_kernels or detects_transpose or qkv_split or every_gemm_kernel or bench_gemm_shapes or batch64" 2>&1 | tail -12 > gpurun_out/r2m/tests.log
cat gpurun_out/r2m/tests.log
timeout 400 python tools/r2_probe.py v3=3:0 v5=5:0 v5noepi=5:4 v3noepi=3:4 > gpurun_out/r2m/probe.log 2>&1; grep -vE "attention|ln_mod|amdgpu" gpurun_out/r2m/probe.log
timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r2m/bench2.json 2>/dev/null; cut -c1-200 gpurun_out/r2m/bench2.json; python -c "
import json; d=json.load(open('gpurun_out/r2m/bench2.json')); print(d['value'], d['split_ms'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
