set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2b
timeout 600 python -m pytest tests/test_gpu_dit.py -x -q -m gpu -k "gemm256_kernels or detects_transpose or qkv_split or every_gemm_kernel" 2>&1 | tail -15 > gpurun_out/r2b/tests_v4.log
cat gpurun_out/r2b/tests_v4.log
timeout 400 python tools/r2_probe.py v3=3:0 v4=4:0 v4prio=4:2048 v4setprio=4:8 v4all=0:8192 > gpurun_out/r2b/probe.log 2>&1
cat gpurun_out/r2b/probe.log
timeout 300 python -m pytest tests/test_gpu_vae.py tests/test_gpu_configs.py -x -q -m gpu -k "vae or dopri5" 2>&1 | tail -8
