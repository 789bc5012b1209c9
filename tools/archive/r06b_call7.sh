R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/f7
timeout 600 python tools/fused_qkv_ab.py DiT-L/2 64 20 8388608 > gpurun_out/f7/ab_prio.log 2>&1; cat gpurun_out/f7/ab_prio.log
timeout 1500 python -m pytest tests/test_gpu_dit.py tests/test_gpu_cosched.py -x -q 2>&1 | tail -8 > gpurun_out/f7/test.log
cat gpurun_out/f7/test.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/f7/bench.log 2>&1; tail -1 gpurun_out/f7/bench.log
timeout 600 python bench.py --no-cpu-baseline --in-flight 1 > gpurun_out/f7/bench1.log 2>&1; tail -1 gpurun_out/f7/bench1.log
