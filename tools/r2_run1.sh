set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 300 python tools/r2_probe.py store8B=1024 > gpurun_out/r2a/probe.log 2>&1
tail -30 gpurun_out/r2a/probe.log
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_dit.py -x -q -m gpu -s 2>&1 | tail -25 > gpurun_out/r2a/tests_new.log
cat gpurun_out/r2a/tests_new.log
timeout 300 python bench.py --steps 5 --warmup 2 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err
cat gpurun_out/r2a/bench.json
