cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2k
timeout 900 python -m pytest tests/test_gpu_dit.py tests/test_gpu_unet.py tests/test_gpu_sampler.py -x -q -m gpu -k "patch4 or options or golden or fused_euler or config1 or fullsize" 2>&1 | tail -12 > gpurun_out/r2k/tests.log
cat gpurun_out/r2k/tests.log
