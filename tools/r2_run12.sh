cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2l
timeout 600 python -m pytest tests/test_gpu_dit.py -x -q -m gpu -k "ln_modulate or golden" 2>&1 | tail -6 > gpurun_out/r2l/tests.log
cat gpurun_out/r2l/tests.log
timeout 300 python tools/r2_probe.py ln_old=0:524288 > gpurun_out/r2l/probe.log 2>&1; grep -E "forward|ln_mod" gpurun_out/r2l/probe.log
