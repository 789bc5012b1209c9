cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout 600 python -m pytest tests/test_gpu_dit.py -x -q -m gpu -k "attention or golden or fullsize" 2>&1 | tail -8 > gpurun_out/r2h/tests.log
cat gpurun_out/r2h/tests.log
timeout 300 python tools/r2_probe.py old_attn=0:32768 > gpurun_out/r2h/probe.log 2>&1; grep -E "forward|attention|ln_mod" gpurun_out/r2h/probe.log
