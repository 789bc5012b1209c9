cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2j
timeout 600 python -m pytest tests/test_gpu_dit.py -x -q -m gpu -k "attention or golden" 2>&1 | tail -8 > gpurun_out/r2j/tests.log
cat gpurun_out/r2j/tests.log
timeout 300 python tools/r2_probe.py oldcore=0:65536 persist=0:131072 > gpurun_out/r2j/probe.log 2>&1; grep -E "forward|attention|ln_mod" gpurun_out/r2j/probe.log
