cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 900 python -m pytest tests/test_gpu_dit.py tests/test_gpu_unet.py tests/test_gpu_edm.py -x -q -m gpu -k "attention or groupnorm or split_k or unet or edm or golden" 2>&1 | tail -15 > gpurun_out/r2g/tests.log
cat gpurun_out/r2g/tests.log
timeout 300 python tools/r2_probe.py old_attn=0:32768 > gpurun_out/r2g/probe.log 2>&1; grep -E "forward|attention|ln_mod" gpurun_out/r2g/probe.log
timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/r2g/bench2.json 2>/dev/null; cut -c1-400 gpurun_out/r2g/bench2.json
timeout 300 python bench.py --config 5 --steps 2 --warmup 1 > gpurun_out/r2g/bench5.json 2>/dev/null; cut -c1-1400 gpurun_out/r2g/bench5.json
