cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2s
timeout 600 python -m pytest tests/test_gpu_dit.py -x -q -m gpu -k "epilogue or bench_gemm_shapes or gemm256_kernels or qkv" 2>&1 | tail -6 > gpurun_out/r2s/tests.log
cat gpurun_out/r2s/tests.log
timeout 120 tools/ubench/valu_rate 2>&1 | tee gpurun_out/r2s/valu_rate.log
timeout 300 python tools/epi_trace.py 2>&1 | grep -v amdgpu | tee gpurun_out/r2s/epi_trace.log
timeout 400 python tools/r2_probe.py nohoist=0:65536 noepi=5:4 2>&1 | grep -vE "amdgpu|attention|ln_mod" | tee gpurun_out/r2s/probe.log
