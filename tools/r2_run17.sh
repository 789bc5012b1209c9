cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2r
timeout 600 python -m pytest tests/test_gpu_dit.py -x -q -m gpu -k "ln_modulate or epilogue or bench_gemm_shapes or gemm256_kernels" 2>&1 | tail -6 > gpurun_out/r2r/tests.log
cat gpurun_out/r2r/tests.log
timeout 300 python tools/epi_trace.py 2>&1 | grep -v amdgpu | tee gpurun_out/r2r/epi_trace.log
timeout 400 python tools/r2_probe.py noepi=5:4 ln4=0:32768 2>&1 | grep -vE "amdgpu" | tee gpurun_out/r2r/probe.log
