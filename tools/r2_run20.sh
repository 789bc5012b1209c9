cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2u
timeout 300 python -m pytest tests/test_gpu_dit.py -x -q -m gpu -k "epilogue or qkv or golden or bench_gemm_shapes or gemm256_kernels" 2>&1 | tail -6 > gpurun_out/r2u/tests.log
cat gpurun_out/r2u/tests.log
timeout 200 python tools/epi_trace.py 2>&1 | grep -v amdgpu | grep -E "rep 1" | grep -E "GELU|qkv" | tee gpurun_out/r2u/epi_trace.log
timeout 300 python tools/r2_probe.py nopipe=0:262144 2>&1 | grep -vE "amdgpu|attention|ln_modulate" | tee gpurun_out/r2u/probe.log
