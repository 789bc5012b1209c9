cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2p
timeout 900 python -m pytest tests/test_gpu_dit.py tests/test_gpu_vae.py tests/test_gpu_unet.py -x -q -m gpu -k "gemm256_kernels or qkv_split or every_gemm_kernel or vae_decode or building_blocks or golden" 2>&1 | tail -6 > gpurun_out/r2p/tests.log
cat gpurun_out/r2p/tests.log
timeout 300 python tools/epi_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/r2p/epi_probe.log
timeout 300 python tools/r2_probe.py v3=3:0 2>&1 | grep -vE "amdgpu|attention|ln_mod" | tee gpurun_out/r2p/probe.log
