R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/f18
timeout 900 python -m pytest tests/test_gpu_dit.py -x -q -k "fused_qkv" 2>&1 | tail -3 > gpurun_out/f18/test.log; cat gpurun_out/f18/test.log
for rep in 1 2 3; do
echo "== new"; timeout 600 python tools/fused_qkv_ab.py DiT-L/2 64 20 2>&1 | grep "fused=1\|bit" | cut -c1-90
echo "== old"; LFM_HIP_LIBRARY=$R/tools/_var/old/liblfm_hip.so timeout 600 python tools/fused_qkv_ab.py DiT-L/2 64 20 2>&1 | grep "fused=1" | cut -c1-90
done > gpurun_out/f18/ab.log 2>&1; cat gpurun_out/f18/ab.log
