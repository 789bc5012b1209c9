set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f1
timeout 900 python -m pytest tests/test_gpu_dit.py -x -q -k "fused_qkv" 2>&1 | tail -15 > gpurun_out/f1/test.log
timeout 600 python tools/fused_qkv_ab.py DiT-L/2 64 20 > gpurun_out/f1/ab_L.log 2>&1
timeout 600 python tools/fused_qkv_ab.py DiT-B/2 512 5 > gpurun_out/f1/ab_B.log 2>&1
cat gpurun_out/f1/*.log
