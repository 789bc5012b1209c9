R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/f17
for rep in 1 2 3; do
echo "== default"; timeout 600 python tools/fused_qkv_ab.py DiT-L/2 64 20 2>&1 | grep "fused=1\|bit" | cut -c1-90
echo "== OPT 5"; LFM_HIP_LIBRARY=$R/tools/_var/opt5/liblfm_hip.so timeout 600 python tools/fused_qkv_ab.py DiT-L/2 64 20 2>&1 | grep "fused=1\|bit" | cut -c1-90
done > gpurun_out/f17/ab.log 2>&1; cat gpurun_out/f17/ab.log
