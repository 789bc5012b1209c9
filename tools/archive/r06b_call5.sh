R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/f5
LFM_HIP_LIBRARY=$R/tools/_var/measure/liblfm_hip.so timeout 300 python tools/fused_qkv_trace.py > gpurun_out/f5/trace.log 2>&1
LFM_HIP_LIBRARY=$R/tools/_var/measure/liblfm_hip.so timeout 300 python tools/fused_qkv_trace.py 4194304 > gpurun_out/f5/trace_per_item.log 2>&1
cat gpurun_out/f5/trace.log; head -12 gpurun_out/f5/trace_per_item.log
