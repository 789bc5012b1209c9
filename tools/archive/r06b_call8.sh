R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/f8
export LFM_HIP_LIBRARY=$R/tools/_var/measure/liblfm_hip.so
for f in 0 262144 524288 786432; do echo "== flags $f"; timeout 300 python tools/fused_qkv_trace.py $f 2>&1 | sed -n 9,16p; done > gpurun_out/f8/handover.log
cat gpurun_out/f8/handover.log
