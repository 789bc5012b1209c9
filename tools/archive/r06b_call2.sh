cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/f2
LFM_HIP_LIBRARY=tools/_var/measure/liblfm_hip.so timeout 600 python tools/fused_qkv_phases.py DiT-L/2 64 > gpurun_out/f2/phases_L.log 2>&1
cat gpurun_out/f2/phases_L.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/f2/prof -o fwd -- python $GRAFT_REPO_ROOT/tools/fwd_probe.py 6 > $GRAFT_REPO_ROOT/gpurun_out/f2/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/f2/prof -name "*kernel_stats.csv" | head -1 | xargs head -8
