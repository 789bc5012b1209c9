R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/f4
timeout 900 python -m pytest tests/test_gpu_dit.py -x -q -k "fused_qkv" 2>&1 | tail -15 > gpurun_out/f4/test.log
cat gpurun_out/f4/test.log
timeout 600 python tools/fused_qkv_ab.py DiT-L/2 64 20 > gpurun_out/f4/ab_L.log 2>&1; cat gpurun_out/f4/ab_L.log
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/f4
export LFM_HIP_LIBRARY=$R/tools/_var/measure/liblfm_hip.so
for mode in 0 4194304 33554432 67108864 100663296; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$mode -o b -- python $R/tools/fused_qkv_phases.py $mode > $O/s$mode.log 2>&1
  echo "== flags $mode"; grep -h "qkv_attention\|EpiGateResidMod\|EpiModGelu" $(find $O/s$mode -name "*kernel_stats.csv" | head -1) | cut -c1-200
done > $O/summary.txt
cat $O/summary.txt
rm -rf $O/s*
