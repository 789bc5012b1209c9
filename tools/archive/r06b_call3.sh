R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/f3; mkdir -p $O
export LFM_HIP_LIBRARY=$R/tools/_var/measure/liblfm_hip.so
for mode in 0 33554432 67108864 100663296; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$mode -o b -- python $R/tools/fused_qkv_phases.py $mode > $O/s$mode.log 2>&1
  echo "== flags $mode"; grep -h "qkv_attention\|EpiGateResidMod\|EpiModGelu" $(find $O/s$mode -name "*kernel_stats.csv" | head -1) | cut -c1-200
done > $O/summary.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/two -o b -- python $R/tools/fused_qkv_phases.py 0 DiT-L/2 64 0 > $O/two.log 2>&1
echo "== two kernels" >> $O/summary.txt; head -8 $(find $O/two -name "*kernel_stats.csv" | head -1) | cut -c1-200 >> $O/summary.txt
cat $O/summary.txt
rm -rf $O/s* $O/two
