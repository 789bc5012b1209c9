R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/f13; mkdir -p $O
for mode in 0 32 64 0 32 64; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$mode -o b -- python $R/tools/fused_qkv_phases.py $mode > $O/s$mode.log 2>&1
  echo "== flags $mode"; grep -h "qkv_attention" $(find $O/s$mode -name "*kernel_stats.csv" | head -1) | cut -d, -f1-4 | cut -c1-120
  rm -rf $O/s$mode
done > $O/gm.txt
cat $O/gm.txt
