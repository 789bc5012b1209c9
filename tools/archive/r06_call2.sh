#!/bin/bash
# round 6, call 2: phase split of the streamed attention kernel (measurement build), in-situ kernel durations with the option on / off
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/c2; export TMPDIR=/tmp
LFM_HIP_LIBRARY=tools/ship_variants/measure/liblfm_hip.so timeout 600 python tools/attn_stream_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c2/attn_stream_phases.txt
cd /tmp
for opt in 0 1; do
  LFM_ATT_STREAM=$opt timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c2/prof$opt -o f -- python $R/tools/fwd_probe.py 4 > /dev/null 2>&1
  f=$(find $R/gpurun_out/c2/prof$opt -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/c2/fwd_kernel_stats_stream$opt.csv; rm -rf $R/gpurun_out/c2/prof$opt
  echo "== in situ (DiT-L/2 batch-64 forward), stream $opt"; grep -i "attention\|EpiGateResid\|EpiQKV\|EpiModGelu" $R/gpurun_out/c2/fwd_kernel_stats_stream$opt.csv | cut -c1-200
done 2>&1 | tee $R/gpurun_out/c2/in_situ.txt
C1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
C0="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for opt in 0 1; do
  LFM_ATT_STREAM=$opt timeout 200 rocprofv3 --pmc $C1 --kernel-trace --output-format csv -d $R/gpurun_out/c2/pmc_sq$opt -o p -- python $R/tools/attn_probe.py 0 6 > /dev/null 2>&1
  LFM_ATT_STREAM=$opt timeout 200 rocprofv3 --pmc $C0 --kernel-trace --output-format csv -d $R/gpurun_out/c2/pmc_mfma$opt -o p -- python $R/tools/attn_probe.py 0 6 > /dev/null 2>&1
done
cd $R
for opt in 0 1; do echo "== PMC stream $opt"; python tools/pmc_parse.py gpurun_out/c2/pmc_sq$opt attention; python tools/pmc_parse.py gpurun_out/c2/pmc_mfma$opt attention; done 2>&1 | tee gpurun_out/c2/attn_pmc.txt
rm -rf gpurun_out/c2/pmc_*
