#!/bin/bash
# round 6, call 3: streamed attention with 72 KiB of LDS (two resident workgroups per CU): parity, A/B, phase split, in-situ durations
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/c8; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dit.py -x -q -m gpu -k "attention" 2>&1 | tail -5 | tee gpurun_out/c8/pytest_attention.txt
timeout 600 python tools/attn_stream_ab.py 5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c8/attn_stream_ab.txt
LFM_HIP_LIBRARY=tools/ship_variants/measure/liblfm_hip.so timeout 600 python tools/attn_stream_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c8/attn_stream_phases.txt
cd /tmp
for opt in 0 1; do
  LFM_ATT_STREAM=$opt timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c8/prof$opt -o f -- python $R/tools/fwd_probe.py 4 > /dev/null 2>&1
  f=$(find $R/gpurun_out/c8/prof$opt -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/c8/fwd_kernel_stats_stream$opt.csv; rm -rf $R/gpurun_out/c8/prof$opt
  echo "== in situ (DiT-L/2 batch-64 forward), stream $opt"; grep -i "attention" $R/gpurun_out/c8/fwd_kernel_stats_stream$opt.csv | cut -c1-200
done 2>&1 | tee $R/gpurun_out/c8/in_situ.txt
C1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
LFM_ATT_STREAM=1 timeout 200 rocprofv3 --pmc $C1 --kernel-trace --output-format csv -d $R/gpurun_out/c8/pmc_sq1 -o p -- python $R/tools/attn_probe.py 0 6 > /dev/null 2>&1
cd $R; python tools/pmc_parse.py gpurun_out/c8/pmc_sq1 attention 2>&1 | tee gpurun_out/c8/attn_pmc.txt; rm -rf gpurun_out/c8/pmc_*
