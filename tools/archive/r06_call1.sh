#!/bin/bash
# round 6, call 1: the streamed attention kernel -- parity, A/B, kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/c1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dit.py -x -q -m gpu -k "attention" 2>&1 | tail -15 > gpurun_out/c1/pytest_attention.txt
cat gpurun_out/c1/pytest_attention.txt
timeout 600 python tools/attn_stream_ab.py 5 2>&1 | tee gpurun_out/c1/attn_stream_ab.txt
cd /tmp
for opt in 0 1; do
  LFM_ATT_STREAM=$opt timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c1/prof_attn$opt -o attn -- python $GRAFT_REPO_ROOT/tools/attn_probe.py 0 30 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
for opt in 0 1; do f=$(find gpurun_out/c1/prof_attn$opt -name "*kernel_stats.csv" | head -1); echo "== stream $opt"; head -5 "$f"; done | tee gpurun_out/c1/attn_kernel_stats.txt
