#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c6
for rnd in 1 2 3; do
  for v in base prio1 prio2 prio3 vpre epi1; do
    LFM_HIP_LIBRARY=tools/ship_variants/$v/liblfm_hip.so timeout 120 python tools/attn_variant_time.py $v 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/c6/attn_variants.txt
