#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c18
for rnd in 1 2 3 4; do
  for v in base dual; do
    LFM_HIP_LIBRARY=tools/ship_variants/$v/liblfm_hip.so timeout 120 python tools/attn_variant_time.py $v 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/c18/attn_variants.txt
