#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c11
for v in packed; do
  timeout 600 python tools/cosched_aggressors.py tools/ship_variants/$v/liblfm_hip.so 24 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/c11/cosched_aggressors_round2.txt
