#!/bin/bash
# ring depth of the 64x64 latency kernel: SQ_STAGES 4 / 5 / 6 / 8 (variant builds) on the stand-alone probe and the 50-step solve
# build the variants first (not tracked): for n in 4 5 6; do bash tools/build_variant.sh sq$n -DSQ_STAGES=$n; done
mkdir -p gpurun_out
for v in sq4 sq5 sq6 default; do
  if [ $v = default ]; then unset LFM_HIP_LIBRARY; else export LFM_HIP_LIBRARY=tools/variants/$v/liblfm_hip.so; fi
  echo "=== $v"
  timeout 200 python tools/latency_gemm_probe.py 2>&1 | grep "kernel 7" | grep "shaped\|4 K-tiles" | cut -c1-150
  timeout 200 python tools/latency_probe.py 2>&1 | grep "DiT-L/2 .64x64"
done > gpurun_out/r05_call20.log 2>&1
cat gpurun_out/r05_call20.log
