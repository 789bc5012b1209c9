#!/bin/bash
# query-split attention for latency mode: tests, latency A/B, kernel stats
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_dit.py -q -x -k "attention or latency or cfg or oracle" > gpurun_out/pytest_call21.txt 2>&1; tail -4 gpurun_out/pytest_call21.txt
timeout 400 python tools/latency_probe.py > gpurun_out/latency_probe.txt 2>&1; grep "64x64" gpurun_out/latency_probe.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/b1stats -o b -- python $R/tools/fwd_probe_b1.py > $R/gpurun_out/b1stats.log 2>&1
cp $(find $R/gpurun_out/b1stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/b1_kernel_stats.csv 2>/dev/null; rm -rf $R/gpurun_out/b1stats
head -7 $R/gpurun_out/b1_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
