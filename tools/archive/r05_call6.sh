#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest "tests/test_gpu_dit.py::test_skinny_latency_kernel_vs_splitk_path_and_oracle" "tests/test_gpu_dit.py::test_dit_matches_oracle_fullsize" tests/test_gpu_downstream.py tests/test_gpu_cli.py -q -x > gpurun_out/pytest_call6.txt 2>&1; tail -12 gpurun_out/pytest_call6.txt
timeout 300 python tools/latency_probe.py > gpurun_out/latency_probe.txt 2>&1; tail -10 gpurun_out/latency_probe.txt
timeout 200 python tools/splitk256_probe.py 20 > gpurun_out/splitk256_probe.txt 2>&1; tail -12 gpurun_out/splitk256_probe.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/b1stats -o b -- python $R/tools/fwd_probe_b1.py > $R/gpurun_out/b1stats.log 2>&1
cp $(find $R/gpurun_out/b1stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/b1_kernel_stats.csv 2>/dev/null; rm -rf $R/gpurun_out/b1stats
head -14 $R/gpurun_out/b1_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
