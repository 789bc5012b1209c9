#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest "tests/test_gpu_dit.py::test_skinny_latency_kernel_vs_splitk_path_and_oracle" "tests/test_gpu_dit.py::test_dit_matches_oracle_fullsize" tests/test_gpu_edm.py -q -x > gpurun_out/pytest_call7.txt 2>&1; tail -8 gpurun_out/pytest_call7.txt
timeout 300 python tools/latency_probe.py > gpurun_out/latency_probe2.txt 2>&1; tail -8 gpurun_out/latency_probe2.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/b1stats -o b -- python $R/tools/fwd_probe_b1.py > $R/gpurun_out/b1stats.log 2>&1
cp $(find $R/gpurun_out/b1stats -name "*kernel_stats.csv" | head -1) $R/gpurun_out/b1_kernel_stats2.csv 2>/dev/null; rm -rf $R/gpurun_out/b1stats
head -9 $R/gpurun_out/b1_kernel_stats2.csv | cut -d, -f1-4 | cut -c1-150
