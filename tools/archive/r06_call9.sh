#!/bin/bash
# round 6, call 9: full GPU suite + bench lines (two lanes default, one lane) after the attention work
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/c9; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -n 15 > gpurun_out/c9/pytest_gpu.txt; cat gpurun_out/c9/pytest_gpu.txt
timeout 400 python bench.py --steps 6 --warmup 2 2>/dev/null | grep '^{' > gpurun_out/c9/bench_line.json; head -c 900 gpurun_out/c9/bench_line.json; echo
timeout 300 python bench.py --steps 6 --warmup 2 --in-flight 1 --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/c9/bench_line_one_lane.json; head -c 600 gpurun_out/c9/bench_line_one_lane.json; echo
