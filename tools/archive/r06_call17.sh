#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/final; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -n 12 > gpurun_out/final/pytest_gpu.txt; cat gpurun_out/final/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 3 | tee gpurun_out/final/smoke.txt
timeout 400 python bench.py --steps 8 --warmup 3 2>/dev/null | grep '^{' > gpurun_out/final/bench_line.json; head -c 300 gpurun_out/final/bench_line.json; echo
