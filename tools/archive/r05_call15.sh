#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_final2.txt 2>&1; tail -4 gpurun_out/pytest_gpu_final2.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/profile_round.sh r05head > gpurun_out/profile_round_head.log 2>&1
python - <<'PY'
import json
for f in ("bench_line.json","bench_line_one_lane.json"):
    d=json.loads(open("gpurun_out/r05head/"+f).read().strip().splitlines()[-1]); print(f, round(d["value"],2), d["clock_mhz_under_mfma_load"], round(d["roofline"]["avg_launch_us"],2), round(d["roofline_block"]["block_us"],1), round(d["roofline_block"]["two_lanes"]["block_us_upper_bound"],1), d.get("cpu_baseline",{}).get("value"))
PY
head -6 gpurun_out/r05head/kernel_stats.csv | cut -d, -f1-4 | cut -c1-140
