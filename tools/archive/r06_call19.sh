#!/bin/bash
# config 6 evidence at HEAD (after the EDM concat change)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r06cfg6; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python bench.py --config 6 --steps 4 --warmup 2 2>/dev/null | grep '^{' > $O/config6_bench_line.json
timeout 300 python bench.py --config 6 --steps 3 --warmup 1 --in-flight 1 --no-roofline 2>/dev/null | grep '^{' > $O/config6_bench_line_one_lane.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats6 -o b -- python $R/bench.py --config 6 --steps 1 --warmup 1 --no-roofline --in-flight 1 > $O/stats6.log 2>&1
cp $(find $O/stats6 -name "*kernel_stats.csv" | head -1) $O/config6_kernel_stats.csv; rm -rf $O/stats6
python -c "
import json
for f in ('config6_bench_line.json','config6_bench_line_one_lane.json'):
    d=json.load(open('$O/'+f)); print(f, round(d['value'],1), round(d['mfma_frac_whole_path'],4), d['clock_mhz_under_mfma_load'])"
head -n 6 $O/config6_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
