#!/bin/bash
# kernel stats of BASELINE config 5 (origin-ADM celeb512, batch 32): rocprofv3 --kernel-trace --stats of bench.py --config 5
TAG=${1:-r2adm}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --config 5 --steps 1 --warmup 1 > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
head -40 $O/kernel_stats.csv | cut -c1-230; tail -2 $O/stats.log | cut -c1-600
