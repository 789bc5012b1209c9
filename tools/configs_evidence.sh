#!/bin/bash
# bench lines of BASELINE.json configs 3 / 4 / 5 (and bench.py's config 6: the EDM-style ADM UNet) and a rocprofv3 kernel-stats summary of each.  usage: tools/configs_evidence.sh <tag> [gains for config 3 ...]
TAG=${1:-r3cfg}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
for g in "$@"; do  # probing the field gain of config 3: steps / accepted / rejected
  timeout 200 python bench.py --config 3 --steps 1 --warmup 1 --no-roofline --field-gain $g 2>/dev/null | grep '^{' | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('gain', j['config'].get('field_gain'), j['config']['dopri5'], round(j['value'],1), 'img/s')" | tee -a $O/config3_gain_probe.log
done
for c in 3 4 5 6; do
  timeout 300 python bench.py --config $c --steps 4 --warmup 2 2>/dev/null | grep '^{' > $O/config${c}_bench_line.json
  cut -c1-400 $O/config${c}_bench_line.json
  [ $c != 3 ] && timeout 300 python bench.py --config $c --steps 3 --warmup 1 --in-flight 1 --no-roofline 2>/dev/null | grep '^{' > $O/config${c}_bench_line_one_lane.json
done
cd /tmp && export TMPDIR=/tmp
for c in 3 4 5 6; do
  # kernel stats with ONE batch in flight: per-kernel durations that add up (the bench lines above run the default: two lanes for configs 4 / 5 / 6)
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$c -o b -- python $R/bench.py --config $c --steps 1 --warmup 1 --no-roofline --in-flight 1 > $O/stats$c.log 2>&1
  cp $(find $O/stats$c -name "*kernel_stats.csv" | head -1) $O/config${c}_kernel_stats.csv 2>/dev/null
  rm -rf $O/stats$c
  head -8 $O/config${c}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
done
