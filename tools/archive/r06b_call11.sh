R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/f11
timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/f11/test.log; cat gpurun_out/f11/test.log
bash tools/profile_round.sh r06c > gpurun_out/f11/profile.log 2>&1
python3 -c "
import json
for f in ['bench_line.json','bench_line_one_lane.json']:
    d=json.load(open('gpurun_out/r06c/'+f)); print(f, d['value'], d['clock_mhz_under_mfma_load'], d['roofline']['avg_launch_us'], d['roofline_block']['block_us'], d['roofline_block']['frac'])
"
head -5 gpurun_out/r06c/kernel_stats.csv | cut -c1-150
