R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/final4
timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 > gpurun_out/final4/pytest_gpu.txt; cat gpurun_out/final4/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final4/smoke.txt 2>&1; tail -2 gpurun_out/final4/smoke.txt
bash tools/profile_round.sh r06f > gpurun_out/final4/profile.log 2>&1
python3 -c "
import json
for f in ['bench_line.json','bench_line_one_lane.json']:
    d=json.load(open('gpurun_out/r06f/'+f)); print(f, d['value'], d['clock_mhz_under_mfma_load'], d['roofline']['avg_launch_us'], d['roofline_block']['block_us'], d['roofline_block']['frac'])
"
head -4 gpurun_out/r06f/kernel_stats.csv | cut -d, -f1-4 | cut -c1-130
