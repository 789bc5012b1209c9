R=$GRAFT_REPO_ROOT
cd $R; mkdir -p gpurun_out/final2
timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/final2/pytest_gpu.txt; cat gpurun_out/final2/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final2/smoke.txt 2>&1; tail -3 gpurun_out/final2/smoke.txt
timeout 900 python bench.py > gpurun_out/final2/bench.log 2>&1; grep '^{' gpurun_out/final2/bench.log > gpurun_out/final2/bench_line.json; cut -c1-300 gpurun_out/final2/bench_line.json
