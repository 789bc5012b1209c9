cd $GRAFT_REPO_ROOT
O=gpurun_out/r2y; mkdir -p $O
( time timeout 240 python -m pytest tests -q -m gpu -x --durations=12 -p no:cacheprovider 2>&1 | tail -24 ) > $O/tests.log 2>&1
tail -26 $O/tests.log
timeout 40 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu | tail -2 | tee $O/smoke.log
for c in 5 4 3; do timeout 60 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -v amdgpu | tee $O/bench_c$c.log | cut -c1-330; done
