cd $GRAFT_REPO_ROOT
O=gpurun_out/r2x; mkdir -p $O
timeout 150 python bench.py --steps 5 --warmup 2 2>&1 | grep -v amdgpu | tee $O/bench.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/stats.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv; rm -rf $O/stats
head -8 $O/kernel_stats.csv | cut -c1-160
( time timeout 170 python -m pytest tests/test_gpu_configs.py tests/test_gpu_sampler.py -q -m gpu -x --durations=12 -p no:cacheprovider 2>&1 | tail -22 ) > $O/tests.log 2>&1
tail -24 $O/tests.log
