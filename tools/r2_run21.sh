cd $GRAFT_REPO_ROOT
O=gpurun_out/r2v; mkdir -p $O
nproc > $O/nproc.log
( time timeout 900 python -m pytest tests -q -m gpu -n 4 --maxfail=20 -p no:cacheprovider 2>&1 | tail -40 ) > $O/tests.log 2>&1
tail -15 $O/tests.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -v amdgpu | tee $O/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/stats.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/stats
head -16 $O/kernel_stats.csv | cut -c1-200
