cd $GRAFT_REPO_ROOT
O=gpurun_out/r2w; mkdir -p $O
( time timeout 700 python -m pytest tests -q -m gpu --maxfail=8 --durations=40 -p no:cacheprovider 2>&1 | tail -80 ) > $O/tests.log 2>&1
tail -70 $O/tests.log
timeout 200 python tools/r2_probe2.py 2>&1 | grep -v amdgpu | tee $O/probe2.log
