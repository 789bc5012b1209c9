cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2t
timeout 600 python -m pytest tests/test_gpu_dit.py -x -q -m gpu -k "epilogue or qkv or golden or patch or fullsize or bench_gemm_shapes" 2>&1 | tail -6 > gpurun_out/r2t/tests.log
cat gpurun_out/r2t/tests.log
timeout 120 tools/ubench/valu_rate 2>&1 | tee gpurun_out/r2t/valu_rate.log
timeout 300 python tools/epi_trace.py 2>&1 | grep -v amdgpu | grep -E "rep 1|skipped" | tee gpurun_out/r2t/epi_trace.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2t/prof -o p -- python $GRAFT_REPO_ROOT/tools/r2_probe.py nohoist=0:65536 > $GRAFT_REPO_ROOT/gpurun_out/r2t/probe.log 2>&1
cd $GRAFT_REPO_ROOT
grep -vE "amdgpu|rocprof|^[EWI]2026" gpurun_out/r2t/probe.log | tail -30
head -25 $(find gpurun_out/r2t/prof -name "*kernel_stats.csv" | head -1) | cut -c1-180
