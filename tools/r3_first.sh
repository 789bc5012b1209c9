# First GPU call of the next round (prepared at the end of round 2, when the GPU budget was spent): validates what was written blind and
# collects the two measurements the round-2 findings ask for.  usage: gpurun --timeout 400 -- 'bash tools/r3_first.sh'
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
# 1. the experimental fused LayerNorm epilogue: correctness first (bounded spin: cannot hang), then the A/B
LFM_EXPERIMENTAL=1 timeout 120 python -m pytest tests/test_gpu_dit.py -q -m gpu -x -k "experimental_fused_ln" -p no:cacheprovider 2>&1 | tail -5 | tee $O/fuse_ln_test.log
timeout 120 python tools/fuse_ln_probe.py 2>&1 | grep -v amdgpu | tee $O/fuse_ln_probe.log
# 2. where an attention workgroup's ~40k cycles go
timeout 60 python tools/attn_trace.py 2>&1 | grep -v amdgpu | tee $O/attn_trace.log
# 3. the LN-modulate / attention phase probe and the bench line with the one-row LN kernel in place
timeout 60 python tools/r2_probe3.py 2>&1 | grep -v amdgpu | tee $O/probe3.log
timeout 150 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -v amdgpu | tee $O/bench.log | cut -c1-300
