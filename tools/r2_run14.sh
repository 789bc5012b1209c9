cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2n
timeout 300 python tools/r2_probe.py nt=0:2097152 nofwd > gpurun_out/r2n/probe.log 2>&1; grep -vE "attention|ln_mod|amdgpu" gpurun_out/r2n/probe.log
