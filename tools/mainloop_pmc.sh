#!/bin/bash
# (M) PMC passes on the main-loop ablations of the LFM_MEASURE build (tools/mainloop_ablation.py): clock, MFMA busy, wait classes.  usage: tools/mainloop_pmc.sh <tag>
TAG=${1:-r3pmc}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
C1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16"
for abl in 0 5 7; do
  sel=$(( 5 | ((4 | (abl << 21)) << 4) ))
  timeout 120 rocprofv3 --pmc $C1 --kernel-trace --output-format csv -d $O/abl$abl -o p -- python $R/tools/gemm_probe.py $sel 16384 1024 4096 1 5 randn > $O/abl$abl.log 2>&1
done
cd $R && python tools/pmc_parse.py $O gemm256 2>&1 | tee $O/pmc_summary.txt | tail -40
