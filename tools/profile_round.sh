#!/bin/bash
# Round profile: kernel stats of the bench command, PMC passes on the dominant GEMM (SQ set incl. MFMA-busy, LDS set, FETCH_SIZE,
# WRITE_SIZE: one pass each, --pmc with --kernel-trace only), the plain bench line.  usage: tools/profile_round.sh <tag>
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
C0="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16"
C1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
C2="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"
pmc() { # tag counters -- probe args
  local tag=$1; shift; local ctr=$1; shift
  timeout 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$tag -o p -- python $R/tools/gemm_probe.py "$@" > $O/$tag.log 2>&1
}
pmc fc1_mfma "$C0" 0 16384 4096 1024 1 5 randn
pmc fc1_sq "$C1" 0 16384 4096 1024 1 5 randn
pmc fc1_lds "$C2" 0 16384 4096 1024 1 5 randn
pmc fc1_fetch "FETCH_SIZE" 0 16384 4096 1024 1 5 randn
pmc fc1_write "WRITE_SIZE" 0 16384 4096 1024 1 5 randn
pmc fc2_mfma "$C0" 0 16384 1024 4096 3 5 randn
pmc proj_mfma "$C0" 0 16384 1024 1024 3 5 randn
cd $R && python tools/pmc_parse.py $O gemm256 > $O/pmc_summary.txt 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
tail -60 $O/pmc_summary.txt; head -30 $O/kernel_stats.csv | cut -c1-200
