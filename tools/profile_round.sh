#!/bin/bash
# Round profile: kernel stats of the bench command, PMC passes on the dominant GEMM, the power-cap A/B, the plain bench line.
# Counters are collected in their own passes (--pmc with --kernel-trace only).  usage: tools/profile_round.sh <tag>
TAG=${1:-r1e}
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
C1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
C2="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16"
pmc() { # tag counters... -- probe args
  local tag=$1; shift; local ctr=$1; shift
  timeout 120 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$tag -o p -- python $R/tools/gemm_probe.py "$@" > $O/$tag.log 2>&1
}
pmc fc1_sq "$C1" 0 16384 4096 1024 1 5 randn
pmc fc1_lds "$C2" 0 16384 4096 1024 1 5 randn
pmc fc1_fetch "FETCH_SIZE" 0 16384 4096 1024 1 5 randn
pmc fc1_write "WRITE_SIZE" 0 16384 4096 1024 1 5 randn
pmc cap_randn "$C1" 3 4096 4096 8192 0 4 randn
pmc cap_zeros "$C1" 3 4096 4096 8192 0 4 zeros
pmc cap_v2_randn "$C1" 2 4096 4096 8192 0 4 randn
cd $R && python tools/pmc_parse.py $O > $O/pmc_summary.txt 2>&1
python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json; echo; tail -40 $O/pmc_summary.txt
