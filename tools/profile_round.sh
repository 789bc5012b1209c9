#!/bin/bash
# Round profile: rocprofv3 kernel stats of the bench command; PMC passes (one counter set per pass, --pmc with --kernel-trace only) on the DiT-L/2
# batch-64 forward IN SITU -- MFMA busy / clock, SQ wait classes, LDS, FETCH_SIZE, WRITE_SIZE for the four block GEMMs and the attention kernel.
# usage: tools/profile_round.sh <tag>
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null; rm -rf $O/stats
grep '^{' $O/stats.log > $O/bench_line_under_rocprof.json
C0="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16"
C1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
C2="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU"
pmc() { # tag counters
  local tag=$1; shift
  timeout 150 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $O/$tag -o p -- python $R/tools/fwd_probe.py 3 > $O/$tag.log 2>&1
}
pmc mfma "$C0"; pmc sq "$C1"; pmc lds "$C2"; pmc fetch "FETCH_SIZE"; pmc write "WRITE_SIZE"
cd $R
for k in EpiModGeluF16 EpiQKVMod EpiGateResidMod dit_attention; do python tools/pmc_parse.py $O $k; done > $O/pmc_summary.txt 2>&1
timeout 200 python bench.py --steps 5 --warmup 2 2>/dev/null | grep '^{' > $O/bench_line.json
head -c 600 $O/bench_line.json; echo; head -40 $O/pmc_summary.txt
