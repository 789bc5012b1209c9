#!/bin/bash
# Round profile: rocprofv3 kernel stats of the bench command; PMC passes (one counter set per pass, --pmc with --kernel-trace only) on the DiT-L/2
# batch-64 forward IN SITU -- MFMA busy / clock, SQ wait classes, LDS, FETCH_SIZE, WRITE_SIZE for the four block GEMMs and the attention kernel.
# usage: tools/profile_round.sh <tag>
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$TAG; mkdir -p $O
# ONE batch in flight: per-kernel durations that add up (comparable with rounds 1-4) -- the block sum the roofline fractions are computed from
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o b -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --in-flight 1 > $O/stats.log 2>&1
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null; rm -rf $O/stats
grep '^{' $O/stats.log > $O/bench_line_under_rocprof.json
# the default (two batches in flight): kernels of the two lanes overlap, so per-kernel durations are NOT additive here -- kept to show what runs
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -o b -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $O/stats2.log 2>&1
cp $(find $O/stats2 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_two_lanes.csv 2>/dev/null; rm -rf $O/stats2
grep '^{' $O/stats2.log > $O/bench_line_two_lanes_under_rocprof.json
C0="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16"
C1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
C2="GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU"
pmc() { # tag counters
  local tag=$1; shift
  timeout 150 rocprofv3 --pmc $1 --kernel-trace --output-format csv -d $O/$tag -o p -- python $R/tools/fwd_probe.py 3 > $O/$tag.log 2>&1
}
pmc mfma "$C0"; pmc sq "$C1"; pmc lds "$C2"; pmc fetch "FETCH_SIZE"; pmc write "WRITE_SIZE"
cd $R
for k in EpiModGeluF16 EpiQKVMod EpiGateResidMod dit_attention qkv_attention_kernel; do python tools/pmc_parse.py $O $k; done > $O/pmc_summary.txt 2>&1
python - "$O" <<'PY'
import json, sys
# FETCH_SIZE / WRITE_SIZE (KiB per launch, mean over the launches after the first) of the folded fc1 GEMM -> the tracked file bench.py reads `traffic` from
o = sys.argv[1]
vals, cur = {}, None
for line in open(o + "/pmc_summary.txt"):
    w = line.split()
    if len(w) > 2 and "kernel" in w[1]:
        cur = w[1]
    elif cur and "gemm256h_tn_kernel" in cur and "EpiModGeluF16" in cur and w and w[0] in ("FETCH_SIZE", "WRITE_SIZE"):
        vals[w[0]] = float(w[1])
if len(vals) == 2:
    tag = o.rstrip("/").split("/")[-1]
    json.dump({"kernel": "gemm256h_tn_kernel<ASrcRowMajor, EpiModGeluF16, false, 0, 1>", "fetch_kib": vals["FETCH_SIZE"], "write_kib": vals["WRITE_SIZE"],
               "source": "profiles/%s_pmc_in_situ.txt (EpiModGeluF16: fetch / write passes)" % tag}, open(o + "/fc1_traffic.json", "w"), indent=1)
    print("fc1 traffic", vals)
PY
timeout 300 python bench.py --steps 5 --warmup 2 2>/dev/null | grep '^{' > $O/bench_line.json
timeout 200 python bench.py --steps 5 --warmup 2 --in-flight 1 --no-cpu-baseline 2>/dev/null | grep '^{' > $O/bench_line_one_lane.json
head -c 600 $O/bench_line.json; echo; head -40 $O/pmc_summary.txt
