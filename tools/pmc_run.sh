#!/bin/bash
# PMC passes over the GEMM probe (counters only + kernel trace; never mixed with other trace domains)
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/pmc4; mkdir -p $O
C1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
C2="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"
run() { # tag sel M N K fill
  for i in 1 2; do
    eval C=\$C$i
    timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$1_c$i -o p -- python $R/tools/gemm_probe.py $2 $3 $4 $5 0 4 $6 > $O/$1_c$i.log 2>&1
  done
}
run t1_v3 3 256 256 8192 randn
run t1_v2 2 256 256 8192 randn
run t256_v3_randn 3 4096 4096 8192 randn
run t256_v3_zeros 3 4096 4096 8192 zeros
cd $R && python tools/pmc_parse.py $O
