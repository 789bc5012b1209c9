R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
C1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"
C2="SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
C3="SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES"
for v in 0 32768; do for i in 1 2 3; do eval C=\$C$i; timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/f${v}_c$i -o p -- python $R/tools/attn_probe.py $v 5 > $O/f${v}_c$i.log 2>&1; done; done
cd $R && python tools/pmc_parse.py $O attention > $O/pmc_summary.txt 2>&1; cat $O/pmc_summary.txt
