#!/bin/bash
# Issue / LDS / matrix-pipe counters of tfront_kernel (csrc/tfront.hip) at the headline shape.  -> gpurun_out/tfront_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/tfront_pmc.txt
: > $OUT
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 150 rocprofv3 --pmc $grp -d $R/gpurun_out/tpmc_$n -o p -- python $R/tools/tfront_one.py > $R/gpurun_out/tpmc_$n.log 2>&1
  python $R/tools/pmc_summary.py $(find $R/gpurun_out/tpmc_$n -name "*.db" | head -1) tfront 2>&1 | tail -3 | tee -a $OUT
  rm -rf $R/gpurun_out/tpmc_$n $R/gpurun_out/tpmc_$n.log
done
