#!/bin/bash
# Issue / LDS / matrix-pipe counters of the fused feed-forward (csrc/ff_fused.hip) at the 64x64 level (M = 32768): the 8-wave
# kernel that ships, the 4-wave kernel, and the GEGLU GEMM of the two-launch chain; one counter group per pass
# (MI355X_MICROARCH.md: --pmc in its own runs).  -> gpurun_out/ff_pmc.txt (copy to profiles/rNN_ff_fused_pmc.txt)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/ff_pmc.txt
echo "# tools/ff_pmc.sh; lib_sha16: $(cd $R && python -c 'from powerpaint_amd import _lib; print(_lib.build_id())')" > $OUT
WHICH=("fused ff_fused8" "fused4 ff_fused_kernel" "chain pp_gemm_kernel_v2"); [ "${ONLY:-}" = "fused" ] && WHICH=("fused ff_fused8")
for which in "${WHICH[@]}"; do
  set -- $which
  echo "## $1 (kernel filter $2)" | tee -a $OUT
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
    n=$(echo $grp | tr ' ' '_' | cut -c1-40)
    timeout 150 rocprofv3 --pmc $grp -d $R/gpurun_out/fpmc_$n -o p -- python $R/tools/ff_one.py --only $1 32768 > $R/gpurun_out/fpmc_$n.log 2>&1
    python $R/tools/pmc_summary.py $(find $R/gpurun_out/fpmc_$n -name "*.db" | head -1) $2 2>&1 | tail -4 | tee -a $OUT
    rm -rf $R/gpurun_out/fpmc_$n $R/gpurun_out/fpmc_$n.log
  done
done
