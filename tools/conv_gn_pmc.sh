#!/bin/bash
# Issue / LDS / matrix-pipe counters of the fused norm -> SiLU -> conv kernel (csrc/conv_gn.hip) at the 64x64 level
# (batch 8, 320 -> 320: 256 workgroups of 256x160, five channel chunks) and on a long-K shape (640 + 320 -> 320), one counter
# group per pass; the same passes on the skeleton without the normalisation (lab build, PP_CONV_GN_NMODE=2).
# -> gpurun_out/conv_gn_pmc.txt (copy to profiles/rNN_gemm_pmc.txt); the header carries pp_build_id of the library it ran
cd /tmp && export TMPDIR=/tmp
R=/root/repo
OUT=$R/gpurun_out/conv_gn_pmc.txt
: > $OUT
echo "# tools/conv_gn_pmc.sh; lib_sha16: $(cd $R && python -c 'from powerpaint_amd import _lib; print(_lib.build_id())')  (shipping library; the nm2 passes run libpp_hip_lab.so of the same sources)" >> $OUT
SHAPES=("64 320 0 320" "64 640 320 320"); [ "${SHAPES_ONE:-0}" = "1" ] && SHAPES=("64 320 0 320")
for shape in "${SHAPES[@]}"; do
 for variant in ship nm2; do
  if [ $variant = nm2 ]; then export PP_LAB=1 PP_LIB=$R/powerpaint_amd/libpp_hip_lab.so PP_CONV_GN_NMODE=2; else unset PP_LAB PP_LIB PP_CONV_GN_NMODE; fi
  echo "## shape (H C1 C2 Cout) = $shape, variant = $variant" | tee -a $OUT
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
    n=$(echo $grp | tr ' ' '_' | cut -c1-40)
    timeout 150 rocprofv3 --pmc $grp -d $R/gpurun_out/cpmc_$n -o p -- python $R/tools/conv_gn_one.py $shape > $R/gpurun_out/cpmc_$n.log 2>&1
    python $R/tools/pmc_summary.py $(find $R/gpurun_out/cpmc_$n -name "*.db" | head -1) pp_conv_gn 2>&1 | tail -3 | tee -a $OUT
    rm -rf $R/gpurun_out/cpmc_$n $R/gpurun_out/cpmc_$n.log
  done
 done
done
