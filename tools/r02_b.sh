#!/bin/bash
# round 2, run B: counters of the ping-pong 256x160 GEMM; split-K sweep on the deep (M <= 2048) convs; linears
mkdir -p gpurun_out/r02b
R=$PWD
( cd /tmp && export TMPDIR=/tmp
  : > $R/gpurun_out/r02b/gemm_pmc_t53.txt
  for tile in 53 33; do
  for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
    n=$(echo $grp | tr ' ' '_' | cut -c1-30)_$tile
    timeout 150 rocprofv3 --pmc $grp -d $R/gpurun_out/r02b/pmc_$n -o p -- python $R/tools/gemm_one.py 8192 1280 11520 $tile > $R/gpurun_out/r02b/pmc_$n.log 2>&1
    echo "# tile $tile" >> $R/gpurun_out/r02b/gemm_pmc_t53.txt
    python $R/tools/pmc_summary.py $(find $R/gpurun_out/r02b/pmc_$n -name "*.db" | head -1) pp_gemm 2>&1 | tail -6 >> $R/gpurun_out/r02b/gemm_pmc_t53.txt
    tail -1 $R/gpurun_out/r02b/pmc_$n.log >> $R/gpurun_out/r02b/gemm_pmc_t53.txt
    rm -rf $R/gpurun_out/r02b/pmc_$n
  done
  done )
cat gpurun_out/r02b/gemm_pmc_t53.txt
python tools/gemm_pp_bench.py --tiles 31,54,33,53,32 --splitk 1,2,4,8 --conv-only --max-m 2048 --rounds 2 --json gpurun_out/r02b/deep.json > gpurun_out/r02b/deep.log 2>&1
tail -120 gpurun_out/r02b/deep.log
python tools/gemm_pp_bench.py --tiles 24,21,31,44,54,33,53 --splitk 1 --lin-only --rounds 2 --json gpurun_out/r02b/lin.json > gpurun_out/r02b/lin.log 2>&1
tail -150 gpurun_out/r02b/lin.log
