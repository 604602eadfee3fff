# LDS / issue counters of the v2 GEMM main loop on one long-K shape (one counter group per pass).  Writes summaries to
# gpurun_out/gemm_pmc.txt and deletes the databases.
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/gemm_pmc.txt
: > $OUT
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 150 rocprofv3 --pmc $grp -d /root/repo/gpurun_out/gpmc_$n -o p -- python /root/repo/tools/gemm_one.py 8192 1280 11520 33 > /root/repo/gpurun_out/gpmc_$n.log 2>&1
  python /root/repo/tools/pmc_summary.py $(find /root/repo/gpurun_out/gpmc_$n -name "*.db" | head -1) pp_gemm 2>&1 | tail -3 | tee -a $OUT
  rm -rf /root/repo/gpurun_out/gpmc_$n
done
