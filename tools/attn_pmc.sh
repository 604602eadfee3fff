cd /tmp && export TMPDIR=/tmp
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SALU"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $grp -d /root/repo/gpurun_out/apmc_$n -o p -- python /root/repo/tools/attn_one.py 40 4096 4096 > /root/repo/gpurun_out/apmc_$n.log 2>&1
  python /root/repo/tools/pmc_summary.py $(find /root/repo/gpurun_out/apmc_$n -name "*.db" | head -1) attn_ 2>&1 | tail -4
done
