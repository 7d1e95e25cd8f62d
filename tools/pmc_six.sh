set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
P2="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS"
P3="TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum GRBM_GUI_ACTIVE"
for mode in stream classic; do
  if [ $mode = classic ]; then export CTC_TC_NO_STREAM=1; else unset CTC_TC_NO_STREAM; fi
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $P -d $R/gpurun_out/pmc_${mode}_$i -o p -- python3 $R/tools/prof_case.py six 3 > $R/gpurun_out/pmc_${mode}_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${mode}_$i.log; exit 1; }
  done
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_stream_1 $R/gpurun_out/pmc_stream_2 $R/gpurun_out/pmc_stream_3 $R/gpurun_out/pmc_classic_1 $R/gpurun_out/pmc_classic_2 $R/gpurun_out/pmc_classic_3
