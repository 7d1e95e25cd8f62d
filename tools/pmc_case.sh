# Counter passes (one rocprofv3 --pmc run each) over one case of tools/prof_case.py:
#   bash tools/pmc_case.sh <case>      -> gpurun_out/pmc_<case>_{1,2,3} and a per-kernel summary
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C=$1
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS"
P3="TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmc_${C}_$i
  timeout -s KILL 120 rocprofv3 --pmc $P -d $R/gpurun_out/pmc_${C}_$i -o p -- python3 $R/tools/prof_case.py $C 3 > $R/gpurun_out/pmc_${C}_$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_${C}_$i.log; exit 1; }
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${C}_1 $R/gpurun_out/pmc_${C}_2 $R/gpurun_out/pmc_${C}_3
