#!/bin/bash
# PMC passes (one counter group per pass, rocprofv3 slot limits) of tools/conv_hot.py <args>; summary to gpurun_out/pmc_<tag>.txt
# usage: tools/pmc_conv.sh <tag> <kernel substring> <conv_hot args...>
R=${GRAFT_REPO_ROOT:-/root/repo}; tag=$1; pat=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_${tag}_$i -- python3 $R/tools/conv_hot.py "$@" > $R/gpurun_out/pmc_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmc_${tag}_$i.log; exit 1; }
done
cd $R
python3 tools/pmc_summarize.py "$pat" gpurun_out/pmc_${tag}_1 gpurun_out/pmc_${tag}_2 gpurun_out/pmc_${tag}_3 gpurun_out/pmc_${tag}_4 > gpurun_out/pmc_${tag}.txt
find gpurun_out -name "*kernel_trace.csv" -delete; find gpurun_out -name "*counter_collection.csv" -delete
cat gpurun_out/pmc_${tag}.txt
