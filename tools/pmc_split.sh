#!/bin/bash
# developer aid (GPU box): SQ counter passes over the split-operand GEMM loop -> gpurun_out/pmc_split_<tag>/
tag=${1:-base}
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $R/gpurun_out/pmc_split_$tag/p$i --output-format csv -- python3 $R/tools/gemm_split_one.py 51200 1024 1024 1 3 > $R/gpurun_out/pmc_split_$tag/log$i.txt 2>&1 || echo "pass $i failed"
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_split_$tag | grep -A40 "skg_gemm_kernel" 
