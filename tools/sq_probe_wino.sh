#!/bin/bash
# SQ-counter probe of the Winograd forward / data-gradient kernel, per dispatch (run through gpurun from the repo root):
#   bash tools/sq_probe_wino.sh <tag>     -> gpurun_out/<tag>_sq_wino.txt (F(2x2,3x3)), <tag>_sq_wino4.txt (F(4x4,3x3))
set -u
TAG=${1:-r03}
ROOT=$(pwd)
export TMPDIR=/tmp GRAFT_REPO_ROOT=$ROOT
mkdir -p $ROOT/gpurun_out
for FAM in ${FAMILIES:-2 4 4v}; do
export FAMILY=$FAM
SUF=$([ $FAM = 2 ] && echo "" || echo $FAM)
rm -rf /tmp/sqw
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/sqw/p$i -- python $ROOT/tools/pmc_probe_wino.py > $ROOT/gpurun_out/${TAG}_sqw${SUF}_$i.log 2>&1)
done
python tools/sq_probe_wino_summary.py /tmp/sqw > gpurun_out/${TAG}_sq_wino${SUF}.txt
cat gpurun_out/${TAG}_sq_wino${SUF}.txt
done
