#!/bin/bash
# SQ-counter probe of the convolution kernels (run through gpurun from the repo root): bash tools/sq_probe.sh r02
# Three rocprofv3 --pmc passes (<= 8 SQ counters each, --kernel-trace only) over tools/pmc_probe.py; summary -> gpurun_out/<tag>_sq_probe.txt
set -u
TAG=${1:-r02}
ROOT=$(pwd)
export TMPDIR=/tmp GRAFT_REPO_ROOT=$ROOT
rm -rf /tmp/sqp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/sqp/p$i -- python $ROOT/tools/pmc_probe.py > $ROOT/gpurun_out/${TAG}_sqp$i.log 2>&1)
done
python tools/pmc_probe_summary.py /tmp/sqp > gpurun_out/${TAG}_sq_probe.txt
cat gpurun_out/${TAG}_sq_probe.txt
