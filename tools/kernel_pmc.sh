#!/bin/bash
# Hardware counters of one kernel under any command: separate rocprofv3 --pmc passes (with --kernel-trace --output-format csv only),
# per-kernel means by tools/pmc_kernels.py.  Through gpurun from the repo root:
#   bash tools/kernel_pmc.sh <tag> <kernel substring> <command ...>      -> gpurun_out/<tag>_pmc.md
set -u
TAG=$1; PAT=$2; shift 2
ROOT=$(pwd)
export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out
rm -rf /tmp/kpmc_$TAG
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/kpmc_$TAG/p$i -- "$@" > $ROOT/gpurun_out/${TAG}_kpmc$i.log 2>&1)
done
python tools/pmc_kernels.py /tmp/kpmc_$TAG "$PAT" > gpurun_out/${TAG}_pmc.md
cat gpurun_out/${TAG}_pmc.md
