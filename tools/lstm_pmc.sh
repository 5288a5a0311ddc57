#!/bin/bash
# Counters of the LSTM recurrence step kernels at 1280 rows (cfg2 / cfg3), the round-4 review's item 7: what bounds a step of 35 / 41 us
# against a 17 us f32 MFMA floor?  Separate rocprofv3 --pmc passes (with --kernel-trace --output-format csv only) over the sequence
# drivers of tools/microbench.py lstm; run through gpurun from the repo root:  bash tools/lstm_pmc.sh r05
set -u
TAG=${1:-r05}
ROOT=$(pwd)
export TMPDIR=/tmp VC_LSTM_MODES=3 VC_LSTM_NS=1280
mkdir -p $ROOT/gpurun_out
rm -rf /tmp/lpmc
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/lpmc/p$i -- python $ROOT/tools/microbench.py lstm > $ROOT/gpurun_out/${TAG}_lpmc$i.log 2>&1)
done
python tools/pmc_kernels.py /tmp/lpmc lstm_rec > gpurun_out/${TAG}_lstm_pmc.md
cat gpurun_out/${TAG}_lstm_pmc.md
