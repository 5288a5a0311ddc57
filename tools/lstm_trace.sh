#!/bin/bash
# Per-kernel durations of the LSTM step kernels (run through gpurun): rocprofv3 kernel trace of tools/microbench.py lstm
#   gpurun --timeout 600 -- 'bash tools/lstm_trace.sh "3,1"'
set -u
ROOT=$(pwd)
export TMPDIR=/tmp
rm -rf /tmp/kt_lstm
(cd /tmp && VC_LSTM_MODES=${1:-3,1} rocprofv3 --kernel-trace --stats -d /tmp/kt_lstm -- python $ROOT/tools/microbench.py lstm > /tmp/kt_lstm.log 2>&1)
DB=$(find /tmp/kt_lstm -name "*_results.db" | head -1)
python tools/rocpd_stats.py "$DB" 40 | grep -E "lstm|gemm_kernel|splitk|^\| kernel" | cut -c1-260
