#!/bin/bash
# Per-layer memory-side traffic of the convolution kernels (run through gpurun from the repo root):
#   bash tools/traffic_probe_layers.sh <tag>   -> gpurun_out/<tag>_traffic_layers.md
set -u
TAG=${1:-r04}
ROOT=$(pwd)
export TMPDIR=/tmp GRAFT_REPO_ROOT=$ROOT
mkdir -p $ROOT/gpurun_out
rm -rf /tmp/trl
i=0
for C in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/trl/p$i -- python $ROOT/tools/traffic_probe_layers.py > $ROOT/gpurun_out/${TAG}_trl_$i.log 2>&1)
done
python tools/traffic_probe_layers.py --summary /tmp/trl > gpurun_out/${TAG}_traffic_layers.md
cat gpurun_out/${TAG}_traffic_layers.md
