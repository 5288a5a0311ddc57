#!/bin/bash
# Collect the round's profiles on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1800 -- 'bash tools/collect_profiles.sh r01'
# Writes gpurun_out/<tag>_*; copy the summaries into profiles/ afterwards (tools/README in profiles/).
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
for WL in cfg4 cfg2; do
  rm -rf /tmp/kt_$WL
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt_$WL -- python $ROOT/bench.py --workload $WL --no-cpu-baseline > $OUT/${TAG}_${WL}_kt.log 2>&1)
  DB=$(find /tmp/kt_$WL -name "*_results.db" | head -1)
  python tools/rocpd_stats.py "$DB" 60 > $OUT/${TAG}_${WL}_kernel_stats.md
done
rm -rf /tmp/pmc
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  D=/tmp/pmc/$(echo $C | tr ' ' '_')
  (cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_pmc_$(echo $C | cut -d' ' -f1).log 2>&1)
done
python tools/pmc_summary.py /tmp/pmc > $OUT/${TAG}_cfg4_pmc.md
# serial reference: one stream, so that per-launch durations are also the family rate
VC_VGG_STREAMS=1 python bench.py --no-cpu-baseline > $OUT/${TAG}_bench_1stream.json 2>/dev/null
rm -rf /tmp/kt_1s
(cd /tmp && VC_VGG_STREAMS=1 rocprofv3 --kernel-trace --stats -d /tmp/kt_1s -- python $ROOT/bench.py --no-cpu-baseline > $OUT/${TAG}_cfg4_1stream_kt.log 2>&1)
python tools/rocpd_stats.py "$(find /tmp/kt_1s -name '*_results.db' | head -1)" 60 > $OUT/${TAG}_cfg4_kernel_stats_1stream.md
