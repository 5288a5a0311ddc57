set -u
TAG=r03; ROOT=$(pwd); OUT=$ROOT/gpurun_out; export TMPDIR=/tmp
for WL in cfg4; do
  rm -rf /tmp/kt_$WL
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt_$WL -- python $ROOT/bench.py --workload $WL --no-cpu-baseline --strong-n1 0 > $OUT/${TAG}_${WL}_kt.log 2>&1)
  python tools/rocpd_stats.py "$(find /tmp/kt_$WL -name "*_results.db" | head -1)" 60 > $OUT/${TAG}_${WL}_kernel_stats.md
done
rm -rf /tmp/kt_1s
(cd /tmp && VC_VGG_STREAMS=1 rocprofv3 --kernel-trace --stats -d /tmp/kt_1s -- python $ROOT/bench.py --no-cpu-baseline --strong-n1 0 > $OUT/${TAG}_cfg4_1stream_kt.log 2>&1)
python tools/rocpd_stats.py "$(find /tmp/kt_1s -name "*_results.db" | head -1)" 60 > $OUT/${TAG}_cfg4_kernel_stats_1stream.md
rm -rf /tmp/pmc
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  D=/tmp/pmc/$(echo $C | tr ' ' '_')
  (cd /tmp && VC_VGG_STREAMS=1 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --strong-n1 0 > $OUT/${TAG}_pmc_$(echo $C | cut -d' ' -f1).log 2>&1)
done
python tools/pmc_summary.py /tmp/pmc $OUT/${TAG}_traffic_cfg4.json > $OUT/${TAG}_cfg4_pmc.md
bash tools/sq_probe_wino.sh ${TAG} > /dev/null 2>&1
python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
tail -3 $OUT/${TAG}_cfg4_pmc.md; tail -4 $OUT/${TAG}_sq_wino4.txt
