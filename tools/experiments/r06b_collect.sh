set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; TAG=r06b
python -m pytest tests -x -q -m gpu > $OUT/${TAG}_pytest.log 2>&1; tail -3 $OUT/${TAG}_pytest.log
python bench.py --workload cfg5 --steps 30 --warmup 4 > $OUT/${TAG}_bench_cfg5.json 2>/dev/null
VC_DECODE_GRAPH=0 python bench.py --no-cpu-baseline --workload cfg5 --steps 30 --warmup 4 > $OUT/${TAG}_bench_cfg5_eager.json 2>/dev/null
VC_DECODE_SLICES=1 python bench.py --no-cpu-baseline --workload cfg5 --steps 30 --warmup 4 > $OUT/${TAG}_bench_cfg5_one_slice.json 2>/dev/null
python bench.py --workload cfg1 --graph 1 > $OUT/${TAG}_bench_cfg1.json 2>/dev/null
rm -rf /tmp/kt_cfg5
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_cfg5 -- python $ROOT/bench.py --workload cfg5 --no-cpu-baseline --steps 30 --warmup 4 > $OUT/${TAG}_cfg5_kt.log 2>&1)
python tools/rocpd_stats.py "$(find /tmp/kt_cfg5 -name "*_results.db" | head -1)" 40 > $OUT/${TAG}_cfg5_kernel_stats.md
for f in cfg5 cfg5_eager cfg5_one_slice cfg1; do python -c "import json,sys; d=json.loads(open('$OUT/${TAG}_bench_$f.json').readline()); print('$f', d['ms_per_step'], d['value'], d.get('greedy_decode'))"; done
python bench.py > $OUT/${TAG}_bench_default.json 2>/dev/null; python -c "import json; d=json.loads(open('$OUT/${TAG}_bench_default.json').readline()); print('default', d['ms_per_step'], d['value'], d['roofline']['frac'])"
