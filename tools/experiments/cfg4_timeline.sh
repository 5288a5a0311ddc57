# launch-order timeline of one cfg4 step, every dispatch (tools/rocpd_timeline.py): gpurun_out/tl4/timeline.txt
ROOT=$PWD
mkdir -p gpurun_out/tl4
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/tl4 -o kt -- python $ROOT/bench.py --no-cpu-baseline --strong-n1 0 --steps 6 --warmup 2 > $ROOT/gpurun_out/tl4/run.log 2>&1)
DB=$(ls gpurun_out/tl4/*.db gpurun_out/tl4/*/*.db 2>/dev/null | head -1)
python tools/rocpd_timeline.py $DB preprocess_kernel ${1:-0} > gpurun_out/tl4/timeline.txt 2>&1
tail -2 gpurun_out/tl4/timeline.txt
rm -f gpurun_out/tl4/*.db gpurun_out/tl4/*/*.db
