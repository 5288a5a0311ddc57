ROOT=$PWD
mkdir -p gpurun_out/tl
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/tl -o kt -- python $ROOT/tools/experiments/prof_beam.py > $ROOT/gpurun_out/tl/run.log 2>&1)
DB=$(ls gpurun_out/tl/*.db gpurun_out/tl/*/*.db 2>/dev/null | head -1)
echo DB=$DB
python tools/rocpd_timeline.py $DB sample_kernel 0 > gpurun_out/tl/timeline.txt 2>&1
sed -n 1,120p gpurun_out/tl/timeline.txt
tail -3 gpurun_out/tl/timeline.txt
rm -f gpurun_out/tl/*.db gpurun_out/tl/*/*.db
