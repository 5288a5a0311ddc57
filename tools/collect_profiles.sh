#!/bin/bash
# Collect the round's profiles on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 2400 -- 'bash tools/collect_profiles.sh r02'
# Writes gpurun_out/<tag>_*; copy the summaries into profiles/ afterwards (profiles/README.md).
set -u
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
# 1. the driver's command (headline line incl. the CPU baseline leg)
python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
# 2. kernel traces of the same command (cfg4) and of cfg2: per-kernel table + per-family sum / union of dispatch intervals; cfg4 also
#    with ONE VGG stream (VC_VGG_STREAMS=1: nothing overlaps, the per-kernel durations are the kernels' own)
for WL in cfg4 cfg2 cfg3 cfg5; do   # (round 6: cfg3 and cfg5 too -- every BASELINE config's bench line follows from a tracked table)
  rm -rf /tmp/kt_$WL
  (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt_$WL -- python $ROOT/bench.py --workload $WL --no-cpu-baseline --strong-n1 0 > $OUT/${TAG}_${WL}_kt.log 2>&1)
  DB=$(find /tmp/kt_$WL -name "*_results.db" | head -1)
  python tools/rocpd_stats.py "$DB" 60 > $OUT/${TAG}_${WL}_kernel_stats.md
done
rm -rf /tmp/kt_1s
(cd /tmp && VC_VGG_STREAMS=1 rocprofv3 --kernel-trace --stats -d /tmp/kt_1s -- python $ROOT/bench.py --no-cpu-baseline --strong-n1 0 > $OUT/${TAG}_cfg4_1stream_kt.log 2>&1)
python tools/rocpd_stats.py "$(find /tmp/kt_1s -name "*_results.db" | head -1)" 60 > $OUT/${TAG}_cfg4_kernel_stats_1stream.md
# 3. PMC passes (each counter set in its own run, with --kernel-trace only; ONE VGG stream so that a kernel's counters are its own)
rm -rf /tmp/pmc
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  D=/tmp/pmc/$(echo $C | tr ' ' '_')
  (cd /tmp && VC_VGG_STREAMS=1 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --strong-n1 0 > $OUT/${TAG}_pmc_$(echo $C | cut -d' ' -f1).log 2>&1)
done
python tools/pmc_summary.py /tmp/pmc $OUT/${TAG}_traffic_cfg4.json > $OUT/${TAG}_cfg4_pmc.md
# 4. comparison points: the NHWC implicit-GEMM kernels behind layout conversions (VC_CONV_WINO=0: the checker / fallback path), one stream,
#    F(2x2,3x3) on the 56-wide layers (the round-3 preference rule)
VC_CONV_WINO=0 python bench.py --no-cpu-baseline --strong-n1 0 > $OUT/${TAG}_bench_nowino.json 2>/dev/null
VC_WINO4_MIN_COVERAGE=0.85 python bench.py --no-cpu-baseline --strong-n1 0 > $OUT/${TAG}_bench_conv3_f23.json 2>/dev/null
VC_VGG_STREAMS=1 python bench.py --no-cpu-baseline --strong-n1 0 > $OUT/${TAG}_bench_1stream.json 2>/dev/null
# 5. the step a user runs: fresh host batches through set_batch inside the timed region
python bench.py --no-cpu-baseline --strong-n1 0 --fresh-batch 4 > $OUT/${TAG}_bench_fresh_cfg4.json 2>/dev/null
python bench.py --no-cpu-baseline --workload cfg2 > $OUT/${TAG}_bench_cfg2.json 2>/dev/null
python bench.py --no-cpu-baseline --workload cfg2 --fresh-batch 4 > $OUT/${TAG}_bench_fresh_cfg2.json 2>/dev/null
# 6. secondary lines: V = 11313, variable lengths, one caption per image; the other workloads
python bench.py --no-cpu-baseline --workload cfg2 --vocab 11313 > $OUT/${TAG}_bench_cfg2_v11313.json 2>/dev/null
python bench.py --no-cpu-baseline --workload cfg2 --variable-len > $OUT/${TAG}_bench_cfg2_varlen.json 2>/dev/null
python bench.py --no-cpu-baseline --workload cfg2 --num-captions 1 > $OUT/${TAG}_bench_cfg2_nc1.json 2>/dev/null
python bench.py --workload cfg1 --graph 1 > $OUT/${TAG}_bench_cfg1.json 2>/dev/null      # (with its CPU-baseline legs: training step + greedy decode of 32 images)
python bench.py --no-cpu-baseline --workload cfg3 > $OUT/${TAG}_bench_cfg3.json 2>/dev/null
python bench.py --workload cfg5 --steps 30 --warmup 4 > $OUT/${TAG}_bench_cfg5.json 2>/dev/null   # (with the beam-search CPU-baseline leg, 8 images)
# 6b. the split-bf16 (bf16x3) mode: every dense product on the bf16 matrix pipe, reported BESIDE the f32 lines above (never instead of them)
for WL in cfg2 cfg3 cfg1; do python bench.py --no-cpu-baseline --workload $WL --precision bf16x3 > $OUT/${TAG}_bench_${WL}_bf16x3.json 2>/dev/null; done
python bench.py --no-cpu-baseline --strong-n1 0 --precision bf16x3 > $OUT/${TAG}_bench_cfg4_bf16x3.json 2>/dev/null
rm -rf /tmp/kt_bx
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt_bx -- python $ROOT/bench.py --workload cfg2 --precision bf16x3 --no-cpu-baseline > $OUT/${TAG}_cfg2_bf16x3_kt.log 2>&1)
python tools/rocpd_stats.py "$(find /tmp/kt_bx -name "*_results.db" | head -1)" 40 > $OUT/${TAG}_cfg2_bf16x3_kernel_stats.md
python tools/microbench.py gemmx 2>/dev/null | grep "^gemm" > $OUT/${TAG}_gemm_bf16x3.txt
if ls vae_captioning_amd/lib/libvaecap_bxabl*.so > /dev/null 2>&1; then bash tools/experiments/bx_ablate.sh 2>/dev/null | grep -E "^==|^gemm" > $OUT/${TAG}_gemm_bf16x3_ablation.txt; fi
# 6c. the convolution kernels of the split-bf16 mode: the direct weight gradient (adopted) per layer, its stages removed one by one
#     (make wbabl), its counters on conv3_2 (full / MFMAs on constants / MFMAs on frozen real operands: the clock is the finding); the two
#     direct forward / data-gradient kernels (not adopted) per layer; the cfg4 step in this mode kernel by kernel
python tools/microbench.py wgbx 2>/dev/null | grep -E "^conv|^sum" > $OUT/${TAG}_wgrad_bx_layers.txt
if ls vae_captioning_amd/lib/libvaecap_wbabl*.so > /dev/null 2>&1; then
  (for n in 1 2 4 7 16; do echo "== WB_ABL=$n (1 split arithmetic, 2 LDS operand reads (and with them the split), 4 staging, 7 all three, 16 all of it with the operands of the first chunk kept: real data)"
     VC_LIB=$ROOT/vae_captioning_amd/lib/libvaecap_wbabl$n.so python tools/microbench.py wgbx 2>/dev/null | grep -E "^conv|^sum" | sed -e "s/ | f32.*//"; done) > $OUT/${TAG}_wgrad_bx_ablation.txt
  (export VC_WG_ONLY=3_2
   echo "## shipped kernel"; bash tools/kernel_pmc.sh ${TAG}_wgbx_full wgrad_bx_kernel python $ROOT/tools/microbench.py wgbx
   echo; echo "## WB_ABL=7: MFMAs only, operands = constants"; VC_LIB=$ROOT/vae_captioning_amd/lib/libvaecap_wbabl7.so bash tools/kernel_pmc.sh ${TAG}_wgbx_abl7 wgrad_bx_kernel python $ROOT/tools/microbench.py wgbx
   echo; echo "## WB_ABL=16: MFMAs only, operands = the first chunk's (real data)"; VC_LIB=$ROOT/vae_captioning_amd/lib/libvaecap_wbabl16.so bash tools/kernel_pmc.sh ${TAG}_wgbx_abl16 wgrad_bx_kernel python $ROOT/tools/microbench.py wgbx) > $OUT/${TAG}_wgrad_bx_pmc.md 2>/dev/null
fi
rm -rf /tmp/kt_c4bx
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt_c4bx -- python $ROOT/bench.py --precision bf16x3 --no-cpu-baseline --strong-n1 0 > $OUT/${TAG}_cfg4_bf16x3_kt.log 2>&1)
python tools/rocpd_stats.py "$(find /tmp/kt_c4bx -name "*_results.db" | head -1)" 40 > $OUT/${TAG}_cfg4_bf16x3_kernel_stats.md
# 7. per-layer tables: F(2x2,3x3) forward / data gradient and the F(3x3,2x2) weight gradient; F(4x4,3x3) against F(2x2,3x3) per layer
#    (forward, data gradient with the float mask, data gradient with mask bits); LSTM recurrence steps (default, two workgroups per CU)
python tools/microbench.py winoab winow 2>/dev/null | grep -v amdgpu.ids > $OUT/${TAG}_wino_layers.txt
(for v in fwd dgrad bits; do echo "== python tools/experiments/wino4_try.py 64 $v"; python tools/experiments/wino4_try.py 64 $v 2>&1 | grep "^conv\|^sum"; done
 for v in fwd bits; do echo "== python tools/experiments/wino4_try.py 32 $v"; python tools/experiments/wino4_try.py 32 $v 2>&1 | grep "^conv\|^sum"; done) > $OUT/${TAG}_wino4_layers.txt
(echo "== python tools/microbench.py lstm (VC_LSTM_MODES=3,1)"; python tools/microbench.py lstm 2>&1 | grep "^lstm") > $OUT/${TAG}_lstm_steps.txt
# 7b. round 6: F(4x4,3x3) on a once-transformed input (MODE 2, vc_conv3x3_wino4v_*) against the fused kernel per layer, and inside the step
(for b in 32 64; do python tools/experiments/wino4v_try.py $b 2>&1 | grep "^conv\|^sum"; done) > $OUT/${TAG}_wino4v_layers_rerun.txt
for v in 1 0; do VC_WINO4V=$v python bench.py --no-cpu-baseline --strong-n1 0 > $OUT/${TAG}_bench_wino4v$v.json 2>/dev/null; done
VC_DECODE_GRAPH=0 python bench.py --no-cpu-baseline --workload cfg5 --steps 30 --warmup 4 > $OUT/${TAG}_bench_cfg5_eager.json 2>/dev/null
# 8. SQ counters of the Winograd forward / data-gradient kernel per layer shape (fused and MODE 2)
bash tools/sq_probe_wino.sh ${TAG} > /dev/null 2>&1
# 9. round 6: counters of the AG heads GEMM (cfg3) and of one decode round's logits GEMM (cfg5)
bash tools/kernel_pmc.sh ${TAG}_heads gemm_kernel python $ROOT/tools/microbench.py gemmshape > /dev/null 2>&1
VC_SHAPE=0,0,640,10000,512 bash tools/kernel_pmc.sh ${TAG}_declogits gemm_kernel python $ROOT/tools/microbench.py gemmshape > /dev/null 2>&1
