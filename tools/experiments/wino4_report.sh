#!/bin/bash
# EXPERIMENT report (HISTORY.md section 4g) -> profiles/r03_wino4_experiment.txt.  Build first (in the container):
#   make -C vae_captioning_amd/csrc wino4; for n in 1 8 32 64 128 15 256; do make -C vae_captioning_amd/csrc wino4 W4FLAGS=-DW4_ABL=$n && \
#     cp vae_captioning_amd/lib/libvaecap_wino4.so vae_captioning_amd/lib/libvaecap_wino4_abl$n.so; done; make -C vae_captioning_amd/csrc wino4
#   hipcc -w --offload-arch=gfx950 -O3 -std=c++17 tools/probes/mfma16_f43.hip -o build/probes/mfma16_f43   (and mfma_specialised)
L=vae_captioning_amd/lib
echo "== tools/probes/mfma16_f43: VALU operations and ds_read_b128 pinned between v_mfma_f32_16x16x4_f32 (two workgroups of four waves per CU)"
./build/probes/mfma16_f43
echo; echo "== tools/probes/mfma_specialised: an MFMA-only wave and a VALU-only wave on the same SIMD"
./build/probes/mfma_specialised
echo; echo "== tools/experiments/wino4_try.py 64: forward (bias + ReLU), 64 images, F(2x2,3x3) = the library's kernel"
VC_LIB=$L/libvaecap_wino4.so python tools/experiments/wino4_try.py 64 2>&1 | grep "^conv\|^sum"
echo; echo "== data gradient (ReLU mask from the float activation in both kernels)"
VC_LIB=$L/libvaecap_wino4.so python tools/experiments/wino4_try.py 64 dgrad 2>&1 | grep "^conv\|^sum"
echo; echo "== F(4x4,3x3) forward with parts of the main loop removed (W4_ABL; results wrong, timing only)"
for n in 1 8 32 64 128 15 256; do
  case $n in 1) t="no transform arithmetic";; 8) t="no staging";; 32) t="no global loads (LDS writes kept)";; 64) t="no patch staging";; 128) t="no weight staging (pieces 1..4)";;
    15) t="MFMAs only";; 256) t="patch loads contiguous over the lanes";; esac
  echo "-- W4_ABL=$n: $t"
  W4_ONLY=conv1_2,conv2_2,conv4_2 VC_LIB=$L/libvaecap_wino4_abl$n.so python tools/experiments/wino4_try.py 64 2>&1 | grep "^conv" | cut -c1-24,54-
done
