#!/bin/bash
# split-bf16 GEMM main-loop ablation (make -C vae_captioning_amd/csrc bxabl): one line per variant and shape
for n in ${BXABLS:-0 1 2 4 6 16 32}; do
  if [ $n = 0 ]; then L=vae_captioning_amd/lib/libvaecap.so; else L=vae_captioning_amd/lib/libvaecap_bxabl$n.so; fi
  echo "== BX_ABL=$n"
  VC_LIB=$L python tools/microbench.py gemmx 2>&1 | grep -E "27520 x  10000|512 x  10000 x  27520|4096 x   4096"
done
