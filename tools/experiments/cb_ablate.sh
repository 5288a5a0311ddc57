#!/bin/bash
# direct split-bf16 convolution ablation (make -C vae_captioning_amd/csrc cbabl): one block per variant
for n in ${CBABLS:-0 1 2 3 4 8 16 32}; do
  if [ $n = 0 ]; then L=vae_captioning_amd/lib/libvaecap.so; else L=vae_captioning_amd/lib/libvaecap_cbabl$n.so; fi
  echo "== CB_ABL=$n"
  VC_LIB=$L python tools/microbench.py convbx 2>&1 | grep -E "^conv(1_2|2_2|3_2|4_2|5_2)|^sum" | sed -e 's/| F(4x4.*//'
done
