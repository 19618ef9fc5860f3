#!/bin/bash
# round 5, session 1: A/B of the interleaved MFMA/harvest order and of the guess's margin / sampling stride on the headline step
bash tools/gpu_ab_opts.sh s1 \
  "base|-|" \
  "ilv12|hashgan_amd/_lib/ab_ilv12.so|" \
  "ilv8|hashgan_amd/_lib/ab_ilv8.so|" \
  "sig4|-|--opt guess_sigma=4" \
  "sig3|-|--opt guess_sigma=3" \
  "str32|-|--opt sample_stride=32" \
  "str48|-|--opt sample_stride=48" \
  "str16|-|--opt sample_stride=16" \
  "str12|-|--opt sample_stride=12" \
  "base2|-|" 2>&1 | tee gpurun_out/s1/summary.txt
