#!/bin/bash
# usage: tools/gpu_ab_real.sh : real-path timing for every hashgan_amd/_lib/ab_*.so
for so in hashgan_amd/_lib/ab_*.so; do echo "== $so"; HG_LIBRARY=$PWD/$so HG_REAL_MFMA=2 timeout 600 python tools/real_prof.py 2>&1 | tail -1 | cut -c1-330; done
