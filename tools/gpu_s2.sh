#!/bin/bash
# round 5, session 2: push / emit diet of k_select_mx3 (straight-line push with the ballot as exec mask, 24-bit multiply in the emit's
# decode), 12-byte queue entries at one address, MFMA/harvest interleave
mkdir -p gpurun_out/s2
bash tools/gpu_ab_opts.sh s2 \
  "new|-|" \
  "i8|hashgan_amd/_lib/ab_i8.so|" \
  "q12|hashgan_amd/_lib/ab_q12.so|" \
  "i8q|hashgan_amd/_lib/ab_i8q.so|" \
  "new2|-|" 2>&1 | tee gpurun_out/s2/summary.txt
HG_LIBRARY=hashgan_amd/_lib/ab_i8q.so timeout 900 python -m pytest tests/test_hip_parity.py tests/test_full_size_properties.py -m gpu -q -x -k "golden or stages or lost or bursts or far or long_codes or odd or ap_through or full or fused_step" > gpurun_out/s2/pytest_i8q.log 2>&1; echo "pytest i8q rc=$?"; tail -3 gpurun_out/s2/pytest_i8q.log
