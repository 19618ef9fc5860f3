#!/bin/bash
# usage: tools/gpu_real.sh <tag> : real-valued path -- parity tests, then the C2-shape timing for each real_mfma mode
TAG=${1:-real}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_real.py tests/test_hip_parity.py -m gpu -x -q -k "real or Real" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
for m in 2 1; do HG_REAL_MFMA=$m timeout 600 python tools/real_prof.py 2>&1 | tail -2; done | tee $OUT/prof.txt
