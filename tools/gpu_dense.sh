#!/bin/bash
# usage: tools/gpu_dense.sh <tag> : the dense regime (N/8 < R <= N) -- its parity tests and the sweep shapes of VERDICT item 7
TAG=${1:-dense}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "dense_regime or long_lists or golden or python_surface" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for shape in "10000 1000000 64 500000" "10000 1000000 64 200000" "10000 1000000 64 1000000" "1000 54000 32 54000" "1000 54000 32 54000 rank_dense=2" "10000 200000 64 100000" "10000 1000000 64 500000 rank_dense_gbm=0" "10000 1000000 64 130000" "10000 1000000 64 100000" "10000 1000000 64 50000" "10000 1000000 64 50000 rank_slices=0" "10000 1000000 64 20000" "10000 1000000 64 8000"; do
  timeout 600 python tools/shape_sweep.py $shape 2>&1 | tail -1
done | tee $OUT/sweep.txt
