#!/bin/bash
OUT=gpurun_out/efence; mkdir -p $OUT
for m in 1 2 3; do
HG_EFENCE=$m timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "two_halves or blind or batch_after or recycles or surface or independent or golden" > $OUT/efence${m}_parity.log 2>&1; echo "efence $m parity rc=$?"; tail -2 $OUT/efence${m}_parity.log
HG_EFENCE=$m timeout 900 python -m pytest tests/test_hip_real.py -m gpu -q -x -k "filter_and_rescore or bucket_of_26 or generic" > $OUT/efence${m}_real.log 2>&1; echo "efence $m real rc=$?"; tail -2 $OUT/efence${m}_real.log
done
