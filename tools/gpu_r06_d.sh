#!/bin/bash
OUT=gpurun_out/r06d; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_real.py -m gpu -q -x > $OUT/pytest_real.log 2>&1; echo "real tests rc=$?"; tail -3 $OUT/pytest_real.log
timeout 300 python tools/literal_breakdown.py cifar nus c2 > $OUT/literal_breakdown.txt 2>&1; echo "breakdown rc=$?"
timeout 300 python tools/real_prof.py > $OUT/real_prof.txt 2>&1
