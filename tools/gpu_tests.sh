#!/bin/bash
# usage: tools/gpu_tests.sh <tag> [extra pytest args]
TAG=${1:-t}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q "$@" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/pytest.log | tail -30
