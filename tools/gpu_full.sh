#!/bin/bash
# usage: tools/gpu_full.sh <tag> : the whole gpu suite, then the headline step three times
TAG=${1:-full}
bash tools/gpu_tests.sh ${TAG}_tests
bash tools/gpu_ab_opts.sh $TAG "prod|-|" 2>&1 | tee gpurun_out/$TAG/summary.txt
