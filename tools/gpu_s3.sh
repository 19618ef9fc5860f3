#!/bin/bash
# round 5, session 3: the re-laid-out queue entries (query tag, ring index, leaner emit) as the production build + the whole gpu suite
mkdir -p gpurun_out/s3
bash tools/gpu_ab_opts.sh s3 "prod|-|" "prod2|-|" 2>&1 | tee gpurun_out/s3/summary.txt
bash tools/gpu_tests.sh s3_tests
