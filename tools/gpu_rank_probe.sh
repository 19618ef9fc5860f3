#!/bin/bash
# usage: tools/gpu_rank_probe.sh <tag> : how k_rank_cnt's time depends on the number of blocks in flight (Q sweep at C2's N, R),
# its phase profile (needs hashgan_amd/_lib/ab_rankprof.so from tools/ab_build.sh rankprof "-DHG_RANK_PROFILE=1")
TAG=${1:-rankprobe}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for Q in 256 512 1280 2560 5120 10000; do python tools/shape_sweep.py $Q 1000000 64 5000; done > $OUT/q_sweep.txt 2>&1
[ -f hashgan_amd/_lib/ab_rankprof.so ] && HG_LIBRARY=hashgan_amd/_lib/ab_rankprof.so python tools/rank_phase_profile.py > $OUT/phase_profile.txt 2>&1
cat $OUT/q_sweep.txt $OUT/phase_profile.txt
