#!/bin/bash
# usage: tools/gpu_ab.sh <tag> [bench args] : every hashgan_amd/_lib/ab_*.so on the same box, alternating, 3 rounds
TAG=${1:-ab}; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
for r in 1 2 3; do for so in hashgan_amd/_lib/ab_*.so; do n=$(basename $so .so)
  HG_LIBRARY=$PWD/$so python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-c4-ref --kernel-timing all "$@" > $OUT/${n}_$r.json 2> $OUT/${n}_$r.err
  python -c "
import json
try:
    d=json.loads(open('$OUT/${n}_$r.json').read().strip().splitlines()[-1]); k=d['kernels']
    print('%-14s r$r step %.4f parity %s fb %s select %.4f rank %.4f' % ('$n', d['ms_per_step'], d.get('parity_vs_reference_golden'), d.get('optimistic_fallbacks'), k.get('k_select_mx',{}).get('avg_ms',0), k.get('k_rank_lds',{}).get('avg_ms',0)))
except Exception as e: print('$n ERR', e, open('$OUT/${n}_$r.err').read()[-600:])
"; done; done
