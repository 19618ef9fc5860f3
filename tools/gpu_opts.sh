#!/bin/bash
# usage: tools/gpu_opts.sh "opt=val [opt=val]" ... : bench.py step time and kernels per option set
for o in "$@"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-c4-ref --kernel-timing all $args | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-36s' % '$o', round(d['ms_per_step'],4), d['parity_vs_reference_golden'], d['optimistic_fallbacks'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
done
