#!/bin/bash
# usage: tools/gpu_probe3.sh <tag> : probe-build timing breakdown of k_select_mx3 (no drain / no emit / no ring writes) + PMC passes
TAG=${1:-p3}; OUT=gpurun_out/$TAG; mkdir -p $OUT
run() { lib=$1; name=$2; shift 2; HG_LIBRARY=$lib python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-c4-ref "$@" > $OUT/$name.json 2> $OUT/$name.err; python -c "
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('%-22s step %.4f  '%('$name', d['ms_per_step']), {k:v['avg_ms'] for k,v in d.get('kernels',{}).items()}, 'fallbacks', d['optimistic_fallbacks'])
except Exception as e: print('$name ERR', e, open('$OUT/$name.err').read()[-800:])
"; }
P=$PWD/hashgan_amd/_lib/libhashgan_amd_probe.so
run $P full
run $P nodrain --opt probe_select=2
run $P noemit --opt probe_select=8
run $P noringwrite --opt probe_select=4
run $P mx_full --opt select_packed=1
run $P mx_nodrain --opt select_packed=1 --opt probe_select=2
