#!/bin/bash
# usage: tools/gpu_probe.sh <tag> : probe-build timing breakdown of the select kernel
TAG=${1:-p}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export HG_LIBRARY=$PWD/hashgan_amd/_lib/libhashgan_amd_probe.so
run() { name=$1; shift; python bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-c4-ref "$@" > $OUT/$name.json 2> $OUT/$name.err; python -c "
import json
try:
    d=json.loads(open('$OUT/$name.json').read().strip().splitlines()[-1]); print('%-28s step %.4f  '%('$name', d['ms_per_step']), {k:v['avg_ms'] for k,v in d.get('kernels',{}).items()}, 'fallbacks', d['optimistic_fallbacks'])
except Exception as e: print('$name ERR', e, open('$OUT/$name.err').read()[-800:])
"; }
run compact0 --opt compact_records=0
run compact0_pad4k --opt compact_records=0 --opt lds_pad=4096
run compact1
run compact1_nodrain --opt probe_select=2
run compact1_noemit --opt probe_select=8
run compact1_nostore --opt probe_select=4
run compact0_nodrain --opt compact_records=0 --opt probe_select=2
run compact0_noemit --opt compact_records=0 --opt probe_select=8
run compact0_nostore --opt compact_records=0 --opt probe_select=4
