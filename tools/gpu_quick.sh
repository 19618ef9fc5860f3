#!/bin/bash
# usage: tools/gpu_quick.sh <tag> : the fast loop -- core parity tests, then one bench line with every kernel timed
TAG=${1:-q}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-c4-ref --kernel-timing all > $OUT/all_$i.json 2> $OUT/all_$i.err; done
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-c4-ref > $OUT/plain.json 2> $OUT/plain.err
for f in $OUT/*.json; do echo "== $f"; python -c "
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print(round(d['ms_per_step'],4), d.get('parity_vs_reference_golden'), {k:v['avg_ms'] for k,v in d.get('kernels',{}).items()})
except Exception as e: print('ERR', e); print(open('$f'.replace('.json','.err')).read()[-1500:])
"; done
