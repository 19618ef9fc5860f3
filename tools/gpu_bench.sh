#!/bin/bash
# usage: tools/gpu_bench.sh <tag> : driver-style bench x3 (graph on), eager, all-kernel timing
TAG=${1:-b}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-c4-ref > $OUT/graph_$i.json 2> $OUT/graph_$i.err; done
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-c4-ref --opt step_graph=0 > $OUT/eager.json 2> $OUT/eager.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-c4-ref --kernel-timing all > $OUT/graph_all.json 2> $OUT/graph_all.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-c4-ref --kernel-timing none > $OUT/graph_none.json 2> $OUT/graph_none.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-c4-ref --kernel-timing all --opt step_graph=0 > $OUT/eager_all.json 2> $OUT/eager_all.err
for f in $OUT/*.json; do echo "== $f"; python -c "
import json
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print(round(d['ms_per_step'],4), d.get('parity_vs_reference_golden'), d.get('step_accounting'), {k:v['avg_ms'] for k,v in d.get('kernels',{}).items()})
except Exception as e: print('ERR', e); print(open('$f'.replace('.json','.err')).read()[-1500:])
"; done
