#!/bin/bash
# usage: tools/gpu_spin_ab.sh <tag> : what the HIP events of each --kernel-timing mode cost the step, interleaved runs
TAG=${1:-evcost}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for i in 1 2 3; do for s in none pair-passes all; do
python bench.py --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-large-r --no-c4-ref --kernel-timing $s > $OUT/${s}_$i.json 2> $OUT/${s}_$i.err
python -c "
import json
d=json.loads(open('$OUT/${s}_$i.json').read().strip().splitlines()[-1]); print('$s', round(d['ms_per_step'],4), d.get('parity_vs_reference_golden'))"
done; done
