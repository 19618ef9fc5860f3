#!/bin/bash
# One gpurun call: gpu tests, driver-style bench runs, forced-sharded (1-rank RCCL) bench, kernel trace.
# usage: tools/gpu_check.sh <tag> [pytest-args]
set -u
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q ${2:-} > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-c4-ref > $OUT/bench_$i.json 2> $OUT/bench_$i.err; done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err
HG_BENCH_FORCE_SHARDED=1 timeout 300 python bench.py --gpus 1 --workload c2 --steps 20 --warmup 5 > $OUT/bench_sharded1.json 2> $OUT/bench_sharded1.err
timeout 600 python bench.py --gpus 1 --workload c4 --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-c4-ref > $OUT/bench_c4.json 2> $OUT/bench_c4.err
HG_BENCH_FORCE_SHARDED=1 timeout 600 python bench.py --gpus 1 --workload c4 --steps 10 --warmup 3 > $OUT/bench_c4_sharded1.json 2> $OUT/bench_c4_sharded1.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-c4-ref > $GRAFT_REPO_ROOT/$OUT/prof.log 2>&1)
find $OUT/prof -name "*.db" | head -3 | while read f; do python tools/prof_summary.py $f > $OUT/prof_summary.txt 2>&1; done
for f in $OUT/bench_*.json; do echo "== $f"; python -c "
import json,sys
try:
    d=json.loads(open('$f').read().strip().splitlines()[-1])
    print({k:d.get(k) for k in ('value','ms_per_step','parity_vs_reference_golden','optimistic_fallbacks','kernels')})
    for k in ('h2d_inclusive',):
        if k in d: print(k, d[k])
except Exception as e: print('ERR', e); print(open('$f'.replace('.json','.err')).read()[-1500:])
"; done
cat $OUT/prof_summary.txt | head -30
