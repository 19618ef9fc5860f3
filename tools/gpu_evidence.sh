#!/bin/bash
# usage: tools/gpu_evidence.sh <tag> : the evidence set of a round -- gpu tests, the driver-style bench line, kernel trace + PMC passes
set -u
TAG=${1:-ev}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -2 $OUT/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-large-r --no-c4-ref > $OUT/bench_$i.json 2> $OUT/bench_$i.err; done
bash tools/gpu_pmc.sh ${TAG}_pmc > $OUT/pmc.log 2>&1
python tools/prof_summary.py $(find gpurun_out/${TAG}_pmc/trace -name "*.db" | head -1) > $OUT/kernel_trace_stats.txt 2>&1
cp gpurun_out/${TAG}_pmc/summary.txt $OUT/pmc_summary.txt; cp gpurun_out/${TAG}_pmc/traffic.json $OUT/traffic.json
for f in $OUT/bench_*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],4), round(d['value']), d['parity_vs_reference_golden'], d['roofline']['frac'], d['roofline'].get('traffic'))"; done
head -12 $OUT/kernel_trace_stats.txt
