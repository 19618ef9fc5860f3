#!/bin/bash
# usage: tools/gpu_evidence_r06.sh <tag> [quick] : round 6's evidence set -- the driver-style bench line, kernel trace + PMC passes of the headline step
# (C2), kernel traces of C1 / C3 / C5, the real-valued call's counter passes, the new-context probes; `quick` stops after the C2 passes
set -u
TAG=${1:-ev6}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err
for i in 1 2; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-large-r --no-c4-ref --no-configs --no-literal > $OUT/bench_$i.json 2> $OUT/bench_$i.err; done
bash tools/gpu_pmc.sh ${TAG}_pmc > $OUT/pmc.log 2>&1
python tools/prof_summary.py $(find gpurun_out/${TAG}_pmc/trace -name "*.db" | head -1) > $OUT/kernel_trace_stats.txt 2>&1
cp gpurun_out/${TAG}_pmc/summary.txt $OUT/pmc_summary.txt; cp gpurun_out/${TAG}_pmc/traffic.json $OUT/traffic.json
timeout 300 python tools/new_context_probe.py --contexts 12 > $OUT/new_context_probe.txt 2>&1
timeout 300 python tools/new_context_probe.py --contexts 12 --sorted --keep-one >> $OUT/new_context_probe.txt 2>&1
if [ "${2:-}" != "quick" ]; then
bash tools/gpu_pmc.sh ${TAG}_pmc_c5 --workload c5 > $OUT/pmc_c5.log 2>&1
cp gpurun_out/${TAG}_pmc_c5/summary.txt $OUT/pmc_c5_summary.txt
python tools/pmc_traffic.py $(find gpurun_out/${TAG}_pmc_c5/pmc3 -name "*.db" | head -1) $(find gpurun_out/${TAG}_pmc_c5/pmc4 -name "*.db" | head -1) $OUT/traffic_with_c5.json c5 $OUT/traffic.json > $OUT/traffic_c5.txt 2>&1
bash tools/gpu_config_traces.sh ${TAG}_cfg > $OUT/cfg.log 2>&1
cp gpurun_out/${TAG}_cfg/c1_kernel_trace_stats.txt gpurun_out/${TAG}_cfg/c3_kernel_trace_stats.txt gpurun_out/${TAG}_cfg/c5_kernel_trace_stats.txt $OUT/ 2>/dev/null
timeout 600 python tools/replica_shard_timing.py > $OUT/replica_shard_timing.txt 2>&1
bash tools/gpu_real_pmc.sh ${TAG}_real > $OUT/real_pmc.txt 2>&1
fi
for f in $OUT/bench_*.json; do python -c "
import json
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],4), round(d['value']), d['parity_vs_reference_golden'], d['roofline']['frac'], d['roofline'].get('traffic'))"; done
head -8 $OUT/kernel_trace_stats.txt
# the raw rocprofv3 databases stay on the box (gpurun merges at most 64 MiB back): their summaries above are what is kept
find gpurun_out/${TAG}_pmc gpurun_out/${TAG}_pmc_c5 gpurun_out/${TAG}_cfg gpurun_out/${TAG}_real -name "*.db" -delete 2>/dev/null
