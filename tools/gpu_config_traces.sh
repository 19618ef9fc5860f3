#!/bin/bash
# usage: tools/gpu_config_traces.sh <tag> : rocprofv3 --kernel-trace --stats of the step at the other BASELINE configurations
# (C1, C3, C5: bench.py --workload cN, side legs off) -> gpurun_out/<tag>/<cN>_kernel_trace_stats.txt
TAG=${1:-cfg}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for wl in c1 c3 c5; do
  B="python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-large-r --no-c4-ref --no-configs --no-literal --no-pipeline-extras --kernel-timing none"
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_$wl -o t -- $B > $OUT/trace_$wl.log 2>&1
  ( echo "# rocprofv3 --kernel-trace --stats -- bench.py --workload $wl --steps 20 --warmup 5 --kernel-timing none (side legs off)"; grep "^{" $OUT/trace_$wl.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('# bench line: ms_per_step %.4f (min %.4f median %.4f), parity %s, untimed steps %s' % (d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_median'], d['parity_vs_reference_golden'], d['untimed_steps']))"
    python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $(find $OUT/trace_$wl -name "*.db" | head -1) ) > $OUT/${wl}_kernel_trace_stats.txt 2>&1
  head -14 $OUT/${wl}_kernel_trace_stats.txt
done
