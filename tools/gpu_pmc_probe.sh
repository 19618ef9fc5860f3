#!/bin/bash
# usage: tools/gpu_pmc_probe.sh <tag> : VALU / SALU / LDS instruction counts of the select kernel in the probe build,
# with the drain cut at different points (probe_select = 2: no drain, 8: pushes only, 0: all)
TAG=${1:-pp}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HG_LIBRARY=$GRAFT_REPO_ROOT/hashgan_amd/_lib/libhashgan_amd_probe.so
cd /tmp
for pr in 2 8 0; do
  B="python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-large-r --no-c4-ref --kernel-timing none --opt probe_select=$pr"
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR -d $OUT/p$pr -o p -- $B > $OUT/p$pr.log 2>&1
  echo "== probe_select=$pr"; python $GRAFT_REPO_ROOT/tools/prof_summary.py $(find $OUT/p$pr -name "*.db" | head -1) | grep -E "k_select_mx3" 
done
