#!/bin/bash
# usage: tools/gpu_ab_pmc.sh <tag> : instruction counters of k_select_mx for every hashgan_amd/_lib/ab_*.so
TAG=${1:-abp}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
for so in $GRAFT_REPO_ROOT/hashgan_amd/_lib/ab_*.so; do n=$(basename $so .so)
  B="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-c4-ref --kernel-timing none"
  HG_LIBRARY=$so timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/$n -o p -- $B > $OUT/$n.log 2>&1
  echo "== $n"; python $GRAFT_REPO_ROOT/tools/prof_summary.py $(find $OUT/$n -name "*.db" | head -1) | grep "k_select_mx" 
done
