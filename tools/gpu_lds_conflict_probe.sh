#!/bin/bash
# usage: tools/gpu_lds_conflict_probe.sh <tag> : where k_select_mx3's LDS bank conflicts come from -- SQ_LDS_BANK_CONFLICT /
# SQ_LDS_IDX_ACTIVE / SQ_INSTS_LDS of the probe build with the drain cut at different points
# (probe_select = 2: no drain at all -- staging + A-fragment reads only; 8: pushes, no emit; 4: emit without ring stores; 0: all)
TAG=${1:-ldsp}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp HG_LIBRARY=$GRAFT_REPO_ROOT/hashgan_amd/_lib/libhashgan_amd_probe.so
cd /tmp
for pr in 2 8 4 0; do
  B="python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-large-r --no-c4-ref --kernel-timing none --opt probe_select=$pr"
  timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d $OUT/p$pr -o p -- $B > $OUT/p$pr.log 2>&1
  echo "== probe_select=$pr"; python $GRAFT_REPO_ROOT/tools/prof_summary.py $(find $OUT/p$pr -name "*.db" | head -1) | grep -E "k_select_mx3"
done
