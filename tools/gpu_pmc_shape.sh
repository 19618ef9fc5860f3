#!/bin/bash
# usage: tools/gpu_pmc_shape.sh <tag> "Q N b R [opts]" : SQ counter passes of tools/shape_sweep.py for one shape
TAG=${1:-pmcs}; SH=$2; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set -d $OUT/pmc$i -o p -- python $GRAFT_REPO_ROOT/tools/shape_sweep.py $SH > $OUT/pmc$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
for f in $(find $OUT -name "*.db" | sort); do python tools/prof_summary.py $f; done > $OUT/summary.txt 2>&1
grep -E "k_rank_cnt|k_select_mx" $OUT/summary.txt
