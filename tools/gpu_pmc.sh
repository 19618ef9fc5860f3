#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> [bench args...] : SQ counter passes + kernel trace of bench.py (no cpu baseline)
#        HG_PMC_CMD="python tools/shape_sweep.py Q N b R" tools/gpu_pmc.sh <tag> : the same passes over another command (relative to the repo root)
TAG=${1:-pmc}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B0="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-real --no-sorted --no-large-r --no-c4-ref --no-configs --no-literal --no-pipeline-extras --kernel-timing none $*"
if [ -n "$HG_PMC_CMD" ]; then B=$(echo "$HG_PMC_CMD" | sed "s#python tools/#python $GRAFT_REPO_ROOT/tools/#"); else B=$B0; fi
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $B > $OUT/trace.log 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set -d $OUT/pmc$i -o p -- $B > $OUT/pmc$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
for f in $(find $OUT -name "*.db" | sort); do python tools/prof_summary.py $f; done > $OUT/summary.txt 2>&1
python tools/pmc_traffic.py $(find $OUT/pmc3 -name "*.db" | head -1) $(find $OUT/pmc4 -name "*.db" | head -1) $OUT/traffic.json > $OUT/traffic.txt 2>&1
grep -E "k_select_mx|k_rank_|== " $OUT/summary.txt | head -80
