#!/bin/bash
# usage: tools/gpu_real_pmc.sh <tag> : kernel trace + SQ / cache counter passes of tools/real_prof.py (the real-valued call at 10k x 1M x 64, R = 5000)
TAG=${1:-realpmc}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="python $GRAFT_REPO_ROOT/tools/real_prof.py"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $B > $OUT/trace.log 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set -d $OUT/pmc$i -o p -- $B > $OUT/pmc$i.log 2>&1
done
cd $GRAFT_REPO_ROOT
for f in $(find $OUT -name "*.db" | sort); do python tools/prof_summary.py $f; done > $OUT/summary.txt 2>&1
grep -E "k_real_|k_ap|== " $OUT/summary.txt | head -400
