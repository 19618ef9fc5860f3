#!/bin/bash
# usage: tools/gpu_real_prof.sh <tag> : the real-valued ranking at the C2 shape per kernel (tools/real_prof.py, the three pair
# passes) + rocprofv3 --kernel-trace --stats of the default sequence, and the same trace at C1 (R = N: tools/c1_real_modes.py)
TAG=${1:-rp}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
{
echo "# tools/real_prof.py on MI355X (tanh of Gaussians, one-hot labels, Q=10000 N=1000000 b=64 R=5000), per call; kernels: avg ms per launch"
for m in 2 1 0; do HG_REAL_MFMA=$m timeout 300 python $GRAFT_REPO_ROOT/tools/real_prof.py 2>&1 | tail -1; done
echo; echo "# rocprofv3 --kernel-trace --stats of the same script, real_mfma=2 (filter + rescore)"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t2 -o t -- python $GRAFT_REPO_ROOT/tools/real_prof.py > $OUT/t2.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $(find $OUT/t2 -name "*.db" | head -1)
echo; echo "# rocprofv3 --kernel-trace --stats of tools/c1_real_modes.py (C1 with real-valued features: Q=1000 N=54000 b=32 R=N; its three modes)"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t1 -o t -- python $GRAFT_REPO_ROOT/tools/c1_real_modes.py > $OUT/t1.log 2>&1
python $GRAFT_REPO_ROOT/tools/prof_summary.py $(find $OUT/t1 -name "*.db" | head -1)
} > $OUT/real_path.txt 2>&1
cat $OUT/real_path.txt | cut -c1-200
