#!/bin/bash
# Developer helper: PC sampling (rocprofv3 beta) of the LZ parse alone; needs a build with line tables
# (VG_LINES=1 python -m vclust_amd.build --force).  usage: pc_sample.sh [families] [method] [interval]
NF=${1:-2000}; METHOD=${2:-host_trap}; IVL=${3:-1}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pcs
rm -rf $OUT; mkdir -p $OUT
UNIT=time; [ "$METHOD" = stochastic ] && UNIT=cycles
NF=$NF REPS=2 timeout 600 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $IVL \
    --kernel-trace -d $OUT -o pcs -- python $GRAFT_REPO_ROOT/tools/micro/parse_ab.py 2>&1 | tail -5
ls -la $OUT $OUT/* | head -30
