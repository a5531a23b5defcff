#!/bin/bash
# kernel trace (rocprofv3 --kernel-trace --stats) of ONE configuration of tools/bench_configs.py -> gpurun_out/<tag>_kernel_stats.csv (top kernels printed)
#   tools/trace_config.sh <tag> "<config name substring>"
set -u
TAG=$1; NAME=$2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$ROOT
cd /tmp
rm -rf $OUT/trace; timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace -- python $ROOT/tools/bench_configs.py "$NAME" > $OUT/trace.log 2>&1; echo "rc=$?"
cd $ROOT
python tools/rocpd_stats.py $(find $OUT/trace -name "*.db" | sort | tail -1) > $ROOT/gpurun_out/${TAG}_kernel_stats.csv
rm -rf $OUT/trace
head -40 $ROOT/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
