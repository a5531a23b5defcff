#!/bin/bash
# kernel timeline of ONE headline step under rocprofv3 --kernel-trace (the profiler serialises the graph's kernel nodes): per-launch durations in launch order
#   tools/overlap_timeline.sh <out file> [ENV=VALUE ...]
set -u
OUTF=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/ov; mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$ROOT
for kv in "$@"; do export "$kv"; done
cd /tmp
rm -rf $OUT/trace; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --in-flight 1 > $OUT/trace.log 2>&1; echo "rc=$?"
cd $ROOT
python tools/rocpd_timeline.py $(find $OUT/trace -name "*.db" | sort | tail -1) > $OUTF
rm -rf $OUT/trace
