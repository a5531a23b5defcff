#!/bin/bash
cd "$(dirname "$0")/.."
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O
export PYTHONPATH=$ROOT TMPDIR=/tmp
(time python -m pytest tests -m gpu -q) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_stg$i.json 2> $O/bench_stg$i.err; tail -1 $O/bench_stg$i.err
VD3D_CONV_NO_STAGGER=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_nostg$i.json 2> $O/bench_nostg$i.err; tail -1 $O/bench_nostg$i.err
done
