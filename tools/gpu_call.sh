#!/bin/bash
cd "$(dirname "$0")/.."
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O
export PYTHONPATH=$ROOT TMPDIR=/tmp
(time python -m pytest tests -m gpu -q) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python bench.py > $O/bench_line.json 2> $O/bench.err; tail -1 $O/bench.err
