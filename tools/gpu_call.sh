#!/bin/bash
# scratch driver of one gpurun call (edited per call); outputs under gpurun_out/
cd "$(dirname "$0")/.."
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O
export PYTHONPATH=$ROOT TMPDIR=/tmp
(time python -m pytest tests -m gpu -q --deselect tests/test_stage_taps_c5_c3_gpu.py) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
(time python -m pytest tests/test_stage_taps_c5_c3_gpu.py -m gpu -q -s) > $O/pytest_taps.log 2>&1; tail -4 $O/pytest_taps.log
python tools/bench_stem.py 40 2>&1 | grep -v amdgpu | tee $O/stem_ab.txt
python tools/bench_block_fusion_bound.py 2>&1 | grep -v amdgpu | tee $O/block_fusion_bound.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_new.json 2> $O/bench_new.err; grep -o '"value": [0-9.]*, "unit"' $O/bench_new.json | head -1; tail -1 $O/bench_new.err
VD3D_STEM_WG8=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_wg8.json 2> $O/bench_wg8.err; grep -o '"value": [0-9.]*, "unit"' $O/bench_wg8.json | head -1; tail -1 $O/bench_wg8.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_new2.json 2> $O/bench_new2.err; grep -o '"value": [0-9.]*, "unit"' $O/bench_new2.json | head -1; tail -1 $O/bench_new2.err
