#!/bin/bash
cd "$(dirname "$0")/.."
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O
export PYTHONPATH=$ROOT TMPDIR=/tmp
python -m pytest tests/test_dcn_gpu.py tests/test_km3d_gpu.py -m gpu -q -x 2>&1 | tail -6
rm -f $O/dcn_lstage.txt
for i in 1 2; do
python tools/bench_dcn.py fp16 20 2>&1 | grep "ida" | sed 's/$/  staged logits/' | tee -a $O/dcn_lstage.txt
VD3D_DCN_NO_LSTAGE=1 python tools/bench_dcn.py fp16 20 2>&1 | grep "ida" | sed 's/$/  strided logits/' | tee -a $O/dcn_lstage.txt
done
python tools/bench_configs.py "fp16 (as BASELINE" 2>&1 | grep img/s | sed 's/$/  staged/' | tee -a $O/dcn_lstage.txt
VD3D_DCN_NO_LSTAGE=1 python tools/bench_configs.py "fp16 (as BASELINE" 2>&1 | grep img/s | sed 's/$/  strided/' | tee -a $O/dcn_lstage.txt
python tools/bench_configs.py "fp16 (as BASELINE" 2>&1 | grep img/s | sed 's/$/  staged/' | tee -a $O/dcn_lstage.txt
