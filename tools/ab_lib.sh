#!/bin/bash
# same-box A/B of the product library against another build next to it:  tools/ab_lib.sh libvd3d_hip_base.so   (headline twice each, C3, C5)
BASE=$1
for i in 1 2; do
python bench.py --no-other-configs --no-cpu-baseline 2>&1 >/dev/null | grep "rank 0" | grep -o "median.*ms/step (all: [0-9. ]*)" | sed 's/^/NEW  C2 /'
VD3D_TUNING_LIB=$BASE python bench.py --no-other-configs --no-cpu-baseline 2>&1 >/dev/null | grep "rank 0" | grep -o "median.*ms/step (all: [0-9. ]*)" | sed 's/^/BASE C2 /'
done
for c in C3 C5; do
python tools/ab_c3.py $c 2>/dev/null | tail -n 1 | cut -c1-120 | sed 's/^/NEW  /'
VD3D_TUNING_LIB=$BASE python tools/ab_c3.py $c 2>/dev/null | tail -n 1 | cut -c1-120 | sed 's/^/BASE /'
done
