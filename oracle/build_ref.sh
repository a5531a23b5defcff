#!/bin/bash
# ORACLE tooling: build host libraries from the REFERENCE'S OWN native sources (device functions only), reading
# them where they lie under /root/reference.  Outputs go to oracle/_ref/ (git-ignored, travels to the GPU box).
# The reference's build system (setup.py + nvcc) is not used: nvcc is absent and the code is CUDA-only, so only the
# arithmetic device functions are compiled for the host through oracle/ref_native_shim.h.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${VD3D_REFERENCE_ROOT:-/root/reference}"
OUT="$HERE/_ref"
mkdir -p "$OUT"
IOU="$REF/visualDet3D/networks/lib/ops/iou3d/src/iou3d_kernel.cu"
DCN="$REF/visualDet3D/networks/lib/ops/dcn/src/cuda/deform_conv_cuda_kernel.cu"
if [ ! -f "$IOU" ]; then echo "reference tree not present: skipping oracle/_ref build"; exit 0; fi
# iou3d: Point/cross/intersection/box_overlap/iou_bev (lines 1-221) + iou_normal (295-303)
( sed -n '1,221p;295,303p' "$IOU"; cat "$HERE/ref_iou3d_wrap.inc" ) | \
  g++ -O2 -fPIC -shared -ffp-contract=off -x c++ -include "$HERE/ref_native_shim.h" - -o "$OUT/libiou3d_ref.so"
echo "built $OUT/libiou3d_ref.so"
if [ -f "$HERE/ref_dcn_wrap.inc" ]; then
  # DCN: bilinear samplers + im2col kernels (v1: 84-243 region, v2: 467-633 region), includes stripped
  ( sed -n "$(cat "$HERE/ref_dcn_lines.txt")" "$DCN"; cat "$HERE/ref_dcn_wrap.inc" ) | \
    g++ -O2 -fPIC -shared -ffp-contract=off -x c++ -include "$HERE/ref_native_shim.h" -include "$HERE/ref_dcn_shim.h" - -o "$OUT/libdcn_ref.so"
  echo "built $OUT/libdcn_ref.so"
fi
