#!/bin/bash
# register / LDS / occupancy report of every kernel in one csrc file:  tools/regs.sh conv_igemm.hip [filter]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -c visualdet3d_amd/csrc/$1 -o /tmp/regs_test.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -i "Function Name\| VGPRs:\|AGPRs\|Spill\|Occupancy" | sed 's/.*remark: [^ ]* *//' | paste - - - - - - | grep "${2:-.}" | sed "s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/_ZN12_GLOBAL__N_1[0-9]*//" | cut -c1-250
