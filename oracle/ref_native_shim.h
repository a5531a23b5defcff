// ORACLE build shim (test infrastructure): lets g++ compile DEVICE FUNCTIONS of the reference's CUDA sources for
// the host, straight from /root/reference (nothing is copied into the repo; see oracle/build_ref.sh).
#pragma once
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <cmath>
using std::max;
using std::min;
#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
