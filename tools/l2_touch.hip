// l2_touch.hip -- stand-alone tool kernel (NOT part of libvd3d_hip.so): read a byte range through EVERY XCD's L2.
// Experiment for VERDICT r5 item 7 (batch-1 cross-boundary weight prefetch): tools/bench_b1_prefetch.py launches it on a side stream for layer i + 1's
// weights while layer i computes.  Block b runs on XCD b % 8 (observed placement, MI355X_MICROARCH.md): blocks b, b + 8, ... of one XCD split the range.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void __launch_bounds__(256) l2_touch_kernel(const uint4* __restrict__ p, int64_t n16, int per_xcd, uint32_t* sink) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    uint32_t acc = 0;
    for (int64_t i = (int64_t)j * 256 + threadIdx.x; i < n16; i += (int64_t)per_xcd * 256) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u && xcd == 9) *sink = acc;      // (never true: keeps the loads)
}

extern "C" int l2_touch(const void* ptr, int64_t bytes, int per_xcd, void* sink, void* stream) {
    hipLaunchKernelGGL(l2_touch_kernel, dim3(8 * per_xcd), dim3(256), 0, (hipStream_t)stream, (const uint4*)ptr, bytes / 16, per_xcd, (uint32_t*)sink);
    return (int)hipGetLastError();
}
