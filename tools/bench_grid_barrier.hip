// tools/bench_grid_barrier.hip -- what does a layer boundary cost on MI355X: a kernel boundary inside a hipGraph against a device-scope
// barrier inside ONE persistent kernel?  (the batch-1 "persistent runner" question, measured instead of argued.)
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/bench_grid_barrier tools/bench_grid_barrier.hip      (cross-compiles without a GPU)
//   tools/bench_grid_barrier [layers=32] [kb_per_workgroup=16]
//
// The "layer" is the communication pattern of a convolution layer at batch 1 without its arithmetic: workgroup w (one per CU, 256 of them)
// reads the segment its LEFT and RIGHT neighbours wrote in the previous layer (its halo) plus its own, and writes its own segment of the next
// buffer.  Neighbouring workgroup ids sit on DIFFERENT XCDs (round-robin dispatch), so every layer's reads cross XCDs: the data must leave the
// writer's L2 and the reader's L2 must not serve a stale line.
//   A. `layers` launches of that layer as kernel nodes of one hipGraph (what the detectors' hipGraph replay does today);
//   B. ONE persistent launch (256 workgroups) with a device-scope barrier between layers: release fence (L2 write-back) + atomic arrive +
//      spin + acquire fence (L2 invalidate).  The spin gives up after ~2 s (no hung box) and the program says so;
//   C. B without the data (barrier only), D. A without the data (empty kernels): the floors.
// Every variant checks its result on the host.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kThreads = 256;

__device__ __forceinline__ void layer_body(const float* __restrict__ in, float* __restrict__ out, int w, int nwg, int n_per) {
    const float* me = in + (size_t)w * n_per;
    const float* le = in + (size_t)((w + nwg - 1) % nwg) * n_per;
    const float* ri = in + (size_t)((w + 1) % nwg) * n_per;
    float* dst = out + (size_t)w * n_per;
    for (int i = threadIdx.x * 4; i < n_per; i += kThreads * 4) {
        const float4 a = *(const float4*)(me + i), b = *(const float4*)(le + i), c = *(const float4*)(ri + i);
        float4 o;
        o.x = (a.x + b.x + c.x) * (1.0f / 3.0f) + 1.0f;
        o.y = (a.y + b.y + c.y) * (1.0f / 3.0f) + 1.0f;
        o.z = (a.z + b.z + c.z) * (1.0f / 3.0f) + 1.0f;
        o.w = (a.w + b.w + c.w) * (1.0f / 3.0f) + 1.0f;
        *(float4*)(dst + i) = o;
    }
}

__global__ void __launch_bounds__(kThreads) layer_kernel(const float* in, float* out, int n_per) { layer_body(in, out, blockIdx.x, gridDim.x, n_per); }
__global__ void __launch_bounds__(kThreads) empty_kernel(unsigned* c) { if (threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *c = 1; }

// device-scope barrier: every workgroup of the grid must be resident (grid <= CUs, one workgroup per CU)
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, unsigned* failed, int nap) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();                                           // release: this workgroup's stores leave its XCD's L2
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (nap) __builtin_amdgcn_s_sleep(1);                      // (nap = 0: poll flat out)
            if (++spins > 40000000L) { ok = false; *failed = 1; break; }    // ~2 s: give up rather than hang the box
        }
        __threadfence();                                           // acquire: later loads must not hit stale lines of this XCD's L2
    }
    __syncthreads();
    return ok;
}

__global__ void __launch_bounds__(kThreads) persistent_kernel(float* a, float* b, unsigned* counter, unsigned* failed, int layers, int n_per, int with_data, int nap) {
    const int w = blockIdx.x, nwg = gridDim.x;
    for (int l = 0; l < layers; ++l) {
        if (with_data) layer_body((l & 1) ? b : a, (l & 1) ? a : b, w, nwg, n_per);
        if (!grid_barrier(counter, (unsigned)(l + 1) * (unsigned)nwg, failed, nap)) return;
    }
}

static double reference(int layers) {        // every element starts at 0: x -> x + 1 per layer (the mean of three equal values)
    return (double)layers;
}

int main(int argc, char** argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 32;
    const int kb = argc > 2 ? atoi(argv[2]) : 16;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int nwg = prop.multiProcessorCount;
    const int n_per = kb * 1024 / 4;
    const size_t n = (size_t)nwg * n_per;
    printf("%s: %d CUs; %d layers; %d KB per workgroup and layer (%.1f MB per buffer)\n", prop.name, nwg, layers, kb, n * 4 / 1e6);
    float *a, *b;
    unsigned *counter, *failed;
    CK(hipMalloc(&a, n * 4));
    CK(hipMalloc(&b, n * 4));
    CK(hipMalloc(&counter, 64));
    CK(hipMalloc(&failed, 64));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    std::vector<float> host(n);
    auto check = [&](const char* what, float* final_buf, bool with_data) {
        if (!with_data) return;
        CK(hipMemcpy(host.data(), final_buf, n * 4, hipMemcpyDeviceToHost));
        double worst = 0;
        for (size_t i = 0; i < n; ++i) { double d = host[i] - reference(layers); if (d < 0) d = -d; if (d > worst) worst = d; }
        printf("    %s: max |result - %d| = %.3g %s\n", what, layers, worst, worst < 1e-3 ? "(ok)" : "(WRONG: a stale line was read)");
    };
    auto time_it = [&](auto&& fn, int reps) {
        fn();
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) fn();
        CK(hipStreamSynchronize(s));
        return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    };
    for (int with_data = 1; with_data >= 0; --with_data) {
        // ---- A / D: a hipGraph of `layers` kernel nodes ----
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        CK(hipMemsetAsync(a, 0, n * 4, s));
        for (int l = 0; l < layers; ++l) {
            if (with_data) hipLaunchKernelGGL(layer_kernel, dim3(nwg), dim3(kThreads), 0, s, (l & 1) ? b : a, (l & 1) ? a : b, n_per);
            else hipLaunchKernelGGL(empty_kernel, dim3(nwg), dim3(kThreads), 0, s, counter);
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const double tg = time_it([&] { CK(hipGraphLaunch(ge, s)); }, 50);
        check("hipGraph", (layers & 1) ? b : a, with_data);
        // the memset node alone, to subtract
        hipGraph_t g0;
        hipGraphExec_t ge0;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        CK(hipMemsetAsync(a, 0, n * 4, s));
        CK(hipStreamEndCapture(s, &g0));
        CK(hipGraphInstantiate(&ge0, g0, nullptr, nullptr, 0));
        const double t0 = time_it([&] { CK(hipGraphLaunch(ge0, s)); }, 50);
        printf("%s hipGraph of %d kernel nodes: %8.1f us per replay = %6.2f us per layer (memset-only graph %.1f us subtracted)\n",
               with_data ? "A. layers as a" : "D. EMPTY kernels, a", layers, tg, (tg - t0) / layers, t0);
        // ---- B / C: one persistent launch with device-scope barriers ----
        const double tm = time_it([&] {
            CK(hipMemsetAsync(a, 0, n * 4, s));
            CK(hipMemsetAsync(counter, 0, 64, s));
            CK(hipMemsetAsync(failed, 0, 64, s));
        }, 50);
        for (int nap = 1; nap >= 0; --nap) {
            const double tp = time_it([&] {
                CK(hipMemsetAsync(a, 0, n * 4, s));
                CK(hipMemsetAsync(counter, 0, 64, s));
                CK(hipMemsetAsync(failed, 0, 64, s));
                hipLaunchKernelGGL(persistent_kernel, dim3(nwg), dim3(kThreads), 0, s, a, b, counter, failed, layers, n_per, with_data, nap);
            }, 50);
            unsigned f = 0;
            CK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost));
            printf("%s ONE persistent launch, %d device-scope barriers (%s): %8.1f us = %6.2f us per layer (three memsets %.1f us subtracted)%s\n",
                   with_data ? "B. layers inside" : "C. barriers only,", layers, nap ? "s_sleep while polling" : "flat-out polling", tp, (tp - tm) / layers, tm,
                   f ? "  ** a barrier TIMED OUT: the grid was not co-resident **" : "");
            check("persistent", (layers & 1) ? b : a, with_data);
        }
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
        CK(hipGraphExecDestroy(ge0));
        CK(hipGraphDestroy(g0));
    }
    return 0;
}
