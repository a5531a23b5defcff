// ORACLE build shim for the DCN device code: the grid-stride loop becomes a plain loop (host, single thread).
#pragma once
#define CUDA_KERNEL_LOOP(i, n) for (int i = 0; i < (n); ++i)
