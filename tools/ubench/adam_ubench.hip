// Diagnostic microbenchmark (not part of the product): cost of the per-chain Adam pass as a function of data and layout.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../../learning_environments_amd/csrc/lenv_gemm.cuh"
using namespace lenv;

__global__ __launch_bounds__(DNT) void adam_k(float *arena, int64_t stride, int P, int reps, unsigned long long *cyc)
{
    float *base = arena + blockIdx.x * stride;
    float *params = base, *target = base + P, *m = base + 2 * (int64_t)P, *v = base + 3 * (int64_t)P, *grad = base + 4 * (int64_t)P;
    AdamConsts ac{ -1e-3f, 0.9f, 0.1f, 0.001f, 0.999f, 1e-8f };
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        wg_adam(params, m, v, grad, 0, P, ac, target, 0.01f, 0.99f);
        __syncthreads();
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
}

int main()
{
    const int P = 68000, chains = 96, reps = 50;
    const int64_t stride = 5 * (int64_t)P + 64 * 1024;
    float *arena; unsigned long long *cyc;
    hipMalloc(&arena, sizeof(float) * stride * chains);
    hipMalloc(&cyc, sizeof(unsigned long long) * chains);
    std::vector<float> h(stride * chains);
    for (int mode = 0; mode < 4; ++mode) {
        for (size_t i = 0; i < h.size(); ++i) {
            float u = (float)rand() / RAND_MAX - 0.5f;
            h[i] = mode == 0 ? 0.0f : mode == 1 ? u * 1e-3f : mode == 2 ? u * 1e-20f : (i % 3 ? 0.0f : u * 1e-4f);
        }
        hipMemcpy(arena, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(adam_k, dim3(chains), dim3(DNT), 0, 0, arena, stride, P, reps, cyc);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
        printf("mode %d: %.3f ms total, %.1f us per pass, %.0f cycles per pass (chain 0)\n", mode, ms, 1e3 * ms / reps, (double)c0 / reps);
    }
    return 0;
}
