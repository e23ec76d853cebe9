// Diagnostic microbenchmark (not part of the product): cost of one wg_gemm call with a hot instruction cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../learning_environments_amd/csrc/lenv_gemm.cuh"
using namespace lenv;

__global__ __launch_bounds__(DNT) void gemm_k(float *arena, int64_t stride, int I, int J, int R, int reps, int mode, unsigned long long *cyc)
{
    extern __shared__ __align__(16) float lds[];
    float *Ps = lds, *Qs = lds + GemmShape<128>::PS_FLOATS;
    GemmQueue gq(reinterpret_cast<GemmCmd *>(Qs + GemmShape<128>::QS_FLOATS));
    float *base = arena + blockIdx.x * stride;
    float *X = base, *W = base + 128 * 128, *bias = W + 128 * 128, *Y = bias + 128, *Y2 = Y + 128 * 128;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        if (mode == 0) gq.gemm(X, R, 1, W, R, 1, I, J, R, epi_bias_act(Y, J, bias, LENV_ACT_RELU, 0.0f));
        else if (mode == 1) gq.gemm(Y, 1, J, X, 1, R, J, R, I, epi_store(Y2, R));                      // dW = dY^T X
        else gq.gemm(Y, J, 1, W, 1, R, I, R, J, epi_act_bwd(Y2, R, X, R, LENV_ACT_RELU, 0.0f));        // dX = dY W
        if ((r & 7) == 7 || r == reps - 1) gq.run<128>(Ps, Qs);                                       // queues of 8 products
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
}

int main()
{
    const int chains = 96, reps = 50;
    const int64_t stride = 6 * 128 * 128;
    float *arena; unsigned long long *cyc;
    hipMalloc(&arena, sizeof(float) * stride * chains);
    hipMalloc(&cyc, sizeof(unsigned long long) * chains);
    std::vector<float> h(stride * chains);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(arena, h.data(), sizeof(float) * h.size(), hipMemcpyHostToDevice);
    const size_t ldsb = (GemmShape<128>::PS_FLOATS + GemmShape<128>::QS_FLOATS) * sizeof(float) + GEMM_QUEUE_MAX * sizeof(GemmCmd);
    hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    const int shapes[][3] = { {128, 128, 128}, {128, 128, 6}, {10, 128, 128}, {1, 128, 128}, {128, 3, 128} };
    for (auto &sh : shapes)
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(gemm_k, dim3(chains), dim3(DNT), ldsb, 0, arena, stride, sh[0], sh[1], sh[2], reps, mode, cyc);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
            printf("I=%d J=%d R=%d mode %d: %.1f us per call, %.0f cycles per call (chain 0)\n", sh[0], sh[1], sh[2], mode, 1e3 * ms / reps, (double)c0 / reps);
        }
    return 0;
}
