// tools/ubench/mfma_chain.hip -- how long is a DEPENDENT chain of v_mfma_f32_32x32x2_f32 (the k-ascending accumulation every product of the
// wave-chain kernels is), per instruction, on one wave per SIMD -- and with 2 / 4 independent chains interleaved in the same wave?
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chain tools/ubench/mfma_chain.hip ; run on the GPU box: /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int WIDE>
__global__ __launch_bounds__(512) void chain_kernel(float *out, unsigned long long *cyc, int n, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int c = 0; c < NACC; ++c) for (int v = 0; v < 16; ++v) acc[c][v] = 0.0f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < n; ++t) {
#pragma unroll
        for (int c = 0; c < NACC; ++c) {
            if constexpr (WIDE == 0) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
            else {
                f32x4 r = { acc[c][0], acc[c][1], acc[c][2], acc[c][3] };
                r = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, r, 0, 0, 0);
                acc[c][0] = r[0]; acc[c][1] = r[1]; acc[c][2] = r[2]; acc[c][3] = r[3];
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
    for (int c = 0; c < NACC; ++c) for (int v = 0; v < 16; ++v) s += acc[c][v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int WIDE> static void run(const char *name, int threads)
{
    float *out; unsigned long long *cyc, h[4];
    hipMalloc(&out, 4 * 512 * sizeof(float)); hipMalloc(&cyc, 4 * sizeof(unsigned long long));
    const int n = 4096;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((chain_kernel<NACC, WIDE>), dim3(1), dim3(threads), 0, 0, out, cyc, n, 1.0f, 0.5f);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, sizeof(unsigned long long), hipMemcpyDeviceToHost);
    printf("%-58s %2d waves/CU: %7.1f cycles per MFMA (%5.1f per chain step)\n", name, threads / 64, (double)h[0] / (n * NACC), (double)h[0] / n);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<1, 0>("32x32x2 f32, ONE dependent chain per wave", 256);
    run<2, 0>("32x32x2 f32, TWO independent chains interleaved per wave", 256);
    run<4, 0>("32x32x2 f32, FOUR independent chains interleaved per wave", 256);
    run<1, 0>("32x32x2 f32, one chain per wave, two waves per SIMD", 512);
    run<2, 0>("32x32x2 f32, two chains per wave, two waves per SIMD", 512);
    run<1, 1>("16x16x4 f32, ONE dependent chain per wave", 256);
    run<2, 1>("16x16x4 f32, TWO independent chains per wave", 256);
    run<4, 1>("16x16x4 f32, FOUR independent chains per wave", 256);
    run<1, 1>("16x16x4 f32, one chain per wave, two waves per SIMD", 512);
    return 0;
}
