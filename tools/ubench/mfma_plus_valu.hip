// tools/ubench/mfma_plus_valu.hip -- do the two fp32 pipes of a CDNA4 SIMD run side by side?  Waves 0-3 (one per SIMD) run a dependent chain
// of v_mfma_f32_32x32x2_f32 (32 MAC / cycle / SIMD), waves 4-7 run v_pk_fma_f32 on 32 independent accumulator pairs with a scalar second
// operand (also 32 MAC / cycle / SIMD: the shape a VALU tile of the wave-chain layers would have -- lane = two units, register = sample,
// x[k][sample] as a scalar).  Timed: each kind alone, then both in one workgroup.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_plus_valu tools/ubench/mfma_plus_valu.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// mode bit 0: waves 0-3 run the MFMA chain; bit 1: waves 4-7 run the packed-FMA tiles
__global__ __launch_bounds__(512) void both_kernel(float *out, unsigned long long *cyc, int n, float a0, float b0, int mode, const float *__restrict__ xs)
{
    const int wave = threadIdx.x >> 6;
    float s = 0.0f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        if (mode & 1) {
            f32x16 acc;
            for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
            const float a = a0 + threadIdx.x * 1e-3f, b = b0;
            for (int t = 0; t < n; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            for (int v = 0; v < 16; ++v) s += acc[v];
        }
    } else if (mode & 2) {
        // n MFMAs are n * 2048 MACs; a packed FMA is 128: 16 per MFMA.  32 accumulator pairs (samples), one weight pair per k.
        f32x2 acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = f32x2{ 0.0f, 0.0f };
        f32x2 w = { a0 + threadIdx.x * 1e-3f, b0 - threadIdx.x * 1e-3f };
        const int iters = n / 2;                          // 32 packed FMAs per iteration = 2 MFMAs' worth
        if (mode & 4) {
            // scalars already in SGPRs (no load in the loop): the pure issue rate of the packed FMAs next to the MFMA chain
            float sc[8];
            for (int i = 0; i < 8; ++i) sc[i] = xs[i];
            for (int t = 0; t < iters; ++t) {
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = __builtin_elementwise_fma(w, f32x2{ sc[i & 7], sc[i & 7] }, acc[i]);
                w.x += 1e-7f;
            }
        } else
        for (int t = 0; t < iters; ++t) {
            const float *xk = xs + (t & 63) * 32;          // uniform address: scalar loads, waited for in every iteration (no prefetch)
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const float x = xk[i];
                acc[i] = __builtin_elementwise_fma(w, f32x2{ x, x }, acc[i]);
            }
            w.x += 1e-7f;
        }
        for (int i = 0; i < 32; ++i) s += acc[i].x + acc[i].y;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main()
{
    float *out, *xs; unsigned long long *cyc, h[8];
    hipMalloc(&out, 512 * sizeof(float)); hipMalloc(&cyc, 8 * sizeof(unsigned long long)); hipMalloc(&xs, 64 * 32 * sizeof(float));
    hipMemset(xs, 0, 64 * 32 * sizeof(float));
    const int n = 8192;
    const char *names[8] = { "", "MFMA chain alone (waves 0-3)", "packed-FMA tiles alone, scalar loads in the loop", "both, scalar loads in the loop", "", "",
                             "packed-FMA tiles alone, scalars resident", "both, scalars resident" };
    for (int mode = 1; mode <= 7; ++mode) {
        if (mode == 4 || mode == 5) continue;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(both_kernel, dim3(1), dim3(512), 0, 0, out, cyc, n, 1.0f, 0.5f, mode, xs);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("%-40s MFMA wave: %6.1f cycles per 2048 MACs   packed-FMA wave: %6.1f cycles per 2048 MACs\n", names[mode],
               (double)h[0] / n, (double)h[4] / n);
    }
    return 0;
}
