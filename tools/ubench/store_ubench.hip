// Diagnostic microbenchmark (not part of the product): cost of a workgroup writing / reading a 64 KB f32 slab in HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __noinline__ void store_generic(float *out, float val, int lane, int wave)
{
#pragma unroll
    for (int v = 0; v < 32; ++v) out[(wave * 32 + v) * 64 + lane] = val + v;       // 256 B per instruction, generic pointer
}
__device__ __noinline__ float load_generic(const float *in, int lane, int wave)
{
    float s = 0.0f;
#pragma unroll
    for (int v = 0; v < 32; ++v) s += in[(wave * 32 + v) * 64 + lane];
    return s;
}

__global__ __launch_bounds__(512) void k(float *arena, int64_t stride, int reps, int mode, unsigned long long *cyc, float *sink)
{
    __shared__ float lds[64];
    float *base = arena + blockIdx.x * stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.0f;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        float *o = base + (r & 3) * 16384;
        if (mode == 0) {
#pragma unroll
            for (int v = 0; v < 32; ++v) o[(wave * 32 + v) * 64 + lane] = (float)r + v;                          // global_store_dword
        } else if (mode == 1) store_generic(threadIdx.x == 9999 ? lds : o, (float)r, lane, wave);               // flat_store_dword
        else if (mode == 2) {
#pragma unroll
            for (int v = 0; v < 8; ++v) reinterpret_cast<float4 *>(o)[(wave * 8 + v) * 64 + lane] = make_float4(r, v, 0, 0);   // dwordx4
        } else if (mode == 3) {
#pragma unroll
            for (int v = 0; v < 32; ++v) acc += o[(wave * 32 + v) * 64 + lane];                                    // global_load_dword
        } else if (mode == 4) acc += load_generic(threadIdx.x == 9999 ? lds : o, lane, wave);                     // flat_load_dword
        else {
#pragma unroll
            for (int v = 0; v < 8; ++v) { float4 t = reinterpret_cast<const float4 *>(o)[(wave * 8 + v) * 64 + lane]; acc += t.x + t.y + t.z + t.w; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = __builtin_readcyclecounter() - t0;
    if (acc == 12345.678f) sink[0] = acc;
}

int main()
{
    const int reps = 200;
    const int64_t stride = 4 * 16384 + 1024;
    float *arena, *sink; unsigned long long *cyc;
    hipMalloc(&arena, sizeof(float) * stride * 256);
    hipMalloc(&sink, 64);
    hipMalloc(&cyc, sizeof(unsigned long long) * 256);
    hipMemset(arena, 0, sizeof(float) * stride * 256);
    const char *names[] = { "global_store_dword", "flat_store_dword", "global_store_dwordx4", "global_load_dword", "flat_load_dword", "global_load_dwordx4" };
    for (int chains : { 1, 96, 192 })
        for (int mode = 0; mode < 6; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(chains), dim3(512), 0, 0, arena, stride, reps, mode, cyc, sink);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
            printf("chains %3d %-22s: %.2f us per 64 KB slab, %.0f cycles (chain 0), %.1f GB/s aggregate\n", chains, names[mode], 1e3 * ms / reps, (double)c0 / reps,
                   chains * 65536.0 * reps / (ms * 1e6));
        }
    return 0;
}
