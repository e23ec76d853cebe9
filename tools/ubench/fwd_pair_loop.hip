// Microbenchmark behind docs/notebook_r01_r04.md section 9 "next (a)": the minibatch forward's hidden-unit pair loop of the DDQN kernel (4-57-2 tanh net,
// pair records + 16 bank-private copies of the canonical tanh table in LDS, packed fp32 math, gathers one or two pairs ahead), run by
//   W waves per workgroup (one workgroup per CU), each lane carrying I items that share every broadcast weight read,
// for (W, I) = (12, 1) [today's occupancy: 168 VGPRs], (8, 1), (8, 2), (4, 2), (4, 4) [256 / 512 VGPRs].
// Prints cycles per (64 items x 1 pair) per CU -- the currency of docs/notebook_r01_r04.md section 8 -- so that "two items per lane at two waves per SIMD"
// can be priced before the kernel is restructured.  Same instruction mix as the kernel's steady-state stage; results are checksummed only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I learning_environments_amd/csrc tools/ubench/fwd_pair_loop.hip -o /tmp/fwd_pair_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "lenv_device.cuh"

using namespace lenv;
typedef float v2f __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

constexpr int S = 4, NPAIRS = 29, PR = 16, PR4 = 4;      // record: [w(8) | b(2) pad(2) | out(4)]
constexpr int REC_FLOATS = NPAIRS * PR + 4;

struct Pipe {
    v2f z, d;
    float4 k0, k1;
    __device__ __forceinline__ void issue(const TanhLds &tl, v2f zz)
    {
        z = zz;
        const float a0 = __builtin_fabsf(zz.x), a1 = __builtin_fabsf(zz.y);
        const v2f t = {a0 < LENV_TANH_TMAX ? a0 : LENV_TANH_TMAX, a1 < LENV_TANH_TMAX ? a1 : LENV_TANH_TMAX};
        const v2f u = t + (v2f){LENV_TANH_MAGIC, LENV_TANH_MAGIC};
        d = t - (u - (v2f){LENV_TANH_MAGIC, LENV_TANH_MAGIC});
        k0 = det_tanh_lds_gather(0u, tl.off(__float_as_uint(u.x)));
        k1 = det_tanh_lds_gather(0u, tl.off(__float_as_uint(u.y)));
    }
    __device__ __forceinline__ v2f finish() const { return (v2f){det_tanh_poly(k0, d.x, z.x), det_tanh_poly(k1, d.y, z.y)}; }
};

// I items per lane, DEPTH pairs of gathers in flight (1 = the kernel's pipeline, 2 = one more activation pipe per item)
template <int I, int DEPTH, int NT>
__global__ __launch_bounds__(NT) void pair_loop_kernel(const float *weights, float *out, unsigned long long *cycles, int reps)
{
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    det_tanh_lds_stage(lds, true, tid, NT);
    float *rec = lds + LENV_TANH16_FLOATS;
    float *hrows = rec + ((REC_FLOATS + 3) & ~3);                  // [256][58]: the pass-0 h stores of the kernel (rows shared modulo 256)
    for (int i = tid; i < REC_FLOATS; i += NT) rec[i] = weights[i];
    __syncthreads();
    const TanhLds tl = TanhLds::make(true, lane);
    const float4 *W4 = reinterpret_cast<const float4 *>(rec);
    float x[I][S], q[I][2];
#pragma unroll
    for (int it = 0; it < I; ++it) {
#pragma unroll
        for (int k = 0; k < S; ++k) x[it][k] = 0.01f * (float)((tid * I + it) % 97) - 0.3f * (float)k;
        q[it][0] = q[it][1] = 0.0f;
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        Pipe p[DEPTH + 1][I];
        float4 r1[3], r2;
        auto load1 = [&](int jp) { r1[0] = W4[jp * PR4]; r1[1] = W4[jp * PR4 + 1]; r1[2] = W4[jp * PR4 + 2]; };
        auto load2 = [&](int jp) { r2 = W4[jp * PR4 + 3]; };
        auto layer1 = [&](int it) -> v2f {
            v2f z = {0.0f, 0.0f};
            z = fma2((v2f){x[it][0], x[it][0]}, (v2f){r1[0].x, r1[0].y}, z);
            z = fma2((v2f){x[it][1], x[it][1]}, (v2f){r1[0].z, r1[0].w}, z);
            z = fma2((v2f){x[it][2], x[it][2]}, (v2f){r1[1].x, r1[1].y}, z);
            z = fma2((v2f){x[it][3], x[it][3]}, (v2f){r1[1].z, r1[1].w}, z);
            return z + (v2f){r1[2].x, r1[2].y};
        };
        auto finish = [&](Pipe (&cur)[I], int jp) {
#pragma unroll
            for (int it = 0; it < I; ++it) {
                const v2f hh = cur[it].finish();
                v2f qq = {q[it][0], q[it][1]};
                qq = fma2((v2f){hh.x, hh.x}, (v2f){r2.x, r2.y}, qq);
                qq = fma2((v2f){hh.y, hh.y}, (v2f){r2.z, r2.w}, qq);
                q[it][0] = qq.x; q[it][1] = qq.y;
                if ((tid & 2) == 0) *reinterpret_cast<v2f *>(hrows + ((tid * I + it) & 255) * 58 + 2 * jp) = hh;      // ~ the pass-0 third of the items
            }
        };
        // prologue: DEPTH pairs issued
        load1(0);
#pragma unroll
        for (int dd = 0; dd < DEPTH; ++dd) {
#pragma unroll
            for (int it = 0; it < I; ++it) p[dd][it].issue(tl, layer1(it));
            load1(dd + 1);
        }
        load2(0);
        // steady state, rotating DEPTH+1 pipes (fully unrolled over the rotation so that the pipes are registers)
        int jp = 0;
#pragma unroll 1
        for (; jp + (DEPTH + 1) + DEPTH < NPAIRS; jp += DEPTH + 1) {
#pragma unroll
            for (int s = 0; s <= DEPTH; ++s) {
                const int cur = s, nxt = (s + DEPTH) % (DEPTH + 1);
#pragma unroll
                for (int it = 0; it < I; ++it) p[nxt][it].issue(tl, layer1(it));          // pair jp+s+DEPTH (r1 holds it)
                load1(jp + s + DEPTH + 1 < NPAIRS ? jp + s + DEPTH + 1 : NPAIRS - 1);
                finish(p[cur], jp + s);
                load2(jp + s + 1 < NPAIRS ? jp + s + 1 : NPAIRS - 1);
            }
        }
        // (the tail pairs are skipped: the benchmark prices the steady state; `done` pairs are reported)
        if (r == reps - 1 && tid == 0 && blockIdx.x == 0) cycles[1] = (unsigned long long)jp;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc = 0.0f;
#pragma unroll
    for (int it = 0; it < I; ++it) acc += q[it][0] + q[it][1];
    out[(size_t)blockIdx.x * NT + tid] = acc + hrows[((tid * I) & 255) * 58];
    if (tid == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int I, int DEPTH, int NT>
static void run(const char *name, const float *dW, float *dOut, unsigned long long *dC, int blocks)
{
    const int reps = 200;
    const size_t lds_bytes = (size_t)(LENV_TANH16_FLOATS + ((REC_FLOATS + 3) & ~3) + 256 * 58) * sizeof(float);
    if (lds_bytes > 160 * 1024) { printf("%-34s does not fit LDS (%zu KB)\n", name, lds_bytes / 1024); return; }
    auto kern = pair_loop_kernel<I, DEPTH, NT>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds_bytes, 0, dW, dOut, dC, 2);           // warm-up
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds_bytes, 0, dW, dOut, dC, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2];
    hipMemcpy(c, dC, sizeof(c), hipMemcpyDeviceToHost);
    const double pairs = (double)c[1] * reps;                                   // pairs per item
    const double item_pairs64 = pairs * (double)(NT * I) / 64.0;                // units of (64 items x 1 pair) per CU
    printf("%-34s %6.2f ms, %9.0f cycles per run of %2llu pairs: %6.1f cycles per (64 items x pair) per CU; per wave-stage %6.1f\n", name, ms,
           (double)c[0] / reps, c[1], (double)c[0] / item_pairs64, (double)c[0] / pairs);
}

int main()
{
    std::vector<float> w(REC_FLOATS);
    for (int i = 0; i < REC_FLOATS; ++i) w[i] = 0.3f * (float)((i * 37) % 23 - 11) / 11.0f;
    float *dW, *dOut;
    unsigned long long *dC;
    hipMalloc(&dW, w.size() * sizeof(float));
    hipMalloc(&dOut, 256 * 768 * sizeof(float));
    hipMalloc(&dC, 2 * sizeof(unsigned long long));
    hipMemcpy(dW, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice);
    const int blocks = 192;
    run<1, 1, 768>("12 waves x 1 item, depth 1", dW, dOut, dC, blocks);
    run<1, 1, 640>("10 waves x 1 item, depth 1", dW, dOut, dC, blocks);
    run<1, 1, 512>(" 8 waves x 1 item, depth 1", dW, dOut, dC, blocks);
    run<1, 2, 512>(" 8 waves x 1 item, depth 2", dW, dOut, dC, blocks);
    run<2, 1, 512>(" 8 waves x 2 items, depth 1", dW, dOut, dC, blocks);
    run<2, 2, 512>(" 8 waves x 2 items, depth 2", dW, dOut, dC, blocks);
    run<2, 1, 256>(" 4 waves x 2 items, depth 1", dW, dOut, dC, blocks);
    run<2, 2, 256>(" 4 waves x 2 items, depth 2", dW, dOut, dC, blocks);
    run<4, 1, 256>(" 4 waves x 4 items, depth 1", dW, dOut, dC, blocks);
    run<1, 1, 256>(" 4 waves x 1 item, depth 1", dW, dOut, dC, blocks);
    run<1, 1, 64>(" 1 wave  x 1 item, depth 1", dW, dOut, dC, blocks);
    return 0;
}
