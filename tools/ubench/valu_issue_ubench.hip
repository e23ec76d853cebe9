// Diagnostic: VALU issue cost on gfx950 with 1 / 2 / 3 waves per SIMD (the DDQN kernel runs 3), for independent v_fma_f32,
// v_pk_fma_f32, dependent v_fma_f32 chains, and an LDS gather + dependent math mix like the tanh-table forward.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_issue_ubench.hip -o /tmp/valu_ub && /tmp/valu_ub
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef float v2f __attribute__((ext_vector_type(2)));

typedef __attribute__((address_space(4))) const float cfloat;
typedef float v16f __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void k(float *out, unsigned long long *cyc, int iters, const float *wglb = nullptr)
{
    __shared__ float tab[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = (float)i * 1e-3f;
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    v2f p0 = {a0, 1}, p1 = {2, 3}, p2 = {4, 5}, p3 = {6, 7};
    const float m = 0.999f, c = 1e-3f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 8) {            // like 6, next record prefetched one iteration ahead (two SGPR sets, loop unrolled by two)
        auto body = [&](const v16f &w) {
            a0 = fmaf(a0, w[0], c); a1 = fmaf(a1, w[1], c); a2 = fmaf(a2, w[2], c); a3 = fmaf(a3, w[3], c);
            a4 = fmaf(a4, w[4], c); a5 = fmaf(a5, w[5], c); a6 = fmaf(a6, w[6], c); a7 = fmaf(a7, w[7], c);
            a0 = fmaf(a0, w[8], c); a1 = fmaf(a1, w[9], c); a2 = fmaf(a2, w[10], c); a3 = fmaf(a3, w[11], c);
            a4 = fmaf(a4, w[12], c); a5 = fmaf(a5, w[13], c); a6 = fmaf(a6, w[14], c); a7 = fmaf(a7, w[15], c);
        };
        v16f wA, wB;
        const float *base = wglb;
        asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(wA) : "s"(base));
        for (int it = 0; it < iters; it += 2) {
            const float *p1 = wglb + ((it + 1) & 31) * 16, *p2 = wglb + ((it + 2) & 31) * 16;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(wA));
            asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(wB) : "s"(p1));
            body(wA);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(wB));
            asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(wA) : "s"(p2));
            body(wB);
        }
    } else if (MODE < 9)
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {        // 8 independent v_fma_f32 x 2
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            }
        } else if (MODE == 1) { // 4 independent v_pk_fma_f32 x 4 (16 instructions, 32 fmas)
            const v2f mm = {m, m}, cc = {c, c};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc));
            }
        } else if (MODE == 2) { // 16 dependent v_fma_f32
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));
        } else if (MODE == 3) { // 2 chains of 8 dependent (two hidden units' polynomials)
#pragma unroll
            for (int r = 0; r < 8; ++r) asm volatile("v_fma_f32 %0, %0, %2, %3\n v_fma_f32 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(m), "v"(c));
        } else if (MODE == 4) { // gather -> 6 dependent fma, software-unpipelined (16 instr incl. index math)
            int idx = (__float_as_int(a0) >> 11) & 4095;
            float t = tab[idx];
#pragma unroll
            for (int r = 0; r < 12; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(m), "v"(c));
            a0 = a0 * m + t;
        } else if (MODE == 5) { // 4 broadcast ds_read_b128 (16 wave-uniform weights) + 16 v_fma on them: the forward's weight path
            const float4 *w4 = reinterpret_cast<const float4 *>(tab) + ((it & 31) * 4);
            float4 w0 = w4[0], w1 = w4[1], w2 = w4[2], w3 = w4[3];
            a0 = fmaf(a0, w0.x, c); a1 = fmaf(a1, w0.y, c); a2 = fmaf(a2, w0.z, c); a3 = fmaf(a3, w0.w, c);
            a4 = fmaf(a4, w1.x, c); a5 = fmaf(a5, w1.y, c); a6 = fmaf(a6, w1.z, c); a7 = fmaf(a7, w1.w, c);
            a0 = fmaf(a0, w2.x, c); a1 = fmaf(a1, w2.y, c); a2 = fmaf(a2, w2.z, c); a3 = fmaf(a3, w2.w, c);
            a4 = fmaf(a4, w3.x, c); a5 = fmaf(a5, w3.y, c); a6 = fmaf(a6, w3.z, c); a7 = fmaf(a7, w3.w, c);
        } else if (MODE == 6) { // the same 16 weights through the scalar cache (s_load_dwordx16) as SGPR operands
            cfloat *w = (cfloat *)wglb + (it & 31) * 16;
            a0 = fmaf(a0, w[0], c); a1 = fmaf(a1, w[1], c); a2 = fmaf(a2, w[2], c); a3 = fmaf(a3, w[3], c);
            a4 = fmaf(a4, w[4], c); a5 = fmaf(a5, w[5], c); a6 = fmaf(a6, w[6], c); a7 = fmaf(a7, w[7], c);
            a0 = fmaf(a0, w[8], c); a1 = fmaf(a1, w[9], c); a2 = fmaf(a2, w[10], c); a3 = fmaf(a3, w[11], c);
            a4 = fmaf(a4, w[12], c); a5 = fmaf(a5, w[13], c); a6 = fmaf(a6, w[14], c); a7 = fmaf(a7, w[15], c);
        } else if (MODE == 7) { // like 6 plus a scalar-cache invalidate every 29 iterations (one per learn step)
            if (it % 29 == 0) { __builtin_amdgcn_s_dcache_inv(); asm volatile("s_waitcnt lgkmcnt(0)"); }
            cfloat *w = (cfloat *)wglb + (it & 31) * 16;
            a0 = fmaf(a0, w[0], c); a1 = fmaf(a1, w[1], c); a2 = fmaf(a2, w[2], c); a3 = fmaf(a3, w[3], c);
            a4 = fmaf(a4, w[4], c); a5 = fmaf(a5, w[5], c); a6 = fmaf(a6, w[6], c); a7 = fmaf(a7, w[7], c);
            a0 = fmaf(a0, w[8], c); a1 = fmaf(a1, w[9], c); a2 = fmaf(a2, w[10], c); a3 = fmaf(a3, w[11], c);
            a4 = fmaf(a4, w[12], c); a5 = fmaf(a5, w[13], c); a6 = fmaf(a6, w[14], c); a7 = fmaf(a7, w[15], c);
        }
    }
    if (MODE == 10 || MODE == 11) {   // a LONG straight-line body (256 instructions per iteration): instruction-fetch bound?
        for (int it = 0; it < iters / 16; ++it) {
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                if (MODE == 10)     // 8-byte VOP3 encodings
                    asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                 "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
                else                // 4-byte VOP2 encodings
                    asm volatile("v_fmac_f32_e32 %0, %8, %9\n v_fmac_f32_e32 %1, %8, %9\n v_fmac_f32_e32 %2, %8, %9\n v_fmac_f32_e32 %3, %8, %9\n"
                                 "v_fmac_f32_e32 %4, %8, %9\n v_fmac_f32_e32 %5, %8, %9\n v_fmac_f32_e32 %6, %8, %9\n v_fmac_f32_e32 %7, %8, %9\n"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            }
        }
    }
    if (MODE == 9) {            // back-to-back v_mfma_f32_32x32x2_f32 (64 shader cycles each on one SIMD): calibrates ticks per cycle
        typedef float v16 __attribute__((ext_vector_type(16)));
        v16 acc0 = {0}, acc1 = {0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, a0, acc1, 0, 0, 0);
            }
        }
        a2 += acc0[0] + acc1[3];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

static float *wglb;
template <int MODE>
static void run(const char *name, float *out, unsigned long long *cyc, int instr_per_iter)
{
    const int iters = 100000;   // long enough, on every CU, for the clocks to ramp like under a real launch
    for (int threads : {256, 512, 768, 1024}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, (const float *)wglb);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters, (const float *)wglb);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h;
        hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        // s_memtime ticks at 100 MHz-class constant clock?  report raw ticks and ticks per (instruction x waves per SIMD)
        printf("%-28s waves/SIMD %d: %9llu ticks (%.2f ticks/ns), %.3f ticks = %.3f ns per wave-instruction, %.3f ns per SIMD-instruction\n", name,
               threads / 256, h, (double)h / (ms * 1e6), (double)h / iters / instr_per_iter, ms * 1e6 / iters / instr_per_iter,
               ms * 1e6 / iters / instr_per_iter / (threads / 256));
    }
}

int main()
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 1024 * sizeof(float)); hipMalloc(&cyc, 8);
    hipMalloc(&wglb, 4096 * sizeof(float));
    { float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = 0.999f; hipMemcpy(wglb, h, sizeof(h), hipMemcpyHostToDevice); }
    run<0>("indep v_fma_f32", out, cyc, 16);
    run<1>("indep v_pk_fma_f32", out, cyc, 16);
    run<2>("dependent v_fma_f32", out, cyc, 16);
    run<3>("2 chains v_fma_f32", out, cyc, 16);
    run<4>("gather + 12 dep fma", out, cyc, 16);
    run<5>("4 bcast ds_read_b128+16fma", out, cyc, 16);
    run<6>("s_load x16 + 16 fma", out, cyc, 16);
    run<7>("s_load x16 + inv/29", out, cyc, 16);
    run<8>("s_load x16 prefetched", out, cyc, 16);
    run<9>("v_mfma_f32_32x32x2_f32", out, cyc, 16);
    run<10>("256 indep v_fma_f32 (VOP3)", out, cyc, 16);
    run<11>("256 indep v_fmac_f32 (VOP2)", out, cyc, 16);
    return 0;
}
