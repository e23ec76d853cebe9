// Microbenchmark + exactness check behind the round-5 forward of the DDQN kernel: the minibatch forward's hidden-unit loop with the two
// Linear layers on v_mfma_f32_4x4x1_16b_f32 (lane = sample: block b = lanes 4b..4b+3, A[i] = a weight of hidden unit 4g+i supplied by lane
// 4b+i -- the same for every block --, B[j] = the lane's own input, D[i] = the four pre-activations of the lane's own sample), against
// the packed-fp32 pair loop of rounds 2-4 (tools/ubench/fwd_pair_loop.hip).  K = 1 per instruction, so a chain of them over k IS the
// canonical fmaf chain; the bias joins as a fifth step fma(b, 1.0, z) = z + b (one rounding, the same value as the add).
//   part 1: bitwise check of the MFMA formulation against the fmaf chain on random data (layout and rounding);
//   part 2: cycles per (64 items x 1 pair) per CU of the quad loop at 12 / 8 / 4 waves, next to the pair loop's.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I learning_environments_amd/csrc tools/ubench/fwd_quad_mfma.hip -o fwd_quad_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "lenv_device.cuh"

using namespace lenv;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
static __device__ __forceinline__ v4f mfma4(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

constexpr int S = 4, A = 2, HQ = 57, NQ = (HQ + 3) / 4;           // 15 quads (units 57..59 are zero rows)
// quad record: [4 units][S weights] | [4 biases] | [4 action rows][4 units]  = 16 + 4 + 16 floats
constexpr int QR = 4 * S + 4 + 16;

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

// ---------------- part 1: exactness ----------------
// W1 [HQ][S], b1 [HQ], W2 [A][HQ], x [n][S] -> q [n][A] and h [n][HQ], both ways
__global__ __launch_bounds__(64) void exact_kernel(const float *W1, const float *b1, const float *W2, const float *x, int n, float *q_ref, float *q_mf, float *h_ref, float *h_mf)
{
    __shared__ __align__(16) float tab[LENV_TANH1_FLOATS];
    __shared__ __align__(16) float rec[NQ * QR];
    const int lane = threadIdx.x;
    for (int e = lane; e < LENV_TANH1_FLOATS; e += 64) tab[e] = lenv_tanh_table[e];
    for (int e = lane; e < NQ * QR; e += 64) rec[e] = 0.0f;
    __syncthreads();
    for (int u = lane; u < HQ; u += 64) {
        float *r = rec + (u >> 2) * QR;
        for (int k = 0; k < S; ++k) r[(u & 3) * S + k] = W1[u * S + k];
        r[4 * S + (u & 3)] = b1[u];
        for (int a = 0; a < A; ++a) r[4 * S + 4 + a * 4 + (u & 3)] = W2[a * HQ + u];
    }
    __syncthreads();
    for (int s0 = blockIdx.x * 64; s0 < n; s0 += gridDim.x * 64) {
        const int s = s0 + lane;
        float xs[S];
        for (int k = 0; k < S; ++k) xs[k] = s < n ? x[s * S + k] : 0.0f;
        // reference: the canonical order (oracle mlp forward): z = fma chain k ascending from 0, + b; h = tanh; q[a] = fma chain u ascending
        float qr[A] = {0.0f, 0.0f};
        for (int u = 0; u < HQ; ++u) {
            float z = 0.0f;
            for (int k = 0; k < S; ++k) z = fma32(xs[k], W1[u * S + k], z);
            z = z + b1[u];
            const float h = det_tanhf(tab, z);
            if (s < n) h_ref[(size_t)s * HQ + u] = h;
            for (int a = 0; a < A; ++a) qr[a] = fma32(h, W2[a * HQ + u], qr[a]);
        }
        // MFMA formulation
        v4f q4 = {0.0f, 0.0f, 0.0f, 0.0f};
        const float one = 1.0f;
        for (int g = 0; g < NQ; ++g) {
            const float *r = rec + g * QR;
            v4f z4 = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int k = 0; k < S; ++k) z4 = mfma4(r[(lane & 3) * S + k], xs[k], z4);
            z4 = mfma4(r[4 * S + (lane & 3)], one, z4);
            float h[4];
            for (int i = 0; i < 4; ++i) h[i] = det_tanhf(tab, z4[i]);
            for (int i = 0; i < 4; ++i) {
                if (s < n && 4 * g + i < HQ) h_mf[(size_t)s * HQ + 4 * g + i] = h[i];
                q4 = mfma4(r[4 * S + 4 + (lane & 3) * 4 + i], h[i], q4);
            }
        }
        if (s < n) for (int a = 0; a < A; ++a) { q_ref[s * A + a] = qr[a]; q_mf[s * A + a] = q4[a]; }
    }
}

// ---------------- part 2: timing ----------------
// MODE 0: both layers on MFMA; MODE 1: layer 1 (+bias) on MFMA, output layer packed fp32 from broadcast reads
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void quad_loop_kernel(const float *weights, float *out, unsigned long long *cycles, int reps)
{
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    det_tanh_lds_stage(lds, true, tid, NT);
    float *rec = lds + LENV_TANH16_FLOATS;
    float *hrows = rec + ((NQ * QR + 3) & ~3);
    for (int i = tid; i < NQ * QR; i += NT) rec[i] = weights[i];
    __syncthreads();
    const TanhLds tl = TanhLds::make(true, lane);
    float x[S];
#pragma unroll
    for (int k = 0; k < S; ++k) x[k] = 0.01f * (float)(tid % 97) - 0.3f * (float)k;
    float one;
    asm volatile("v_mov_b32 %0, 1.0" : "=v"(one));
    v4f q4 = {0.0f, 0.0f, 0.0f, 0.0f};
    const float *myrec = rec + (lane & 3) * S;           // + g * QR: this lane's layer-1 row of quad g
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        float4 aw; float ab; float4 aw2; float4 bw2[2];
        auto loadA = [&](int g) { aw = *reinterpret_cast<const float4 *>(myrec + g * QR); ab = rec[g * QR + 4 * S + (lane & 3)]; };
        auto loadW2 = [&](int g) {
            if constexpr (MODE == 0) aw2 = *reinterpret_cast<const float4 *>(rec + g * QR + 4 * S + 4 + (lane & 3) * 4);
            else { bw2[0] = *reinterpret_cast<const float4 *>(rec + g * QR + 4 * S + 4); bw2[1] = *reinterpret_cast<const float4 *>(rec + g * QR + 4 * S + 8); }
        };
        auto layer1 = [&]() -> v4f {
            v4f z4 = {0.0f, 0.0f, 0.0f, 0.0f};
            z4 = mfma4(aw.x, x[0], z4); z4 = mfma4(aw.y, x[1], z4); z4 = mfma4(aw.z, x[2], z4); z4 = mfma4(aw.w, x[3], z4);
            return mfma4(ab, one, z4);
        };
        auto out2 = [&](v2f hh, int half, int jp) {
            if constexpr (MODE == 0) {
                q4 = mfma4(half ? aw2.z : aw2.x, hh.x, q4);
                q4 = mfma4(half ? aw2.w : aw2.y, hh.y, q4);
            } else {
                v2f qq = {q4.x, q4.y};
                const float4 w = bw2[half];
                qq = fma2((v2f){hh.x, hh.x}, (v2f){w.x, w.y}, qq);
                qq = fma2((v2f){hh.y, hh.y}, (v2f){w.z, w.w}, qq);
                q4.x = qq.x; q4.y = qq.y;
            }
            if ((tid & 2) == 0) *reinterpret_cast<v2f *>(hrows + (tid & 255) * 58 + 2 * jp) = hh;
        };
        Pipe pa, pb;
        loadA(0);
        v4f z4 = layer1();
        loadA(1);
        pa.issue(tl, (v2f){z4.x, z4.y});
        v2f zhi = {z4.z, z4.w};
        loadW2(0);
        int g = 0;
#pragma unroll 1
        for (; g + 2 < NQ; ++g) {
            pb.issue(tl, zhi);                                   // pair 2g+1
            out2(pa.finish(), 0, 2 * g);                         // pair 2g
            z4 = layer1();                                       // quad g+1
            loadA(g + 2);
            pa.issue(tl, (v2f){z4.x, z4.y});                     // pair 2g+2
            zhi = (v2f){z4.z, z4.w};
            out2(pb.finish(), 1, 2 * g + 1);                     // pair 2g+1
            loadW2(g + 1);
        }
        if (r == reps - 1 && tid == 0 && blockIdx.x == 0) cycles[1] = (unsigned long long)(2 * g);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[(size_t)blockIdx.x * NT + tid] = q4.x + q4.y + hrows[(tid & 255) * 58];
    if (tid == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int MODE, int NT>
static void run(const char *name, const float *dW, float *dOut, unsigned long long *dC, int blocks)
{
    const int reps = 200;
    const size_t lds_bytes = (size_t)(LENV_TANH16_FLOATS + ((NQ * QR + 3) & ~3) + 256 * 58) * sizeof(float);
    auto kern = quad_loop_kernel<MODE, NT>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds_bytes, 0, dW, dOut, dC, 2);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), lds_bytes, 0, dW, dOut, dC, reps);
    hipDeviceSynchronize();
    unsigned long long c[2];
    hipMemcpy(c, dC, sizeof(c), hipMemcpyDeviceToHost);
    const double pairs = (double)c[1] * reps;
    const double item_pairs64 = pairs * (double)NT / 64.0;
    printf("%-44s %9.0f cycles per run of %2llu pairs: %6.1f cycles per (64 items x pair) per CU; per wave-stage %6.1f\n", name,
           (double)c[0] / reps, c[1], (double)c[0] / item_pairs64, (double)c[0] / pairs);
}

int main()
{
    // ---- part 1 ----
    const int n = 64 * 64;
    std::vector<float> W1(HQ * S), b1(HQ), W2(A * HQ), x(n * S);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)((st >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
    for (auto &v : W1) v = 2.0f * rnd();
    for (auto &v : b1) v = rnd();
    for (auto &v : W2) v = rnd();
    for (int i = 0; i < n * S; ++i) x[i] = (i % 7 == 0 ? 40.0f : 3.0f) * rnd();
    x[5] = 1e-41f; x[6] = -1e-42f;          // denormal inputs: products underflow, both forms must agree
    float *dW1, *db1, *dW2, *dx, *dq0, *dq1, *dh0, *dh1;
    hipMalloc(&dW1, W1.size() * 4); hipMalloc(&db1, b1.size() * 4); hipMalloc(&dW2, W2.size() * 4); hipMalloc(&dx, x.size() * 4);
    hipMalloc(&dq0, n * A * 4); hipMalloc(&dq1, n * A * 4); hipMalloc(&dh0, (size_t)n * HQ * 4); hipMalloc(&dh1, (size_t)n * HQ * 4);
    hipMemcpy(dW1, W1.data(), W1.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db1, b1.data(), b1.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW2, W2.data(), W2.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(exact_kernel, dim3(16), dim3(64), 0, 0, dW1, db1, dW2, dx, n, dq0, dq1, dh0, dh1);
    hipDeviceSynchronize();
    std::vector<float> q0(n * A), q1(n * A), h0((size_t)n * HQ), h1((size_t)n * HQ);
    hipMemcpy(q0.data(), dq0, q0.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(q1.data(), dq1, q1.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h0.data(), dh0, h0.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), dh1, h1.size() * 4, hipMemcpyDeviceToHost);
    size_t badq = 0, badh = 0;
    for (size_t i = 0; i < q0.size(); ++i) badq += memcmp(&q0[i], &q1[i], 4) != 0;
    for (size_t i = 0; i < h0.size(); ++i) badh += memcmp(&h0[i], &h1[i], 4) != 0;
    printf("exactness: %zu of %zu q words differ, %zu of %zu h words differ (q[0] = %g %g | %g %g)\n", badq, q0.size(), badh, h0.size(), q0[0], q0[1], q1[0], q1[1]);
    // ---- part 2 ----
    std::vector<float> w(NQ * QR);
    for (int i = 0; i < NQ * QR; ++i) w[i] = 0.3f * (float)((i * 37) % 23 - 11) / 11.0f;
    float *dW, *dOut;
    unsigned long long *dC;
    hipMalloc(&dW, w.size() * sizeof(float));
    hipMalloc(&dOut, 256 * 768 * sizeof(float));
    hipMalloc(&dC, 2 * sizeof(unsigned long long));
    hipMemcpy(dW, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice);
    const int blocks = 192;
    run<0, 768>("quad loop, both layers MFMA, 12 waves", dW, dOut, dC, blocks);
    run<1, 768>("quad loop, layer 1 MFMA / output packed, 12 waves", dW, dOut, dC, blocks);
    run<0, 640>("quad loop, both layers MFMA, 10 waves", dW, dOut, dC, blocks);
    run<0, 512>("quad loop, both layers MFMA, 8 waves", dW, dOut, dC, blocks);
    run<1, 512>("quad loop, layer 1 MFMA / output packed, 8 waves", dW, dOut, dC, blocks);
    run<0, 256>("quad loop, both layers MFMA, 4 waves", dW, dOut, dC, blocks);
    run<0, 64>("quad loop, both layers MFMA, 1 wave", dW, dOut, dC, blocks);
    return 0;
}
