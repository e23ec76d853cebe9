// lenv_gemm.cuh -- workgroup-cooperative, LDS-tiled GEMM with a canonical (ascending, single-thread) reduction order.
//
//   C[i][j] = epi(i, j, sum_{r<R} P[i*sPi + r*sPr] * Q[j*sQj + r*sQr])       I, J <= 128, any R
//
// 512 threads, 4x8 outputs per thread, the reduction staged through LDS 64 deep (Ps/Qs: GT_RB*GT_LD floats each).  Every
// output is accumulated by ONE thread with r ascending, i.e. the sequential fmaf chain of oracle/lenv_oracle.h, whatever
// the strides -- the same routine serves forward (r = input feature), input-gradient (r = output unit) and
// weight-gradient (r = sample) products of the big-net agents (DuelingDDQN, TD3).
#pragma once

#include "lenv_device.cuh"

namespace lenv {

constexpr int DNT = 512;          // threads per chain (8 waves)
constexpr int GT_I = 128, GT_J = 128, GT_RB = 64, GT_LD = 132;   // 128x128 outputs, 64-deep stages, padded LDS rows

// ---- workgroup-cooperative GEMM: C[i][j] = epi(i, j, sum_{r<R} P[i*sPi + r*sPr] * Q[j*sQj + r*sQr]), r ascending -------
template <class Epi>
__device__ __forceinline__ void wg_gemm(const float *P, int sPi, int sPr, const float *Q, int sQj, int sQr, int I, int J, int R,
                                        float *Ps, float *Qs, Epi epi)
{
    const int tid = threadIdx.x;
    const int ti = tid & 31, tj = tid >> 5;              // 32 x 16 thread grid, 4 x 8 outputs each
    float acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = 0.0f;
    const bool active = 4 * ti < I && 8 * tj < J;
    for (int r0 = 0; r0 < R; r0 += GT_RB) {
        const int rb = R - r0 < GT_RB ? R - r0 : GT_RB;
        __syncthreads();                                   // previous stage fully consumed
        // stage P -> Ps[r][i], Q -> Qs[r][j] (zero padded); the unit-stride index runs fastest across threads
        if (sPr == 1) { for (int e = tid; e < GT_I * rb; e += DNT) { int i = e / rb, r = e - i * rb; Ps[r * GT_LD + i] = i < I ? P[(int64_t)i * sPi + (r0 + r)] : 0.0f; } }
        else { for (int e = tid; e < GT_I * rb; e += DNT) { int r = e >> 7, i = e & 127; Ps[r * GT_LD + i] = i < I ? P[(int64_t)i * sPi + (int64_t)(r0 + r) * sPr] : 0.0f; } }
        if (sQr == 1) { for (int e = tid; e < GT_J * rb; e += DNT) { int j = e / rb, r = e - j * rb; Qs[r * GT_LD + j] = j < J ? Q[(int64_t)j * sQj + (r0 + r)] : 0.0f; } }
        else { for (int e = tid; e < GT_J * rb; e += DNT) { int r = e >> 7, j = e & 127; Qs[r * GT_LD + j] = j < J ? Q[(int64_t)j * sQj + (int64_t)(r0 + r) * sQr] : 0.0f; } }
        __syncthreads();
        if (active) {
            for (int r = 0; r < rb; ++r) {
                const float4 p4 = *reinterpret_cast<const float4 *>(Ps + r * GT_LD + 4 * ti);
                const float4 q0 = *reinterpret_cast<const float4 *>(Qs + r * GT_LD + 8 * tj);
                const float4 q1 = *reinterpret_cast<const float4 *>(Qs + r * GT_LD + 8 * tj + 4);
                const float pv[4] = { p4.x, p4.y, p4.z, p4.w };
                const float qv[8] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w };
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 8; ++b) acc[a][b] = fma32(pv[a], qv[b], acc[a][b]);
            }
        }
    }
    if (active) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int i = 4 * ti + a, j = 8 * tj + b;
                if (i < I && j < J) epi(i, j, acc[a][b]);
            }
    }
}

}  // namespace lenv
