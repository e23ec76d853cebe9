// lenv_ln.cuh -- the shared nn.LayerNorm(hidden) of build_nn_from_config (models/model_utils.py:22-37: eps 1e-5, biased variance) as ROW
// routines of the GEMM-queue kernels: the layer products stay workgroup GEMMs, a LayerNorm position runs the queue and then these.  Same
// sequence of operations as the oracle's mlp_forward_one_ex / mlp_backward_one_ex + mlp_fold_ln_grads (sequential sums along the row, column
// sums rows ascending).  Used by td3_discrete_inner_loop.hip (where they were written) and dueling_se_inner_loop.hip.
#pragma once

#include "lenv_gemm.cuh"

namespace lenv {

// ---- LayerNorm rows (forward): z [I][H] holds Linear + bias.  A chunk of rows is copied into LDS (coalesced; the product buffers Ps / Qs
// -- LNBUF floats from Ps on -- are free between queue runs; odd row stride = conflict-free), thread b then owns row b: sequential mean /
// variance and the normalised row in place, and the copy back applies weight, bias and activation: xh <- normalised rows, rstd[b],
// z <- act(fma(xn, w, b)) ----
// (round 6: the copies request eight rows' words before the first LDS store -- one round trip per eight rows instead of one per row, which was
// ~30 k of the routine's ~55 k cycles at 128 x 128 --, and the staged rows are addressed as LDS (ds_ instead of flat instructions); the
// arithmetic and its order are unchanged)
__device__ __forceinline__ lfloat *ln_lds_ptr(float *p) { return lds_offset_ptr(p); }      // (not a cast: see lenv_gemm.cuh)

template <int NB>
__device__ __forceinline__ void ln_copy_rows_in(const float *src, int ld, int nr, int Hh, lfloat *buf, int Hp, int lane_, int wave_)
{
    for (int b0 = wave_; b0 < nr; b0 += DNW * NB)
        for (int jj = lane_; jj < Hh; jj += 64) {
            float v[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) { const int b = b0 + DNW * u; v[u] = b < nr ? src[(int64_t)b * ld + jj] : 0.0f; }
#pragma unroll
            for (int u = 0; u < NB; ++u) { const int b = b0 + DNW * u; if (b < nr) buf[b * Hp + jj] = v[u]; }
        }
}

template <int LNBUF>
__device__ __forceinline__ void ln_rows_forward(float *Ps, float *z, int I, int Hh, const float *w, const float *bb, float *xh, float *rstd, int act, float pr)
{
    const int tid = threadIdx.x, lane_ = tid & 63, wave_ = uni(tid >> 6);
    const int Hp = Hh | 1, R = LNBUF / Hp;
    lfloat *buf = ln_lds_ptr(Ps);
    for (int r0 = 0; r0 < I; r0 += R) {
        const int nr = I - r0 < R ? I - r0 : R;
        ln_copy_rows_in<8>(z + (int64_t)r0 * Hh, Hh, nr, Hh, buf, Hp, lane_, wave_);
        __syncthreads();
        for (int b = tid; b < nr; b += DNT) {
            lfloat *zr = buf + b * Hp;
            float sm = 0.0f, sv = 0.0f;
#pragma unroll 16
            for (int jj = 0; jj < Hh; ++jj) sm = sm + zr[jj];
            const float mean = sm / (float)Hh;
#pragma unroll 16
            for (int jj = 0; jj < Hh; ++jj) { const float dj = zr[jj] - mean; sv = fma32(dj, dj, sv); }
            const float r = 1.0f / __builtin_sqrtf(sv / (float)Hh + 1e-5f);
            if (rstd) rstd[r0 + b] = r;
            int jj = 0;
            for (; jj + 8 <= Hh; jj += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = zr[jj + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) zr[jj + u] = (v[u] - mean) * r;
            }
            for (; jj < Hh; ++jj) zr[jj] = (zr[jj] - mean) * r;
        }
        __syncthreads();
        for (int b = wave_; b < nr; b += DNW) {
            float *dst = z + (int64_t)(r0 + b) * Hh;
            float *dxh = xh ? xh + (int64_t)(r0 + b) * Hh : nullptr;
            for (int jj = lane_; jj < Hh; jj += 64) {
                const float xnrm = buf[b * Hp + jj];
                if (dxh) dxh[jj] = xnrm;
                dst[jj] = act_fwd(act, pr, fma32(xnrm, w[jj], bb[jj]));
            }
        }
        __syncthreads();
    }
}

// ---- LayerNorm rows (backward): d [I][H] holds the gradient of the LayerNorm OUTPUT.  Column sums first (the shared weight / bias gradient
// of this position, rows ascending; first: store, else add to what the positions above left), then -- chunks of rows of d and xh staged
// in LDS as above -- thread b turns row b into the gradient of the LayerNorm input ----
template <int LNBUF, int MAXW>
__device__ __forceinline__ void ln_rows_backward(float *Ps, float *d, int I, int Hh, const float *w, const float *xh, const float *rstd, float *g_w, float *g_b, bool first)
{
    const int tid = threadIdx.x, lane_ = tid & 63, wave_ = uni(tid >> 6);
    if (g_w) {
        for (int jj = tid; jj < Hh; jj += DNT) {
            float sw = 0.0f, sb = 0.0f;
            int b = 0;
            for (; b + 8 <= I; b += 8) {
                float dv[8], xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { dv[u] = d[(int64_t)(b + u) * Hh + jj]; xv[u] = xh[(int64_t)(b + u) * Hh + jj]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { sw = fma32(dv[u], xv[u], sw); sb = sb + dv[u]; }
            }
            for (; b < I; ++b) {
                const float dv = d[(int64_t)b * Hh + jj];
                sw = fma32(dv, xh[(int64_t)b * Hh + jj], sw);
                sb = sb + dv;
            }
            if (first) { g_w[jj] = sw; g_b[jj] = sb; }
            else { g_w[jj] = g_w[jj] + sw; g_b[jj] = g_b[jj] + sb; }
        }
        __syncthreads();
    }
    const int Hp = Hh | 1, R = (LNBUF - MAXW) / (2 * Hp);
    lfloat *wl = ln_lds_ptr(Ps), *bufd = wl + MAXW, *bufx = bufd + R * Hp;
    for (int jj = tid; jj < Hh; jj += DNT) wl[jj] = w[jj];
    for (int r0 = 0; r0 < I; r0 += R) {
        const int nr = I - r0 < R ? I - r0 : R;
        ln_copy_rows_in<4>(d + (int64_t)r0 * Hh, Hh, nr, Hh, bufd, Hp, lane_, wave_);
        ln_copy_rows_in<4>(xh + (int64_t)r0 * Hh, Hh, nr, Hh, bufx, Hp, lane_, wave_);
        __syncthreads();
        for (int b = tid; b < nr; b += DNT) {
            lfloat *dr = bufd + b * Hp;
            const lfloat *xr = bufx + b * Hp;
            float s1 = 0.0f, s2 = 0.0f;
            {
                int jj = 0;
                for (; jj + 8 <= Hh; jj += 8) {
                    float v[8], wv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { v[u] = dr[jj + u]; wv[u] = wl[jj + u]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) dr[jj + u] = v[u] * wv[u];
                }
                for (; jj < Hh; ++jj) dr[jj] = dr[jj] * wl[jj];
            }
#pragma unroll 16
            for (int jj = 0; jj < Hh; ++jj) s1 = s1 + dr[jj];
#pragma unroll 16
            for (int jj = 0; jj < Hh; ++jj) s2 = fma32(dr[jj], xr[jj], s2);
            const float m1 = s1 / (float)Hh, m2 = s2 / (float)Hh, r = rstd[r0 + b];
            {
                int jj = 0;
                for (; jj + 8 <= Hh; jj += 8) {
                    float v[8], xv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { v[u] = dr[jj + u]; xv[u] = xr[jj + u]; }
#pragma unroll
                    for (int u = 0; u < 8; ++u) dr[jj + u] = fma32(-xv[u], m2, v[u] - m1) * r;
                }
                for (; jj < Hh; ++jj) dr[jj] = fma32(-xr[jj], m2, dr[jj] - m1) * r;
            }
        }
        __syncthreads();
        for (int b = wave_; b < nr; b += DNW) {
            float *dst = d + (int64_t)(r0 + b) * Hh;
            for (int jj = lane_; jj < Hh; jj += 64) dst[jj] = bufd[b * Hp + jj];
        }
        __syncthreads();
    }
}


}  // namespace lenv
