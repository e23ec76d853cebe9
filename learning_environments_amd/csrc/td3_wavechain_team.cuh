// td3_wavechain_team.cuh -- the learn step of td3_wavechain_kernel for a chain on a team of G >= 3 workgroups (a member owns one
// or two of the six -- batch 256: eight -- 32-sample blocks of the minibatch).  Included by td3_wavechain.hip; same products and the same canonical
// k-ascending chains as the one-workgroup routines there -- hence the same bits -- but built for a member that has few samples and
// a whole CU to itself:
//
//   * NO LDS weight images.  A sample block runs on a QUAD of waves (wave 4 q + jt computes output tile jt = 32 units of every layer);
//     the MFMA A operand of a wave -- its 32 columns of the K-major weight matrix -- comes straight from the arena (L2) into 64 VGPRs
//     (lane (unit li, half h) reads Wt[(2 t + h) * 128 + 32 jt + li]: two 128-byte segments per instruction, every weight read once
//     per member) while layer 1 runs; nothing is staged, no barrier waits for an image.  The backward chains reduce over the units and
//     need the transposed matrix: the optimizer epilogue below keeps a second, unit-major copy W2u[j][k] of every online W2 for them.
//   * Two quads = two independent passes side by side (the twin critics, their targets): the second quad's epilogues, exchanges and
//     loads fill the matrix-pipe bubbles of the first.
//   * The quad exchanges the operand registers of the next layer through a 16 KB LDS buffer per quad and stage (bufA).
//   * The 1- and 6-output layers (critic / actor heads, d/da of the policy step) run on v_mfma_f32_16x16x4_f32, one 16-sample half per
//     wave: 32 instructions of 32 cycles instead of 64 of 64 for a tile that is 3-19 % full.
//   * The parameter gradients are a list of wave jobs dealt over ALL waves of the team (16 W2 tiles + 4 W1 tiles on the matrix cores,
//     bias / output-layer sums on the vector unit): operands straight from the row-major activation / gradient arrays in the arena,
//     rolling register prefetch, one i-ascending chain per output.  The job that produced a gradient applies torch.optim.Adam and the
//     Polyak update to ITS parameters right away (tile transposed through LDS so that all arena accesses are 16 bytes per lane): the
//     gradient never reaches the arena and there is no barrier between the gradient and the optimizer phases.
#pragma once

namespace lenv {

using namespace wc;

// head on v_mfma_f32_16x16x4_f32: out^T[c][sample] = sum_u Wl[u][c] * x[sample][u] (u ascending), c < nout <= 8, for the samples
// 16 half .. 16 half + 15 of the block whose activations sit in the exchange buffer `xch` in operand order (xch_put).  wl: element
// (u, c) at wl[u * ldw + c * ldc] in the arena.  Lane l gets outputs c = 4 (l >> 4) + v, v = 0..3, of sample 16 half + (l & 15).
// The A operand (32 registers) is loaded by head16_load long before the chain needs it.
__device__ __forceinline__ void head16_load(const float *wl_, int ldw, int ldc, int lane, int nout, float (&a)[32])
{
    const int n = lane & 15, j = lane >> 4;
    const int cc = n < nout ? n : nout - 1;
    const gfloat *pa = (const gfloat *)wl_ + j * ldw + cc * ldc;
#pragma unroll
    for (int t = 0; t < 32; ++t) a[t] = pa[4 * t * ldw];
}
__device__ __forceinline__ f32x4 head16_chain(const float (&a)[32], const float *xch_, int half, int lane)
{
    const int n = lane & 15, j = lane >> 4;
    const lfloat *pb = (const lfloat *)xch_ + 128 * (j >> 1) + 32 * (j & 1) + 16 * half + n;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < 32; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], pb[(4 * (t >> 1) + (t & 1)) * 64], acc, 0, 0, 0);
    return acc;
}

// ---- one or two network passes (quad q runs job q) over this member's sample blocks ----------------------------------------------
// mode 0 (Critic_Q): q_out[i] = net(x)      mode 1 (Actor_TD3): Y[i][ocol + c] = tanh(net(x)) * max_action, th_out[i][c] = tanh (if given)
// d_h1 / r_h2 (>= 0): plain [sample][unit] copies of the hidden activations in the dump area (for the backward pass)
template <int ACT, int IN, int OUT, int NBK>
__device__ __noinline__ void t3v_forward(const T3wCtx *ctx_, int njobs_, int ldx_, int mode_,
                                         const float *par0_, const float *X0_, float *q_out0_, int d_h1_0_, int r_h2_0_,
                                         const float *par1_, const float *X1_, float *q_out1_, int d_h1_1_, int r_h2_1_,
                                         float *Y0_, float *th_out0_, float *Y1_, float *th_out1_, int ldy_, int ocol_)
{
    T3W_CTX_PROLOGUE;
    const int njobs = uni(njobs_), ldx = uni(ldx_), mode = uni(mode_), ldy = uni(ldy_), ocol = uni(ocol_);
    const int quad = wave >> 2, jt = wave & 3;
    const bool active = quad < njobs;
    const bool q1 = active && quad == 1;
    const float *par = uni_ptr(q1 ? par1_ : par0_), *X = uni_ptr(q1 ? X1_ : X0_);
    lfloat *q_out = (lfloat *)uni_ptr(q1 ? q_out1_ : q_out0_);
    const int d_h1 = uni(q1 ? d_h1_1_ : d_h1_0_), r_h2 = uni(q1 ? r_h2_1_ : r_h2_0_);
    float *Y = uni_ptr(q1 ? Y1_ : Y0_), *th_out = uni_ptr(q1 ? th_out1_ : th_out0_);
    constexpr int in = IN, out = OUT;
    const int nb = NBK / TG;
    float *xch0 = bufA + quad * 4096, *xch1 = bufA + 8192 + quad * 4096;
    TSUB_DECL;
    // the wave's weights: its 32 columns of W1t and W2t, the two bias slices, the output layer (waves that run a head).  Issue order =
    // use order: loads return in order, so what layer 1 of the first block needs goes first and the 64 W2 loads behind it
    float a2[64], wa[in >> 1], ha[32], xb0[in >> 1], xl0 = 0.0f;
    f32x4 b1v[4], b2v[4], wlv[4];
    const bool head_wave = active && jt < 2;
    if (active) {
        const gfloat *w1 = (const gfloat *)par + oW1t + L.h * W + 32 * jt + L.li;
#pragma unroll
        for (int t = 0; t < (in >> 1); ++t) wa[t] = w1[2 * t * W];
        {
            const int row = 32 * (tg * nb) + L.li;
            const gfloat *xr = (const gfloat *)X + row * ldx + L.h;
#pragma unroll
            for (int t = 0; t < (in >> 1); ++t) xb0[t] = xr[2 * t];
            if (in & 1) xl0 = ((const gfloat *)X)[row * ldx + in - 1];
        }
        if (in & 1) {
            const gfloat *wl = (const gfloat *)par + oW1t + (in - 1) * W + 32 * jt + 4 * L.h;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) wlv[g4] = *(const gf4 *)(wl + 8 * g4);
        }
        const gfloat *bp = (const gfloat *)par + ob1 + 32 * jt + 4 * L.h;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) b1v[g4] = *(const gf4 *)(bp + 8 * g4);
        __builtin_amdgcn_sched_barrier(0);
        const gfloat *w2 = (const gfloat *)par + oW2t + L.h * W + 32 * jt + L.li;
#pragma unroll
        for (int t = 0; t < 64; ++t) a2[t] = w2[2 * t * W];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) b2v[g4] = *(const gf4 *)(bp + (ob2 - ob1) + 8 * g4);
        if (head_wave) head16_load(par + oWo, 8, 1, L.lane, out, ha);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 1
    for (int bi = 0; bi < nb; ++bi) {
        const int blk = tg * nb + bi, row = 32 * blk + L.li;
        float r16[16], rf[64];
        f32x16 acc;
        if (active) {
            float xb[in >> 1], xl = xl0;
#pragma unroll
            for (int t = 0; t < (in >> 1); ++t) xb[t] = xb0[t];
            if (bi > 0) {
                const gfloat *xr = (const gfloat *)X + row * ldx + L.h;
#pragma unroll
                for (int t = 0; t < (in >> 1); ++t) xb[t] = xr[2 * t];
                if (in & 1) xl = ((const gfloat *)X)[row * ldx + in - 1];
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#ifdef LENV_PHASE_TIMING_SUB
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (diagnostic build: how long until ALL the weights are there)
            TSUB_MARK(42);
#endif
#pragma unroll
            for (int t = 0; t < (in >> 1); ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[t], xb[t], acc, 0, 0, 0);
            if (in & 1) {                                  // an odd last k is one fmaf per output
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) acc[4 * g4 + cc] = fma32(xl, wlv[g4][cc], acc[4 * g4 + cc]);
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) r16[4 * g4 + cc] = act_fwd(ACT, prelu, acc[4 * g4 + cc] + b1v[g4][cc]);
            if (d_h1 >= 0) {
                gfloat *rm = (gfloat *)dump_of(d_h1, 0) + row * W + 32 * jt + 4 * L.h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) *(gf4 *)(rm + 8 * g4) = f32x4{r16[4 * g4], r16[4 * g4 + 1], r16[4 * g4 + 2], r16[4 * g4 + 3]};
            }
            tile16_to_operand(r16);
            xch_put(xch0, jt, L.lane, r16);
        }
        TSUB_MARK(43);
        barrier_lds();
        TSUB_MARK(30);
        if (active) {
            xch_get(xch0, L.lane, rf);
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll
            for (int t = 0; t < 64; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[t], rf[breg_of(t)], acc, 0, 0, 0);
#ifdef LENV_PHASE_TIMING_SUB
            asm volatile("s_nop 0" :: "v"(acc[0]) : "memory");      // (diagnostic build: the chain has drained)
            TSUB_MARK(44);
#endif
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) r16[4 * g4 + cc] = act_fwd(ACT, prelu, acc[4 * g4 + cc] + b2v[g4][cc]);
            if (r_h2 >= 0) {
                gfloat *rm = (gfloat *)dump_of(r_h2, 0) + row * W + 32 * jt + 4 * L.h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) *(gf4 *)(rm + 8 * g4) = f32x4{r16[4 * g4], r16[4 * g4 + 1], r16[4 * g4 + 2], r16[4 * g4 + 3]};
            }
            tile16_to_operand(r16);
            xch_put(xch1, jt, L.lane, r16);
        }
        TSUB_MARK(45);
        barrier_lds();
        TSUB_MARK(31);
        if (head_wave) {
            // output layer: 16 samples per wave
            const f32x4 hacc = head16_chain(ha, xch1, jt, L.lane);
            const int n = L.lane & 15, orow = 32 * blk + 16 * jt + n;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int cidx = 4 * (L.lane >> 4) + v;
                if (cidx < out) {
                    const float z = hacc[v] + ((const gfloat *)par)[obo + cidx];
                    if (mode == 0) q_out[orow] = z;
                    else {
                        const float th = det_tanhf(lenv_tanh_table, z);
                        if (th_out) th_out[orow * out + cidx] = th;
                        Y[orow * ldy + ocol + cidx] = th * ma;
                    }
                }
            }
        }
        TSUB_MARK(32);
    }
    __syncthreads();
    TSUB_MARK(33);
}

// ---- backward of one or two networks, first half: the per-sample chain from dOut[i][out] (LDS) back to dz2 and dh1 (row-major copies
// r_dz2 / r_dh1 in the arena for the weight gradients; < 0: not needed) and, for the policy step (job 0, dx_n > 0), the action part of the
// input gradient turned into the actor's output gradient dz (LDS).  w2u: the unit-major copy W2u[j][k] of the net's second layer. ----
template <int ACT, int IN, int OUT, int NBK>
__device__ __noinline__ void t3v_backward(const T3wCtx *ctx_, int njobs_,
                                          const float *par0_, const float *w2u0_, const float *dOut0_, int d_h1_0_, int r_h2_0_, int r_dz2_0_, int r_dh1_0_,
                                          const float *par1_, const float *w2u1_, const float *dOut1_, int d_h1_1_, int r_h2_1_, int r_dz2_1_, int r_dh1_1_,
                                          int dx_col_, int dx_n_, const float *th_, float *dz_out_)
{
    T3W_CTX_PROLOGUE;
    const int njobs = uni(njobs_), dx_col = uni(dx_col_), dx_n = uni(dx_n_);
    const int quad = wave >> 2, jt = wave & 3;
    const bool active = quad < njobs;
    const bool q1 = active && quad == 1;
    const float *par = uni_ptr(q1 ? par1_ : par0_), *w2u = uni_ptr(q1 ? w2u1_ : w2u0_);
    const lfloat *dOut = (const lfloat *)uni_ptr(q1 ? dOut1_ : dOut0_);
    const int d_h1 = uni(q1 ? d_h1_1_ : d_h1_0_), r_h2 = uni(q1 ? r_h2_1_ : r_h2_0_), r_dz2 = uni(q1 ? r_dz2_1_ : r_dz2_0_),
              r_dh1 = uni(q1 ? r_dh1_1_ : r_dh1_0_);
    const float *th = uni_ptr(th_);
    lfloat *dz_out = (lfloat *)uni_ptr(dz_out_);
    constexpr int out = OUT;
    const int nb = NBK / TG;
    float *xch0 = bufA + quad * 4096, *xch1 = bufA + 8192 + quad * 4096;
    TSUB_DECL;
    // the wave's weights: the output-layer rows of its 16 units, its 32 columns k of W2u (A[k][j] = W2[j][k]) and -- policy step -- the
    // action rows of W1 for d/da.  Issue order = use order (loads return in order): the first block's activations before the 64 W2u loads
    float a2[64], ha[32];
    f32x4 wov[4][4][(out > 4) ? 2 : 1], hv2f[4], hv1f[4];
    const bool dx_wave = active && quad == 0 && jt < 2 && dx_n > 0;
    if (active) {
        const gfloat *wp = (const gfloat *)par + oWo + (32 * jt + 4 * L.h) * 8;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                wov[g4][cc][0] = *(const gf4 *)(wp + (8 * g4 + cc) * 8);
                if (out > 4) wov[g4][cc][(out > 4) ? 1 : 0] = *(const gf4 *)(wp + (8 * g4 + cc) * 8 + 4);
            }
        {
            const int row = 32 * (tg * nb) + L.li;
            const gfloat *hd = (const gfloat *)dump_of(r_h2, 0) + row * W + 32 * jt + 4 * L.h;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) hv2f[g4] = *(const gf4 *)(hd + 8 * g4);
            const gfloat *h1p = (const gfloat *)dump_of(d_h1, 0) + row * W + 32 * jt + 4 * L.h;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) hv1f[g4] = *(const gf4 *)(h1p + 8 * g4);
        }
        __builtin_amdgcn_sched_barrier(0);
        const gfloat *w2 = (const gfloat *)w2u + L.h * W + 32 * jt + L.li;
#pragma unroll
        for (int t = 0; t < 64; ++t) a2[t] = w2[2 * t * W];
        if (dx_wave) head16_load(par + oW1t + dx_col * W, 1, W, L.lane, dx_n, ha);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 1
    for (int bi = 0; bi < nb; ++bi) {
        const int blk = tg * nb + bi, row = 32 * blk + L.li;
        float r16[16], rf[64];
        f32x16 acc;
        f32x4 hv1[4];
        if (active) {
            // dz2 = act'(h2) * (sum_c dOut[i][c] Wo[c][unit], c ascending from 0), the 16 units of this lane in tile jt
            f32x4 hv2[4];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) { hv2[g4] = hv2f[g4]; hv1[g4] = hv1f[g4]; }
            if (bi > 0) {
                const gfloat *hd = (const gfloat *)dump_of(r_h2, 0) + row * W + 32 * jt + 4 * L.h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) hv2[g4] = *(const gf4 *)(hd + 8 * g4);
                const gfloat *h1p = (const gfloat *)dump_of(d_h1, 0) + row * W + 32 * jt + 4 * L.h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) hv1[g4] = *(const gf4 *)(h1p + 8 * g4);
            }
            float dO[out];
#pragma unroll
            for (int cc = 0; cc < out; ++cc) dO[cc] = dOut[row * out + cc];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    float up = 0.0f;
#pragma unroll
                    for (int o = 0; o < out; ++o) up = fma32(dO[o], wov[g4][cc][o >> 2][o & 3], up);
                    r16[4 * g4 + cc] = act_bwd(ACT, prelu, hv2[g4][cc], up);
                }
            if (r_dz2 >= 0) {
                gfloat *rm = (gfloat *)dump_of(r_dz2, 0) + row * W + 32 * jt + 4 * L.h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) *(gf4 *)(rm + 8 * g4) = f32x4{r16[4 * g4], r16[4 * g4 + 1], r16[4 * g4 + 2], r16[4 * g4 + 3]};
            }
            tile16_to_operand(r16);
            xch_put(xch0, jt, L.lane, r16);
        }
        barrier_lds();
        TSUB_MARK(35);
        if (active) {
            xch_get(xch0, L.lane, rf);
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
#pragma unroll
            for (int t = 0; t < 64; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[t], rf[breg_of(t)], acc, 0, 0, 0);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) r16[4 * g4 + cc] = act_bwd(ACT, prelu, hv1[g4][cc], acc[4 * g4 + cc]);
            if (r_dh1 >= 0) {
                gfloat *rm = (gfloat *)dump_of(r_dh1, 0) + row * W + 32 * jt + 4 * L.h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) *(gf4 *)(rm + 8 * g4) = f32x4{r16[4 * g4], r16[4 * g4 + 1], r16[4 * g4 + 2], r16[4 * g4 + 3]};
            }
            if (dx_n > 0) {
                tile16_to_operand(r16);
                xch_put(xch1, jt, L.lane, r16);
            }
        }
        barrier_lds();
        TSUB_MARK(36);
        if (dx_wave) {
            // dX[i][dx_col + c] = sum_u dh1[i][u] W1[u][dx_col + c] (u ascending), c < dx_n, then dz = (dX * max_action) * (1 - th^2)
            const f32x4 hacc = head16_chain(ha, xch1, jt, L.lane);
            const int n = L.lane & 15, orow = 32 * blk + 16 * jt + n;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int cidx = 4 * (L.lane >> 4) + v;
                if (cidx < dx_n) {
                    const float t_ = th[orow * dx_n + cidx];
                    dz_out[orow * dx_n + cidx] = (hacc[v] * ma) * fma32(-t_, t_, 1.0f);
                }
            }
        }
        TSUB_MARK(37);
    }
    __syncthreads();
    TSUB_MARK(38);
}

// ---- parameter gradients + optimizer step of one or two networks as wave jobs over the whole team --------------------------------
// Per network 26 jobs: 0-15 W2 tiles (kt = n / 4, jt = n % 4), 16-19 W1 column tiles, 20/21 b1 halves, 22/23 b2 halves, 24/25 output-layer
// halves (24 also the output bias).  Matrix-core jobs come first in the global list (nets interleaved) so that every SIMD of the team
// gets its share of them before the vector jobs fill the remaining waves: job n runs on SIMD n % (4 G) of the team, round n / (4 G);
// even rounds on the SIMD's first wave, odd rounds on its second.
struct T3vNet {
    float *par;            // this net's parameters in the arena (targets / Adam m, v at the same offsets of their arrays)
    float *w2u;            // unit-major copy of W2 (online nets)
    const float *dOut;     // [B][OUT] in LDS
    int d_h1, r_h2, r_dz2, r_dh1;
};

// acc = sum_{i < B} A[i][ca] * Bm[i][cb] (B = 2 NS rows), i ascending in ONE chain: lane (li, h) reads rows 2 t + h of the two row-major arrays
// (pa / pb already point at its row h and column), D k-steps of operands in flight (more in flight did not help: 28 deep the eight waves' outstanding lines
// overran the 32 KB L1 and the chain got slower)
#ifdef T3V_DIAG_WIDE
// timing experiment (results are garbage): the same bytes per k-step, but fetched as ONE 16-byte load per operand every FOUR steps -- a
// quarter of the vector-memory instructions
template <int LDA, int NS>
__device__ __forceinline__ f32x16 t3v_wgrad_chain(const gfloat *pa, const gfloat *pb)
{
    constexpr int D = 16;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
    f32x4 ra[D / 4], rb[D / 4];
    const gf4 *qa = (const gf4 *)((uintptr_t)pa & ~(uintptr_t)15), *qb = (const gf4 *)((uintptr_t)pb & ~(uintptr_t)15);
#pragma unroll
    for (int u = 0; u < D / 4; ++u) { ra[u] = qa[u * 2 * LDA / 4 * 4 / 4 + u]; rb[u] = qb[u * 64 + u]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[(t % D) / 4][t & 3], rb[(t % D) / 4][t & 3], acc, 0, 0, 0);
        if ((t & 3) == 3 && t + D - 3 < NS) { const int u = ((t - 3) % D) / 4; ra[u] = qa[((t + D - 3) / 4) * 8 * ((LDA + 3) / 4)]; rb[u] = qb[((t + D - 3) / 4) * 8 * 32]; }
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}
#else
template <int LDA, int NS>
__device__ __forceinline__ f32x16 t3v_wgrad_chain(const gfloat *pa, const gfloat *pb)
{
    constexpr int D = 14;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
    float ra[D], rb[D];
#pragma unroll
    for (int u = 0; u < D; ++u) { ra[u] = pa[2 * u * LDA]; rb[u] = pb[2 * u * W]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < NS; ++t) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[t % D], rb[t % D], acc, 0, 0, 0);
#ifdef T3V_DIAG_SAME_ROWS                                   // timing experiment: every operand load hits the same two rows (L1 hits)
        if (t + D < NS) { ra[t % D] = pa[2 * ((t + D) & 1) * LDA]; rb[t % D] = pb[2 * ((t + D) & 1) * W]; }
#else
        if (t + D < NS) { ra[t % D] = pa[2 * (t + D) * LDA]; rb[t % D] = pb[2 * (t + D) * W]; }
#endif
        // keep the written order -- one matrix instruction, then the two loads D k-steps ahead: left alone, the scheduler
        // bunches the loads and drains the queue (s_waitcnt vmcnt(0)) every dozen instructions
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}
#endif

#ifdef LENV_PHASE_TIMING_SUB
// diagnostic: cycles per job class of the gradient phase -- [phase (critics / actor)][class: W2 tile, W1 tile, bias, output layer, epilogue
// share of the tile jobs][sum, count] over all waves of chain 0's team
__device__ unsigned long long g_t3v_jobs[2][5][2];
#define TJOB_T0 const unsigned long long jb0 = __builtin_readcyclecounter()
#define TJOB_ADD(cls) do { if (L.lane == 0 && blockIdx.x < 8u * (unsigned)TG && (blockIdx.x & 7u) == 0u) { atomicAdd(&g_t3v_jobs[nnets == 2 ? 0 : 1][cls][0], __builtin_readcyclecounter() - jb0); atomicAdd(&g_t3v_jobs[nnets == 2 ? 0 : 1][cls][1], 1ull); } } while (0)
#else
#define TJOB_T0
#define TJOB_ADD(cls)
#endif

// polyak_ == 0 (a learn step without the delayed policy update, TD3.py:101): the targets stay as they are (tau = 0, 1 - tau = 1: the
// epilogue's tau * w + (1 - tau) * t returns t)
template <int ACT, int IN, int OUT, int LDX, int NBK>
__device__ __noinline__ void t3v_wgrad(const T3wCtx *ctx_, int nnets_, const T3vNet n0_, const T3vNet n1_, const float *X_, int ldx_, int ac_slot_, int polyak_)
{
    T3W_CTX_PROLOGUE;
    const int nnets = uni(nnets_), ac_slot = uni(ac_slot_);
    (void)ldx_;
    constexpr int in = IN, out = OUT, B = 32 * NBK, CH = B % 48 == 0 ? 48 : 32;      // (CH: rows per batch of loads of the vector jobs) X = the gathered [s, a] rows (LDX = S + A); the actor reads their first S columns
    const float *X = uni_ptr(X_);
    const int64_t o_t = 3 * PN, o_m = 6 * PN, o_v = 9 * PN;     // params | targets | adam_m | adam_v (3 nets each)
    volatile lfloat *ctrl = (volatile lfloat *)uni_ptr(c->ctrl);
    const AdamConsts ac{ ctrl[ac_slot], ctrl[ac_slot + 1], unif(c->w1), unif(c->w2), unif(c->beta2), unif(c->aeps) };
    const bool polyak = uni(polyak_) != 0;
    const float tau = polyak ? unif(c->tau) : 0.0f, omt = polyak ? unif(c->omt) : 1.0f;
    float *tile = bufB + wave * 1024, *tileT = bufA + wave * 1056;     // the wave's gradient tile and its transposed copy ([32][33])
    constexpr int NJ = 24 + 2 * out;                       // jobs per net: 16 + 4 matrix-core tiles, 2 + 2 bias halves, 2 per output unit
    // Dealing the jobs.  Matrix-core jobs in the order (net, W2 tiles by (kt, jt), W1 tiles) go to the members in BLOCKS: a member's six or
    // seven tiles share their operand panels (the four jt tiles of one kt read the same B x 32 slice of h1, tiles of one jt the same slice
    // of dz2), its waves walk them in step, and what one wave has pulled into the CU's L1 the next finds there -- dealt round robin over
    // the team every tile fetched both its panels through the L1's 64-byte port and the phase ran at that port's speed.  The vector jobs
    // are dealt round robin and run on the member's remaining waves.  Wave w of a member runs entries w, w + 8, ... of its list.
    const int n_mfma = 20 * nnets, n_valu = (NJ - 20) * nnets;
    const int mlo = (tg * n_mfma + TG - 1) / TG, mhi = ((tg + 1) * n_mfma + TG - 1) / TG, nm = mhi - mlo;
    const int nv = (n_valu - tg + TG - 1) / TG;              // vector jobs tg, tg + TG, ...
    // Wave w of a member: with fewer matrix-core jobs than waves (G = 6: six or seven) wave w < nm runs tile job w and the vector jobs
    // share the waves behind them (a tile job with its optimizer epilogue is the longest job: nobody runs anything after one);
    // otherwise (G = 3) entries w, w + 8, ... of the member's list, tiles first.
    const bool spare = nm < NW;
    const int q0 = spare ? (wave < nm ? wave : nm + (wave - nm)) : wave, qstep = spare ? (wave < nm ? NW * 1024 : NW - nm) : NW;
    TSUB_DECL;
#pragma unroll 1
    for (int q = q0; q < nm + nv; q += qstep) {
        int net, job;
        if (q < nm) { const int m = mlo + q; net = m / 20; job = m - 20 * net; }
        else { const int v = tg + (q - nm) * TG; net = v % nnets; job = 20 + v / nnets; }
        const bool second = net == 1;
        float *par = uni_ptr(second ? n1_.par : n0_.par), *w2u = uni_ptr(second ? n1_.w2u : n0_.w2u);
        const lfloat *dOut = (const lfloat *)uni_ptr(second ? n1_.dOut : n0_.dOut);
        const int d_h1 = uni(second ? n1_.d_h1 : n0_.d_h1), r_h2 = uni(second ? n1_.r_h2 : n0_.r_h2), r_dz2 = uni(second ? n1_.r_dz2 : n0_.r_dz2),
                  r_dh1 = uni(second ? n1_.r_dh1 : n0_.r_dh1);
        float *tgt = par + o_t, *am = par + o_m, *av = par + o_v;
        L.refresh();
        TJOB_T0;
        if (job < 20) {
            // job < 16: gW2t[k][j] = sum_i h1[i][k] dz2[i][j], tile (kt, jt); else gW1t[k][j] = sum_i x[i][k] dh1[i][j] (k < in): A = the
            // minibatch inputs (lane = input column k, clamped), column tile jt
            const bool w2 = job < 16;
            const int kt = w2 ? job >> 2 : 0, jt = w2 ? job & 3 : job - 16;
            const int off = w2 ? oW2t + (32 * kt) * W + 32 * jt : oW1t + 32 * jt;
            T3vTileState st;
            t3v_tile_state(L, par + off, am + off, av + off, tgt + off, w2 ? 32 : in, st);     // (in flight behind the chain's operands)
            f32x16 acc;
            if (w2) {
                const gfloat *pa = (const gfloat *)dump_of(d_h1, 0) + L.h * W + 32 * kt + L.li;
                const gfloat *pb = (const gfloat *)dump_of(r_dz2, 0) + L.h * W + 32 * jt + L.li;
                acc = t3v_wgrad_chain<W, B / 2>(pa, pb);
            } else {
                const gfloat *pa = (const gfloat *)X + L.h * LDX + (L.li < in ? L.li : in - 1);
                const gfloat *pb = (const gfloat *)dump_of(r_dh1, 0) + L.h * W + 32 * jt + L.li;
                acc = t3v_wgrad_chain<LDX, B / 2>(pa, pb);
            }
            TJOB_ADD(4);
            t3v_tile_adam(acc, st, tile, tileT, L, par + off, am + off, av + off, tgt + off, w2 ? 32 : in, w2 ? w2u + (32 * jt) * W + 32 * kt : nullptr, ac, tau,
                          omt);
            TJOB_ADD(w2 ? 0 : 1);
        } else if (job < 24) {
            // gb1[j] = sum_i dh1[i][j], gb2[j] = sum_i dz2[i][j] (plain adds, i ascending): lane = unit, 64 units per job
            const int j = 64 * (job & 1) + L.lane;
            const gfloat *src = (const gfloat *)dump_of(job < 22 ? r_dh1 : r_dz2, 0) + j;
            const int off = (job < 22 ? ob1 : ob2) + j;
            const T3vAdam1State st1 = t3v_adam1_load(par, am, av, tgt, off);      // (in flight behind the column's loads)
            float s = 0.0f;
#pragma unroll 1
            for (int i0 = 0; i0 < B; i0 += CH) {
                float x[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) x[u] = src[(i0 + u) * W];
#pragma unroll
                for (int u = 0; u < CH; ++u) s = s + x[u];
            }
            t3v_adam1_step(s, st1, par, am, av, tgt, off, ac, tau, omt);
            TJOB_ADD(2);
        } else {
            // output layer, output unit cc: gWo[k][cc] = sum_i dOut[i][cc] h2[i][k] (i ascending), lane = k, 64 per job; the job of the
            // first half also sums gbo[cc] = sum_i dOut[i][cc] (plain adds) on the way
            const int cc = (job - 24) >> 1, k = 64 * (job & 1) + L.lane;
            const gfloat *src = (const gfloat *)dump_of(r_h2, 0) + k;
            const T3vAdam1State st1 = t3v_adam1_load(par, am, av, tgt, oWo + k * 8 + cc);
            const bool bias_lane = (job & 1) == 0 && L.lane == 0;
            T3vAdam1State stb{};
            if (bias_lane) stb = t3v_adam1_load(par, am, av, tgt, obo + cc);
            float s = 0.0f, sb = 0.0f;
#pragma unroll 1
            for (int i0 = 0; i0 < B; i0 += CH) {
                float x[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) x[u] = src[(i0 + u) * W];
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const float d = dOut[(i0 + u) * out + cc];
                    s = fma32(d, x[u], s);
                    sb = sb + d;
                }
            }
            t3v_adam1_step(s, st1, par, am, av, tgt, oWo + k * 8 + cc, ac, tau, omt);
            if (bias_lane) t3v_adam1_step(sb, stb, par, am, av, tgt, obo + cc, ac, tau, omt);
            TJOB_ADD(3);
        }
    }
    TSUB_MARK(40);
    __syncthreads();
    TSUB_MARK(41);
}

}  // namespace lenv
