// lenv_icm.cuh -- the reference's Intrinsic Curiosity Module (models/icm_baseline.py:8-172) inside the GEMM-tiled inner loops:
// ICM.train on the gathered minibatch + compute_intrinsic_rewards with the updated model (agents/DDQN.py:74-76,
// agents/TD3.py:68-70).  oracle/lenv_oracle_icm.inc is the canonical restatement; arithmetic and order are the same.
#pragma once

#include "lenv_gemm.cuh"

namespace lenv {

// ---- Intrinsic Curiosity Module (models/icm_baseline.py:8-172; oracle/lenv_oracle_icm.inc is the canonical restatement) ----
struct IcmLin { int in, out, oW, ob; };
struct IcmNet { int S, A, Ai, F, H, C, P; IcmLin feat[3], inv[3], pre[3], res[4][2], post[2]; };

__host__ __device__ inline void icm_lin_set(IcmLin &L, int in, int out, int &o) { L.in = in; L.out = out; L.oW = o; o += in * out; L.ob = o; o += out; }

// ICMModel.__init__ in state-dict order: features S-H-H-F, inverse 2F-H-H-Ai, forward_pre (F+Ai)-H-H-F, four residual blocks
// {fc1, fc2: (F+Ai)-F}, forward_post F-H-F; Ai = 1 for a two-action discrete env (icm_baseline.py:39-40)
__host__ __device__ inline void icm_build(IcmNet &n, int S, int A, int F, int H, bool discrete)
{
    int o = 0;
    n.S = S; n.A = A; n.Ai = (discrete && A == 2) ? 1 : A; n.F = F; n.H = H; n.C = F + n.Ai;
    icm_lin_set(n.feat[0], S, H, o); icm_lin_set(n.feat[1], H, H, o); icm_lin_set(n.feat[2], H, F, o);
    icm_lin_set(n.inv[0], 2 * F, H, o); icm_lin_set(n.inv[1], H, H, o); icm_lin_set(n.inv[2], H, n.Ai, o);
    icm_lin_set(n.pre[0], n.C, H, o); icm_lin_set(n.pre[1], H, H, o); icm_lin_set(n.pre[2], H, F, o);
    for (int k = 0; k < 4; ++k) { icm_lin_set(n.res[k][0], n.C, F, o); icm_lin_set(n.res[k][1], n.C, F, o); }
    icm_lin_set(n.post[0], F, H, o); icm_lin_set(n.post[1], H, F, o);
    n.P = o;
}

// per-chain ICM buffers in the arena
enum { IB_P, IB_M, IB_V, IB_G, IB_X2, IB_ACT, IB_FH0, IB_FH1, IB_FE, IB_IIN, IB_IH0, IB_IH1, IB_IZ, IB_PIN, IB_PH0, IB_PH1,
       IB_XC0, IB_XC1, IB_XC2, IB_XC3, IB_XC4, IB_HC0, IB_HC1, IB_HC2, IB_HC3, IB_QH, IB_PRED,
       IB_DPRED, IB_DZ, IB_DQH, IB_DX, IB_DH, IB_DPH1, IB_DPH0, IB_DFE, IB_DIH1, IB_DIH0, IB_DIIN, IB_DFH1, IB_DFH0, IB_COUNT };


// floats of every ICM buffer of one chain (minibatch B)
__host__ __device__ inline void icm_buffer_sizes(const IcmNet &n, int B, int64_t (&sz)[IB_COUNT])
{
    const int64_t b = B, b2 = 2 * (int64_t)B, H = n.H, F = n.F, C = n.C, Ai = n.Ai;
    const int64_t v[IB_COUNT] = { n.P, n.P, n.P, n.P, b2 * n.S, b * Ai, b2 * H, b2 * H, b2 * F, b * 2 * F, b * H, b * H, b * Ai, b * C, b * H, b * H,
                                  b * C, b * C, b * C, b * C, b * C, b * C, b * C, b * C, b * C, b * H, b * F,
                                  b * F, b * Ai, b * H, b * F, b * F, b * H, b * H, b2 * F, b * H, b * H, b * 2 * F, b2 * H, b2 * H };
    for (int i = 0; i < IB_COUNT; ++i) sz[i] = v[i];
}

// what one ICM step needs from the kernel around it
struct IcmStep {
    GemmQueue &gq; float *Ps, *Qs;            // the kernel's GEMM queue and its LDS staging buffers
    float *arena; const int64_t *a_icm;       // the chain's arena and the offsets of its ICM buffers (IB_*)
    volatile float *ctrl; int ci;             // LDS control words: ctrl[ci], ctrl[ci+1] carry the Adam bias corrections
    const IcmNet &icm; double *pows;          // layout; running beta1^t, beta2^t (thread 0's copy is the live one)
    double lr, beta, eta, adam_beta1, adam_beta2, adam_eps;
    int B;
    const float *s_rows; int lds;             // states [B][lds] and next states [B][ldn] of the minibatch
    const float *n_rows; int ldn;
    bool continuous;                          // TD3: the action vector is the input and the inverse loss is an MSE
};

// action_of(b, i): action input i of sample b; add_reward(b, r): rewards[b] += r (called by one thread per sample)
template <int MAXI, class ActFn, class RewFn>
__device__ __forceinline__ void icm_train_and_reward(const IcmStep &c, ActFn action_of, RewFn add_reward)
{
    const int tid = (int)threadIdx.x;
    GemmQueue &gq = c.gq;
    float *const Ps = c.Ps, *const Qs = c.Qs, *const arena = c.arena;
    const IcmNet &icm = c.icm;
    volatile float *ctrl = c.ctrl;
    const int B = c.B, S = icm.S, A = icm.A;
    const int F = icm.F, Hi = icm.H, Ai = icm.Ai, C = icm.C;
    auto buf = [&](int i) { return arena + c.a_icm[i]; };
    float *ip = buf(IB_P), *ig = buf(IB_G), *X2 = buf(IB_X2), *actin = buf(IB_ACT);
    float *fh0 = buf(IB_FH0), *fh1 = buf(IB_FH1), *fe = buf(IB_FE), *iin = buf(IB_IIN), *ih0 = buf(IB_IH0), *ih1 = buf(IB_IH1), *iz = buf(IB_IZ);
    float *pin = buf(IB_PIN), *ph0 = buf(IB_PH0), *ph1 = buf(IB_PH1), *qh = buf(IB_QH), *pred = buf(IB_PRED);
    float *xc[5], *hc[4];
    for (int k = 0; k < 5; ++k) xc[k] = buf(IB_XC0 + k);
    for (int k = 0; k < 4; ++k) hc[k] = buf(IB_HC0 + k);
    constexpr int LK = LENV_ACT_LEAKYRELU, RL = LENV_ACT_RELU;
    auto lin = [&](const IcmLin &L, const float *X, int ldx, int rows, float *Y, int ldy, int act) {
        if (act >= 0) gq.gemm(X, ldx, 1, ip + L.oW, L.in, 1, rows, L.out, L.in, epi_bias_act(Y, ldy, ip + L.ob, act, 0.25f));
        else gq.gemm(X, ldx, 1, ip + L.oW, L.in, 1, rows, L.out, L.in, epi_bias(Y, ldy, 0, ip + L.ob));
    };
    // stacked states, action inputs (one logit target for two actions, else one-hot: icm_baseline.py:39-40,134-136) and
    // the action columns of every concatenated buffer
    for (int e = tid; e < B * S; e += DNT) { const int b = e / S, i = e - b * S; X2[e] = c.s_rows[b * c.lds + i]; X2[B * S + e] = c.n_rows[b * c.ldn + i]; }
    for (int e = tid; e < B * Ai; e += DNT) {
        const int b = e / Ai, i = e - b * Ai;
        const float av = action_of(b, i);
        actin[e] = av;
        pin[b * C + F + i] = av;
        for (int k = 0; k < 5; ++k) xc[k][b * C + F + i] = av;
        for (int k = 0; k < 4; ++k) hc[k][b * C + F + i] = av;
    }
    __syncthreads();
    auto icm_forward = [&]() {                                  // ICMModel.forward (icm_baseline.py:80-102)
        lin(icm.feat[0], X2, S, 2 * B, fh0, Hi, LK); lin(icm.feat[1], fh0, Hi, 2 * B, fh1, Hi, LK); lin(icm.feat[2], fh1, Hi, 2 * B, fe, F, -1);
        gq.run<MAXI>(Ps, Qs);
        for (int e = tid; e < B * F; e += DNT) {
            const int b = e / F, f = e - b * F;
            const float vs = fe[e], vn = fe[B * F + e];
            iin[b * 2 * F + f] = vs; iin[b * 2 * F + F + f] = vn; pin[b * C + f] = vs;
        }
        __syncthreads();
        lin(icm.inv[0], iin, 2 * F, B, ih0, Hi, RL); lin(icm.inv[1], ih0, Hi, B, ih1, Hi, RL); lin(icm.inv[2], ih1, Hi, B, iz, Ai, -1);
        lin(icm.pre[0], pin, C, B, ph0, Hi, LK); lin(icm.pre[1], ph0, Hi, B, ph1, Hi, LK); lin(icm.pre[2], ph1, Hi, B, xc[0], C, -1);
        for (int k = 0; k < 4; ++k) {
            lin(icm.res[k][0], xc[k], C, B, hc[k], C, LK);
            const IcmLin &L2 = icm.res[k][1];                   // x_{k+1} = x_k + fc2([fc1([x_k | a]) | a])
            gq.gemm(hc[k], C, 1, ip + L2.oW, L2.in, 1, B, L2.out, L2.in, epi_bias_add(xc[k + 1], C, ip + L2.ob, xc[k], C));
        }
        lin(icm.post[0], xc[4], C, B, qh, Hi, LK); lin(icm.post[1], qh, Hi, B, pred, F, -1);
        gq.run<MAXI>(Ps, Qs);
    };
    icm_forward();
    // ---- loss gradients: beta * mse(forward, features(s')) and (1 - beta) * BCEWithLogits | CrossEntropy(inverse, action) ----
    float *dpred = buf(IB_DPRED), *dz = buf(IB_DZ), *dqh = buf(IB_DQH), *dx = buf(IB_DX), *dh = buf(IB_DH), *dph1 = buf(IB_DPH1);
    float *dph0 = buf(IB_DPH0), *dfe = buf(IB_DFE), *dih1 = buf(IB_DIH1), *dih0 = buf(IB_DIH0), *diin = buf(IB_DIIN), *dfh1 = buf(IB_DFH1), *dfh0 = buf(IB_DFH0);
    const float c_mse = (float)(c.beta * 2.0 / ((double)B * F)), c_act = (float)((1.0 - c.beta) / (double)B);
    const float c_amse = (float)((1.0 - c.beta) * 2.0 / ((double)B * A));     // continuous actions: MSELoss on [B, A]
    for (int e = tid; e < B * F; e += DNT) dpred[e] = c_mse * (pred[e] - fe[B * F + e]);
    for (int b = tid; b < B; b += DNT) {
        if (c.continuous) {
            for (int i = 0; i < A; ++i) dz[b * A + i] = c_amse * (iz[b * A + i] - actin[b * A + i]);
        } else if (Ai == 1) {
            const float sg = fma32(0.5f, det_tanhf(lenv_tanh_table, 0.5f * iz[b]), 0.5f);
            dz[b] = c_act * (sg - actin[b]);
        } else {
            float mx = iz[b * A];
            for (int i = 1; i < A; ++i) if (iz[b * A + i] > mx) mx = iz[b * A + i];
            float ex[4], sm = 0.0f;                           // A <= 3 for the supported envs
            for (int i = 0; i < A; ++i) { ex[i] = det_expf(iz[b * A + i] - mx); sm = sm + ex[i]; }
            for (int i = 0; i < A; ++i) dz[b * A + i] = c_act * (ex[i] / sm - actin[b * A + i]);
        }
    }
    __syncthreads();
    // ---- backward: per Linear dW = dY^T X (reduction over the rows), db = column sums, dX = dY W ----
    auto bw_w = [&](const IcmLin &L, const float *dY, int ldd, const float *X, int ldx, int rows) {
        gq.gemm(dY, 1, ldd, X, 1, ldx, L.out, L.in, rows, epi_store(ig + L.oW, L.in));
        gq.colsum(dY, rows, ldd, L.out, ig + L.ob);
    };
    auto bw_x = [&](const IcmLin &L, const float *dY, int ldd, int rows, int ncols, const GemmEpi &ep) {
        gq.gemm(dY, ldd, 1, ip + L.oW, 1, L.in, rows, ncols, L.out, ep);
    };
    bw_w(icm.post[1], dpred, F, qh, Hi, B); bw_x(icm.post[1], dpred, F, B, Hi, epi_act_bwd(dqh, Hi, qh, Hi, LK, 0.25f));
    bw_w(icm.post[0], dqh, Hi, xc[4], C, B); bw_x(icm.post[0], dqh, Hi, B, F, epi_store(dx, F));
    for (int k = 3; k >= 0; --k) {
        bw_w(icm.res[k][1], dx, F, hc[k], C, B); bw_x(icm.res[k][1], dx, F, B, F, epi_act_bwd(dh, F, hc[k], C, LK, 0.25f));
        bw_w(icm.res[k][0], dh, F, xc[k], C, B); bw_x(icm.res[k][0], dh, F, B, F, epi_accum(dx, F));
    }
    gq.run<MAXI>(Ps, Qs);
    bw_w(icm.pre[2], dx, F, ph1, Hi, B); bw_x(icm.pre[2], dx, F, B, Hi, epi_act_bwd(dph1, Hi, ph1, Hi, LK, 0.25f));
    bw_w(icm.pre[1], dph1, Hi, ph0, Hi, B); bw_x(icm.pre[1], dph1, Hi, B, Hi, epi_act_bwd(dph0, Hi, ph0, Hi, LK, 0.25f));
    bw_w(icm.pre[0], dph0, Hi, pin, C, B); bw_x(icm.pre[0], dph0, Hi, B, F, epi_store(dfe, F));
    bw_w(icm.inv[2], dz, Ai, ih1, Hi, B); bw_x(icm.inv[2], dz, Ai, B, Hi, epi_act_bwd(dih1, Hi, ih1, Hi, RL, 0.25f));
    bw_w(icm.inv[1], dih1, Hi, ih0, Hi, B); bw_x(icm.inv[1], dih1, Hi, B, Hi, epi_act_bwd(dih0, Hi, ih0, Hi, RL, 0.25f));
    bw_w(icm.inv[0], dih0, Hi, iin, 2 * F, B); bw_x(icm.inv[0], dih0, Hi, B, 2 * F, epi_store(diin, 2 * F));
    gq.run<MAXI>(Ps, Qs);
    for (int e = tid; e < B * F; e += DNT) {                    // gradient at features([s; s'])
        const int b = e / F, f = e - b * F;
        dfe[e] = dfe[e] + diin[b * 2 * F + f];                 // features(s): forward_pre + inverse
        dfe[B * F + e] = diin[b * 2 * F + F + f] - dpred[e];   // features(s'): inverse + mse target
    }
    __syncthreads();
    bw_w(icm.feat[2], dfe, F, fh1, Hi, 2 * B); bw_x(icm.feat[2], dfe, F, 2 * B, Hi, epi_act_bwd(dfh1, Hi, fh1, Hi, LK, 0.25f));
    bw_w(icm.feat[1], dfh1, Hi, fh0, Hi, 2 * B); bw_x(icm.feat[1], dfh1, Hi, 2 * B, Hi, epi_act_bwd(dfh0, Hi, fh0, Hi, LK, 0.25f));
    bw_w(icm.feat[0], dfh0, Hi, X2, S, 2 * B);
    gq.run<MAXI>(Ps, Qs);
    // ---- torch.optim.Adam over all ICM parameters (lr = icm.lr), then the intrinsic rewards of the UPDATED model ----
    if (tid == 0) {
        c.pows[0] *= c.adam_beta1; c.pows[1] *= c.adam_beta2;
        ctrl[c.ci] = (float)(-(c.lr / (1.0 - c.pows[0])));
        ctrl[c.ci + 1] = (float)__builtin_sqrt(1.0 - c.pows[1]);
    }
    __syncthreads();
    {
        const AdamConsts ac{ ctrl[c.ci], ctrl[c.ci + 1], (float)(1.0 - c.adam_beta1), (float)(1.0 - c.adam_beta2), (float)c.adam_beta2, (float)c.adam_eps };
        wg_adam(ip, buf(IB_M), buf(IB_V), ig, 0, icm.P, ac, nullptr, 0.0f, 0.0f);
    }
    __syncthreads();
    icm_forward();
    const float eta = (float)c.eta;
    for (int b = tid; b < B; b += DNT) {                        // rewards += eta * mean_f (features(s') - forward)^2
        float sm = 0.0f;
        for (int f = 0; f < F; ++f) { const float d = fe[(B + b) * F + f] - pred[b * F + f]; sm = fma32(d, d, sm); }
        add_reward(b, eta * (sm / (float)F));
    }
    __syncthreads();
}

}  // namespace lenv
