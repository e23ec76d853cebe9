// td3_discrete_inner_loop.hip -- fused NES inner loop for TD3_discrete_vary: TD3 on a discrete action space through a
// Gumbel-softmax actor, trained on a VirtualEnv (three SE nets) and tested on the real env; one 512-thread workgroup per chain.
//
// Replaces GTN_Worker.calc_score (agents/GTN_worker.py:187-221) with
//   TD3_discrete_vary.learn / select_train_action / select_test_action   agents/TD3_discrete_vary.py:62-117,159-171
//   Actor_TD3_discrete (gumbel_softmax head), Critic_Q                    models/actor_critic.py:22-35,64-71
//   build_nn_from_config with the shared nn.LayerNorm                     models/model_utils.py:4-39
//   BaseAgent.train / test with discretize_action, ReplayBuffer            agents/base_agent.py:64-227, utils.py:9-72
//   EnvWrapper.step (virtual branch: one-hot of the index) -> VirtualEnv  envs/env_wrapper.py:16-47, envs/virtual_env.py:43-54
//   real env of the test phases: gym 0.17.3 CartPole-v0 / Acrobot-v1 / MountainCar-v0 (lenv_device.cuh)
//
// Structure of the GEMM-queue TD3 kernel (td3_rn_inner_loop.hip): parameters, targets, Adam state and activations of a chain
// live in its HBM arena, every layer product is the canonical-order workgroup GEMM of lenv_gemm.cuh.  New here: the LayerNorm
// rows between a product and its activation (forward: one thread per row, sequential sums as the oracle; backward: column
// sums for the shared weight / bias per LayerNorm position, then one thread per row), the Gumbel-softmax head with its
// backward, the annealed temperature, and the argmax hand-over to the SE / the real env.
#include "lenv_gemm.cuh"
#include "lenv_ln.cuh"

// Diagnostic build only (-DLENV_PHASE_TIMING): per-phase shader-clock totals of chain 0, never in the shipped library.
#ifdef LENV_PHASE_TIMING
#define TDP_DECL unsigned long long pt_last = __builtin_readcyclecounter(), pt_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define TDP_MARK(i) do { unsigned long long pt_now = __builtin_readcyclecounter(); pt_acc[i] += pt_now - pt_last; pt_last = pt_now; } while (0)
#else
#define TDP_DECL
#define TDP_MARK(i)
#endif

namespace lenv {
#ifdef LENV_PHASE_TIMING
__device__ unsigned long long g_td3d_phase_cycles[16];
#endif

constexpr int TD_MAXL = 3;     // hidden layers of actor / critic (vary_hyperparameters draws hidden_layer + 1)
constexpr int TD_MAXW = 512;   // max hidden_size
constexpr int TD_MAXB = 768;   // max batch size
constexpr int TD_MAXI = 256;   // rows of one product block
constexpr int TD_MAXA = 8;     // max action_dim

enum { STREAM_TD3D_GUMBEL_ACT = 13, STREAM_TD3D_GUMBEL_TEST = 14, STREAM_TD3D_GUMBEL_TARGET = 15, STREAM_TD3D_GUMBEL_ACTOR = 16 };

// flat layout of one net in Module.parameters() order: W0 b0 [W1 b1 [LNw LNb] W2 b2 ...] Wout bout
struct DMlpOff { int in, H, L, out, ln; int oW[TD_MAXL + 1], ob[TD_MAXL + 1], oLN; int P; };

__host__ __device__ inline void dmlp_off(DMlpOff &m, int in, int H, int L, int out, int use_ln)
{
    m.in = in; m.H = H; m.L = L; m.out = out; m.ln = (use_ln && L >= 2) ? 1 : 0; m.oLN = 0;
    int o = 0, n_in = in;
    for (int l = 0; l <= TD_MAXL; ++l) m.oW[l] = m.ob[l] = 0;
    for (int l = 0; l < L; ++l) {
        m.oW[l] = o; o += H * n_in; m.ob[l] = o; o += H; n_in = H;
        if (m.ln && l == 1) { m.oLN = o; o += 2 * H; }
    }
    m.oW[L] = o; o += out * H; m.ob[L] = o; o += out;
    m.P = o;
}

struct Td3dArgs {
    lenv_td3d_cfg cfg;
    const float *theta, *eps; const int32_t *worker; const float *sign;
    const float *agent_init; const uint64_t *rng_keys;
    lenv_td3d_tapes tapes;
    float *arena; int64_t arena_stride;
    lenv_td3_out out;
    int64_t rb_cap; int RS;
    int P, P_se;                               // row stride of agent_init / final_params (cfg's maximal shapes); parameters of theta
    const double *hp_lr; const int32_t *hp_batch, *hp_hidden, *hp_layers;
    int64_t a_params, a_targets, a_m, a_v, a_grad, a_replay, a_xc, a_xn, a_xa, a_hc1[TD_MAXL], a_hc2[TD_MAXL], a_ha[TD_MAXL], a_ht[TD_MAXL],
        a_xh1[TD_MAXL], a_xh2[TD_MAXL], a_xha[TD_MAXL], a_rs1, a_rs2, a_rsa, a_d[2], a_dx, a_raw, a_ys, a_dz, a_meter, a_se;
};

// natural-log based Gumbel(0,1) draw from one counter value (oracle: orc_gumbel)
__device__ __forceinline__ float det_gumbel(uint64_t key, uint32_t stream, uint64_t n)
{
    const double u = ((double)(rng_u64(key, stream, n) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
    return (float)(-det_log(-det_log(u)));
}

// np.linspace(temp, temp / 20, 2000)[min(n, 1999)] rounded to fp32 (oracle: orc_td3d_temperature)
__device__ __forceinline__ float td3d_temperature(double temp0, int64_t n)
{
    const double stop = temp0 / 20;
    const double step = (stop - temp0) / 1999.0;
    if (n >= 1999) return (float)stop;
    return (float)((double)n * step + temp0);
}

__device__ __forceinline__ int argmax_first(const float *v, int n)
{
    int best = 0;
    for (int k = 1; k < n; ++k) if (v[k] > v[best]) best = k;
    return best;
}

// gumbel_softmax(raw * max_action, tau, hard) for one row (oracle: td3d_actor_one): out = the returned action, ys = y_soft
__device__ __forceinline__ void gumbel_softmax_row(const float *raw, const float *gum, int A, float ma, float tau, bool hard, float *out, float *ys)
{
    float t[TD_MAXA], e[TD_MAXA], y[TD_MAXA];
    for (int k = 0; k < A; ++k) t[k] = (raw[k] * ma + gum[k]) / tau;
    float mx = t[0];
    for (int k = 1; k < A; ++k) if (t[k] > mx) mx = t[k];
    float sm = 0.0f;
    for (int k = 0; k < A; ++k) { e[k] = det_expf(t[k] - mx); sm = sm + e[k]; }
    for (int k = 0; k < A; ++k) y[k] = e[k] / sm;
    if (hard) {
        int idx = 0;
        for (int k = 1; k < A; ++k) if (y[k] > y[idx]) idx = k;
        for (int k = 0; k < A; ++k) out[k] = ((k == idx ? 1.0f : 0.0f) - y[k]) + y[k];
    } else for (int k = 0; k < A; ++k) out[k] = y[k];
    if (ys) for (int k = 0; k < A; ++k) ys[k] = y[k];
}

template <int ENVD>
__global__ __launch_bounds__(DNT) void td3_discrete_inner_kernel(const Td3dArgs a)
{
    extern __shared__ __align__(16) float lds[];
    const lenv_td3d_cfg &cfg = a.cfg;
    const int tid = threadIdx.x;
    const int64_t chain = blockIdx.x;
    if (threadIdx.x == 0 && a.out.status) a.out.status[chain] = 0;
    constexpr int S = ENVD == LENV_ENV_CARTPOLE ? 4 : (ENVD == LENV_ENV_ACROBOT ? 6 : 2);
    constexpr int A = ENVD == LENV_ENV_CARTPOLE ? 2 : 3;
    constexpr int SA = S + A;
    const bool vary = a.hp_batch != nullptr;
    const int H = vary ? a.hp_hidden[chain] : cfg.hidden, L = vary ? a.hp_layers[chain] : cfg.layers;
    const int Bm = cfg.batch_size;
    const int B = vary ? a.hp_batch[chain] : cfg.batch_size;
    const double lr = vary ? a.hp_lr[chain] : cfg.lr;
    const int T = cfg.test_episodes, Hse = cfg.se_hidden, Lse = cfg.se_layers, se_act = cfg.se_act, RS = a.RS;
    const int policy_delay = cfg.policy_delay;
    if (vary && (H < 1 || H > cfg.hidden || L < 1 || L > cfg.layers || B < 1 || B > Bm)) {   // uniform per chain
        if (tid == 0) { if (a.out.status) a.out.status[chain] = -8; a.out.score[chain] = 0.0; }
        return;
    }
    DMlpOff mo_actor, mo_critic, mo_se[3];
    dmlp_off(mo_actor, S, H, L, A, cfg.use_layer_norm);
    dmlp_off(mo_critic, SA, H, L, 1, cfg.use_layer_norm);
    dmlp_off(mo_se[0], SA, Hse, Lse, S, 0);
    dmlp_off(mo_se[1], SA, Hse, Lse, 1, 0);
    dmlp_off(mo_se[2], SA, Hse, Lse, 1, 0);
    // cfg.se_layer_norm: the SE nets' own LayerNorm -- never perturbed by NES (nn.Linear modules only, GTN_worker.py:156-175), so no parameters
    // in theta: ln == 2 marks a position that normalises with the constructor's weight 1 / bias 0
    if (cfg.se_layer_norm != 0 && Lse >= 2) mo_se[0].ln = mo_se[1].ln = mo_se[2].ln = 2;
    const int Pa = mo_actor.P, Pc = mo_critic.P, P = Pa + 2 * Pc;
    const int act_id = cfg.act;
    const float prelu = cfg.prelu, ma = (float)cfg.max_action;
    const bool hard = cfg.gumbel_hard != 0;

    // ---- LDS carve-up ----
    float *Ps = lds, *Qs = Ps + GemmShape<TD_MAXI>::PS_FLOATS;
    GemmCmd *cmds = reinterpret_cast<GemmCmd *>(Qs + GemmShape<TD_MAXI>::QS_FLOATS);   // [GEMM_QUEUE_MAX] command queue
    float *rowb = reinterpret_cast<float *>(cmds + GEMM_QUEUE_MAX);                    // [2][TD_MAXW] single-row activations
    float *q1 = rowb + 2 * TD_MAXW;                                                     // [Bm] each
    float *q2 = q1 + Bm, *tq1 = q2 + Bm, *tq2 = tq1 + Bm, *rr = tq2 + Bm, *dd = rr + Bm, *dq1 = dd + Bm, *dq2 = dq1 + Bm;
    float *misc = dq2 + Bm;                                // [64]
    double *xs_d = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(misc + 64) + 7) & ~(uintptr_t)7);   // [4] train env state
    double *xt_d = xs_d + 4;                              // [T][4] test env states
    double *ret = xt_d + 4 * T;                           // [T]
    float *ep_rew = reinterpret_cast<float *>(ret + T);   // [T]
    int *tlen = reinterpret_cast<int *>(ep_rew + T);      // [T] env steps of each test episode
    int *tflag = tlen + T;                                // [T] test episode still running
    float *state = reinterpret_cast<float *>(tflag + T);  // [8] current observation (fp32)
    float *action = state + 8;                            // [8]
    float *newrow = action + 8;                           // [32] replay row [s | a | s' | r | done]
    float *xse = newrow + 32;                             // [16] SE input cat(one_hot, state)
    float *nse = xse + 16;                                // [16] SE outputs [s' | r | done]
    volatile float *ctrl = misc;
    volatile int *ictrl = reinterpret_cast<volatile int *>(misc + 32);

    float *arena = a.arena + chain * a.arena_stride;
    float *params = arena + a.a_params, *targets = arena + a.a_targets, *adam_m = arena + a.a_m, *adam_v = arena + a.a_v;
    float *grad = arena + a.a_grad, *rb = arena + a.a_replay;
    float *xc = arena + a.a_xc, *xn = arena + a.a_xn, *xa = arena + a.a_xa;      // [B][SA] critic inputs
    float *dxb = arena + a.a_dx, *rawb = arena + a.a_raw, *ysb = arena + a.a_ys, *dzb = arena + a.a_dz;
    float *hc1[TD_MAXL], *hc2[TD_MAXL], *ha[TD_MAXL], *ht[TD_MAXL], *xh1[TD_MAXL], *xh2[TD_MAXL], *xha[TD_MAXL];
    float *dbuf[2] = { arena + a.a_d[0], arena + a.a_d[1] };
    for (int l = 0; l < TD_MAXL; ++l) {
        hc1[l] = arena + a.a_hc1[l]; hc2[l] = arena + a.a_hc2[l]; ha[l] = arena + a.a_ha[l]; ht[l] = arena + a.a_ht[l];
        xh1[l] = arena + a.a_xh1[l]; xh2[l] = arena + a.a_xh2[l]; xha[l] = arena + a.a_xha[l];
    }
    float *rs1 = arena + a.a_rs1, *rs2 = arena + a.a_rs2, *rsa = arena + a.a_rsa;   // [TD_MAXL][B] 1/sqrt(var + eps)
    double *meter = reinterpret_cast<double *>(arena + a.a_meter);
    float *sep = arena + a.a_se;

    // ---- stage the perturbed SE (GTN_worker.py:165-175) and the fresh agent ----
    {
        const float sg = a.eps ? a.sign[chain] : 0.0f;
        const float *e = a.eps ? a.eps + (int64_t)a.worker[chain] * a.P_se : nullptr;
        for (int i = tid; i < a.P_se; i += DNT) sep[i] = e ? fma32(sg, e[i], a.theta[i]) : a.theta[i];
    }
    for (int p = tid; p < P; p += DNT) {
        const float w = a.agent_init[chain * a.P + p];
        params[p] = w; targets[p] = w; adam_m[p] = 0.0f; adam_v[p] = 0.0f;
    }
    if (tid < 64) misc[tid] = 0.0f;
    __syncthreads();

    const uint64_t key = a.rng_keys ? a.rng_keys[chain] : 0;
    const bool tape = cfg.rng_mode == LENV_RNG_TAPE;
    int status = 0;
    int64_t n_rand = 0, n_actn = 0, n_testn = 0, n_test_ep = 0, learn_it = 0, policy_it = 0;
    int train_steps = 0, test_steps = 0, episodes_run = 0;
    double pows[4] = { 1.0, 1.0, 1.0, 1.0 };
    const int rb_cap = (int)a.rb_cap;
    float temp = td3d_temperature(cfg.gumbel_temp, 0);                  // gumbel_temp_annealed = steps[0] (TD3_discrete_vary.py:60)
    const float astd = (float)cfg.action_std;

    GemmQueue gq(cmds);

    // ---- LayerNorm rows (forward): z [I][H] holds Linear + bias.  A chunk of rows is copied into LDS (coalesced; the product
    // buffers Ps / Qs are free between queue runs; odd row stride = conflict-free), thread b then owns row b: sequential mean /
    // variance and the normalised row in place (oracle: mlp_forward_one_ex), and the copy back applies weight, bias and activation:
    // xh <- normalised rows, rstd[b], z <- act(fma(xn, w, b)) ----
    constexpr int LNBUF = GemmShape<TD_MAXI>::PS_FLOATS + GemmShape<TD_MAXI>::QS_FLOATS;
    auto ln_forward = [&](float *z, int I, int Hh, const float *w, const float *bb, float *xh, float *rstd, int act, float pr) {
        ln_rows_forward<LNBUF>(Ps, z, I, Hh, w, bb, xh, rstd, act, pr);
    };

    // ---- LayerNorm rows (backward): d [I][H] holds the gradient of the LayerNorm OUTPUT.  Column sums first (the shared weight /
    // bias gradient of this position, rows ascending: oracle mlp_backward_one_ex + mlp_fold_ln_grads), then -- chunks of rows of d
    // and xh staged in LDS as above -- thread b turns row b into the gradient of the LayerNorm input ----
    auto ln_backward = [&](float *d, int I, int Hh, const float *w, const float *xh, const float *rstd, float *g_w, float *g_b, bool first) {
        ln_rows_backward<LNBUF, TD_MAXW>(Ps, d, I, Hh, w, xh, rstd, g_w, g_b, first);
    };

    // ---- MLP forward over I rows (row stride ldx): hidden activations to hid[l][I][H]; xh / rstd (may be null): what the
    // backward of the LayerNorm positions needs.  The products are queued; a LayerNorm position runs the queue, the output
    // layer's product is left queued for the caller ----
    auto mlp_forward = [&](const float *par, const DMlpOff &mo, const float *X, int ldx, int I, float *const *hid, float *const *xh,
                           float *rstd, float *out, int ldo, int ocol) {
        const float *in = X;
        int n_in = mo.in, ldin = ldx;
        for (int l = 0; l < mo.L; ++l) {
            if (mo.ln && l >= 1) {
                gq.gemm(in, ldin, 1, par + mo.oW[l], n_in, 1, I, mo.H, n_in, epi_bias(hid[l], mo.H, 0, par + mo.ob[l]));
                gq.run<TD_MAXI>(Ps, Qs);
                ln_forward(hid[l], I, mo.H, par + mo.oLN, par + mo.oLN + mo.H, xh ? xh[l] : nullptr, rstd ? rstd + (int64_t)l * Bm : nullptr, act_id, prelu);
            } else gq.gemm(in, ldin, 1, par + mo.oW[l], n_in, 1, I, mo.H, n_in, epi_bias_act(hid[l], mo.H, par + mo.ob[l], act_id, prelu));
            in = hid[l]; n_in = mo.H; ldin = mo.H;
        }
        gq.gemm(in, ldin, 1, par + mo.oW[mo.L], n_in, 1, I, mo.out, n_in, epi_bias(out, ldo, ocol, par + mo.ob[mo.L]));
    };

    // ---- one row through a net in LDS row buffers: thread j owns output j of a layer and runs the k-ascending fmaf chain of the
    // product (bit-identical to mlp_forward with I = 1); a LayerNorm position is reduced by thread 0 ----
    auto mlp_row1 = [&](const float *par, const DMlpOff &mo, const float *x, float *out, int ocol, int act, float pr) {
        const float *in = x;
        int n_in = mo.in;
        for (int l = 0; l <= mo.L; ++l) {
            const bool last = l == mo.L;
            const int n_out = last ? mo.out : mo.H;
            const float *W = par + mo.oW[l], *bb = par + mo.ob[l];
            float *h = last ? out + ocol : rowb + (l & 1) * TD_MAXW;
            const bool lnl = !last && mo.ln && l >= 1;
            for (int j = tid; j < n_out; j += DNT) {
                const float *w = W + (int64_t)j * n_in;
                float z = 0.0f;
                int k = 0;
                for (; k + 32 <= n_in; k += 32) {              // 32 terms requested before the first fmaf (one round trip instead of 32)
                    float wv[32], xv[32];
#pragma unroll
                    for (int u = 0; u < 32; ++u) { wv[u] = w[k + u]; xv[u] = in[k + u]; }
#pragma unroll
                    for (int u = 0; u < 32; ++u) z = fma32(xv[u], wv[u], z);
                }
                for (; k < n_in; ++k) z = fma32(in[k], w[k], z);
                z = z + bb[j];
                h[j] = (last || lnl) ? z : act_fwd(act, pr, z);
            }
            __syncthreads();
            if (lnl) {
                if (tid == 0) {
                    float sm = 0.0f, sv = 0.0f;
                    for (int j = 0; j < n_out; ++j) sm = sm + h[j];
                    const float mean = sm / (float)n_out;
                    for (int j = 0; j < n_out; ++j) { const float dj = h[j] - mean; sv = fma32(dj, dj, sv); }
                    ctrl[14] = mean; ctrl[15] = 1.0f / __builtin_sqrtf(sv / (float)n_out + 1e-5f);
                }
                __syncthreads();
                const float mean = ctrl[14], r = ctrl[15];
                const float *lw = par + mo.oLN, *lb = lw + mo.H;
                if (mo.ln == 2) { for (int j = tid; j < n_out; j += DNT) h[j] = act_fwd(act, pr, fma32((h[j] - mean) * r, 1.0f, 0.0f)); }
                else for (int j = tid; j < n_out; j += DNT) h[j] = act_fwd(act, pr, fma32((h[j] - mean) * r, lw[j], lb[j]));
                __syncthreads();
            }
            in = h; n_in = mo.H;
        }
    };

    // ---- MLP backward: dOut [I][out] -> parameter gradients gpar (may be null) and input gradient dX (may be null) ----
    auto mlp_backward = [&](const float *par, const DMlpOff &mo, const float *X, int ldx, int I, float *const *hid, float *const *xh,
                            const float *rstd, const float *dOut, float *gpar, float *dX) {
        const int Hh = mo.H, O = mo.out;
        if (gpar) {
            gq.gemm(dOut, 1, O, hid[mo.L - 1], 1, Hh, O, Hh, I, epi_store(gpar + mo.oW[mo.L], Hh));
            gq.colsum(dOut, I, O, O, gpar + mo.ob[mo.L]);
        }
        float *dcur = dbuf[0];
        gq.gemm(dOut, O, 1, par + mo.oW[mo.L], 1, Hh, I, Hh, O, epi_act_bwd(dcur, Hh, hid[mo.L - 1], Hh, act_id, prelu));
        for (int l = mo.L - 1; l >= 0; --l) {
            const int n_in = l == 0 ? mo.in : Hh;
            const float *inp = l == 0 ? X : hid[l - 1];
            const int ldin = l == 0 ? ldx : Hh;
            if (mo.ln && l >= 1) {
                gq.run<TD_MAXI>(Ps, Qs);
                ln_backward(dcur, I, Hh, par + mo.oLN, xh[l], rstd + (int64_t)l * Bm, gpar ? gpar + mo.oLN : nullptr, gpar ? gpar + mo.oLN + Hh : nullptr, l == mo.L - 1);
            }
            if (gpar) {
                gq.gemm(dcur, 1, Hh, inp, 1, ldin, Hh, n_in, I, epi_store(gpar + mo.oW[l], n_in));
                gq.colsum(dcur, I, Hh, Hh, gpar + mo.ob[l]);
            }
            if (l > 0) {
                float *dn = dbuf[(mo.L - l) & 1];
                gq.gemm(dcur, Hh, 1, par + mo.oW[l], 1, n_in, I, n_in, Hh, epi_act_bwd(dn, n_in, hid[l - 1], n_in, act_id, prelu));
                dcur = dn;
            } else if (dX) {
                gq.gemm(dcur, Hh, 1, par + mo.oW[0], 1, n_in, I, n_in, Hh, epi_store(dX, n_in));
            }
        }
    };

    // torch.optim.Adam single-tensor step on params[p0, p0+n) (pows index pi), thread 0 publishes the bias corrections
    auto adam = [&](int p0, int n, int pi) {
        if (tid == 0) {
            pows[pi] *= cfg.adam_beta1; pows[pi + 1] *= cfg.adam_beta2;
            ctrl[10] = (float)(-(lr / (1.0 - pows[pi])));
            ctrl[11] = (float)__builtin_sqrt(1.0 - pows[pi + 1]);
        }
        __syncthreads();
        const float neg_step = ctrl[10], bc2_sqrt = ctrl[11];
        const float w1 = (float)(1.0 - cfg.adam_beta1), w2 = (float)(1.0 - cfg.adam_beta2), beta2 = (float)cfg.adam_beta2, aeps = (float)cfg.adam_eps;
        const AdamConsts ac{ neg_step, bc2_sqrt, w1, w2, beta2, aeps };
        wg_adam(params, adam_m, adam_v, grad, p0, n, ac, nullptr, 0.0f, 0.0f);
        __syncthreads();
    };

    // actor(obs row(s), temp) + randn(A) * action_std for Tg rows (select_train_action / select_test_action :164-171): raw rows in
    // `rawrows` [Tg][A] -> action vectors at[Tg][A].  Row te draws its Gumbel / Gaussian values at index cnt(te).
    auto noisy_actions = [&](const float *rawrows, float *at, int Tg, auto cnt, const float *gtape, int64_t gstride, uint32_t gstream,
                             const float *ntape, int64_t nstride, uint32_t nstream) {
        if (tid < Tg) {
            const int64_t c = cnt(tid);
            float gm[TD_MAXA], pa[TD_MAXA];
            for (int k = 0; k < A; ++k) {
                if (tape) { if (c >= gstride) { status = -9; gm[k] = 0.0f; } else gm[k] = gtape[(chain * gstride + c) * A + k]; }
                else gm[k] = det_gumbel(key, gstream, (uint64_t)(c * A + k));
            }
            gumbel_softmax_row(rawrows + tid * A, gm, A, ma, temp, hard, pa, nullptr);
            for (int k = 0; k < A; ++k) {
                float zn;
                if (tape) { if (c >= nstride) { status = -7; zn = 0.0f; } else zn = ntape[(chain * nstride + c) * A + k]; }
                else zn = (float)det_normal(key, nstream, (uint64_t)(c * A + k));
                at[tid * A + k] = pa[k] + zn * astd;
            }
        }
        __syncthreads();
    };

    // ---- real-env test phase (BaseAgent.test, base_agent.py:155-227): T episodes in lock-step; the env gets action.argmax().
    // With a tape the reference's draws are consumed episode by episode, so the episodes run one after the other there. ----
    auto test_phase = [&]() {
        const bool serial = tape;
        const int Tg = serial ? 1 : T;
        float *xt = xn;                                    // [Tg][S] fp32 observations (xn is free outside learn)
        float *at = xa;                                    // [Tg][A] action vectors
        float *rawt = rawb;                                // [Tg][A] raw actor outputs
        for (int g0 = 0; g0 < T; g0 += Tg) {
            if (tid < Tg) {
                const int64_t row = n_test_ep + g0 + tid;
                double st4[4];
                if (tape) {
                    if (row >= a.tapes.test_reset_stride) { status = -5; st4[0] = st4[1] = st4[2] = st4[3] = 0.0; }
                    else for (int i = 0; i < 4; ++i) st4[i] = a.tapes.test_reset[(chain * a.tapes.test_reset_stride + row) * 4 + i];
                } else real_env_reset_draw(ENVD, key, STREAM_TEST_RESET, row, st4);
                for (int i = 0; i < 4; ++i) xt_d[tid * 4 + i] = st4[i];
                ep_rew[g0 + tid] = 0.0f; tflag[tid] = 1; tlen[g0 + tid] = 0;
            }
            __syncthreads();
            for (int tt = 0; tt < cfg.max_steps; ++tt) {
                if (tid < Tg) { float ob[8]; real_env_obs(ENVD, xt_d + tid * 4, ob); for (int i = 0; i < S; ++i) xt[tid * S + i] = ob[i]; }
                __syncthreads();
                if (Tg == 1) mlp_row1(params, mo_actor, xt, rawt, 0, act_id, prelu);
                else {
                    mlp_forward(params, mo_actor, xt, S, Tg, ht, nullptr, nullptr, rawt, A, 0);
                    gq.run<TD_MAXI>(Ps, Qs);
                }
                const int64_t base_ep = n_test_ep + g0;
                noisy_actions(rawt, at, Tg, [&](int te) { return serial ? n_testn : (base_ep + te) * (int64_t)cfg.max_steps + tt; },
                              a.tapes.gumbel_test, a.tapes.gumbel_test_stride, STREAM_TD3D_GUMBEL_TEST,
                              a.tapes.test_noise, a.tapes.test_noise_stride, STREAM_TD3_TEST_NOISE);
                if (tid < Tg && tflag[tid]) {
                    double st4[4], rew; int dn;
                    for (int i = 0; i < 4; ++i) st4[i] = xt_d[tid * 4 + i];
                    real_env_step(ENVD, st4, argmax_first(at + tid * A, A), rew, dn);
                    for (int i = 0; i < 4; ++i) xt_d[tid * 4 + i] = st4[i];
                    ep_rew[g0 + tid] = ep_rew[g0 + tid] + (float)rew;
                    tlen[g0 + tid] = tlen[g0 + tid] + 1;
                    if (dn) tflag[tid] = 0;
                }
                if (serial) ++n_testn;
                __syncthreads();
                if (tid == 0) { int c = 0; for (int te = 0; te < Tg; ++te) c += tflag[te]; ictrl[4] = c; }
                __syncthreads();
                const int alive = ictrl[4];
                __syncthreads();
                if (alive == 0) break;
            }
        }
        if (tid < T) ret[tid] = (double)ep_rew[tid];
        n_test_ep += T;
        __syncthreads();
        for (int te = 0; te < T; ++te) test_steps += tlen[te];
        __syncthreads();
    };

    const float g32 = (float)cfg.gamma;
    const bool budgeted = cfg.step_budget > 0;
    int timed_out_at = -1;
    const bool no_test_env = cfg.test_mode == 1;      // BaseAgent.train(env, test_env=None): lenv_ddqn_cfg::test_mode
    TDP_DECL;
    for (int episode = 0; episode < cfg.train_episodes; ++episode) {
        if (budgeted && (int64_t)train_steps + test_steps > cfg.step_budget) { timed_out_at = episode; break; }   // uniform
        const bool learning = episode >= cfg.init_episodes;
        // env.reset(): VirtualEnv.reset -> reset_env.reset() (virtual_env.py:33-41)
        if (tid == 0) {
            double st4[4];
            if (tape) {
                if (episode >= a.tapes.train_reset_stride) { status = -5; st4[0] = st4[1] = st4[2] = st4[3] = 0.0; }
                else for (int i = 0; i < 4; ++i) st4[i] = a.tapes.train_reset[(chain * a.tapes.train_reset_stride + episode) * 4 + i];
            } else real_env_reset_draw(ENVD, key, STREAM_TRAIN_RESET, (int64_t)episode, st4);
            float ob[8];
            real_env_obs(ENVD, st4, ob);
            for (int i = 0; i < S; ++i) state[i] = ob[i];
        }
        __syncthreads();
        int ep_len = 0;
        float tr_reward = 0.0f;                                  // base_agent.py:102,121 episode_reward += reward (fp32 tensors; uniform over the threads)
        for (int t = 0; t < cfg.max_steps; ++t) {
            const int size_after = train_steps + 1 < rb_cap ? train_steps + 1 : rb_cap;
            const int new_pos = train_steps % rb_cap;
            // ---- select_train_action (:159-167) ----
            if (!learning) {
                if (tid == 0) {
                    int idx;
                    if (tape) { if (n_rand >= a.tapes.rand_action_stride) { status = -3; idx = 0; } else idx = a.tapes.rand_action[chain * a.tapes.rand_action_stride + n_rand]; }
                    else idx = (int)u64_to_below(rng_u64(key, STREAM_ACTION, (uint64_t)n_rand), (uint32_t)A);
                    if (idx < 0 || idx >= A) { status = -3; idx = 0; }
                    for (int k = 0; k < A; ++k) action[k] = k == idx ? 1.0f : 0.0f;
                }
                ++n_rand;
                __syncthreads();
            } else {
                TDP_MARK(7);
                mlp_row1(params, mo_actor, state, nse, 0, act_id, prelu);               // raw actor outputs -> nse[0..A)
                const int64_t c0 = n_actn;
                noisy_actions(nse, action, 1, [&](int) { return c0; }, a.tapes.gumbel_act, a.tapes.gumbel_act_stride, STREAM_TD3D_GUMBEL_ACT,
                              a.tapes.act_noise, a.tapes.act_noise_stride, STREAM_TD3_ACT_NOISE);
                ++n_actn;
            }
            TDP_MARK(0);                                   // select_train_action
            // ---- env.step(action.argmax()) -> EnvWrapper.step: one-hot of the index -> VirtualEnv.step (virtual_env.py:43-54): the
            // three SE nets on cat(one_hot, state); reward / done see the pre-transition state ----
            if (tid == 0) {
                const int a_idx = argmax_first(action, A);
                for (int k = 0; k < A; ++k) xse[k] = k == a_idx ? 1.0f : 0.0f;
                for (int i = 0; i < S; ++i) { xse[A + i] = state[i]; newrow[i] = state[i]; }
                for (int k = 0; k < A; ++k) newrow[S + k] = action[k];
            }
            __syncthreads();
            mlp_row1(sep, mo_se[0], xse, nse, 0, se_act, cfg.se_prelu);
            mlp_row1(sep + mo_se[0].P, mo_se[1], xse, nse, S, se_act, cfg.se_prelu);
            mlp_row1(sep + mo_se[0].P + mo_se[1].P, mo_se[2], xse, nse, S + 1, se_act, cfg.se_prelu);
            if (tid < S) newrow[S + A + tid] = nse[tid];
            if (tid == 64) { newrow[2 * S + A] = nse[S]; newrow[2 * S + A + 1] = nse[S + 1]; }
            __syncthreads();
            if (tid < 2 * S + A + 2) rb[(int64_t)new_pos * RS + tid] = newrow[tid];
            if (a.out.trace_reward && train_steps < a.out.trace_cap) {
                const int64_t k = chain * a.out.trace_cap + train_steps;
                if (tid < S) { a.out.trace_state[k * S + tid] = newrow[tid]; a.out.trace_next_state[k * S + tid] = newrow[S + A + tid]; }
                if (tid < A) a.out.trace_action[k * A + tid] = newrow[S + tid];
                if (tid == 0) a.out.trace_reward[k] = newrow[2 * S + A];
            }
            const float done_now = newrow[2 * S + A + 1];
            tr_reward = tr_reward + newrow[2 * S + A];
            __syncthreads();
            if (tid < S) state[tid] = newrow[S + A + tid];
            ++ep_len; ++train_steps;
            __syncthreads();

            TDP_MARK(1);                                   // SE step + append
            if (learning) {
                // ================= TD3_discrete_vary.learn (:62-117) =================
                temp = td3d_temperature(cfg.gumbel_temp, learn_it);                   // read before total_it += 1 (:64-69)
                const bool policy_step = (learn_it + 1) % policy_delay == 0;
                for (int b = tid; b < B; b += DNT) {
                    const int64_t n = learn_it * B + b;
                    int idx;
                    if (tape) {
                        if (n >= a.tapes.replay_idx_stride) { status = -4; idx = 0; } else idx = a.tapes.replay_idx[chain * a.tapes.replay_idx_stride + n];
                        if (idx < 0 || idx >= size_after) { status = -6; idx = 0; }
                    } else idx = (int)rng_replay_below(key, (uint64_t)n, (uint32_t)size_after);
                    const float *row = rb + (int64_t)idx * RS;
                    for (int i = 0; i < SA; ++i) xc[b * SA + i] = row[i];                  // [s, a]
                    for (int i = 0; i < S; ++i) xn[b * SA + i] = row[S + A + i];           // s' (action part filled below)
                    rr[b] = row[2 * S + A]; dd[b] = row[2 * S + A + 1];
                }
                __syncthreads();
                // next_actions = actor_target(s', temp) + (randn_like * policy_std).clamp(-clip, clip)   (no clamp to the action range)
                mlp_forward(targets, mo_actor, xn, SA, B, ht, nullptr, nullptr, rawb, A, 0);
                gq.run<TD_MAXI>(Ps, Qs);
                TDP_MARK(8);                               // (sub) gather + actor_target forward
                for (int b = tid; b < B; b += DNT) {
                    const int64_t n = learn_it * B + b;
                    float gm[TD_MAXA], na[TD_MAXA];
                    for (int k = 0; k < A; ++k) {
                        if (tape) { if (n >= a.tapes.gumbel_target_stride) { status = -9; gm[k] = 0.0f; } else gm[k] = a.tapes.gumbel_target[(chain * a.tapes.gumbel_target_stride + n) * A + k]; }
                        else gm[k] = det_gumbel(key, STREAM_TD3D_GUMBEL_TARGET, (uint64_t)(n * A + k));
                    }
                    gumbel_softmax_row(rawb + b * A, gm, A, ma, temp, hard, na, nullptr);
                    for (int k = 0; k < A; ++k) {
                        float zn;
                        if (tape) { if (n >= a.tapes.policy_noise_stride) { status = -8; zn = 0.0f; } else zn = a.tapes.policy_noise[(chain * a.tapes.policy_noise_stride + n) * A + k]; }
                        else zn = (float)det_normal(key, STREAM_TD3_POLICY_NOISE, (uint64_t)(n * A + k));
                        float nz = zn * (float)cfg.policy_std;
                        const float clipv = (float)cfg.policy_std_clip;
                        nz = nz < -clipv ? -clipv : (nz > clipv ? clipv : nz);
                        xn[b * SA + S + k] = na[k] + nz;
                    }
                }
                __syncthreads();
                // the two target critics share ht: the queue runs in order, so critic_1's output product reads its rows before
                // critic_2's products overwrite them
                mlp_forward(targets + Pa, mo_critic, xn, SA, B, ht, nullptr, nullptr, tq1, 1, 0);
                mlp_forward(targets + Pa + Pc, mo_critic, xn, SA, B, ht, nullptr, nullptr, tq2, 1, 0);
                mlp_forward(params + Pa, mo_critic, xc, SA, B, hc1, xh1, rs1, q1, 1, 0);
                mlp_forward(params + Pa + Pc, mo_critic, xc, SA, B, hc2, xh2, rs2, q2, 1, 0);
                gq.run<TD_MAXI>(Ps, Qs);
                TDP_MARK(9);                               // (sub) gumbel/noise + four critic forwards
                {
                    const float norm = (float)(2.0 / (double)B);
                    for (int b = tid; b < B; b += DNT) {
                        const float tq = tq1[b] < tq2[b] ? tq1[b] : tq2[b];
                        const float y = rr[b] + ((1.0f - dd[b]) * g32) * tq;       // rewards + (1 - dones) * gamma * target_Q
                        dq1[b] = norm * (q1[b] - y);
                        dq2[b] = norm * (q2[b] - y);
                    }
                }
                __syncthreads();
                mlp_backward(params + Pa, mo_critic, xc, SA, B, hc1, xh1, rs1, dq1, grad + Pa, nullptr);
                gq.run<TD_MAXI>(Ps, Qs);                                             // dbuf is shared by the two backward passes
                TDP_MARK(10);                              // (sub) TD error + critic_1 backward
                mlp_backward(params + Pa + Pc, mo_critic, xc, SA, B, hc2, xh2, rs2, dq2, grad + Pa + Pc, nullptr);
                gq.run<TD_MAXI>(Ps, Qs);
                TDP_MARK(2);                               // critics: forwards, TD error, backwards
                adam(Pa, 2 * Pc, 0);                       // critic_optimizer: critic_1 then critic_2 parameters
                TDP_MARK(3);
                ++learn_it;
                if (policy_step) {
                    // actor_loss = (-critic_1(states, actor(states, temp))).mean() with the updated critic_1
                    for (int e = tid; e < B * S; e += DNT) { const int b = e / S, i = e - b * S; xa[b * SA + i] = xc[b * SA + i]; }
                    mlp_forward(params, mo_actor, xc, SA, B, ha, xha, rsa, rawb, A, 0);
                    gq.run<TD_MAXI>(Ps, Qs);
                    for (int b = tid; b < B; b += DNT) {
                        const int64_t n = policy_it * B + b;
                        float gm[TD_MAXA];
                        for (int k = 0; k < A; ++k) {
                            if (tape) { if (n >= a.tapes.gumbel_actor_stride) { status = -9; gm[k] = 0.0f; } else gm[k] = a.tapes.gumbel_actor[(chain * a.tapes.gumbel_actor_stride + n) * A + k]; }
                            else gm[k] = det_gumbel(key, STREAM_TD3D_GUMBEL_ACTOR, (uint64_t)(n * A + k));
                        }
                        gumbel_softmax_row(rawb + b * A, gm, A, ma, temp, hard, xa + b * SA + S, ysb + b * A);
                    }
                    __syncthreads();
                    mlp_forward(params + Pa, mo_critic, xa, SA, B, hc1, xh1, rs1, q1, 1, 0);
                    gq.run<TD_MAXI>(Ps, Qs);
                    const float dqa = -(1.0f / (float)B);
                    for (int b = tid; b < B; b += DNT) dq1[b] = dqa;
                    __syncthreads();
                    mlp_backward(params + Pa, mo_critic, xa, SA, B, hc1, xh1, rs1, dq1, nullptr, dxb);
                    gq.run<TD_MAXI>(Ps, Qs);
                    // through the Gumbel softmax: softmax backward, / temp, * max_action (oracle: td3d_actor_head_bwd)
                    for (int b = tid; b < B; b += DNT) {
                        const float *dact = dxb + b * SA + S, *y = ysb + b * A;
                        float dot = 0.0f;
                        for (int k = 0; k < A; ++k) dot = fma32(dact[k], y[k], dot);
                        for (int k = 0; k < A; ++k) dzb[b * A + k] = (((dact[k] - dot) * y[k]) / temp) * ma;
                    }
                    __syncthreads();
                    mlp_backward(params, mo_actor, xc, SA, B, ha, xha, rsa, dzb, grad, nullptr);
                    gq.run<TD_MAXI>(Ps, Qs);
                    adam(0, Pa, 2);
                    const float tau = (float)cfg.tau, omt = (float)(1.0 - cfg.tau);
                    wg_polyak(params, targets, P, tau, omt);
                    __syncthreads();
                    ++policy_it;
                    TDP_MARK(4);                           // policy update
                }
            }
            if (done_now > 0.5f) break;
        }
        ++episodes_run;
        if (tid == 0 && a.out.episode_len) a.out.episode_len[chain * cfg.train_episodes + episode] = ep_len;
        __syncthreads();
        TDP_MARK(7);
        if (!no_test_env) test_phase();                    // per-episode test on the real env (base_agent.py:134-136)
        TDP_MARK(5);
        if (tid == 0) {
            double tm;
            if (no_test_env) tm = (double)tr_reward;       // train(env, test_env=None): avg_meter_reward.update(episode_reward) (base_agent.py:138)
            else {
                double sm = 0.0;
                for (int i = 0; i < T; ++i) sm += ret[i];
                tm = sm / (double)T;
            }
            meter[episode] = tm;
            if (a.out.episode_test_mean) a.out.episode_test_mean[chain * cfg.train_episodes + episode] = tm;
            // early out (base_agent.py:49-62,141-148): break_env = the test env (real rule) or, without one, the training env itself
            ictrl[3] = learning && meter_env_solved(meter, episode + 1, cfg.early_out_num, no_test_env && (true), cfg.solved_reward,
                                                    cfg.early_out_virtual_diff, episode, cfg.init_episodes);
        }
        __syncthreads();
        const int brk = ictrl[3];
        __syncthreads();
        if (brk) break;
    }
    const int64_t remaining = cfg.step_budget - ((int64_t)train_steps + test_steps);     // time_remaining - elapsed
    const int test_before = test_steps;
    test_phase();
    if (budgeted) {
        // BaseAgent.test under the time-out (base_agent.py:177-184), as in the TD3 kernel
        int64_t used = 0;
        int stop = T;
        for (int te = 0; te < T; ++te) {
            if (used > remaining) { stop = te; break; }
            used += tlen[te];
        }
        if (tid == 0) {
            double mn = -1e9;
            if (stop > 0) { mn = ret[0]; for (int i = 1; i < stop; ++i) if (ret[i] < mn) mn = ret[i]; }
            for (int te = stop; te < T; ++te) ret[te] = mn;
        }
        test_steps = test_before + (int)used;
        __syncthreads();
    }
    if (tid == 0) {
        double sm = 0.0;
        for (int i = 0; i < T; ++i) sm += ret[i];
        a.out.score[chain] = sm / (double)T;
        if (a.out.final_returns) for (int i = 0; i < T; ++i) a.out.final_returns[chain * T + i] = ret[i];
        if (a.out.stats) {
            a.out.stats[chain * 4 + 0] = episodes_run; a.out.stats[chain * 4 + 1] = train_steps;
            a.out.stats[chain * 4 + 2] = learn_it; a.out.stats[chain * 4 + 3] = test_steps;
        }
        double pad_r = __builtin_nan("");
        int pad_l = 0;
        if (timed_out_at >= 0) {
            pad_r = -1e9; pad_l = 1000000000;
            if (episodes_run > 0) { pad_r = meter[0]; for (int i = 1; i < episodes_run; ++i) if (meter[i] < pad_r) pad_r = meter[i]; }
            if (episodes_run > 0 && a.out.episode_len) {
                pad_l = a.out.episode_len[chain * cfg.train_episodes];
                for (int i = 1; i < episodes_run; ++i) { const int l = a.out.episode_len[chain * cfg.train_episodes + i]; if (l > pad_l) pad_l = l; }
            }
        }
        for (int e = episodes_run; e < cfg.train_episodes; ++e) {
            if (a.out.episode_test_mean) a.out.episode_test_mean[chain * cfg.train_episodes + e] = pad_r;
            if (a.out.episode_len) a.out.episode_len[chain * cfg.train_episodes + e] = pad_l;
        }
    }
#ifdef LENV_PHASE_TIMING
    if (tid == 0 && chain == 0) for (int pi = 0; pi < 16; ++pi) g_td3d_phase_cycles[pi] = pt_acc[pi];
#endif
    if (a.out.final_params) for (int p = tid; p < P; p += DNT) a.out.final_params[chain * a.P + p] = params[p];
    if (a.out.status && status != 0) atomicMin(&a.out.status[chain], status);
}

// Fresh agents (actor | critic_1 | critic_2): nn.Linear's default init U(-1/sqrt(fan_in), 1/sqrt(fan_in)) from the chain key's
// STREAM_AGENT_INIT draws (element i of the row draws value i), nn.LayerNorm's weight 1 / bias 0
__global__ void td3d_agent_init_kernel(lenv_td3d_cfg cfg, const int32_t *hp_hidden, const int32_t *hp_layers, const uint64_t *rng_keys,
                                       int64_t chains, int64_t row_stride, float *agent_init)
{
    const int64_t c = blockIdx.y;
    if (c >= chains) return;
    const int H = hp_hidden ? hp_hidden[c] : cfg.hidden, L = hp_layers ? hp_layers[c] : cfg.layers;
    if (H < 1 || H > cfg.hidden || L < 1 || L > cfg.layers) return;            // the inner loop reports status -8 for this chain
    DMlpOff ma, mc;
    const int S = cfg.state_dim, A = cfg.action_dim, SA = S + A;
    dmlp_off(ma, S, H, L, A, cfg.use_layer_norm);
    dmlp_off(mc, SA, H, L, 1, cfg.use_layer_norm);
    const int P = ma.P + 2 * mc.P;
    const uint64_t key = rng_keys[c];
    const float bS = (float)(1.0 / __builtin_sqrt((double)S)), bSA = (float)(1.0 / __builtin_sqrt((double)SA)),
                bH = (float)(1.0 / __builtin_sqrt((double)H));
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x) {
        const bool is_actor = i < ma.P;
        const DMlpOff &m = is_actor ? ma : mc;
        const int j = is_actor ? i : (i - ma.P) % mc.P;
        float v;
        if (m.ln && j >= m.oLN && j < m.oLN + 2 * H) v = j < m.oLN + H ? 1.0f : 0.0f;
        else {
            const float bound = j < m.oW[1] ? (is_actor ? bS : bSA) : bH;
            const float u = (float)u64_to_unit(rng_u64(key, STREAM_AGENT_INIT, (uint64_t)i));
            v = (u * 2.0f - 1.0f) * bound;
        }
        agent_init[c * row_stride + i] = v;
    }
}

}  // namespace lenv

using namespace lenv;

static int td3d_layout(const lenv_td3d_cfg *cfg, Td3dArgs &a, size_t *lds_bytes)
{
    const int H = cfg->hidden, L = cfg->layers, B = cfg->batch_size, T = cfg->test_episodes, Hse = cfg->se_hidden;
    const int S = cfg->state_dim, A = cfg->action_dim, SA = S + A;
    if (!((cfg->env_id == LENV_ENV_CARTPOLE && S == 4 && A == 2) || (cfg->env_id == LENV_ENV_ACROBOT && S == 6 && A == 3) ||
          (cfg->env_id == LENV_ENV_MOUNTAINCAR && S == 2 && A == 3)))
        return LENV_ERR_UNSUPPORTED;
    if (cfg->act == LENV_ACT_PRELU) return LENV_ERR_UNSUPPORTED;   // trained PReLU slope of the agent nets: not a parameter here
    if (L < 1 || L > TD_MAXL || H < 1 || H > TD_MAXW || B < 1 || B > TD_MAXB || T < 1 || T > 64 || cfg->se_layers < 1 || cfg->se_layers > TD_MAXL ||
        Hse > TD_MAXW || Hse < 1 || cfg->policy_delay < 1 || cfg->max_steps < 1 || cfg->train_episodes < 0 || !(cfg->gumbel_temp > 0.0))
        return LENV_ERR_UNSUPPORTED;
    DMlpOff ma, mc, ms;
    dmlp_off(ma, S, H, L, A, cfg->use_layer_norm);
    dmlp_off(mc, SA, H, L, 1, cfg->use_layer_norm);
    a.P = ma.P + 2 * mc.P;
    dmlp_off(ms, SA, Hse, cfg->se_layers, S, 0); a.P_se = ms.P;
    dmlp_off(ms, SA, Hse, cfg->se_layers, 1, 0); a.P_se += 2 * ms.P;
    a.RS = (2 * S + A + 2 + 3) & ~3;
    int64_t cap = (int64_t)cfg->train_episodes * cfg->max_steps;
    if (cap > cfg->rb_size) cap = cfg->rb_size;
    a.rb_cap = cap < 1 ? 1 : cap;
    const int RB = B > T ? B : T;                               // rows of the batch buffers
    const bool ln = cfg->use_layer_norm && L >= 2;
    int64_t off = 0;
    auto take = [&](int64_t n) { int64_t r = off; off += (n + 3) & ~(int64_t)3; return r; };
    a.a_params = take(a.P); a.a_targets = take(a.P); a.a_m = take(a.P); a.a_v = take(a.P); a.a_grad = take(a.P);
    a.a_replay = take(a.rb_cap * a.RS);
    a.a_xc = take((int64_t)RB * SA); a.a_xn = take((int64_t)RB * SA); a.a_xa = take((int64_t)RB * SA);
    for (int l = 0; l < TD_MAXL; ++l) {
        a.a_hc1[l] = take((int64_t)RB * H); a.a_hc2[l] = take((int64_t)RB * H); a.a_ha[l] = take((int64_t)RB * H); a.a_ht[l] = take((int64_t)RB * H);
        a.a_xh1[l] = a.a_xh2[l] = a.a_xha[l] = 0;
        if (ln && l >= 1) { a.a_xh1[l] = take((int64_t)RB * H); a.a_xh2[l] = take((int64_t)RB * H); a.a_xha[l] = take((int64_t)RB * H); }
    }
    a.a_rs1 = take((int64_t)TD_MAXL * B); a.a_rs2 = take((int64_t)TD_MAXL * B); a.a_rsa = take((int64_t)TD_MAXL * B);
    a.a_d[0] = take((int64_t)RB * H); a.a_d[1] = take((int64_t)RB * H);
    a.a_dx = take((int64_t)RB * SA); a.a_raw = take((int64_t)RB * A); a.a_ys = take((int64_t)RB * A); a.a_dz = take((int64_t)RB * A);
    a.a_meter = take(2 * (int64_t)(cfg->train_episodes > 0 ? cfg->train_episodes : 1));
    a.a_se = take(a.P_se);
    a.arena_stride = (off + 63) & ~(int64_t)63;
    const size_t lds_floats = GemmShape<TD_MAXI>::PS_FLOATS + GemmShape<TD_MAXI>::QS_FLOATS + GEMM_QUEUE_MAX * sizeof(GemmCmd) / sizeof(float) + 2 * TD_MAXW +
                              8 * (size_t)B + 64 + 2 + 2 * (4 + 4 * (size_t)T + T) + 3 * (size_t)T + 8 + 8 + 32 + 16 + 16 + 16;
    *lds_bytes = lds_floats * sizeof(float);
    if (*lds_bytes > 160 * 1024) return LENV_ERR_UNSUPPORTED;
    return LENV_OK;
}

extern "C" size_t lenv_td3d_workspace_bytes(const lenv_td3d_cfg *cfg, int64_t chains)
{
    if (!cfg || chains < 0) return 0;
    Td3dArgs a;
    size_t lds;
    if (td3d_layout(cfg, a, &lds) != LENV_OK) return 0;
    return (size_t)chains * a.arena_stride * sizeof(float) + 256;
}

extern "C" int64_t lenv_td3d_num_params(const lenv_td3d_cfg *cfg, int64_t *actor_params, int64_t *critic_params)
{
    if (!cfg) return LENV_ERR_INVALID;
    Td3dArgs a;
    size_t lds;
    const int rc = td3d_layout(cfg, a, &lds);
    if (rc != LENV_OK) return rc;
    DMlpOff ma, mc;
    dmlp_off(ma, cfg->state_dim, cfg->hidden, cfg->layers, cfg->action_dim, cfg->use_layer_norm);
    dmlp_off(mc, cfg->state_dim + cfg->action_dim, cfg->hidden, cfg->layers, 1, cfg->use_layer_norm);
    if (actor_params) *actor_params = ma.P;
    if (critic_params) *critic_params = mc.P;
    return a.P;
}

extern "C" int64_t lenv_td3d_se_num_params(const lenv_td3d_cfg *cfg)
{
    if (!cfg) return LENV_ERR_INVALID;
    Td3dArgs a;
    size_t lds;
    const int rc = td3d_layout(cfg, a, &lds);
    return rc != LENV_OK ? rc : a.P_se;
}

extern "C" int lenv_td3d_agent_init(const lenv_td3d_cfg *cfg, const lenv_chain_hp *hp, const uint64_t *rng_keys, int64_t chains,
                                    float *agent_init, void *stream)
{
    if (!cfg || !rng_keys || !agent_init || chains < 0) return LENV_ERR_INVALID;
    if (hp && (!hp->q_hidden || !hp->q_layers)) return LENV_ERR_INVALID;
    if (chains == 0) return LENV_OK;
    Td3dArgs a;
    size_t lds_bytes;
    const int rc = td3d_layout(cfg, a, &lds_bytes);
    if (rc != LENV_OK) return rc;
    hipLaunchKernelGGL(td3d_agent_init_kernel, dim3(64, (unsigned)chains), dim3(256), 0, static_cast<hipStream_t>(stream), *cfg,
                       hp ? hp->q_hidden : nullptr, hp ? hp->q_layers : nullptr, rng_keys, chains, (int64_t)a.P, agent_init);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}

extern "C" int lenv_td3d_inner_loop(const lenv_td3d_cfg *cfg, const lenv_chain_hp *hp, const float *theta, const float *eps,
                                    const int32_t *worker, const float *sign, const float *agent_init, const uint64_t *rng_keys,
                                    const lenv_td3d_tapes *tapes, int64_t chains, void *workspace, size_t workspace_bytes,
                                    const lenv_td3_out *out, void *stream)
{
    if (hp && (!hp->lr || !hp->batch_size || !hp->q_hidden || !hp->q_layers)) return LENV_ERR_INVALID;
    if (!cfg || !theta || !agent_init || !out || !out->score || !workspace || chains < 0) return LENV_ERR_INVALID;
    if (eps && (!worker || !sign)) return LENV_ERR_INVALID;
    if (cfg->rng_mode == LENV_RNG_TAPE && !tapes) return LENV_ERR_INVALID;
    if (cfg->rng_mode == LENV_RNG_COUNTER && !rng_keys) return LENV_ERR_INVALID;
    if (chains == 0) return LENV_OK;
    Td3dArgs a;
    size_t lds_bytes;
    const int rc = td3d_layout(cfg, a, &lds_bytes);
    if (rc != LENV_OK) return rc;
    if (workspace_bytes < (size_t)chains * a.arena_stride * sizeof(float)) return LENV_ERR_WORKSPACE;
    a.cfg = *cfg;
    a.theta = theta; a.eps = eps; a.worker = worker; a.sign = sign; a.agent_init = agent_init; a.rng_keys = rng_keys;
    if (tapes) a.tapes = *tapes; else a.tapes = lenv_td3d_tapes{};
    a.arena = static_cast<float *>(workspace);
    a.out = *out;
    a.hp_lr = hp ? hp->lr : nullptr; a.hp_batch = hp ? hp->batch_size : nullptr;
    a.hp_hidden = hp ? hp->q_hidden : nullptr; a.hp_layers = hp ? hp->q_layers : nullptr;
    void (*kern)(const Td3dArgs) = nullptr;
    if (cfg->env_id == LENV_ENV_CARTPOLE) kern = td3_discrete_inner_kernel<LENV_ENV_CARTPOLE>;
    else if (cfg->env_id == LENV_ENV_ACROBOT) kern = td3_discrete_inner_kernel<LENV_ENV_ACROBOT>;
    else kern = td3_discrete_inner_kernel<LENV_ENV_MOUNTAINCAR>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return LENV_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)chains), dim3(DNT), lds_bytes, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}

#ifdef LENV_PHASE_TIMING
extern "C" int lenv_debug_td3d_phase_cycles(unsigned long long *host_out)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lenv::g_td3d_phase_cycles), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -4;
}
#endif
