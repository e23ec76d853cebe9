// dueling_se_inner_loop.hip -- fused NES inner loop for DuelingDDQN agents on a synthetic environment (BASELINE config 3:
// Acrobot-v1 SE + DuelingDDQN), one 512-thread workgroup per chain.
//
// Replaces GTN_Worker.calc_score (agents/GTN_worker.py:187-221) with
//   DuelingDDQN.learn / select_*_action        agents/DuelingDDQN.py:59-110
//   Critic_DuelingDQN                          models/actor_critic.py:94-122  (q = V + (Adv - Adv.mean()), GLOBAL mean)
//   BaseAgent.train / test, ReplayBuffer       agents/base_agent.py:64-227, utils.py:9-72
//   EnvWrapper.step -> VirtualEnv.step         envs/env_wrapper.py:16-47, envs/virtual_env.py:43-54
//
// Unlike the DDQN kernel, the agent (67 460 parameters at the config-3 shapes, x5 with target/Adam/grad) does not fit
// LDS: parameters, Adam state and minibatch activations live in a per-chain arena in HBM (L2/MALL-resident while the
// chain runs) and every layer is a workgroup-cooperative, LDS-tiled GEMM  C[i][j] = sum_r P[i][r]*Q[j][r]  whose
// reduction index r runs in ascending order inside one thread (32 accumulators per thread, K-blocked through LDS) --
// i.e. the canonical sequential fmaf chain of oracle/lenv_oracle.h, so results are bit-identical to the oracle.
// The same routine serves forward (r = input feature), input-gradient (r = output unit) and weight-gradient (r = sample)
// products by changing strides.  This is the configuration where HBM traffic is real: ~5 MB of weight/Adam streaming
// per learn step and chain.
#include "lenv_gemm.cuh"
#include "lenv_ln.cuh"
#include "lenv_icm.cuh"
#include "lenv_wavechain_host.h"

namespace lenv {

constexpr int D_MAXL = 3;         // feature-stream hidden layers supported (the *_vary agents draw hidden_layer + 1)
constexpr int D_MAXW = 128;       // max feature_dim (head width) and test episodes
constexpr int D_MAXH = 512;       // max hidden_size  (outputs wider than 128 run as several 128-column blocks)
constexpr int D_MAXB = 640;       // max batch size   (more than 128 rows run as several 128-row blocks)
constexpr int D_MAXI = 128;       // rows of one product block

// Arena offsets (floats) of everything whose size follows from the network / batch shapes; the replay buffer, the episode meter and
// the ICM buffers depend on run-time sizes and are carved after these.  constexpr: literal in the SHAPE 1 instantiation.
struct DuelArena { int64_t a_online, a_target, a_m, a_v, a_grad, a_xs, a_xs2, a_act[D_MAXL], a_feat, a_v1, a_a1, a_t[6], a_dbuf[5], a_xh[D_MAXL], a_rstd, end; };
constexpr DuelArena duel_arena(int S, int H, int F, int L, int B, int T, int64_t P, bool ln = false)
{
    DuelArena a{};
    int64_t off = 0;
#define LENV_TAKE64(n) ([&]() { int64_t r_ = off; off += ((int64_t)(n) + 3) & ~(int64_t)3; return r_; }())
    a.a_online = LENV_TAKE64(P); a.a_target = LENV_TAKE64(P); a.a_m = LENV_TAKE64(P); a.a_v = LENV_TAKE64(P); a.a_grad = LENV_TAKE64(P);
    const int64_t rows = B > T ? B : T;                    // minibatch rows / lock-step test episodes
    const int W = H > F ? H : F;
    a.a_xs = LENV_TAKE64(rows * S); a.a_xs2 = LENV_TAKE64(rows * S);
    for (int l = 0; l < D_MAXL; ++l) a.a_act[l] = LENV_TAKE64(l < L ? (int64_t)B * H : 0);
    a.a_feat = LENV_TAKE64((int64_t)B * F); a.a_v1 = LENV_TAKE64((int64_t)B * F); a.a_a1 = LENV_TAKE64((int64_t)B * F);
    for (int l = 0; l < 6; ++l) a.a_t[l] = LENV_TAKE64(l < L || l >= 3 ? rows * W : 0);
    for (int l = 0; l < 5; ++l) a.a_dbuf[l] = LENV_TAKE64(rows * W);
    // q_layer_norm: the normalised rows and 1 / sqrt(var + eps) of the LayerNorm positions (hidden layers 1 .. L-1) of the online pass over s
    for (int l = 0; l < D_MAXL; ++l) a.a_xh[l] = LENV_TAKE64(ln && l >= 1 && l < L ? (int64_t)B * H : 0);
    a.a_rstd = LENV_TAKE64(ln ? (int64_t)D_MAXL * B : 0);
#undef LENV_TAKE64
    a.end = off;
    return a;
}

struct DuelArgs {
    lenv_ddqn_cfg cfg;
    const float *theta, *eps; const int32_t *worker; const float *sign;
    const float *agent_init; const uint64_t *rng_keys;
    lenv_tapes tapes;
    float *arena; int64_t arena_stride;       // per-chain arena (floats)
    lenv_inner_out out;
    int64_t rb_cap; int RS;
    int P, P_se, se_net_size[3];              // P: parameters at cfg's (maximal) shapes = row stride of agent_init / final_online
    // per-chain hyper-parameters (device arrays [chains], all or none): the *_vary agents (agents/DDQN_vary.py:26-59)
    const double *hp_lr; const int32_t *hp_batch, *hp_hidden, *hp_layers;
    // arena offsets (floats), sized for cfg's (maximal) shapes
    DuelArena A;                              // shape-determined part (see duel_arena)
    int64_t a_replay, a_meter;                // run-time sized: carved after A.end
    int64_t a_se_mid;                         // SEs with several hidden layers: [3][se_layers - 1][Hse*Hse + Hse] perturbed weights | biases
    // ICM agents (cfg.icm_enabled): fresh parameters per chain, optional final parameters, arena offsets of the ICM buffers
    const float *icm_init; float *icm_final; int P_icm;
    int64_t a_icm[IB_COUNT];
};

// parameter offsets inside one parameter vector (state-dict order)
struct DuelOffsets {
    int oWf[D_MAXL + 1], obf[D_MAXL + 1];     // feature stream: hidden layers 0..L-1, then the output Linear (index L)
    int oWv1, obv1, oWv2, obv2, oWa1, oba1, oWa2, oba2;
    int oLN;                                  // q_layer_norm with L >= 2: weight [H] | bias [H] of the ONE shared nn.LayerNorm, behind the second Linear (Module.parameters() order)
    int P;
};

__host__ __device__ constexpr DuelOffsets duel_param_offsets(int S, int A, int H, int F, int L, bool plain, bool ln = false)
{
    DuelOffsets d{};
    int o = 0, n_in = S;
    for (int l = 0; l <= D_MAXL; ++l) d.oWf[l] = d.obf[l] = 0;
    for (int l = 0; l < L; ++l) {
        d.oWf[l] = o; o += H * n_in; d.obf[l] = o; o += H; n_in = H;
        if (ln && l == 1) { d.oLN = o; o += 2 * H; }
    }
    d.oWf[L] = o; o += F * H; d.obf[L] = o; o += F;
    if (plain) { d.oWv1 = d.obv1 = d.oWv2 = d.obv2 = d.oWa1 = d.oba1 = d.oWa2 = d.oba2 = o; }       // no heads
    else {
        d.oWv1 = o; o += F * F; d.obv1 = o; o += F; d.oWv2 = o; o += F; d.obv2 = o; o += 1;
        d.oWa1 = o; o += F * F; d.oba1 = o; o += F; d.oWa2 = o; o += A * F; d.oba2 = o; o += A;
    }
    d.P = o;
    return d;
}

// Diagnostic build only (-DLENV_PHASE_TIMING): per-phase shader-clock totals of chain 0, never in the shipped library.
#ifdef LENV_PHASE_TIMING
__device__ unsigned long long g_duel_phase_cycles[16];
#define PT_DECL unsigned long long pt_last = __builtin_readcyclecounter(), pt_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PT_MARK(i) do { unsigned long long pt_now = __builtin_readcyclecounter(); pt_acc[i] += pt_now - pt_last; pt_last = pt_now; } while (0)
#else
#define PT_DECL
#define PT_MARK(i)
#endif

// SHAPE 1 = the published Acrobot DuelingDDQN configuration (default_config_acrobot.yaml's duelingddqn section = BASELINE
// configs[2]: Critic_DuelingDQN 6-128-128 / feature 128 / 3 actions, relu, batch 128; SE hidden 128 leakyrelu; 10 test episodes) in
// production form (counter RNG, no step trace, no per-chain hyper-parameters, no ICM): its dimensions are literals, which removes
// a third of the scalar-register reloads and folds the orchestration arithmetic (646 -> 578 us per learn step, tools/ubench/ab_duel.sh).
struct DuelShape { int env, kind, S, A, F, H, L, B, Hse, T, q_act, se_act; };        // kind 1 = DuelingDDQN, 0 = DDQN (plain mode: F = A)
constexpr DuelShape kDuelShapes[] = {
    { -1, 1, 4, 2, 1, 1, 1, 1, 1, 1, 0, 0 },                                                                          // 0: generic (unused entry)
    { LENV_ENV_ACROBOT, 1, 6, 3, 128, 128, 2, 128, 128, 10, LENV_ACT_RELU, LENV_ACT_LEAKYRELU },        // 1: default_config_acrobot.yaml duelingddqn = BASELINE configs[2]
    { LENV_ENV_MOUNTAINCAR, 0, 2, 3, 3, 256, 2, 128, 128, 10, LENV_ACT_RELU, LENV_ACT_LEAKYRELU },      // 2: default_config_mountaincar.yaml (DDQN 2-256-256-3)
};

template <bool ICM, int SHAPE = 0>
__global__ __launch_bounds__(DNT) void dueling_se_inner_kernel(const DuelArgs a)
{
    extern __shared__ __align__(16) float lds[];
    const lenv_ddqn_cfg &cfg = a.cfg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t chain = blockIdx.x;
    // the chain's status word starts at 0 (ok); written here rather than by a memset node in front of the launch (a captured
    // generation replayed under rocprofv3 did not run the memset)
    if (threadIdx.x == 0 && a.out.status) a.out.status[chain] = 0;
    constexpr bool FIXED = SHAPE != 0;
    constexpr DuelShape kDuelShape = kDuelShapes[SHAPE];
    static_assert(!(FIXED && ICM), "the specialised instantiations have no ICM");
    // agent_kind 1 = DuelingDDQN (Critic_DuelingDQN); agent_kind 0 = DDQN whose Critic_DQN (models/actor_critic.py:84-91:
    // build_nn_from_config(S -> A) with `hidden_layer` hidden layers) does not fit the register-resident small kernel
    // (hidden_layer >= 2, wide layers): the "feature stream" IS the Q-net then (output width A, no heads, no advantage mean).
    const bool plain = FIXED ? kDuelShape.kind == 0 : cfg.agent_kind == 0;
    const int S = FIXED ? kDuelShape.S : cfg.state_dim, A = FIXED ? kDuelShape.A : cfg.num_actions, K = S + A;
    const int F = FIXED ? kDuelShape.F : (plain ? A : cfg.feature_dim);
    const int CFG_B = FIXED ? kDuelShape.B : cfg.batch_size;                     // the launch's (maximal) batch: LDS is carved for it
    // the chain's own lr / batch_size / hidden_size / hidden_layer when the launch carries per-chain arrays, else cfg's
    const bool vary = FIXED ? false : a.hp_batch != nullptr;
    const int H = FIXED ? kDuelShape.H : (vary ? a.hp_hidden[chain] : cfg.q_hidden), L = FIXED ? kDuelShape.L : (vary ? a.hp_layers[chain] : cfg.q_layers);
    const int B = FIXED ? kDuelShape.B : (vary ? a.hp_batch[chain] : cfg.batch_size);
    const double lr = vary ? a.hp_lr[chain] : cfg.lr;
    if (vary && (H < 1 || H > cfg.q_hidden || L < 1 || L > cfg.q_layers || B < 1 || B > CFG_B)) {   // uniform per chain
        if (tid == 0) { if (a.out.status) a.out.status[chain] = -8; a.out.score[chain] = 0.0; }
        return;
    }
    // `use_layer_norm` of the agent's section (model_utils.py:22-37): the shared LayerNorm behind hidden Linear 2..L of the Q-net / the feature stream
    const bool ln = FIXED ? false : (cfg.q_layer_norm != 0 && L >= 2);
    const DuelOffsets po = duel_param_offsets(S, A, H, F, L, plain, ln);
    const int Hse = FIXED ? kDuelShape.Hse : cfg.se_hidden, RS = a.RS, P = po.P, T = FIXED ? kDuelShape.T : cfg.test_episodes;
    const int act_id = FIXED ? kDuelShape.q_act : cfg.q_act, se_act_id = FIXED ? kDuelShape.se_act : cfg.se_act;
    const float prelu = cfg.q_prelu;

    // ---- LDS carve-up ----
    float *Ps = lds, *Qs = Ps + GemmShape<D_MAXI>::PS_FLOATS;
    GemmCmd *cmds = reinterpret_cast<GemmCmd *>(Qs + GemmShape<D_MAXI>::QS_FLOATS);   // [GEMM_QUEUE_MAX] command queue
    float *se_w0T = reinterpret_cast<float *>(cmds + GEMM_QUEUE_MAX);   // [3][K][Hse]
    float *se_b0 = se_w0T + 3 * K * Hse;                  // [3][Hse]
    float *se_wout = se_b0 + 3 * Hse;                     // [S+2][Hse]
    float *se_bout = se_wout + (S + 2) * Hse;             // [S+2] (padded to 16)
    float *se_h = se_bout + 16;                           // [3][Hse]
    const int seL = FIXED ? 1 : cfg.se_layers;              // hidden layers of the SE's three nets (virtual_env.py:16-33)
    float *se_h2 = se_h + 3 * Hse;                        // [3][Hse] second hidden row (se_layers > 1 only)
    const int RBH = CFG_B > T ? CFG_B : T;                  // LDS is carved for cfg's (maximal) batch
    float *qv = se_h2 + (seL > 1 ? 3 * Hse : 0);                           // [3][B][A]   q(s), q_online(s'), q_target(s'); [T][A] in the test phase
    float *Vb = qv + 3 * RBH * A;                           // [3][RBH]  value-head outputs of the three passes (slot 0 reused as scratch)
    float *Advb = Vb + 3 * RBH;                           // [3][RBH][A] advantage-head outputs (RBH = max(B, T) rows per slot)
    float *dq = Advb + 3 * RBH * A;                       // [B]
    float *dAdv = dq + CFG_B;                             // [B][A]
    float *misc = dAdv + CFG_B * A;                       // [64] control words
    double *dstate = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(misc + 64) + 7) & ~(uintptr_t)7);   // [T][4] real-env states (tests)
    double *ret = dstate + 4 * T;                         // [T]
    float *ep_rew = reinterpret_cast<float *>(ret + T);   // [T]
    int *alive = reinterpret_cast<int *>(ep_rew + T);     // [T]
    float *state = reinterpret_cast<float *>(alive + T);  // [8] current SE state
    float *newrow = state + 8;                            // [16]
    int *tlen = reinterpret_cast<int *>(newrow + 16);     // [T] lengths of the episodes of the last test phase (time-out cut)
    double *tstate = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(tlen + T) + 7) & ~(uintptr_t)7);   // [4] RewardEnv: the real training env's state
    volatile float *ctrl = misc;
    volatile int *ictrl = reinterpret_cast<volatile int *>(misc + 32);
    float *se_ln_stat = misc + 20;                        // [3][2] mean | rstd of the SE nets' LayerNorm rows (cfg.se_layer_norm)
    const bool se_ln = FIXED ? false : (cfg.se_layer_norm != 0);

    float *arena = a.arena + chain * a.arena_stride;
    constexpr DuelArena AC = duel_arena(kDuelShape.S, kDuelShape.H, kDuelShape.F, kDuelShape.L, kDuelShape.B, kDuelShape.T,
                                        duel_param_offsets(kDuelShape.S, kDuelShape.A, kDuelShape.H, kDuelShape.F, kDuelShape.L, kDuelShape.kind == 0).P);
#define AV(f) (FIXED ? AC.f : a.A.f)
    float *online = arena + AV(a_online), *target = arena + AV(a_target), *adam_m = arena + AV(a_m), *adam_v = arena + AV(a_v);
    float *grad = arena + AV(a_grad), *rb = arena + a.a_replay, *xs = arena + AV(a_xs), *xs2 = arena + AV(a_xs2);
    float *feat_s = arena + AV(a_feat), *v1_s = arena + AV(a_v1), *a1_s = arena + AV(a_a1);
    double *meter = reinterpret_cast<double *>(arena + a.a_meter);

    // gtn.synthetic_env_type 1: the agent trains on a RewardEnv over the REAL env (envs/reward_env.py:61-133): the transition is
    // the real one, the reward goes through the perturbed reward network (state_dim -> se_hidden -> 1; reward types 0,1,2,5,6)
    const bool reward_env = FIXED ? false : cfg.synthetic_env_type == 1;      // the specialised shapes are VirtualEnv configurations
    const int k_rep = FIXED ? 1 : (cfg.same_action_num > 1 ? cfg.same_action_num : 1);   // env steps per chosen action (same_action_num)
    const int rtype = cfg.reward_env_type;
    const int Drn = rtype == 0 ? 1 : S;                   // RewardEnv.build_reward_net: a 1-input dummy net for type 0
    float *rn_w = se_w0T, *rn_h = se_h;                   // RewardEnv: flat reward-net parameters / its hidden layer (LDS)
    // ---- stage the perturbed SE (GTN_worker.py:165-175) ----
    if (reward_env) {
        const float sg = a.eps ? a.sign[chain] : 0.0f;
        const float *e = a.eps ? a.eps + (int64_t)a.worker[chain] * a.P_se : nullptr;
        float *dst = seL > 1 ? arena + a.a_se_mid : rn_w;       // a reward net with several hidden layers does not fit the LDS rows: arena
        for (int i = tid; i < a.P_se; i += DNT) dst[i] = e ? fma32(sg, e[i], a.theta[i]) : a.theta[i];
    } else {
        const float sg = a.eps ? a.sign[chain] : 0.0f;
        const float *e = a.eps ? a.eps + (int64_t)a.worker[chain] * a.P_se : nullptr;
        for (int i = tid; i < a.P_se; i += DNT) {
            const float w = e ? fma32(sg, e[i], a.theta[i]) : a.theta[i];
            int net = 0, r = i;
            if (r >= a.se_net_size[0]) { r -= a.se_net_size[0]; net = 1; if (r >= a.se_net_size[1]) { r -= a.se_net_size[1]; net = 2; } }
            const int orow = net == 0 ? 0 : (net == 1 ? S : S + 1);
            const int mid = (seL - 1) * (Hse * Hse + Hse);        // hidden-to-hidden layers of this net (state-dict order: W_l | b_l)
            if (r < Hse * K) { int j = r / K, k = r - j * K; se_w0T[(net * K + k) * Hse + j] = w; }
            else if ((r -= Hse * K) < Hse) se_b0[net * Hse + r] = w;
            else if ((r -= Hse) < mid) arena[a.a_se_mid + (int64_t)net * mid + r] = w;
            else {
                r -= mid;
                const int n_out = net == 0 ? S : 1;
                if (r < n_out * Hse) { int o = r / Hse, j = r - o * Hse; se_wout[(orow + o) * Hse + j] = w; }
                else se_bout[orow + (r - n_out * Hse)] = w;
            }
        }
    }
    // ---- fresh agent: online = target = agent_init, Adam state 0 (DuelingDDQN.py:31-36) ----
    for (int p = tid; p < P; p += DNT) {
        const float w = a.agent_init[chain * a.P + p];
        online[p] = w; target[p] = w; adam_m[p] = 0.0f; adam_v[p] = 0.0f;
    }
    // ---- fresh ICM (ICM.__init__, icm_baseline.py:108-133): parameters from icm_init, Adam state 0 ----
    IcmNet icm;
    double icm_pows[2] = { 1.0, 1.0 };
    if constexpr (ICM) {
        icm_build(icm, S, A, cfg.icm_feature_dim, cfg.icm_hidden, /*discrete=*/true);
        float *ip = arena + a.a_icm[IB_P], *im = arena + a.a_icm[IB_M], *iv = arena + a.a_icm[IB_V];
        for (int p = tid; p < icm.P; p += DNT) { ip[p] = a.icm_init[chain * a.P_icm + p]; im[p] = 0.0f; iv[p] = 0.0f; }
    }
    if (tid < 64) misc[tid] = 0.0f;
    __syncthreads();

    const uint64_t key = a.rng_keys ? a.rng_keys[chain] : 0;
    const bool tape = FIXED ? false : cfg.rng_mode == LENV_RNG_TAPE;
    const int env_id = FIXED ? kDuelShape.env : cfg.env_id;
    int status = 0;
    PT_DECL;
    int train_steps = 0, n_act = 0, learn_it = 0, n_test_ep = 0, test_steps = 0, episodes_run = 0;
    double eps_g = cfg.eps_init, b1pow = 1.0, b2pow = 1.0;
    const int rb_cap = (int)a.rb_cap;

    auto obs_of = [&](const double *st, float *obs) { real_env_obs(env_id, st, obs); };

    // ---- Critic_DuelingDQN forward of I rows X[I][S] with parameters `par` (actor_critic.py:117-122) -------------------
    // queue_forward queues the six layer products (intermediate activations to the given buffers; stored ones are kept for
    // the backward pass), head outputs to slot `slot` of Vb / Advb; after gq.run, finish_q combines them into q_out[I][A]
    // (LDS).  global_mean: mean over all I*A advantages (learn, the reference's batch quirk) or per row.
    GemmQueue gq(cmds);
    constexpr int LNBUF = GemmShape<D_MAXI>::PS_FLOATS + GemmShape<D_MAXI>::QS_FLOATS;
    auto queue_forward = [&](const float *par, const float *X, int I, float *const *hid, float *featb, float *v1b, float *a1b, int slot,
                             float *const *xhs = nullptr, float *rstds = nullptr) {
        const float *in = X;
        int n_in = S;
        for (int l = 0; l < L; ++l) {
            if (ln && l >= 1) {
                // a LayerNorm position: Linear + bias, the queue runs (everything queued so far, in order), then the row routine
                // normalises, scales and applies the activation (lenv_ln.cuh); xhs / rstds: kept for the backward pass
                gq.gemm(in, n_in, 1, par + po.oWf[l], n_in, 1, I, H, n_in, epi_bias(hid[l], H, 0, par + po.obf[l]));
                gq.run<D_MAXI>(Ps, Qs);
                ln_rows_forward<LNBUF>(Ps, hid[l], I, H, par + po.oLN, par + po.oLN + H, xhs ? xhs[l] : nullptr, rstds ? rstds + (int64_t)l * CFG_B : nullptr, act_id, prelu);
            } else gq.gemm(in, n_in, 1, par + po.oWf[l], n_in, 1, I, H, n_in, epi_bias_act(hid[l], H, par + po.obf[l], act_id, prelu));
            in = hid[l]; n_in = H;
        }
        // feature_stream's last Linear: no activation (build_nn_from_config ends with a Linear)
        if (plain) {                                       // Critic_DQN: that Linear's output is Q(s, .) -> slot `slot` of Advb
            gq.gemm(in, n_in, 1, par + po.oWf[L], n_in, 1, I, A, n_in, epi_bias(Advb + slot * RBH * A, A, 0, par + po.obf[L]));
            return;
        }
        gq.gemm(in, n_in, 1, par + po.oWf[L], n_in, 1, I, F, n_in, epi_bias(featb, F, 0, par + po.obf[L]));
        gq.gemm(featb, F, 1, par + po.oWv1, F, 1, I, F, F, epi_bias_act(v1b, F, par + po.obv1, act_id, prelu));
        gq.gemm(featb, F, 1, par + po.oWa1, F, 1, I, F, F, epi_bias_act(a1b, F, par + po.oba1, act_id, prelu));
        gq.gemm(v1b, F, 1, par + po.oWv2, F, 1, I, 1, F, epi_bias(Vb + slot * RBH, 1, 0, par + po.obv2));
        gq.gemm(a1b, F, 1, par + po.oWa2, F, 1, I, A, F, epi_bias(Advb + slot * RBH * A, A, 0, par + po.oba2));
    };
    auto finish_q = [&](int slot, int I, float *q_out, bool global_mean) {
        if (plain) {                                       // Q itself sits in the "advantage" slot (sized for max(B, T) rows)
            const float *src = Advb + slot * RBH * A;
            for (int e = tid; e < I * A; e += DNT) q_out[e] = src[e];
            __syncthreads();
            return;
        }
        const float *Vs = Vb + slot * RBH, *As = Advb + slot * RBH * A;
        if (global_mean) {
            if (tid == 0) {
                float sum = 0.0f;
                for (int e = 0; e < I * A; ++e) sum = sum + As[e];
                ctrl[8] = sum / (float)(I * A);
            }
            __syncthreads();
            const float mean = ctrl[8];
            for (int e = tid; e < I * A; e += DNT) q_out[e] = Vs[e / A] + (As[e] - mean);
        } else {
            for (int i = tid; i < I; i += DNT) {
                float sum = 0.0f;
                for (int aa = 0; aa < A; ++aa) sum = sum + As[i * A + aa];
                const float mean = sum / (float)A;
                for (int aa = 0; aa < A; ++aa) q_out[i * A + aa] = Vs[i] + (As[i * A + aa] - mean);
            }
        }
        __syncthreads();
    };
    auto forward = [&](const float *par, const float *X, int I, float *const *hid, float *featb, float *v1b, float *a1b,
                       float *q_out, bool global_mean) {
        queue_forward(par, X, I, hid, featb, v1b, a1b, 0);
        gq.run<D_MAXI>(Ps, Qs);
        finish_q(0, I, q_out, global_mean);
    };

    // ---- ICM.train + compute_intrinsic_rewards on the gathered minibatch (agents/DDQN.py:74-76): lenv_icm.cuh ----
    auto icm_step = [&]() {
        if constexpr (ICM) {
            IcmStep st{ gq, Ps, Qs, arena, a.a_icm, ctrl, 12, icm, icm_pows, cfg.icm_lr, cfg.icm_beta, cfg.icm_eta, cfg.adam_beta1, cfg.adam_beta2,
                        cfg.adam_eps, B, xs, S, xs2, S, /*continuous=*/false };
            const int Ai = icm.Ai;
            // action inputs: the index itself for two actions (one BCE logit), else its one-hot (icm_baseline.py:39-40,134-136);
            // the sampled (a, r, done) sit in dAdv / dq until the TD step
            icm_train_and_reward<D_MAXI>(st, [&](int b, int i) { return Ai == 1 ? dAdv[b * A] : ((int)dAdv[b * A] == i ? 1.0f : 0.0f); },
                                         [&](int b, float r) { dAdv[b * A + 1] = dAdv[b * A + 1] + r; });
        }
    };

    float *hid_s[D_MAXL], *hid_t[D_MAXL];
    for (int l = 0; l < D_MAXL; ++l) { hid_s[l] = arena + AV(a_act[l]); hid_t[l] = arena + AV(a_t[l]); }
    float *xh_s[D_MAXL], *rstd_s = arena + AV(a_rstd);
    for (int l = 0; l < D_MAXL; ++l) xh_s[l] = arena + AV(a_xh[l]);
    float *feat_t = arena + AV(a_t[3]), *v1_t = arena + AV(a_t[4]), *a1_t = arena + AV(a_t[5]);   // temporaries of non-stored passes

    // ---- real-env test phase: the T episodes advance in lock-step as one batch (weights are streamed once per step) ----
    auto test_phase = [&]() {
        if (tid < T) {
            double st[4];
            const int64_t row = (int64_t)n_test_ep + tid;
            if (tape) {
                if (row >= a.tapes.test_reset_stride) { status = -5; for (int i = 0; i < 4; ++i) st[i] = 0.0; }
                else for (int i = 0; i < 4; ++i) st[i] = a.tapes.test_reset[(chain * a.tapes.test_reset_stride + row) * 4 + i];
            } else {
                real_env_reset_draw(env_id, key, STREAM_TEST_RESET, row, st);
            }
            for (int i = 0; i < 4; ++i) dstate[tid * 4 + i] = st[i];
            ep_rew[tid] = 0.0f; alive[tid] = 1; tlen[tid] = 0;
        }
        if (tid == 0) ictrl[0] = 0;
        __syncthreads();
        float *xt = xs2;                                   // [T][S] observation batch (xs2 is free outside learn)
        for (int t = 0; t < cfg.max_steps; t += k_rep) {   // base_agent.py:194 range(0, max_steps, same_action_num)
            if (tid < T) { float obs[8]; obs_of(dstate + tid * 4, obs); for (int i = 0; i < S; ++i) xt[tid * S + i] = obs[i]; }
            __syncthreads();
            forward(online, xt, T, hid_t, feat_t, v1_t, a1_t, qv, false);
            if (tid < T && alive[tid]) {
                int am = 0; float best = qv[tid * A];
                for (int aa = 1; aa < A; ++aa) { const float v = qv[tid * A + aa]; if (v > best) { best = v; am = aa; } }
                double st[4] = { dstate[tid * 4], dstate[tid * 4 + 1], dstate[tid * 4 + 2], dstate[tid * 4 + 3] };
                // EnvWrapper.step on the real env (env_wrapper.py:56-61): the action same_action_num times or until done (TimeLimit:
                // done after max_steps env steps), python-float reward sum
                double rsum = 0.0;
                int nstep = 0;
                for (int r_ = 0; r_ < k_rep; ++r_) {
                    double rew; int dn;
                    real_env_step(env_id, st, am, rew, dn);
                    rsum = rsum + rew;
                    ++nstep;
                    if (tlen[tid] + nstep >= cfg.max_steps) dn = 1;
                    if (dn) { alive[tid] = 0; break; }
                }
                for (int i = 0; i < 4; ++i) dstate[tid * 4 + i] = st[i];
                ep_rew[tid] = ep_rew[tid] + (float)rsum;
                tlen[tid] = tlen[tid] + nstep;
                atomicAdd(const_cast<int *>(&ictrl[0]), nstep);
            }
            __syncthreads();
            int any = 0;
            for (int e = 0; e < T; ++e) any |= alive[e];
            if (!any) break;
        }
        if (tid < T) ret[tid] = (double)ep_rew[tid];
        n_test_ep += T;
        __syncthreads();
        test_steps += ictrl[0];
        __syncthreads();
    };

    // Deterministic time-out (lenv_ddqn_cfg::step_budget, base_agent.py:30-47): elapsed = env steps taken so far
    const bool budgeted = cfg.step_budget > 0;
    const bool no_test_env = FIXED ? false : cfg.test_mode == 1;      // BaseAgent.train(env, test_env=None): lenv_ddqn_cfg::test_mode
    int timed_out_at = -1;
    for (int episode = 0; episode < cfg.train_episodes; ++episode) {
        if (budgeted && (int64_t)train_steps + test_steps > cfg.step_budget) { timed_out_at = episode; break; }   // uniform
        if (episode == 0) eps_g = cfg.eps_init;            // DuelingDDQN.update_parameters_per_episode (:112-117)
        else { eps_g *= cfg.eps_decay; if (eps_g < cfg.eps_min) eps_g = cfg.eps_min; }
        const bool learning = episode >= cfg.init_episodes;
        if (tid == 0) {                                    // env.reset(): VirtualEnv.reset (virtual_env.py:35-41)
            double st0[4];
            if (tape) {
                if (episode >= a.tapes.train_reset_stride) { status = -5; for (int i = 0; i < 4; ++i) st0[i] = 0.0; }
                else for (int i = 0; i < 4; ++i) st0[i] = a.tapes.train_reset[(chain * a.tapes.train_reset_stride + episode) * 4 + i];
            } else {
                real_env_reset_draw(env_id, key, STREAM_TRAIN_RESET, episode, st0);
            }
            float obs[8];
            obs_of(st0, obs);
            for (int i = 0; i < S; ++i) state[i] = obs[i];
            if (reward_env) { for (int i = 0; i < 4; ++i) tstate[i] = st0[i]; ictrl[5] = 0; }   // RewardEnv.reset: real env state; no phi(s) yet
        }
        __syncthreads();
        int ep_len = 0, env_steps = 0;
        float tr_reward = 0.0f;                                  // base_agent.py:102,121 episode_reward += reward (fp32 tensors; uniform over the threads)
        for (int t = 0; t < cfg.max_steps; t += k_rep) {         // base_agent.py:104 range(0, max_steps, same_action_num)
            PT_MARK(9);
            const int size_after = train_steps + 1 < rb_cap ? train_steps + 1 : rb_cap;
            const int new_pos = train_steps % rb_cap;
            // ---- select_train_action (DuelingDDQN.py:96-103) ----
            if (tid == 0) {
                double u;
                if (tape) { if (train_steps >= a.tapes.eps_uniform_stride) { status = -2; u = 1.0; } else u = a.tapes.eps_uniform[chain * a.tapes.eps_uniform_stride + train_steps]; }
                else u = u64_to_unit(rng_u64(key, STREAM_EPS, (uint64_t)train_steps));
                int explored = u < eps_g, action = -1;
                if (explored) {
                    if (tape) { if (n_act >= a.tapes.rand_action_stride) { status = -3; action = 0; } else action = a.tapes.rand_action[chain * a.tapes.rand_action_stride + n_act]; }
                    else action = (int)u64_to_below(rng_u64(key, STREAM_ACTION, (uint64_t)n_act), (uint32_t)A);
                }
                ictrl[1] = explored; ictrl[2] = action;
            }
            __syncthreads();
            const int explored = ictrl[1];
            if (explored) ++n_act;
            if (!explored) {
                for (int i = tid; i < S; i += DNT) xs2[i] = state[i];
                __syncthreads();
                forward(online, xs2, 1, hid_t, feat_t, v1_t, a1_t, qv, false);
                if (tid == 0) {
                    int am = 0; float best = qv[0];
                    for (int aa = 1; aa < A; ++aa) if (qv[aa] > best) { best = qv[aa]; am = aa; }
                    ictrl[2] = am;
                }
                __syncthreads();
            }
            const int action = ictrl[2];
            PT_MARK(0);
            if (reward_env) {
                // ---- EnvWrapper.step -> RewardEnv.step (reward_env.py:61-66): real transition (TimeLimit: done at max_steps),
                // reward = _calc_reward(state, next_state, reward) -- oracle: rn_shape_one ----
                // build_nn_from_config (model_utils.py:16-37): Linear(D, H) | [Linear(H, H) (+ the shared LayerNorm)] x (layers - 1) | Linear(H, 1),
                // flat in Module.parameters() order; the LayerNorm is never perturbed (weight 1 / bias 0, not in theta: cfg.se_layer_norm)
                auto rn_phi = [&](const float *obs, int slot) {       // phi = reward_net(obs) -> ctrl[slot]
                    const float *rnp = seL > 1 ? arena + a.a_se_mid : rn_w;
                    const float *W0 = rnp, *b0 = rnp + Hse * Drn;
                    for (int j = tid; j < Hse; j += DNT) {
                        float z = 0.0f;
                        for (int k = 0; k < Drn; ++k) z = fma32(obs[k], W0[j * Drn + k], z);
                        rn_h[j] = act_fwd(se_act_id, cfg.se_prelu, z + b0[j]);
                    }
                    __syncthreads();
                    const float *hp = rn_h, *Wl = b0 + Hse;
                    float *hn = se_h2;
                    for (int l = 1; l < seL; ++l) {
                        const float *bl = Wl + Hse * Hse;
                        for (int j = tid; j < Hse; j += DNT) {
                            float z = 0.0f;
                            for (int k = 0; k < Hse; ++k) z = fma32(hp[k], Wl[j * Hse + k], z);
                            z = z + bl[j];
                            hn[j] = se_ln ? z : act_fwd(se_act_id, cfg.se_prelu, z);
                        }
                        __syncthreads();
                        if (se_ln) {
                            if (tid == 0) {
                                float sm = 0.0f, sv = 0.0f;
                                for (int jj = 0; jj < Hse; ++jj) sm = sm + hn[jj];
                                const float mean = sm / (float)Hse;
                                for (int jj = 0; jj < Hse; ++jj) { const float dj = hn[jj] - mean; sv = fma32(dj, dj, sv); }
                                se_ln_stat[0] = mean; se_ln_stat[1] = 1.0f / __builtin_sqrtf(sv / (float)Hse + 1e-5f);
                            }
                            __syncthreads();
                            for (int j = tid; j < Hse; j += DNT) hn[j] = act_fwd(se_act_id, cfg.se_prelu, fma32((hn[j] - se_ln_stat[0]) * se_ln_stat[1], 1.0f, 0.0f));
                            __syncthreads();
                        }
                        const float *t2 = hp; hp = hn; hn = const_cast<float *>(t2);
                        Wl = bl + Hse;
                    }
                    const float *Wo = Wl, *bo = Wo + Hse;
                    if (tid == 0) {
                        float acc = 0.0f;
                        for (int j = 0; j < Hse; ++j) acc = fma32(hp[j], Wo[j], acc);
                        ctrl[slot] = acc + bo[0];
                    }
                    __syncthreads();
                };
                if (tid >= 64 && tid < 64 + S) newrow[tid - 64] = state[tid - 64];
                if (tid == 128) newrow[S] = (float)action;
                // EnvWrapper.step (env_wrapper.py:56-61): the action same_action_num times or until done, python-float sum of the
                // shaped rewards (thread 0 keeps it)
                double rsum = 0.0;
                for (int r_ = 0; r_ < k_rep; ++r_) {
                    ++env_steps;
                    if (tid == 0) {
                        double st[4] = { tstate[0], tstate[1], tstate[2], tstate[3] };
                        double rew; int dn;
                        real_env_step(env_id, st, action, rew, dn);
                        if (env_steps >= cfg.max_steps) dn = 1;
                        for (int i = 0; i < 4; ++i) tstate[i] = st[i];
                        float obs[8];
                        obs_of(st, obs);
                        for (int i = 0; i < S; ++i) newrow[S + 1 + i] = obs[i];
                        ctrl[16] = (float)rew; newrow[2 * S + 2] = dn ? 1.0f : 0.0f;
                    }
                    __syncthreads();
                    if (rtype != 0) {
                        if ((rtype == 1 || rtype == 2) && ictrl[5] == 0) rn_phi(state, 14);      // phi(s): carried over after the first step
                        rn_phi(newrow + S + 1, 15);
                    }
                    if (tid == 0) {
                        const float g32 = (float)cfg.gamma, r32 = ctrl[16], phi_s = ctrl[14], phi_s2 = ctrl[15];
                        float shaped;
                        switch (rtype) {
                        case 0: shaped = r32; break;
                        case 1: shaped = g32 * phi_s2 - phi_s; break;
                        case 2: shaped = (r32 + g32 * phi_s2) - phi_s; break;
                        case 5: shaped = phi_s2; break;
                        default: shaped = r32 + phi_s2; break;             // 6
                        }
                        rsum = rsum + (double)shaped;
                        newrow[2 * S + 1] = (float)rsum;
                        ctrl[14] = phi_s2; ictrl[5] = 1;
                    }
                    if (r_ + 1 < k_rep) {
                        __syncthreads();
                        if (newrow[2 * S + 2] > 0.5f) break;                          // uniform: the repeats stop at done
                        __syncthreads();
                    }
                }
            } else {
            // ---- EnvWrapper.step -> VirtualEnv.step: x = [onehot(action), state]; same_action_num SE steps whatever the done flag says,
            // fp32 reward sum (env_wrapper.py:24-29) ----
            if (tid >= 64 && tid < 64 + S) newrow[tid - 64] = state[tid - 64];
            if (tid == 128) newrow[S] = (float)action;
            for (int r_ = 0; r_ < k_rep; ++r_) {
            for (int uu = tid; uu < 3 * Hse; uu += DNT) {
                const int net = uu / Hse, j = uu - net * Hse;
                const float *w = se_w0T + net * K * Hse + j;
                float z = 0.0f;
                for (int k = 0; k < K; ++k) z = fma32(k < A ? (k == action ? 1.0f : 0.0f) : state[k - A], w[k * Hse], z);
                z = z + se_b0[uu];
                se_h[uu] = act_fwd(se_act_id, cfg.se_prelu, z);
            }
            __syncthreads();
            const float *se_hl = se_h;                     // the last hidden row of the three nets
            for (int l = 1; l < seL; ++l) {                // further hidden layers (build_nn_from_config): k-ascending chains over the arena copy
                const float *hin = (l & 1) ? se_h : se_h2;
                float *hout = (l & 1) ? se_h2 : se_h;
                for (int uu = tid; uu < 3 * Hse; uu += DNT) {
                    const int net = uu / Hse, j = uu - net * Hse;
                    const float *wm = arena + a.a_se_mid + ((int64_t)net * (seL - 1) + (l - 1)) * (Hse * Hse + Hse);
                    const float *w = wm + (int64_t)j * Hse, *hi = hin + net * Hse;
                    float z = 0.0f;
                    for (int k = 0; k < Hse; ++k) z = fma32(hi[k], w[k], z);
                    z = z + wm[Hse * Hse + j];
                    hout[uu] = se_ln ? z : act_fwd(se_act_id, cfg.se_prelu, z);
                }
                __syncthreads();
                if (se_ln) {
                    // the SE nets' shared nn.LayerNorm (model_utils.py:22-37): NES perturbs nn.Linear modules only (GTN_worker.py:156-175), so
                    // its weight / bias are the constructor's 1 / 0 for every worker; sequential row sums as in the oracle's mlp_forward_one_ex
                    if (tid < 3) {
                        const float *zr = hout + tid * Hse;
                        float sm = 0.0f, sv = 0.0f;
                        for (int jj = 0; jj < Hse; ++jj) sm = sm + zr[jj];
                        const float mean = sm / (float)Hse;
                        for (int jj = 0; jj < Hse; ++jj) { const float dj = zr[jj] - mean; sv = fma32(dj, dj, sv); }
                        se_ln_stat[2 * tid] = mean; se_ln_stat[2 * tid + 1] = 1.0f / __builtin_sqrtf(sv / (float)Hse + 1e-5f);
                    }
                    __syncthreads();
                    for (int uu = tid; uu < 3 * Hse; uu += DNT) {
                        const int net = uu / Hse;
                        hout[uu] = act_fwd(se_act_id, cfg.se_prelu, fma32((hout[uu] - se_ln_stat[2 * net]) * se_ln_stat[2 * net + 1], 1.0f, 0.0f));
                    }
                    __syncthreads();
                }
                se_hl = hout;
            }
            if (tid < S + 2) {
                const int net = tid < S ? 0 : (tid == S ? 1 : 2);
                const float *h = se_hl + net * Hse, *w = se_wout + tid * Hse;
                float acc = 0.0f;
                for (int j = 0; j < Hse; ++j) acc = fma32(h[j], w[j], acc);
                acc = acc + se_bout[tid];
                // ReplayBuffer.add (utils.py:24-32): row = [s, a, s', r, done]
                if (tid < S) newrow[S + 1 + tid] = acc;
                else if (tid == S) newrow[2 * S + 1] = r_ == 0 ? acc : newrow[2 * S + 1] + acc;
                else newrow[2 * S + 2] = acc;
            }
            if (r_ + 1 < k_rep) {                              // the next repeat starts from the state the SE just produced
                __syncthreads();
                if (tid < S) state[tid] = newrow[S + 1 + tid];
                __syncthreads();
            }
            }
            }
            __syncthreads();
            if (tid < 2 * S + 3) rb[(int64_t)new_pos * RS + tid] = newrow[tid];
            if (!FIXED && tid == 0 && a.out.trace_action && train_steps < a.out.trace_cap) {
                const int64_t k = chain * a.out.trace_cap + train_steps;
                a.out.trace_action[k] = action | (explored << 16);
                for (int i = 0; i < S; ++i) { a.out.trace_state[k * S + i] = newrow[i]; a.out.trace_next_state[k * S + i] = newrow[S + 1 + i]; }
                a.out.trace_reward_done[k * 2] = newrow[2 * S + 1]; a.out.trace_reward_done[k * 2 + 1] = newrow[2 * S + 2];
            }
            const float done_now = newrow[2 * S + 2];
            tr_reward = tr_reward + newrow[2 * S + 1];
            __syncthreads();
            if (tid < S) state[tid] = newrow[S + 1 + tid];
            ep_len += k_rep; ++train_steps;                  // base_agent.py:122: episode_length += same_action_num
            __syncthreads();
            PT_MARK(1);

            if (learning) {
                // ================= DuelingDDQN.learn (DuelingDDQN.py:59-94) =================
                // ReplayBuffer.sample: gather rows; keep a, r, done in LDS (dAdv/dq reuse below), states to xs / xs2
                for (int b = tid; b < B; b += DNT) {
                    const int64_t n = (int64_t)learn_it * B + b;
                    int idx;
                    if (tape) {
                        if (n >= a.tapes.replay_idx_stride) { status = -4; idx = 0; } else idx = a.tapes.replay_idx[chain * a.tapes.replay_idx_stride + n];
                        if (idx < 0 || idx >= size_after) { status = -6; idx = 0; }
                    } else idx = (int)rng_replay_below(key, (uint64_t)n, (uint32_t)size_after);
                    const float *row = rb + (int64_t)idx * RS;
                    for (int i = 0; i < S; ++i) { xs[b * S + i] = row[i]; xs2[b * S + i] = row[S + 1 + i]; }
                    dAdv[b * A + 0] = row[S];               // stash (a, r, done) in dAdv/dq until the TD step
                    dAdv[b * A + 1] = row[2 * S + 1];
                    dq[b] = row[2 * S + 2];
                }
                __syncthreads();
                icm_step();                                 // ICM agents: train the ICM, add the intrinsic rewards (DDQN.py:74-76)
                PT_MARK(2);
                queue_forward(online, xs2, B, hid_t, feat_t, v1_t, a1_t, 1);                 // next_q_values (online)
                queue_forward(target, xs2, B, hid_t, feat_t, v1_t, a1_t, 2);                 // next_q_values_target
                queue_forward(online, xs, B, hid_s, feat_s, v1_s, a1_s, 0, xh_s, rstd_s);    // q_values, activations kept
                gq.run<D_MAXI>(Ps, Qs);                                                       // 18 products, one call
                finish_q(1, B, qv + B * A, true);
                finish_q(2, B, qv + 2 * B * A, true);
                finish_q(0, B, qv, true);
                PT_MARK(3);
                // TD error (DuelingDDQN.py:80-85) and the gradient of the loss w.r.t. V / Adv
                for (int b = tid; b < B; b += DNT) {        // one thread per sample
                    const float g32 = (float)cfg.gamma, norm = (float)(2.0 / (double)B);
                    const int ab = (int)dAdv[b * A + 0];
                    const float r = dAdv[b * A + 1], d = dq[b];
                    int am = 0; float best = qv[(B + b) * A];
                    for (int aa = 1; aa < A; ++aa) { const float v = qv[(B + b) * A + aa]; if (v > best) { best = v; am = aa; } }
                    const float t1 = g32 * qv[(2 * B + b) * A + am];
                    const float t2 = 1.0f - d;
                    const float y = r + t1 * t2;
                    dq[b] = norm * (qv[b * A + ab] - y);
                    Vb[b] = (float)ab;                      // keep the action index
                }
                __syncthreads();
                if (tid == 0) {
                    float s_dq = 0.0f;                      // sequential in b (canonical order of the sum)
                    for (int b = 0; b < B; ++b) s_dq = s_dq + dq[b];
                    ctrl[9] = plain ? 0.0f : (-s_dq) / (float)(B * A);     // backward of `- advantages.mean()` (dueling only)
                    b1pow *= cfg.adam_beta1; b2pow *= cfg.adam_beta2;
                    ctrl[10] = (float)(-(lr / (1.0 - b1pow)));
                    ctrl[11] = (float)__builtin_sqrt(1.0 - b2pow);
                }
                __syncthreads();
                {
                    const float mean_grad = ctrl[9];
                    // dueling: dL/dAdv; plain DQN: dL/dQ (only entry a_b of a row is non-zero, no mean term)
                    for (int e = tid; e < B * A; e += DNT) {
                        const int b = e / A, aa = e - b * A;
                        const float g = aa == (int)Vb[b] ? dq[b] : 0.0f;
                        dAdv[e] = plain ? g : g + mean_grad;
                    }
                }
                __syncthreads();
                PT_MARK(4);
                float *d_a1 = arena + AV(a_dbuf[0]), *d_v1 = arena + AV(a_dbuf[1]), *d_feat = arena + AV(a_dbuf[2]);
                float *dh[2] = { arena + AV(a_dbuf[3]), arena + AV(a_dbuf[4]) };
                // ---- heads, output layers: db = column sums; d hidden of the heads = act'(h) * sum_o dOut[o] * W2[o][k]
                // (reduction over the few outputs).  Everything GEMM-shaped of the backward pass is queued below.
                if (!plain && tid < A) { float s = 0.0f; for (int b = 0; b < B; ++b) s = s + dAdv[b * A + tid]; grad[po.oba2 + tid] = s; }
                if (!plain && tid == A) { float s = 0.0f; for (int b = 0; b < B; ++b) s = s + dq[b]; grad[po.obv2] = s; }
                if (!plain) {
                    // thread = (column k, row group): the head weights of column k are loop invariants, a1/v1 reads are coalesced
                    // along k and 8 rows are in flight per thread (A <= 3 for the supported envs)
                    const int k = tid & 127, rg = tid >> 7, nrg = DNT >> 7;
                    if (k < F) {
                        float wa[3] = { 0.0f, 0.0f, 0.0f };
                        for (int aa = 0; aa < A; ++aa) wa[aa] = online[po.oWa2 + aa * F + k];
                        const float wv = online[po.oWv2 + k];
                        for (int b0 = rg; b0 < B; b0 += 8 * nrg) {
                            float ha[8], hv[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const int b = b0 + u * nrg;
                                ha[u] = b < B ? a1_s[b * F + k] : 0.0f;
                                hv[u] = b < B ? v1_s[b * F + k] : 0.0f;
                            }
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                const int b = b0 + u * nrg;
                                if (b < B) {
                                    float acc = 0.0f;
                                    for (int aa = 0; aa < A; ++aa) acc = fma32(dAdv[b * A + aa], wa[aa], acc);
                                    d_a1[b * F + k] = act_bwd(act_id, prelu, ha[u], acc);
                                    d_v1[b * F + k] = act_bwd(act_id, prelu, hv[u], fma32(dq[b], wv, 0.0f));
                                }
                            }
                        }
                    }
                }
                PT_MARK(5);
                if (!plain) {
                    // heads: dW2 = dOut^T . hidden (reduction over the batch); dW1 = dHid^T . feat, db1; dfeat = d_v1 . Wv1 + d_a1 . Wa1
                    gq.gemm(dAdv, 1, A, a1_s, 1, F, A, F, B, epi_store(grad + po.oWa2, F));
                    gq.gemm(dq, 1, 1, v1_s, 1, F, 1, F, B, epi_store(grad + po.oWv2, F));
                    gq.gemm(d_a1, 1, F, feat_s, 1, F, F, F, B, epi_store(grad + po.oWa1, F));
                    gq.gemm(d_v1, 1, F, feat_s, 1, F, F, F, B, epi_store(grad + po.oWv1, F));
                    gq.colsum(d_a1, B, F, F, grad + po.oba1);
                    gq.colsum(d_v1, B, F, F, grad + po.obv1);
                    gq.gemm(d_v1, F, 1, online + po.oWv1, 1, F, B, F, F, epi_store(d_feat, F));
                    gq.gemm(d_a1, F, 1, online + po.oWa1, 1, F, B, F, F, epi_accum(d_feat, F));
                }
                // feature stream: output Linear (no activation), then the hidden layers downwards.  Plain DQN: the stream's
                // output is Q itself, so its output gradient is the masked dq rows (dAdv, LDS) -- DDQN.py:82-94
                {
                    const float *dcur = plain ? dAdv : d_feat;            // dL/d(output of layer l+1)'s pre-activation
                    int n_out = F;
                    for (int l = L; l >= 0; --l) {
                        const int n_in = l == 0 ? S : H;
                        const float *inp = l == 0 ? xs : hid_s[l - 1];
                        gq.gemm(dcur, 1, n_out, inp, 1, n_in, n_out, n_in, B, epi_store(grad + po.oWf[l], n_in));
                        gq.colsum(dcur, B, n_out, n_out, grad + po.obf[l]);
                        if (l > 0) {
                            float *dn = dh[l & 1];
                            gq.gemm(dcur, n_out, 1, online + po.oWf[l], 1, n_in, B, n_in, n_out, epi_act_bwd(dn, n_in, hid_s[l - 1], n_in, act_id, prelu));
                            if (ln && l - 1 >= 1) {
                                // dn = gradient of the LayerNorm OUTPUT of hidden layer l-1: run the queue, then the row routine turns it
                                // into the gradient of the Linear's output and adds this position's share of the shared weight / bias
                                // gradient (positions from the top down, as the oracle folds them)
                                gq.run<D_MAXI>(Ps, Qs);
                                ln_rows_backward<LNBUF, D_MAXH>(Ps, dn, B, H, online + po.oLN, xh_s[l - 1], rstd_s + (int64_t)(l - 1) * CFG_B, grad + po.oLN,
                                                                grad + po.oLN + H, l - 1 == L - 1);
                            }
                            dcur = dn; n_out = n_in;
                        }
                    }
                }
                gq.run<D_MAXI>(Ps, Qs);
                PT_MARK(6);
                // ---- torch.optim.Adam + Polyak (DuelingDDQN.py:87-93) ----
                {
                    const float neg_step = ctrl[10], bc2_sqrt = ctrl[11];
                    const float w1 = (float)(1.0 - cfg.adam_beta1), w2 = (float)(1.0 - cfg.adam_beta2), beta2 = (float)cfg.adam_beta2;
                    const float adam_eps = (float)cfg.adam_eps, tau = (float)cfg.tau, omt = (float)(1.0 - cfg.tau);
                    const AdamConsts ac{ neg_step, bc2_sqrt, w1, w2, beta2, adam_eps };
                    wg_adam(online, adam_m, adam_v, grad, 0, P, ac, target, tau, omt);
                }
                ++learn_it;
                __syncthreads();
                PT_MARK(7);
            }
            if (done_now > 0.5f) break;
        }
        ++episodes_run;
        if (tid == 0 && a.out.episode_len) a.out.episode_len[chain * cfg.train_episodes + episode] = ep_len;
        __syncthreads();
        PT_MARK(9);
        if (!no_test_env) test_phase();                    // per-episode test on the real env (base_agent.py:134-136)
        PT_MARK(8);
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
            ictrl[3] = learning && meter_env_solved(meter, episode + 1, cfg.early_out_num, no_test_env && !reward_env, cfg.solved_reward,
                                                    cfg.early_out_virtual_diff, episode, cfg.init_episodes);
        }
        __syncthreads();
        const int brk = ictrl[3];
        __syncthreads();
        if (brk) break;
    }
    PT_MARK(9);
    const int64_t remaining = cfg.step_budget - ((int64_t)train_steps + test_steps);     // time_remaining - elapsed
    const int test_before = test_steps;
    test_phase();
    if (budgeted) {
        // BaseAgent.test under the time-out (base_agent.py:177-184): episode e starts only while the earlier episodes of this
        // test used <= remaining steps; the rest is padded with the minimum so far (-1e9 if empty).  Episodes are independent,
        // so the list rolled out above is cut here.
        if (tid == 0) {
            int64_t used = 0;
            int stop = T;
            for (int te = 0; te < T; ++te) {
                if (used > remaining) { stop = te; break; }
                used += tlen[te];
            }
            double mn = -1e9;
            if (stop > 0) { mn = ret[0]; for (int i = 1; i < stop; ++i) if (ret[i] < mn) mn = ret[i]; }
            for (int te = stop; te < T; ++te) ret[te] = mn;
            ictrl[4] = (int)used;
        }
        __syncthreads();
        test_steps = test_before + ictrl[4];
    }
    PT_MARK(8);
#ifdef LENV_PHASE_TIMING
    if (tid == 0 && chain == 0) for (int pi = 0; pi < 10; ++pi) g_duel_phase_cycles[pi] = pt_acc[pi];
#endif
    if (tid == 0) {
        double sm = 0.0;
        for (int i = 0; i < T; ++i) sm += ret[i];
        a.out.score[chain] = sm / (double)T;
        if (a.out.final_returns) for (int i = 0; i < T; ++i) a.out.final_returns[chain * T + i] = ret[i];
        if (a.out.stats) {
            a.out.stats[chain * 4 + 0] = episodes_run; a.out.stats[chain * 4 + 1] = train_steps;
            a.out.stats[chain * 4 + 2] = learn_it; a.out.stats[chain * 4 + 3] = test_steps;
        }
        // episodes that never ran: NaN / 0, or -- after a time-out -- time_is_up's padding (base_agent.py:33-44)
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
    if (a.out.final_online) for (int p = tid; p < P; p += DNT) a.out.final_online[chain * a.P + p] = online[p];
    if constexpr (ICM) {
        if (a.icm_final) for (int p = tid; p < icm.P; p += DNT) a.icm_final[chain * a.P_icm + p] = arena[a.a_icm[IB_P] + p];
    }
    if (a.out.status && status != 0) atomicMin(&a.out.status[chain], status);
    (void)lane; (void)wave;
}

}  // namespace lenv

using namespace lenv;

// Fresh agents for chains with their own network shapes: nn.Linear's default init U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for
// weight and bias (the bounds agents/nes_common.py:linear_init_bounds tabulates for ONE shape), drawn like lenv_nes_draw:
// agent_init[c][i] = (2u - 1) * bound(i), u = unit(rng(key_c, STREAM_AGENT_INIT, i)).
__global__ void dueling_agent_init_kernel(lenv_ddqn_cfg cfg, const int32_t *hp_hidden, const int32_t *hp_layers, const uint64_t *rng_keys,
                                          int64_t chains, int64_t row_stride, float *agent_init)
{
    const int64_t c = blockIdx.y;
    if (c >= chains) return;
    const bool plain = cfg.agent_kind == 0;
    const int S = cfg.state_dim, A = cfg.num_actions, F = plain ? A : cfg.feature_dim;
    const int H = hp_hidden ? hp_hidden[c] : cfg.q_hidden, L = hp_layers ? hp_layers[c] : cfg.q_layers;
    if (H < 1 || H > cfg.q_hidden || L < 1 || L > cfg.q_layers) return;        // the inner loop reports status -8 for this chain
    const bool ln = cfg.q_layer_norm != 0 && L >= 2;
    const DuelOffsets po = duel_param_offsets(S, A, H, F, L, plain, ln);
    const uint64_t key = rng_keys[c];
    // (float)(1/sqrt(fan_in)) in double like the host table (both IEEE-exact)
    const float bS = (float)(1.0 / __builtin_sqrt((double)S)), bH = (float)(1.0 / __builtin_sqrt((double)H)), bF = (float)(1.0 / __builtin_sqrt((double)F));
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < po.P; i += gridDim.x * blockDim.x) {
        if (ln && i >= po.oLN && i < po.oLN + 2 * H) { agent_init[c * row_stride + i] = i < po.oLN + H ? 1.0f : 0.0f; continue; }   // nn.LayerNorm: weight 1, bias 0
        const float bound = i < po.oWf[1] ? bS : (i < (plain ? po.P : po.oWv1) ? bH : bF);
        const float u = (float)u64_to_unit(rng_u64(key, STREAM_AGENT_INIT, (uint64_t)i));
        agent_init[c * row_stride + i] = (u * 2.0f - 1.0f) * bound;
    }
}

static int64_t d_mlp_params(int in, int H, int L, int out) { return (int64_t)in * H + H + (int64_t)(L - 1) * ((int64_t)H * H + H) + (int64_t)H * out + out; }

static int dueling_layout(const lenv_ddqn_cfg *cfg, DuelArgs &a, size_t *lds_bytes)
{
    const bool plain = cfg->agent_kind == 0;               // DDQN with a multi-layer / wide Critic_DQN (see the kernel)
    const int S = cfg->state_dim, A = cfg->num_actions, H = cfg->q_hidden, F = plain ? A : cfg->feature_dim, L = cfg->q_layers, B = cfg->batch_size;
    const int Hse = cfg->se_hidden, T = cfg->test_episodes, K = S + A;
    if (cfg->agent_kind != 1 && cfg->agent_kind != 0) return LENV_ERR_INVALID;
    // the agent's shared nn.PReLU slope is a TRAINED parameter in the reference (model.parameters() -> Adam); a fixed slope
    // would diverge silently, so agent-net PReLU is refused (SE / reward nets keep theirs: the reference never updates those)
    if (cfg->q_act == LENV_ACT_PRELU) return LENV_ERR_UNSUPPORTED;
    if (cfg->grad_chunk != 0 && cfg->grad_chunk < B) return LENV_ERR_UNSUPPORTED;   // batch gradient = one sequential chunk here
    if (L < 1 || L > D_MAXL || H < 1 || H > D_MAXH || F < 1 || F > D_MAXW || B < 1 || B > D_MAXB || T < 1 || T > D_MAXW || cfg->se_layers < 1 || cfg->se_layers > D_MAXL)
        return LENV_ERR_UNSUPPORTED;
    if (cfg->same_action_num < 0 || cfg->same_action_num > 64) return LENV_ERR_UNSUPPORTED;
    if (cfg->test_mode < 0 || cfg->test_mode > 1) return LENV_ERR_INVALID;
    if (!((cfg->env_id == LENV_ENV_CARTPOLE && S == 4 && A == 2) || (cfg->env_id == LENV_ENV_ACROBOT && S == 6 && A == 3) ||
          (cfg->env_id == LENV_ENV_MOUNTAINCAR && S == 2 && A == 3)))
        return LENV_ERR_UNSUPPORTED;
    const bool ln = cfg->q_layer_norm != 0 && L >= 2;      // use_layer_norm: one shared LayerNorm behind hidden Linear 2..L (none with one hidden layer)
    a.P = duel_param_offsets(S, A, H, F, L, plain, ln).P;
    a.se_net_size[0] = (int)d_mlp_params(K, Hse, cfg->se_layers, S);
    a.se_net_size[1] = a.se_net_size[2] = (int)d_mlp_params(K, Hse, cfg->se_layers, 1);
    a.P_se = a.se_net_size[0] + a.se_net_size[1] + a.se_net_size[2];
    if (cfg->synthetic_env_type == 1) {
        // RewardEnv over the real env: theta = the reward network (reward_env.py:29-46; a 1-input dummy for type 0); the real
        // CartPole / Acrobot step has no info vector, so the info types (3, 4, 7, 8, 101, 102) cannot be evaluated
        const int t = cfg->reward_env_type;
        if (!(t == 0 || t == 1 || t == 2 || t == 5 || t == 6)) return LENV_ERR_UNSUPPORTED;
        a.P_se = (int)d_mlp_params(t == 0 ? 1 : S, Hse, cfg->se_layers, 1);
    } else if (cfg->synthetic_env_type != 0) return LENV_ERR_INVALID;
    a.RS = (2 * S + 3 + 3) & ~3;
    int64_t cap = (int64_t)cfg->train_episodes * cfg->max_steps;
    if (cap > cfg->rb_size) cap = cfg->rb_size;
    a.rb_cap = cap < 1 ? 1 : cap;
    a.A = duel_arena(S, H, F, L, B, T, a.P, ln);
    int64_t off = a.A.end;
    auto take = [&](int64_t n) { int64_t r = off; off += (n + 3) & ~(int64_t)3; return r; };
    a.a_replay = take(a.rb_cap * a.RS);
    a.a_meter = take(2 * (int64_t)(cfg->train_episodes > 0 ? cfg->train_episodes : 1));
    // several hidden layers: the SEs' hidden-to-hidden blocks, or (RewardEnv mode) the whole reward net
    a.a_se_mid = take(cfg->se_layers > 1 ? (cfg->synthetic_env_type == 1 ? (int64_t)a.P_se : 3 * (int64_t)(cfg->se_layers - 1) * ((int64_t)Hse * Hse + Hse)) : 0);
    a.P_icm = 0;
    for (int i = 0; i < IB_COUNT; ++i) a.a_icm[i] = 0;
    if (cfg->icm_enabled) {
        const int Fi = cfg->icm_feature_dim, Hi = cfg->icm_hidden;
        if (Fi < 1 || Fi > D_MAXW || Hi < 1 || Hi > D_MAXW) return LENV_ERR_UNSUPPORTED;
        IcmNet n;
        icm_build(n, S, A, Fi, Hi, /*discrete=*/true);
        a.P_icm = n.P;
        int64_t sz[IB_COUNT];
        icm_buffer_sizes(n, B, sz);
        for (int i = 0; i < IB_COUNT; ++i) a.a_icm[i] = take(sz[i]);
    }
    a.arena_stride = (off + 63) & ~(int64_t)63;
    {   // the wave-chain kernel (dueling_wavechain.hip) keeps its own arena layout inside the same workspace
        const int wshape = lenv_wc_dueling_shape(cfg);
        if (wshape) {
            const int64_t wfl = lenv_wc_dueling_arena_floats(cfg, wshape, a.rb_cap, a.RS, a.P_se);
            if (wfl > a.arena_stride) a.arena_stride = wfl;
        }
    }
    const size_t lds_floats = GemmShape<D_MAXI>::PS_FLOATS + GemmShape<D_MAXI>::QS_FLOATS + GEMM_QUEUE_MAX * sizeof(GemmCmd) / sizeof(float) +
                              3 * K * Hse + 3 * Hse + (S + 2) * Hse + 16 + 3 * Hse + (cfg->se_layers > 1 ? 3 * Hse : 0) + 3 * (size_t)(B > T ? B : T) * A + 3 * (size_t)(B > T ? B : T) * (1 + A) +
                              B + (size_t)B * A + 64 + 2 * (4 * (size_t)T + T) + 2 * T + 8 + 16 + 16 + (size_t)T + 10;
    *lds_bytes = lds_floats * sizeof(float);
    if (*lds_bytes > 160 * 1024) return LENV_ERR_UNSUPPORTED;
    return LENV_OK;
}

extern "C" size_t lenv_dueling_se_workspace_bytes(const lenv_ddqn_cfg *cfg, int64_t chains)
{
    if (!cfg || chains < 0) return 0;
    DuelArgs a;
    size_t lds;
    if (dueling_layout(cfg, a, &lds) != LENV_OK) return 0;
    return (size_t)chains * a.arena_stride * sizeof(float) + 256;
}

extern "C" int lenv_dueling_team_size(const lenv_ddqn_cfg *cfg, int64_t chains)
{
    if (!cfg || chains < 1) return LENV_ERR_INVALID;
    if ((cfg->kernel_variant & (LENV_VARIANT_NO_WAVECHAIN | LENV_VARIANT_GENERIC)) || cfg->icm_enabled || cfg->rng_mode != LENV_RNG_COUNTER || cfg->synthetic_env_type != 0 ||
        cfg->same_action_num > 1 || !lenv_wc_dueling_shape(cfg))
        return 1;
    return lenv_wc_dueling_team(cfg, lenv_wc_dueling_shape(cfg), chains);
}

extern "C" int64_t lenv_dueling_num_params(const lenv_ddqn_cfg *cfg)
{
    if (!cfg) return LENV_ERR_INVALID;
    DuelArgs a;
    size_t lds;
    const int rc = dueling_layout(cfg, a, &lds);
    return rc != LENV_OK ? rc : a.P;
}

extern "C" int lenv_dueling_se_inner_loop(const lenv_ddqn_cfg *cfg, const float *theta, const float *eps, const int32_t *worker,
                                          const float *sign, const float *agent_init, const uint64_t *rng_keys,
                                          const lenv_tapes *tapes, int64_t chains, void *workspace, size_t workspace_bytes,
                                          const lenv_inner_out *out, void *stream)
{
    return lenv_dueling_se_inner_loop_hp(cfg, nullptr, theta, eps, worker, sign, agent_init, rng_keys, tapes, chains, workspace,
                                         workspace_bytes, out, stream);
}

extern "C" int64_t lenv_icm_num_params(const lenv_ddqn_cfg *cfg)
{
    if (!cfg || !cfg->icm_enabled) return LENV_ERR_INVALID;
    DuelArgs a;
    size_t lds;
    const int rc = dueling_layout(cfg, a, &lds);
    return rc != LENV_OK ? rc : a.P_icm;
}

extern "C" int lenv_dueling_se_inner_loop_hp(const lenv_ddqn_cfg *cfg, const lenv_chain_hp *hp, const float *theta, const float *eps,
                                             const int32_t *worker, const float *sign, const float *agent_init, const uint64_t *rng_keys,
                                             const lenv_tapes *tapes, int64_t chains, void *workspace, size_t workspace_bytes,
                                             const lenv_inner_out *out, void *stream)
{
    if (cfg && cfg->icm_enabled) return LENV_ERR_INVALID;             // an ICM agent needs lenv_dueling_se_inner_loop_icm
    return lenv_dueling_se_inner_loop_icm(cfg, hp, nullptr, theta, eps, worker, sign, agent_init, rng_keys, tapes, chains, workspace,
                                          workspace_bytes, out, stream);
}

extern "C" int lenv_dueling_se_inner_loop_icm(const lenv_ddqn_cfg *cfg, const lenv_chain_hp *hp, const lenv_icm_io *icm, const float *theta,
                                              const float *eps, const int32_t *worker, const float *sign, const float *agent_init,
                                              const uint64_t *rng_keys, const lenv_tapes *tapes, int64_t chains, void *workspace,
                                              size_t workspace_bytes, const lenv_inner_out *out, void *stream)
{
    if (!cfg || !theta || !agent_init || !out || !out->score || !workspace || chains < 0) return LENV_ERR_INVALID;
    if (cfg->icm_enabled && (!icm || !icm->icm_init)) return LENV_ERR_INVALID;
    if (hp && (!hp->lr || !hp->batch_size || !hp->q_hidden || !hp->q_layers)) return LENV_ERR_INVALID;
    if (eps && (!worker || !sign)) return LENV_ERR_INVALID;
    if (cfg->rng_mode == LENV_RNG_TAPE && !tapes) return LENV_ERR_INVALID;
    if (cfg->rng_mode == LENV_RNG_COUNTER && !rng_keys) return LENV_ERR_INVALID;
    if (chains == 0) return LENV_OK;
    DuelArgs a;
    size_t lds_bytes;
    const int rc = dueling_layout(cfg, a, &lds_bytes);
    if (rc != LENV_OK) return rc;
    if (workspace_bytes < (size_t)chains * a.arena_stride * sizeof(float)) return LENV_ERR_WORKSPACE;
    a.cfg = *cfg;
    a.theta = theta; a.eps = eps; a.worker = worker; a.sign = sign; a.agent_init = agent_init; a.rng_keys = rng_keys;
    if (tapes) a.tapes = *tapes; else a.tapes = lenv_tapes{};
    a.arena = static_cast<float *>(workspace);
    a.out = *out;
    a.hp_lr = hp ? hp->lr : nullptr; a.hp_batch = hp ? hp->batch_size : nullptr;
    a.hp_hidden = hp ? hp->q_hidden : nullptr; a.hp_layers = hp ? hp->q_layers : nullptr;
    a.icm_init = cfg->icm_enabled ? icm->icm_init : nullptr; a.icm_final = cfg->icm_enabled ? icm->icm_final : nullptr;
    void (*kern)(const DuelArgs) = cfg->icm_enabled ? dueling_se_inner_kernel<true> : dueling_se_inner_kernel<false>;
    {
        // the published Acrobot DuelingDDQN shape in production form takes the shape-specialised instantiation
        const bool off = (cfg->kernel_variant & LENV_VARIANT_GENERIC) != 0;
        auto matches = [&](const DuelShape &sp) {
            return cfg->agent_kind == sp.kind && cfg->env_id == sp.env && cfg->state_dim == sp.S && cfg->num_actions == sp.A &&
                   (sp.kind == 0 || cfg->feature_dim == sp.F) && cfg->q_hidden == sp.H && cfg->q_layers == sp.L && cfg->batch_size == sp.B &&
                   cfg->se_hidden == sp.Hse && cfg->test_episodes == sp.T && cfg->q_act == sp.q_act && cfg->se_act == sp.se_act && cfg->same_action_num <= 1 &&
                   !cfg->q_layer_norm && cfg->se_layers == 1 && cfg->test_mode == 0;     // (the FIXED builds hard-code a one-hidden-layer SE without LayerNorm)
        };
        // production launches of a wave-chain shape (dueling_wavechain.hip): kernel_variant NO_WAVECHAIN keeps the GEMM-queue kernel (A/B runs)
        const bool no_wc = (cfg->kernel_variant & LENV_VARIANT_NO_WAVECHAIN) != 0;
        if (!off && !no_wc && !cfg->icm_enabled && !hp && cfg->rng_mode == LENV_RNG_COUNTER && !out->trace_action && cfg->synthetic_env_type == 0 &&
            cfg->same_action_num <= 1) {
            const int wshape = lenv_wc_dueling_shape(cfg);
            if (wshape) {
                return lenv_wc_dueling_launch(wshape, cfg, theta, eps, worker, sign, agent_init, rng_keys, chains, a.arena, a.arena_stride, a.rb_cap,
                                              a.RS, a.P, a.P_se, a.se_net_size, out, static_cast<hipStream_t>(stream));
            }
        }
        if (!off && !cfg->icm_enabled && !hp && cfg->rng_mode == LENV_RNG_COUNTER && !out->trace_action && cfg->synthetic_env_type == 0) {
            if (matches(kDuelShapes[1])) kern = dueling_se_inner_kernel<false, 1>;
            else if (matches(kDuelShapes[2])) kern = dueling_se_inner_kernel<false, 2>;
        }
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return LENV_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)chains), dim3(DNT), lds_bytes, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}

extern "C" int lenv_dueling_agent_init_hp(const lenv_ddqn_cfg *cfg, const lenv_chain_hp *hp, const uint64_t *rng_keys, int64_t chains,
                                          float *agent_init, void *stream)
{
    if (!cfg || !rng_keys || !agent_init || chains < 0) return LENV_ERR_INVALID;
    if (hp && (!hp->q_hidden || !hp->q_layers)) return LENV_ERR_INVALID;
    if (chains == 0) return LENV_OK;
    DuelArgs a;
    size_t lds_bytes;
    const int rc = dueling_layout(cfg, a, &lds_bytes);
    if (rc != LENV_OK) return rc;
    hipLaunchKernelGGL(dueling_agent_init_kernel, dim3(64, (unsigned)chains), dim3(256), 0, static_cast<hipStream_t>(stream), *cfg,
                       hp ? hp->q_hidden : nullptr, hp ? hp->q_layers : nullptr, rng_keys, chains, (int64_t)a.P, agent_init);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}

#ifdef LENV_PHASE_TIMING
extern "C" int lenv_debug_duel_phase_cycles(unsigned long long *host_out)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lenv::g_duel_phase_cycles), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -4;
}
#endif
