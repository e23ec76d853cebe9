// dueling_wavechain.hip -- the DuelingDDQN inner loop of BASELINE configs[2] (Acrobot-v1 SE + DuelingDDQN, default_config_acrobot.yaml)
// rebuilt on the wave-chain primitives of lenv_wavechain.cuh.  Same semantics, same canonical arithmetic order and therefore the
// same bits as dueling_se_inner_kernel (which stays the generic path: other shapes, tapes, traces, *_vary, ICM, RewardEnv), but a
// different execution structure:
//
//   reference                                   here
//   DuelingDDQN.learn  agents/DuelingDDQN.py:59-94     learn_step(): target pass (4 sample blocks) + online pass (8 blocks: s and s'
//   Critic_DuelingDQN  models/actor_critic.py:94-122     share every staged weight image), activations chained in registers
//   loss.backward()                                     backward(): per layer one resident transposed weight image for the input
//                                                         gradients (4 waves, one per SIMD) and two [sample][unit] images for the
//                                                         weight gradients (8 waves)
//   optimizer.step + Polyak  :87-93                     wg_adam over the arena-layout parameter vector
//   select_train_action / BaseAgent.test greedy rows    forward_thin(): A operand straight from the K-major arena arrays
//
// Arena layout of a parameter vector (online, target, Adam m / v, grad all alike; every 128x128 matrix K-MAJOR, Wt[k][unit]):
//   W1t[8][128] (rows >= S are zero) b1 | W2t b2 | W3t b3 | Wv1t bv1 | Wa1t ba1 | Wh[128][4] = (wv2, wa2_0..2)[k] | bh[4]
#include "lenv_wavechain.cuh"
// Team launches of the DuelingDDQN shapes run torch's Adam + the Polyak update as the epilogue of the weight-gradient routines (round 4:
// -1.6 % at the configs[2] shard); -DWCT_NO_FUSED_OPT keeps the separate optimizer pass for A/B timing.  The plain-DQN shape keeps the pass.
#ifndef WCT_NO_FUSED_OPT
#define WCT_FUSED_OPT
#endif
#include "lenv_wavechain_host.h"

namespace lenv {

using namespace wc;

// fixed: H = F = 128, L = 2, B = 128; kind 1 = DuelingDDQN (Critic_DuelingDQN), kind 0 = DDQN whose Critic_DQN is S-128-128-A (the plain-DQN
// mode of lenv_dueling_se_inner_loop: models/actor_critic.py:84-91, agents/DDQN.py:60-94): layers 1 and 2, then the A-column output layer
// where the dueling net has its advantage head -- no feature / stream layers, no value head, no advantage mean.  The arena keeps the
// dueling layout (the unused matrices stay zero and are skipped by the optimizer pass); teams of two as the dueling shape (the chain
// round stores d_h2, member 0 takes the output layer's and layer 1's gradients, member 1 W2's).
struct WcShape { int env, S, A, Hse, T, q_act, se_act, kind; };
constexpr WcShape kWcShapes[] = {
    { -1, 2, 2, 1, 1, 0, 0, 1 },
    { LENV_ENV_ACROBOT, 6, 3, 128, 10, LENV_ACT_RELU, LENV_ACT_LEAKYRELU, 1 },     // default_config_acrobot.yaml duelingddqn = BASELINE configs[2]
    { LENV_ENV_ACROBOT, 6, 3, 128, 10, LENV_ACT_RELU, LENV_ACT_LEAKYRELU, 0 },     // default_config_acrobot.yaml ddqn (6-128-128-3 relu, B = 128)
};
constexpr int WC_B = 128, WC_H = 128;

namespace wcp {     // arena-layout parameter offsets (floats)
constexpr int oW1t = 0, ob1 = 8 * W, oW2t = ob1 + W, ob2 = oW2t + IMG, oW3t = ob2 + W, ob3 = oW3t + IMG, oWv1t = ob3 + W, obv1 = oWv1t + IMG,
              oWa1t = obv1 + W, oba1 = oWa1t + IMG, oWh = oba1 + W, obh = oWh + 4 * W, PW = obh + 4;
static_assert(PW % 4 == 0, "float4 passes");
}

enum { D_H1 = 0, D_H2, D_FEAT, D_V1, D_A1, S_F1, S_DFEAT, S_DH2, S_DH1,
       R_V1, R_A1,          // v1 / a1 once more as plain [sample][unit] arrays: the head weight gradients walk them unit-wise
       WC_NDUMP };

struct WcArgs {
    lenv_ddqn_cfg cfg;
    const float *theta, *eps; const int32_t *worker; const float *sign;
    const float *agent_init; const uint64_t *rng_keys;
    float *arena; int64_t arena_stride;
    lenv_inner_out out;
    int64_t rb_cap; int RS;
    int P, P_se, se_net_size[3];
    int64_t a_par, a_xs, a_xs2, a_dump, a_se, a_replay, a_meter;      // arena offsets (floats)
    int64_t a_gx, a_bar, a_xtm;                                       // team exchange (V | Adv of the three passes), barrier word, per-member scratch rows
    int G;                                                            // workgroups per chain: 1 or 2
    int64_t chains;
};

// state-dict index (duel_param_offsets order) -> arena-layout index
__device__ __forceinline__ int wc_sd_to_arena(int p, int S, int A)
{
    using namespace wcp;
    int o = p;
    if (o < WC_H * S) { const int j = o / S, k = o - j * S; return oW1t + k * W + j; }
    o -= WC_H * S;
    if (o < W) return ob1 + o;
    o -= W;
    const int mats[4] = { oW2t, oW3t, oWv1t, -1 };
    const int bias[4] = { ob2, ob3, obv1, -1 };
    for (int l = 0; l < 3; ++l) {
        if (o < IMG) { const int j = o >> 7, k = o & 127; return mats[l] + k * W + j; }
        o -= IMG;
        if (o < W) return bias[l] + o;
        o -= W;
    }
    if (o < W) return oWh + o * 4;                       // Wv2[0][k]
    o -= W;
    if (o < 1) return obh;
    o -= 1;
    if (o < IMG) { const int j = o >> 7, k = o & 127; return oWa1t + k * W + j; }
    o -= IMG;
    if (o < W) return oba1 + o;
    o -= W;
    if (o < A * W) { const int aa = o >> 7, k = o & 127; return oWh + k * 4 + 1 + aa; }
    o -= A * W;
    return obh + 1 + o;
}


// the same for a Critic_DQN S-128-128-A: net.0.weight [128][S], net.0.bias, net.2.weight [128][128], net.2.bias, net.4.weight [A][128], net.4.bias
__device__ __forceinline__ int wc_sd_to_arena_plain(int p, int S, int A)
{
    using namespace wcp;
    int o = p;
    if (o < WC_H * S) { const int j = o / S, k = o - j * S; return oW1t + k * W + j; }
    o -= WC_H * S;
    if (o < W) return ob1 + o;
    o -= W;
    if (o < IMG) { const int j = o >> 7, k = o & 127; return oW2t + k * W + j; }
    o -= IMG;
    if (o < W) return ob2 + o;
    o -= W;
    if (o < A * W) { const int aa = o >> 7, k = o & 127; return oWh + k * 4 + 1 + aa; }
    o -= A * W;
    return obh + 1 + o;
}

// Everything the phase routines need, written once into LDS by thread 0.  The phases are OUT-OF-LINE functions on purpose: inlined into
// the one big kernel body their per-lane address sets (swizzled image positions, staging slots) are loop invariants of the episode
// loops, get hoisted to the kernel prologue and spilled (2.4 KB of scratch per lane in the first build); behind a call each phase
// computes them where it uses them and the register file belongs to the MFMA chains.
struct WcCtx {
    float *bufA, *bufB, *sm_w1t, *sm_bias, *sm_wh, *sm_bh, *qv, *Vb, *Advb, *dq, *dAdv;
    float *online, *target, *grad, *xs, *xs2, *dumps, *adam_m, *adam_v;
    volatile float *ctrl;
    float prelu;
    float w1, w2, beta2, adam_eps, tau, omt;            // Adam / Polyak constants (torch single-tensor Adam, DuelingDDQN.py:87-93)
    int g, G;                                           // this workgroup's place in its chain's team
    float *gva;                                         // team exchange: Vb [3][B] | Advb [3][B][A] (arena)
    double *dstate;                                     // the lock-step test episodes' state (wc_test_steps): [T][4], returns, flags, lengths
    float *ep_rew;
    int *alive, *tlen;
    double *ret;
    int *test_steps;                                    // out: environment steps of the phase
    int max_steps;
};

template <class T> __device__ __forceinline__ T *lds_uni_ptr(T *const *field) { return uni_ptr(*field); }

__device__ __forceinline__ float *wc_dump(float *dumps, int which, int blk) { return dumps + ((int64_t)which * 4 + blk) * BLK; }

#ifdef LENV_PHASE_TIMING
__device__ unsigned long long g_wc_phase_cycles[64];      // [0,16): kernel phases; [16,64): sub-phases inside the out-of-line routines (chain 0)
#define WSUB_DECL unsigned long long sp_last = __builtin_readcyclecounter(), sp8_last = sp_last
#define WSUB_MARK(i) do { unsigned long long sp_now = __builtin_readcyclecounter(); if (blockIdx.x == 0 && threadIdx.x == 0) g_wc_phase_cycles[i] += sp_now - sp_last; sp_last = sp_now; } while (0)
#define WSUB_MARK4(i) do { unsigned long long sp_now = __builtin_readcyclecounter(); if (blockIdx.x == 0 && threadIdx.x == 256) g_wc_phase_cycles[i] += sp_now - sp_last; sp_last = sp_now; } while (0)
#define WSUB_MARK8(i) do { unsigned long long sp_now = __builtin_readcyclecounter(); if (blockIdx.x == 8 && threadIdx.x == 0) g_wc_phase_cycles[i] += sp_now - sp8_last; sp8_last = sp_now; } while (0)   /* member 1 of chain 0 at G = 2 */
#define WPT_DECL unsigned long long pt_last = __builtin_readcyclecounter(), pt_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define WPT_MARK(i) do { unsigned long long pt_now = __builtin_readcyclecounter(); pt_acc[i] += pt_now - pt_last; pt_last = pt_now; } while (0)
// a team barrier of the learn step with its waiting time: slot i for member 0 of chain 0 (block 0), i - 48 for member 1 (block 8)
#define WBAR(i) do { const unsigned long long b0_ = __builtin_readcyclecounter(); team_barrier(); const unsigned long long b1_ = __builtin_readcyclecounter(); \
                     if (threadIdx.x == 0 && blockIdx.x == 0) g_wc_phase_cycles[i] += b1_ - b0_; if (threadIdx.x == 0 && blockIdx.x == 8) g_wc_phase_cycles[(i) - 48] += b1_ - b0_; } while (0)
#else
#define WBAR(i) team_barrier()
#define WPT_DECL
#define WPT_MARK(i)
#define WSUB_DECL
#define WSUB_MARK(i)
#define WSUB_MARK4(i)
#define WSUB_MARK8(i)
#endif


// ---- thin forward: I <= 32 rows X[I][S] through the ONLINE net, head outputs to slot 0, q to qv (per-row advantage mean) ----
template <int SHAPE> static __device__ __noinline__ void wc_forward_thin_layers(const WcCtx *ctx_, const float *X_, int I_)
{
    using namespace wcp;
    constexpr WcShape SP = kWcShapes[SHAPE];
    constexpr int S = SP.S, A = SP.A, T = SP.T, B = WC_B, ACT = SP.q_act, RBH = B > T ? B : T;
    (void)S; (void)A; (void)T; (void)B; (void)ACT; (void)RBH;
    Lane L;
    L.init();
    const int tid = L.tid, wave = L.wave;
    (void)tid; (void)wave;
    typedef __attribute__((address_space(3))) const WcCtx LCtx;
    LCtx *c = (LCtx *)uni_ptr(ctx_);
    float *bufA = uni_ptr(c->bufA), *bufB = uni_ptr(c->bufB), *sm_w1t = uni_ptr(c->sm_w1t), *sm_bias = uni_ptr(c->sm_bias), *sm_wh = uni_ptr(c->sm_wh),
          *sm_bh = uni_ptr(c->sm_bh), *qv = uni_ptr(c->qv), *Vb = uni_ptr(c->Vb), *Advb = uni_ptr(c->Advb), *dq = uni_ptr(c->dq), *dAdv = uni_ptr(c->dAdv);
    float *online = uni_ptr(c->online), *target = uni_ptr(c->target), *grad = uni_ptr(c->grad), *xs = uni_ptr(c->xs), *xs2 = uni_ptr(c->xs2),
          *dumps = uni_ptr(c->dumps), *adam_m = uni_ptr(c->adam_m), *adam_v = uni_ptr(c->adam_v);
    const float prelu = unif(c->prelu);
    (void)adam_m; (void)adam_v;
    (void)bufA; (void)bufB; (void)sm_w1t; (void)sm_bias; (void)sm_wh; (void)sm_bh; (void)qv; (void)Vb; (void)Advb; (void)dq; (void)dAdv;
    (void)online; (void)target; (void)grad; (void)xs; (void)xs2; (void)dumps; (void)prelu;
    // LDS arrays through address-space-3 pointers: a generic pointer makes every access a FLAT instruction, which the hardware
    // completes out of order with respect to both counters -- each one then waits for every global load in flight
    lfloat *qv_l = (lfloat *)qv, *Vb_l = (lfloat *)Vb, *Advb_l = (lfloat *)Advb, *dq_l = (lfloat *)dq, *dAdv_l = (lfloat *)dAdv, *sm_wh_l = (lfloat *)sm_wh,
           *sm_bh_l = (lfloat *)sm_bh;
    (void)qv_l; (void)Vb_l; (void)Advb_l; (void)dq_l; (void)dAdv_l; (void)sm_wh_l; (void)sm_bh_l;
    const float *X = uni_ptr(X_);
    const int I = uni(I_);
    float *imgX = bufA;                                  // thin-product activation images [unit][16 or 32] live in bufA
    WSUB_DECL;
    constexpr int IW = T <= 16 ? 16 : 32;               // samples per image row (the 16x16x4 tiles take up to 16)
    constexpr bool PL = SP.kind == 0;                    // plain DQN: layers 1, 2 and the output layer
    static_assert(!PL || IW == 16, "plain shapes: the 16-sample path");
    float *imgY = bufA + IW * W, *imgZ = bufA + 2 * IW * W;
    float a2[32], a3[32], av[32], aa[32];                // this wave's weight tiles of the four 128x128 layers
    if constexpr (IW == 16) {
        thin_load16(online + oW2t, wave, L, a2);
        if constexpr (!PL) {
            thin_load16(online + oW3t, wave, L, a3);
            thin_load16(online + oWv1t, wave, L, av); thin_load16(online + oWa1t, wave, L, aa);
        }
    }
    {   // layer 1 (K = S): one thread per (unit, sample); the head output layer's weights go to LDS alongside
        const int j = tid & (W - 1);
        float w[S];
#pragma unroll
        for (int k = 0; k < S; ++k) w[k] = online[oW1t + k * W + j];
        const float bj = online[ob1 + j];
        for (int i = tid >> 7; i < I; i += NT >> 7) {
            float z = 0.0f;
#pragma unroll
            for (int k = 0; k < S; ++k) z = fma32(X[i * S + k], w[k], z);
            ((lfloat *)imgX)[j * IW + i] = act_fwd(ACT, prelu, z + bj);
        }
        for (int e = tid; e < 4 * W + 4; e += NT) sm_wh_l[e] = online[oWh + e];
    }
    __syncthreads();
    WSUB_MARK(24);
    if constexpr (PL) {
        thin_layer16<ACT, 1>(a2, online + ob2, imgY, a2, nullptr, nullptr, imgX, wave, L, prelu);
        imgZ = imgY;                                     // the output layer reads h2
    } else if constexpr (IW == 16) {
        thin_layer16<ACT, 1>(a2, online + ob2, imgY, a2, nullptr, nullptr, imgX, wave, L, prelu);
        __syncthreads();
        WSUB_MARK(25);
        thin_layer16<LENV_ACT_IDENTITY, 1>(a3, online + ob3, imgX, a3, nullptr, nullptr, imgY, wave, L, prelu);
        __syncthreads();
        WSUB_MARK(26);
        thin_layer16<ACT, 2>(av, online + obv1, imgY, aa, online + oba1, imgZ, imgX, wave, L, prelu);
    } else {
        if (wave < 4) thin_layer<ACT>(online + oW2t, online + ob2, imgX, imgY, wave, L, prelu);
        __syncthreads();
        WSUB_MARK(25);
        if (wave < 4) thin_layer<LENV_ACT_IDENTITY>(online + oW3t, online + ob3, imgY, imgX, wave, L, prelu);
        __syncthreads();
        WSUB_MARK(26);
        if (wave < 4) thin_layer<ACT>(online + oWv1t, online + obv1, imgX, imgY, wave, L, prelu);
        else thin_layer<ACT>(online + oWa1t, online + oba1, imgX, imgZ, wave - 4, L, prelu);
    }
    __syncthreads();
    WSUB_MARK(27);
    if (tid < 4 * I) {
        const int i = tid >> 2, o = tid & 3;
        if (o <= A && !(PL && o == 0)) {
            const lfloat *img = (const lfloat *)(o == 0 ? imgY : imgZ) + i;
            const lfloat *wh = (const lfloat *)sm_wh + o;
            float acc = 0.0f;
#pragma unroll 16
            for (int k = 0; k < W; ++k) acc = fma32(img[k * IW], wh[k * 4], acc);
            acc = acc + sm_bh_l[o];
            if (o == 0) Vb_l[i] = acc; else Advb_l[i * A + (o - 1)] = acc;
        }
    }
    __syncthreads();
    WSUB_MARK(28);
}

// ---- the steps of one test phase (T <= 16 episodes in lock-step, BaseAgent.test agents/base_agent.py:155-227) as ONE routine: the
// online net does not change during a test phase, so each wave keeps its tiles of the four 128x128 layers in registers (128 VGPRs) for
// all max_steps forwards instead of fetching 256 KB per forward -- the same chains on the same operands as wc_forward_thin_layers.
// Lane e < T owns episode e: it writes the observation rows (LDS), picks the greedy action from its row's head outputs and steps the
// real environment.  Reset draws, returns and the step count included: the call site in the kernel body stays straight-line code (a
// live-range split copy that the register allocator placed in the divergent reset block in front of the call -- before the block's
// exec restore -- lost the kernel's zero register in the inactive lanes: rocgdb, first Adam pass after the first test phase).
template <int SHAPE> static __device__ __noinline__ void wc_test_steps(const WcCtx *ctx_, uint32_t key_lo_, uint32_t key_hi_, int first_episode_)
{
    using namespace wcp;
    constexpr WcShape SP = kWcShapes[SHAPE];
    constexpr int S = SP.S, A = SP.A, T = SP.T, ACT = SP.q_act, env_id = SP.env;
    static_assert(T <= 16, "one 16-sample tile");
    Lane L;
    L.init();
    const int tid = L.tid, wave = L.wave;
    typedef __attribute__((address_space(3))) const WcCtx LCtx;
    LCtx *c = (LCtx *)uni_ptr(ctx_);
    float *bufA = uni_ptr(c->bufA);
    lfloat *sm_wh_l = (lfloat *)uni_ptr(c->sm_wh), *sm_bh_l = (lfloat *)uni_ptr(c->sm_bh), *Vb_l = (lfloat *)uni_ptr(c->Vb), *Advb_l = (lfloat *)uni_ptr(c->Advb);
    const float *online = uni_ptr(c->online);
    const float prelu = unif(c->prelu);
    typedef __attribute__((address_space(3))) double ldouble;
    typedef __attribute__((address_space(3))) int lint;
    ldouble *dstate = (ldouble *)uni_ptr(c->dstate);
    lfloat *ep_rew = (lfloat *)uni_ptr(c->ep_rew);
    lint *alive = (lint *)uni_ptr(c->alive), *tlen = (lint *)uni_ptr(c->tlen);
    const int max_steps = uni(c->max_steps);
    ldouble *ret = (ldouble *)uni_ptr(c->ret);
    const uint64_t key = ((uint64_t)uni((int)key_hi_) << 32) | (uint32_t)uni((int)key_lo_);
    const int first_episode = uni(first_episode_);
    if (tid < T) {
        double st[4];
        real_env_reset_draw(env_id, key, STREAM_TEST_RESET, (int64_t)first_episode + tid, st);
#pragma unroll
        for (int i = 0; i < 4; ++i) dstate[tid * 4 + i] = st[i];
        ep_rew[tid] = 0.0f; alive[tid] = 1; tlen[tid] = 0;
    }
    constexpr int IW = 16;
    float *imgX = bufA, *imgY = bufA + IW * W, *imgZ = bufA + 2 * IW * W;
    lfloat *Xl = (lfloat *)(bufA + 3 * IW * W);          // observation rows [T][S]
    constexpr bool PL = SP.kind == 0;                    // plain DQN: layers 1, 2, output layer on h2
    float a2[32], a3[32], av[32], aa[32];
    thin_load16(online + oW2t, wave, L, a2);
    if constexpr (!PL) {
        thin_load16(online + oW3t, wave, L, a3);
        thin_load16(online + oWv1t, wave, L, av); thin_load16(online + oWa1t, wave, L, aa);
    } else imgZ = imgY;
    const int j = tid & (W - 1);
    float w[S];
#pragma unroll
    for (int k = 0; k < S; ++k) w[k] = online[oW1t + k * W + j];
    const float bj = online[ob1 + j];
    const f32x4 bv2 = thin_bias16(online + ob2, wave, L), bv3 = PL ? bv2 : thin_bias16(online + ob3, wave, L),
                bvv = PL ? bv2 : thin_bias16(online + obv1, wave, L), bva = PL ? bv2 : thin_bias16(online + oba1, wave, L);
    for (int e = tid; e < 4 * W + 4; e += NT) sm_wh_l[e] = online[oWh + e];
    auto put_obs = [&]() {
        double st[4] = { dstate[tid * 4], dstate[tid * 4 + 1], dstate[tid * 4 + 2], dstate[tid * 4 + 3] };
        float obs[8];
        real_env_obs(env_id, st, obs);
#pragma unroll
        for (int i = 0; i < S; ++i) Xl[tid * S + i] = obs[i];
    };
    if (tid < T) put_obs();
    __syncthreads();
    for (int t = 0; t < max_steps; ++t) {
        for (int i = tid >> 7; i < T; i += NT >> 7) {   // layer 1 (K = S): one thread per (unit, sample)
            float z = 0.0f;
#pragma unroll
            for (int k = 0; k < S; ++k) z = fma32(Xl[i * S + k], w[k], z);
            ((lfloat *)imgX)[j * IW + i] = act_fwd(ACT, prelu, z + bj);
        }
        __syncthreads();
        thin_layer16v<ACT, 1>(a2, bv2, imgY, a2, bv2, nullptr, imgX, wave, L, prelu);
        __syncthreads();
        if constexpr (!PL) {
            thin_layer16v<LENV_ACT_IDENTITY, 1>(a3, bv3, imgX, a3, bv3, nullptr, imgY, wave, L, prelu);
            __syncthreads();
            thin_layer16v<ACT, 2>(av, bvv, imgY, aa, bva, imgZ, imgX, wave, L, prelu);
            __syncthreads();
        }
        if (tid < 4 * T) {
            const int i = tid >> 2, o = tid & 3;
            if (o <= A && !(PL && o == 0)) {
                const lfloat *img = (const lfloat *)(o == 0 ? imgY : imgZ) + i;
                const lfloat *wh = (const lfloat *)sm_wh_l + o;
                float acc = 0.0f;
#pragma unroll 16
                for (int k = 0; k < W; ++k) acc = fma32(img[k * IW], wh[k * 4], acc);
                acc = acc + sm_bh_l[o];
                if (o == 0) Vb_l[i] = acc; else Advb_l[i * A + (o - 1)] = acc;
            }
        }
        __syncthreads();
        if (tid < T && alive[tid]) {                      // q = V + (Adv - mean Adv) (models/actor_critic.py:117-122; plain DQN: q = the output layer), greedy action, env.step
            int am = 0;
            if constexpr (PL) {
                float best = Advb_l[tid * A];
                for (int b = 1; b < A; ++b) { const float v = Advb_l[tid * A + b]; if (v > best) { best = v; am = b; } }
            } else {
                float sum = 0.0f;
                for (int b = 0; b < A; ++b) sum = sum + Advb_l[tid * A + b];
                const float mean = sum / (float)A;
                float best = Vb_l[tid] + (Advb_l[tid * A] - mean);
                for (int b = 1; b < A; ++b) { const float v = Vb_l[tid] + (Advb_l[tid * A + b] - mean); if (v > best) { best = v; am = b; } }
            }
            double st[4] = { dstate[tid * 4], dstate[tid * 4 + 1], dstate[tid * 4 + 2], dstate[tid * 4 + 3] };
            double rew; int dn;
            real_env_step(env_id, st, am, rew, dn);
#pragma unroll
            for (int i = 0; i < 4; ++i) dstate[tid * 4 + i] = st[i];
            ep_rew[tid] = ep_rew[tid] + (float)rew;
            tlen[tid] = tlen[tid] + 1;
            if (dn) alive[tid] = 0;
            put_obs();
        }
        __syncthreads();
        int any = 0;
        for (int e = 0; e < T; ++e) any |= alive[e];
        if (!any) break;
    }
    if (tid < T) ret[tid] = (double)ep_rew[tid];
    if (tid == 0) {
        int n = 0;
        for (int e = 0; e < T; ++e) n += tlen[e];
        *(lint *)uni_ptr(c->test_steps) = n;
    }
    __syncthreads();
}

// ---- one pass of Critic_DuelingDQN over sample blocks: pass 0 = target net on s' (waves 0-3 -> slot 2), pass 1 = online net on
// s (waves 0-3 -> slot 0, activations dumped for the backward pass) and on s' (waves 4-7 -> slot 1) ----
template <int SHAPE> static __device__ __noinline__ void wc_forward_big(const WcCtx *ctx_, int pass_)
{
    using namespace wcp;
    constexpr WcShape SP = kWcShapes[SHAPE];
    constexpr int S = SP.S, A = SP.A, T = SP.T, B = WC_B, ACT = SP.q_act, RBH = B > T ? B : T;
    (void)S; (void)A; (void)T; (void)B; (void)ACT; (void)RBH;
    Lane L;
    L.init();
    const int tid = L.tid, wave = L.wave;
    (void)tid; (void)wave;
    typedef __attribute__((address_space(3))) const WcCtx LCtx;
    LCtx *c = (LCtx *)uni_ptr(ctx_);
    float *bufA = uni_ptr(c->bufA), *bufB = uni_ptr(c->bufB), *sm_w1t = uni_ptr(c->sm_w1t), *sm_bias = uni_ptr(c->sm_bias), *sm_wh = uni_ptr(c->sm_wh),
          *sm_bh = uni_ptr(c->sm_bh), *qv = uni_ptr(c->qv), *Vb = uni_ptr(c->Vb), *Advb = uni_ptr(c->Advb), *dq = uni_ptr(c->dq), *dAdv = uni_ptr(c->dAdv);
    float *online = uni_ptr(c->online), *target = uni_ptr(c->target), *grad = uni_ptr(c->grad), *xs = uni_ptr(c->xs), *xs2 = uni_ptr(c->xs2),
          *dumps = uni_ptr(c->dumps), *adam_m = uni_ptr(c->adam_m), *adam_v = uni_ptr(c->adam_v);
    const float prelu = unif(c->prelu);
    (void)adam_m; (void)adam_v;
    (void)bufA; (void)bufB; (void)sm_w1t; (void)sm_bias; (void)sm_wh; (void)sm_bh; (void)qv; (void)Vb; (void)Advb; (void)dq; (void)dAdv;
    (void)online; (void)target; (void)grad; (void)xs; (void)xs2; (void)dumps; (void)prelu;
    // LDS arrays through address-space-3 pointers: a generic pointer makes every access a FLAT instruction, which the hardware
    // completes out of order with respect to both counters -- each one then waits for every global load in flight
    lfloat *qv_l = (lfloat *)qv, *Vb_l = (lfloat *)Vb, *Advb_l = (lfloat *)Advb, *dq_l = (lfloat *)dq, *dAdv_l = (lfloat *)dAdv, *sm_wh_l = (lfloat *)sm_wh,
           *sm_bh_l = (lfloat *)sm_bh;
    (void)qv_l; (void)Vb_l; (void)Advb_l; (void)dq_l; (void)dAdv_l; (void)sm_wh_l; (void)sm_bh_l;
    const int pass = uni(pass_);
    auto dump_of = [&](int which, int blk) { return wc_dump(dumps, which, blk); };
    const float *par = pass ? online : target;
    for (int i = tid; i < 8 * W; i += NT) ((lfloat *)sm_w1t)[i] = par[oW1t + i];
    for (int i = tid; i < 5 * W; i += NT) {
        const int l = i >> 7, j = i & 127;
        const int off = l == 0 ? ob1 : (l == 1 ? ob2 : (l == 2 ? ob3 : (l == 3 ? obv1 : oba1)));
        ((lfloat *)sm_bias)[i] = par[off + j];
    }
    for (int i = tid; i < 4 * W + 4; i += NT) sm_wh_l[i] = par[oWh + i];       // Wh and bh are contiguous in both places
    StageRegs sr;
    WSUB_DECL;
    stage_load_direct(par + oW2t, L, sr);
    stage_store_direct(bufA, L, sr);
    __syncthreads();
    WSUB_MARK(32);
    constexpr bool PL = SP.kind == 0;                    // plain DQN: W2, then the A-column output layer on h2
    constexpr int NL = PL ? 1 : 4;
    const bool active = pass == 1 || wave < 4;
    const bool stored = pass == 1 && wave < 4;
    const int blk = wave & 3, slot = pass == 0 ? 2 : (wave < 4 ? 0 : 1);
    const float *X = stored ? xs : xs2;
    float bin[64], r[64];
    f32x16 acc[4];
    if (active) {                                      // layer 1: K = S
        acc_zero(acc);
        const lfloat *w1 = (const lfloat *)sm_w1t + L.h * W + L.li;
#pragma unroll
        for (int t = 0; t < S / 2; ++t) {
            const float xb = X[(32 * blk + L.li) * S + 2 * t + L.h];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[2 * t * W + 32 * jt], xb, acc[jt], 0, 0, 0);
        }
        tile_bias_act<ACT>(acc, sm_bias, L, prelu, r);
        if (stored) dump_store(dump_of(D_H1, blk), L, r);
        tile_to_operand(r);
#pragma unroll
        for (int v = 0; v < 64; ++v) bin[v] = r[v];
    }
#pragma unroll 1
    for (int l = 0; l < NL; ++l) {                     // W2 (bufA), W3 (bufB), Wv1 (bufA), Wa1 (bufB)
        float *cur = (l & 1) ? bufB : bufA, *nxt = (l & 1) ? bufA : bufB;
        if (l + 1 < NL) stage_load_direct(par + (l == 0 ? oW3t : (l == 1 ? oWv1t : oWa1t)), L, sr);
        if (active) {
            acc_zero(acc);
            chain128(cur, L, bin, acc);
            if (l == 1) tile_bias_act<LENV_ACT_IDENTITY>(acc, sm_bias + 2 * W, L, prelu, r);
            else tile_bias_act<ACT>(acc, sm_bias + (l == 0 ? 1 : (l == 2 ? 3 : 4)) * W, L, prelu, r);
            if (stored) dump_store(dump_of(l == 0 ? D_H2 : (l == 1 ? D_FEAT : (l == 2 ? D_V1 : D_A1)), blk), L, r);
            if (stored && (l >= 2 || PL)) {             // row-major copy for the head output layer's weight gradient (plain: h2, in R_A1's place)
                gfloat *rm = (gfloat *)dump_of(l == 2 ? R_V1 : R_A1, 0) + (32 * blk + L.li) * W + 4 * L.h;
#pragma unroll
                for (int pc = 0; pc < 16; ++pc)
                    *(gf4 *)(rm + 32 * (pc >> 2) + 8 * (pc & 3)) = f32x4{r[4 * pc], r[4 * pc + 1], r[4 * pc + 2], r[4 * pc + 3]};
            }
            tile_to_operand(r);
            if (l < 2 && !PL) {
#pragma unroll
                for (int v = 0; v < 64; ++v) bin[v] = r[v];
            } else {
                // head output layer on the fresh hidden block: V = wv2 . v1 + bv2 (row 0 of the tile), Adv = Wa2 . a1 + ba2 (rows 0..A-1)
                // (plain DQN: Q = W3 . h2 + b3 in the advantage columns)
                f32x16 hacc;
#pragma unroll
                for (int v = 0; v < 16; ++v) hacc[v] = 0.0f;
                const int col = (l == 2 && !PL) ? 0 : 1 + (L.li < A ? L.li : A - 1);
                const lfloat *wh = (const lfloat *)sm_wh + L.h * 4 + col;
#pragma unroll
                for (int t = 0; t < 64; ++t) hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wh[2 * t * 4], r[breg_of(t)], hacc, 0, 0, 0);
                if (L.h == 0) {
                    const int row = 32 * blk + L.li;
                    if (l == 2 && !PL) Vb_l[slot * RBH + row] = hacc[0] + sm_bh_l[0];
                    else {
#pragma unroll
                        for (int aa = 0; aa < A; ++aa) Advb_l[slot * RBH * A + row * A + aa] = hacc[aa] + sm_bh_l[1 + aa];
                    }
                }
            }
        }
        if (l + 1 < NL) stage_store_direct(nxt, L, sr);
        __syncthreads();
        WSUB_MARK(33 + l);
    }
}

// ---- backward pass of the TD loss through the online net on the s blocks (waves 0-3 own the sample blocks) ----
template <int SHAPE> static __device__ __noinline__ void wc_backward_big(const WcCtx *ctx_)
{
    using namespace wcp;
    constexpr WcShape SP = kWcShapes[SHAPE];
    constexpr int S = SP.S, A = SP.A, T = SP.T, B = WC_B, ACT = SP.q_act, RBH = B > T ? B : T;
    (void)S; (void)A; (void)T; (void)B; (void)ACT; (void)RBH;
    Lane L;
    L.init();
    const int tid = L.tid, wave = L.wave;
    (void)tid; (void)wave;
    typedef __attribute__((address_space(3))) const WcCtx LCtx;
    LCtx *c = (LCtx *)uni_ptr(ctx_);
    float *bufA = uni_ptr(c->bufA), *bufB = uni_ptr(c->bufB), *sm_w1t = uni_ptr(c->sm_w1t), *sm_bias = uni_ptr(c->sm_bias), *sm_wh = uni_ptr(c->sm_wh),
          *sm_bh = uni_ptr(c->sm_bh), *qv = uni_ptr(c->qv), *Vb = uni_ptr(c->Vb), *Advb = uni_ptr(c->Advb), *dq = uni_ptr(c->dq), *dAdv = uni_ptr(c->dAdv);
    float *online = uni_ptr(c->online), *target = uni_ptr(c->target), *grad = uni_ptr(c->grad), *xs = uni_ptr(c->xs), *xs2 = uni_ptr(c->xs2),
          *dumps = uni_ptr(c->dumps), *adam_m = uni_ptr(c->adam_m), *adam_v = uni_ptr(c->adam_v);
    const float prelu = unif(c->prelu);
    (void)adam_m; (void)adam_v;
    (void)bufA; (void)bufB; (void)sm_w1t; (void)sm_bias; (void)sm_wh; (void)sm_bh; (void)qv; (void)Vb; (void)Advb; (void)dq; (void)dAdv;
    (void)online; (void)target; (void)grad; (void)xs; (void)xs2; (void)dumps; (void)prelu;
    // LDS arrays through address-space-3 pointers: a generic pointer makes every access a FLAT instruction, which the hardware
    // completes out of order with respect to both counters -- each one then waits for every global load in flight
    lfloat *qv_l = (lfloat *)qv, *Vb_l = (lfloat *)Vb, *Advb_l = (lfloat *)Advb, *dq_l = (lfloat *)dq, *dAdv_l = (lfloat *)dAdv, *sm_wh_l = (lfloat *)sm_wh,
           *sm_bh_l = (lfloat *)sm_bh;
    (void)qv_l; (void)Vb_l; (void)Advb_l; (void)dq_l; (void)dAdv_l; (void)sm_wh_l; (void)sm_bh_l;
    auto dump_of = [&](int which, int blk) { return wc_dump(dumps, which, blk); };
    float r[64];
    f32x16 acc[4];
    StageRegs sr;
    const int blk = wave & 3;
    // torch.optim.Adam + the Polyak update run INSIDE the backward pass: the first half of layer q-1's matrix is updated by the
    // idle half of the workgroup while waves 0-3 run layer q's input-gradient chain (an element-wise pass bound by HBM / L2
    // latency next to a product bound by the matrix pipe), the rest at the end; element-wise, so the schedule changes no bit
    // (ctrl[10], ctrl[11]: this step's bias corrections)
    volatile float *ctrl = uni_ptr(c->ctrl);
    const AdamConsts ac{ ctrl[10], ctrl[11], unif(c->w1), unif(c->w2), unif(c->beta2), unif(c->adam_eps) };
    const float tau = unif(c->tau), omt = unif(c->omt);
    WSUB_DECL;
    constexpr bool PL = SP.kind == 0;                    // plain DQN: the output layer (advantage columns) on h2, then W2 and layer 1
    constexpr int NQ = PL ? 1 : 4;
    {   // head output layer, one element of gWh per thread (i ascending): gWh[k][0] = sum_i dq[i] v1[i][k], gWh[k][1+aa] = sum_i
        // dAdv[i][aa] a1[i][k], from the row-major copies of v1 / a1 (coalesced along k); plain DQN: gW3[aa][k] = sum_i dQ[i][aa] h2[i][k]
        const int k = tid & 127, col = tid >> 7;
        if (col <= A && !(PL && col == 0)) {
            const gfloat *rm = (const gfloat *)dump_of(col == 0 ? R_V1 : R_A1, 0) + k;
            float s = 0.0f;
            for (int i0 = 0; i0 < B; i0 += 64) {
                float hv[64];                          // 64 row reads in flight, then the ordered chain
#pragma unroll
                for (int u = 0; u < 64; ++u) hv[u] = rm[(i0 + u) * W];
#pragma unroll
                for (int u = 0; u < 64; ++u) s = fma32(col == 0 ? dq_l[i0 + u] : dAdv_l[(i0 + u) * A + col - 1], hv[u], s);
            }
            grad[oWh + k * 4 + col] = s;
        }
    }
    WSUB_MARK(22);
#pragma unroll 1
    for (int qi = 0; qi < NQ; ++qi) {                  // Wv1, Wa1, W3, W2
        // plain DQN: ONE round -- the upstream gradient comes from the output layer (as the advantage stream's does, with h2 in a1's
        // place), everything behind it is the dueling net's W2 round
        const int q = PL ? 3 : qi, qu = PL ? 1 : qi;
        const int oWt = q == 0 ? oWv1t : (q == 1 ? oWa1t : (q == 2 ? oW3t : oW2t));
        const int ob = q == 0 ? obv1 : (q == 1 ? oba1 : (q == 2 ? ob3 : ob2));
        L.refresh();
        stage_load_transposed(online + oWt, L, sr);
        if (wave < 4) {
            // upstream gradient block dz (lane = sample, register = unit)
            if (qu < 2) {
                const gf4 *hd = (const gf4 *)dump_of(PL ? D_H2 : (qu == 0 ? D_V1 : D_A1), blk) + L.lane;
                const int i = 32 * blk + L.li;
                const float dqi = dq_l[i];
                float da[A];
#pragma unroll
                for (int aa = 0; aa < A; ++aa) da[aa] = dAdv_l[i * A + aa];
#pragma unroll
                for (int pc = 0; pc < 16; ++pc) {            // piece (jt, g) = 16 B of the dump = units 32 jt + 8 g + 4 h + c
                    const f32x4 hv = hd[pc * 64];
                    const lfloat *whp = (const lfloat *)sm_wh + (32 * (pc >> 2) + 8 * (pc & 3) + 4 * L.h) * 4;
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const f32x4 wk = *(const lf4 *)(whp + 4 * cc);      // (wv2, wa2_0, wa2_1, wa2_2)[unit]
                        float up;
                        if (qu == 0) up = fma32(dqi, wk[0], 0.0f);
                        else {
                            up = 0.0f;
#pragma unroll
                            for (int aa = 0; aa < A; ++aa) up = fma32(da[aa], wk[1 + aa], up);
                        }
                        r[4 * pc + cc] = act_bwd(ACT, prelu, hv[cc], up);
                    }
                }
            } else dump_load(dump_of(q == 2 ? S_DFEAT : S_DH2, blk), L, r);
            tile_to_image(bufB, blk, L, r);
            tile_to_operand(r);
        }
        stage_store_transposed(bufA, L, sr);
        __syncthreads();
        WSUB_MARK(16);
        L.refresh();
        if (wave < 4) {
            // the epilogue's operand (f1, h2 or h1 dump) is requested before the chain: its latency hides behind the 256 MFMAs
            const gf4 *src = (const gf4 *)dump_of(q == 1 ? S_F1 : (q == 2 ? D_H2 : D_H1), blk) + L.lane;
            f32x4 hv[16];
#pragma unroll
            for (int pc = 0; pc < 16; ++pc) hv[pc] = src[pc * 64];      // (q 0 reads and ignores the h1 dump: no conditional array)
            acc_zero(acc);
            chain128(bufA, L, r, acc);
            WSUB_MARK(40);
            // epilogue, 16 bytes at a time: q 0: f1 = acc -> S_F1; q 1: d_feat = f1 + acc (epi_store, then epi_accum: old + new)
            // -> S_DFEAT; q 2 / 3: d_h = act'(h) * acc -> S_DH2 / S_DH1
            gf4 *dst = (gf4 *)dump_of(q == 0 ? S_F1 : (q == 1 ? S_DFEAT : (q == 2 ? S_DH2 : S_DH1)), blk) + L.lane;
#pragma unroll
            for (int pc = 0; pc < 16; ++pc) {
                f32x4 o;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const float g = acc[pc >> 2][4 * (pc & 3) + cc];
                    o[cc] = q == 0 ? g : (q == 1 ? hv[pc][cc] + g : act_bwd(ACT, prelu, hv[pc][cc], g));
                }
                dst[pc * 64] = o;
            }
            WSUB_MARK(41);
        } else {
            // the idle half: bias gradient of this layer (column sums of the dz image), half of the previous layer's optimizer step
            // (raised priority: next to an MFMA-paced wave on the same SIMD an un-prioritised VALU / memory wave only gets the
            // left-over issue slots and ran 4-5x slower than alone; the chain needs one issue slot per 64 cycles)
            __builtin_amdgcn_s_setprio(3);
            if (tid < 256 + W) grad[ob + (tid - 256)] = image_colsum(bufB, tid - 256, B);
            WSUB_MARK4(44);
            if (qi == 0) {
                if (tid >= 256 && tid < 256 + 1 + A && !(PL && tid == 256)) {
                    const int col = tid - 256;
                    float s = 0.0f;
                    for (int i = 0; i < B; ++i) s = s + (col == 0 ? dq_l[i] : dAdv_l[i * A + col - 1]);
                    grad[obh + col] = s;
                }
            }
            WSUB_MARK4(45);
            if (!PL && q >= 1) {                           // first half of layer q-1's matrix: its gradient is complete, the chain hides the pass
                const int oPrev = q == 1 ? oWv1t : (q == 2 ? oWa1t : oW3t);
                wg_adam_t(online, adam_m, adam_v, grad, oPrev, IMG / 2, ac, target, tau, omt, tid - 256, 256);
            }
            WSUB_MARK4(46);
            __builtin_amdgcn_s_setprio(0);
            WSUB_MARK4(46);
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        WSUB_MARK(17);
        L.refresh();
        if (wave < 4) {                                // image of the layer's input: feat, feat, h2, h1
            dump_load(dump_of(q < 2 ? D_FEAT : (q == 2 ? D_H2 : D_H1), blk), L, r);
            tile_to_image(bufA, blk, L, r);
        }
        __syncthreads();
        WSUB_MARK(18);
        L.refresh();
        wgrad_tiles(bufA, bufB, B, L, grad + oWt);
        __syncthreads();
        WSUB_MARK(19);
    }
    // layer 1: gW1t[k][j] = sum_i x[i][k] d_h1[i][j], gb1 = column sums of d_h1
    if (wave < 4) {
        dump_load(dump_of(S_DH1, blk), L, r);
        tile_to_image(bufB, blk, L, r);
    }
    for (int e = tid; e < B * S; e += NT) qv_l[e] = xs[e];                  // the minibatch states (qv is free after the TD step)
    __syncthreads();
    {
        const int j = tid & 127, kq = tid >> 7;
        const lfloat *img = (const lfloat *)bufB;
        for (int k = kq; k < S; k += 4) {
            float s = 0.0f;
            for (int i0 = 0; i0 < B; i0 += 8) {
                float dv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) dv[u] = img[(i0 + u) * W + (j ^ (u << 2))];
#pragma unroll
                for (int u = 0; u < 8; ++u) s = fma32(dv[u], qv_l[(i0 + u) * S + k], s);
            }
            grad[oW1t + k * W + j] = s;
        }
        if (tid >= 384) grad[ob1 + j] = image_colsum(bufB, j, B);
    }
    __syncthreads();
    WSUB_MARK(20);
    // what is left of the optimizer step
    wg_adam_t(online, adam_m, adam_v, grad, oW1t, ob2 + W - oW1t, ac, target, tau, omt, tid, NT);                 // W1t b1 W2t b2
    if constexpr (PL) wg_adam_t(online, adam_m, adam_v, grad, oWh, PW - oWh, ac, target, tau, omt, tid, NT);      // the output layer (column 0 of Wh / bh stays zero)
    else {
    wg_adam_t(online, adam_m, adam_v, grad, oW3t + IMG / 2, IMG / 2 + W, ac, target, tau, omt, tid, NT);          // second halves + biases
    wg_adam_t(online, adam_m, adam_v, grad, oWv1t + IMG / 2, IMG / 2 + W, ac, target, tau, omt, tid, NT);
    wg_adam_t(online, adam_m, adam_v, grad, oWa1t + IMG / 2, PW - (oWa1t + IMG / 2), ac, target, tau, omt, tid, NT);   // ... ba1 Wh bh
    }
    __syncthreads();
    WSUB_MARK(21);
}

// =====================================================================================================================================
// A chain on a TEAM of two workgroups (BASELINE configs[2]'s shard is 96 chains on 256 CUs).  Member g owns the sample blocks 2g, 2g+1
// of the 128-row minibatch and runs each of them on FOUR waves (wave 4q + jt = output tile jt of block 2g + q; the quad exchanges
// operand registers through bufB, see lenv_wavechain.cuh); the per-parameter work (weight gradients: reductions over all samples)
// is dealt by layer: member 0 takes the two stream layers and the head, member 1 the feature layers.  Same products, same chains,
// same bits as the one-workgroup kernel.
// =====================================================================================================================================
#define WCT_PROLOGUE                                                                                                                       \
    using namespace wcp;                                                                                                                   \
    constexpr WcShape SP = kWcShapes[SHAPE];                                                                                               \
    constexpr int S = SP.S, A = SP.A, T = SP.T, B = WC_B, ACT = SP.q_act, RBH = B > T ? B : T;                                             \
    (void)S; (void)A; (void)T; (void)B; (void)ACT; (void)RBH;                                                                              \
    Lane L;                                                                                                                                \
    L.init();                                                                                                                              \
    const int tid = L.tid, wave = L.wave;                                                                                                  \
    (void)tid; (void)wave;                                                                                                                 \
    typedef __attribute__((address_space(3))) const WcCtx LCtx;                                                                            \
    LCtx *c = (LCtx *)uni_ptr(ctx_);                                                                                                       \
    float *bufA = uni_ptr(c->bufA), *bufB = uni_ptr(c->bufB);                                                                              \
    lfloat *sm_w1t = (lfloat *)uni_ptr(c->sm_w1t), *sm_bias = (lfloat *)uni_ptr(c->sm_bias), *sm_wh = (lfloat *)uni_ptr(c->sm_wh),         \
           *sm_bh = (lfloat *)uni_ptr(c->sm_bh), *Vb_l = (lfloat *)uni_ptr(c->Vb), *Advb_l = (lfloat *)uni_ptr(c->Advb),                   \
           *dq_l = (lfloat *)uni_ptr(c->dq), *dAdv_l = (lfloat *)uni_ptr(c->dAdv), *qv_l = (lfloat *)uni_ptr(c->qv);                       \
    float *online = uni_ptr(c->online), *target = uni_ptr(c->target), *grad = uni_ptr(c->grad), *xs = uni_ptr(c->xs), *xs2 = uni_ptr(c->xs2), \
          *dumps = uni_ptr(c->dumps), *adam_m = uni_ptr(c->adam_m), *adam_v = uni_ptr(c->adam_v), *gva = uni_ptr(c->gva);                  \
    const float prelu = unif(c->prelu);                                                                                                    \
    const int tg = uni(c->g);                                                                                                              \
    (void)bufA; (void)bufB; (void)sm_w1t; (void)sm_bias; (void)sm_wh; (void)sm_bh; (void)Vb_l; (void)Advb_l; (void)dq_l; (void)dAdv_l;     \
    (void)qv_l; (void)online; (void)target; (void)grad; (void)xs; (void)xs2; (void)dumps; (void)adam_m; (void)adam_v; (void)gva;           \
    (void)prelu; (void)tg;                                                                                                                 \
    auto dump_of = [&](int which, int blk) { return wc_dump(dumps, which, blk); };                                                         \
    (void)dump_of

// one piece (16 bytes per lane) of a register-order dump: tile jt, group g4
__device__ __forceinline__ void piece_store(float *dump, int jt, int lane, const float (&r)[16])
{
    gf4 *d = (gf4 *)dump + lane;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) d[(4 * jt + g4) * 64] = f32x4{r[4 * g4], r[4 * g4 + 1], r[4 * g4 + 2], r[4 * g4 + 3]};
}
__device__ __forceinline__ void piece_load(const float *dump, int jt, int lane, f32x4 (&v)[4])
{
    const gf4 *d = (const gf4 *)dump + lane;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) v[g4] = d[(4 * jt + g4) * 64];
}

// ---- team forward: pass 0 = target net on s' (slot 2), pass 1 = online net on s (slot 0, activations dumped) and on s' (slot 1):
// the two inputs of pass 1 share every staged image (two jobs per wave).  The head outputs of this member's rows go to LDS and to
// the team's exchange arrays ----
template <int SHAPE> static __device__ __noinline__ void wct_forward(const WcCtx *ctx_, int pass_)
{
    WCT_PROLOGUE;
    const int pass = uni(pass_);
    constexpr bool PL = SP.kind == 0;                    // plain DQN: W2, then the A-column output layer on h2 (the advantage head's instructions)
    constexpr int NL = PL ? 1 : 4;
    const float *par = pass ? online : target;
    const int nj = pass ? 2 : 1;
    const int quad = wave >> 2, jt = wave & 3, blk = 2 * tg + quad, row = 32 * blk + L.li;
    for (int i = tid; i < 8 * W; i += NT) sm_w1t[i] = par[oW1t + i];
    for (int i = tid; i < 5 * W; i += NT) {
        const int l = i >> 7, j = i & 127;
        const int off = l == 0 ? ob1 : (l == 1 ? ob2 : (l == 2 ? ob3 : (l == 3 ? obv1 : oba1)));
        sm_bias[i] = par[off + j];
    }
    for (int i = tid; i < 4 * W + 4; i += NT) sm_wh[i] = par[oWh + i];
    StageRegs sr;
    WSUB_DECL;
    stage_load_direct(par + oW2t, L, sr);
    stage_store_direct(bufA, L, sr);
    __syncthreads();
    WSUB_MARK(48);
    float r16[2][16], rf[2][64];
    f32x16 acc;
    // exchange slots: [quad][job] x 16 KB = all of bufB
    auto xch = [&](int job) { return bufB + (quad * 2 + job) * 4096; };
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (j < nj) {                                   // layer 1: K = S
            const float *X = pass == 0 ? xs2 : (j == 0 ? xs : xs2);
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
            const lfloat *w1 = sm_w1t + L.h * W + 32 * jt + L.li;
#pragma unroll
            for (int t = 0; t < S / 2; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[2 * t * W], X[row * S + 2 * t + L.h], acc, 0, 0, 0);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 bv = *(const lf4 *)(sm_bias + 32 * jt + 8 * g4 + 4 * L.h);
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) r16[j][4 * g4 + cc] = act_fwd(ACT, prelu, acc[4 * g4 + cc] + bv[cc]);
            }
            if (pass == 1 && j == 0) piece_store(dump_of(D_H1, blk), jt, L.lane, r16[j]);
            tile16_to_operand(r16[j]);
            xch_put(xch(j), jt, L.lane, r16[j]);
        }
    }
    barrier_lds();
#pragma unroll
    for (int j = 0; j < 2; ++j) if (j < nj) xch_get(xch(j), L.lane, rf[j]);
    barrier_lds();
    WSUB_MARK(49);
#pragma unroll 1
    for (int l = 0; l < NL; ++l) {                     // W2, W3, Wv1, Wa1 (image of layer l in bufA)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j < nj) {
#pragma unroll
                for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
                chain_tile(bufA, jt, L, rf[j], acc);
                const lfloat *bb = sm_bias + (l + 1) * W + 32 * jt + 4 * L.h;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 bv = *(const lf4 *)(bb + 8 * g4);
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const float z = acc[4 * g4 + cc] + bv[cc];
                        r16[j][4 * g4 + cc] = l == 1 ? z : act_fwd(ACT, prelu, z);
                    }
                }
                if (pass == 1 && j == 0) {
                    piece_store(dump_of(l == 0 ? D_H2 : (l == 1 ? D_FEAT : (l == 2 ? D_V1 : D_A1)), blk), jt, L.lane, r16[j]);
                    if (l >= 2 || PL) {                 // row-major copy for the head output layer's weight gradient (plain: h2, in R_A1's place)
                        gfloat *rm = (gfloat *)dump_of(l == 2 ? R_V1 : R_A1, 0) + row * W + 32 * jt + 4 * L.h;
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) *(gf4 *)(rm + 8 * g4) = f32x4{r16[j][4 * g4], r16[j][4 * g4 + 1], r16[j][4 * g4 + 2], r16[j][4 * g4 + 3]};
                    }
                }
                tile16_to_operand(r16[j]);
            }
        }
        WSUB_MARK(50);
        if (!PL && l < 2) {                             // h2 / feat: the next layer's operand
#pragma unroll
            for (int j = 0; j < 2; ++j) if (j < nj) xch_put(xch(j), jt, L.lane, r16[j]);
            barrier_lds();
#pragma unroll
            for (int j = 0; j < 2; ++j) if (j < nj) xch_get(xch(j), L.lane, rf[j]);
            stage_load_direct(par + (l == 0 ? oW3t : oWv1t), L, sr);
            stage_store_direct(bufA, L, sr);
            barrier_lds();
        } else if (!PL && l == 2) {
            // v1 waits in the exchange slots (bufB) for the head output layer; the Wa1 image replaces Wv1 (both layers read feat)
#pragma unroll
            for (int j = 0; j < 2; ++j) if (j < nj) xch_put(xch(j), jt, L.lane, r16[j]);
            barrier_lds();                                 // every wave is through with the Wv1 image
            stage_load_direct(par + oWa1t, L, sr);
            stage_store_direct(bufA, L, sr);
            barrier_lds();
        } else {
            // a1 goes to exchange slots in bufA (the last image is used up), then the 2 quads x nj jobs x {V, Adv} head output layers
            // run one per wave: V = wv2 . v1 + bv2 (row 0 of the tile), Adv = Wa2 . a1 + ba2 (rows 0..A-1)
            barrier_lds();                                 // every wave is through with the Wa1 image
#pragma unroll
            for (int j = 0; j < 2; ++j) if (j < nj) xch_put(bufA + (quad * 2 + j) * 4096, jt, L.lane, r16[j]);
            barrier_lds();
            // task = wave: pass 1 (nj = 2): head = wave & 1, job = (wave >> 1) & 1, quad = wave >> 2; pass 0: head = wave & 1, quad = wave >> 1 (waves 0-3)
            const int hd = wave & 1, tj = nj == 2 ? (wave >> 1) & 1 : 0, tq = nj == 2 ? wave >> 2 : wave >> 1;
            if ((nj == 2 || wave < 4) && !(PL && hd == 0)) {      // (plain DQN: no value head)
                float rh[64];
                xch_get((hd == 0 ? bufB : bufA) + (tq * 2 + tj) * 4096, L.lane, rh);
                f32x16 hacc;
#pragma unroll
                for (int v = 0; v < 16; ++v) hacc[v] = 0.0f;
                const int col = hd == 0 ? 0 : 1 + (L.li < A ? L.li : A - 1);
                const lfloat *wh = sm_wh + L.h * 4 + col;
#pragma unroll
                for (int t = 0; t < 64; ++t) hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wh[2 * t * 4], rh[breg_of(t)], hacc, 0, 0, 0);
                if (L.h == 0) {
                    const int slot = pass == 0 ? 2 : tj, hrow = 32 * (2 * tg + tq) + L.li;
                    if (hd == 0) {
                        const float v = hacc[0] + sm_bh[0];
                        Vb_l[slot * RBH + hrow] = v;
                        gva[slot * B + hrow] = v;
                    } else {
#pragma unroll
                        for (int aa = 0; aa < A; ++aa) {
                            const float v = hacc[aa] + sm_bh[1 + aa];
                            Advb_l[slot * RBH * A + hrow * A + aa] = v;
                            gva[3 * B + (slot * B + hrow) * A + aa] = v;
                        }
                    }
                }
            }
        }
        WSUB_MARK(51);
    }
    __syncthreads();
    WSUB_MARK(52);
}

// ---- team backward, per-sample half: the input-gradient chain of this member's blocks through Wv1, Wa1 (-> d_feat), W3 (-> d_h2), W2
// (-> d_h1); the gradients go to the register-order dumps S_DFEAT / S_DH2 / S_DH1 the weight-gradient phase reads ----
template <int SHAPE> static __device__ __noinline__ void wct_backward_chain(const WcCtx *ctx_)
{
    WCT_PROLOGUE;
    const int quad = wave >> 2, jt = wave & 3, blk = 2 * tg + quad, row = 32 * blk + L.li;
    constexpr bool PL = SP.kind == 0;                    // plain DQN: ONE round -- upstream gradient from the output layer (h2 in a1's place), then the W2 round
    constexpr int NQ = PL ? 1 : 4;
    float *xch = bufB + quad * 4096;
    for (int i = tid; i < 4 * W + 4; i += NT) sm_wh[i] = online[oWh + i];
    StageRegs sr;
    WSUB_DECL;
    stage_load_transposed(online + (PL ? oW2t : oWv1t), L, sr);
    __syncthreads();
    WSUB_MARK(53);
    float r16[16], f1[16], rf[64];
    f32x16 acc;
#pragma unroll 1
    for (int qi = 0; qi < NQ; ++qi) {                  // Wv1, Wa1, W3, W2
        const int q = PL ? 3 : qi, qu = PL ? 1 : qi;
        if (qu < 2) {
            // upstream gradient tile dz (lane = sample, register = unit) from the stream's hidden activations and the head gradient
            f32x4 hv[4];
            piece_load(dump_of(PL ? D_H2 : (qu == 0 ? D_V1 : D_A1), blk), jt, L.lane, hv);
            const float dqi = dq_l[row];
            float da[A];
#pragma unroll
            for (int aa = 0; aa < A; ++aa) da[aa] = dAdv_l[row * A + aa];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const lfloat *whp = sm_wh + (32 * jt + 8 * g4 + 4 * L.h) * 4;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const f32x4 wk = *(const lf4 *)(whp + 4 * cc);      // (wv2, wa2_0, wa2_1, wa2_2)[unit]
                    float up;
                    if (qu == 0) up = fma32(dqi, wk[0], 0.0f);
                    else {
                        up = 0.0f;
#pragma unroll
                        for (int aa = 0; aa < A; ++aa) up = fma32(da[aa], wk[1 + aa], up);
                    }
                    r16[4 * g4 + cc] = act_bwd(ACT, prelu, hv[g4][cc], up);
                }
            }
        }
        // (q >= 2: r16 holds the previous layer's input gradient tile)
        if constexpr (PL) piece_store(dump_of(S_DH2, blk), jt, L.lane, r16);      // plain DQN: this IS d_h2 (the W2 weight gradient reads it)
        tile16_to_operand(r16);
        xch_put(xch, jt, L.lane, r16);
        stage_store_transposed(bufA, L, sr);
        barrier_lds();
        WSUB_MARK(54);
        xch_get(xch, L.lane, rf);
        f32x4 hv[4];
        if (q >= 2) piece_load(dump_of(q == 2 ? D_H2 : D_H1, blk), jt, L.lane, hv);
        if (q < 3) stage_load_transposed(online + (q == 0 ? oWa1t : (q == 1 ? oW3t : oW2t)), L, sr);
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.0f;
        chain_tile(bufA, jt, L, rf, acc);
        // epilogue: q 0: f1 = acc; q 1: d_feat = f1 + acc -> S_DFEAT; q 2 / 3: d_h = act'(h) * acc -> S_DH2 / S_DH1
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const float gq = acc[4 * g4 + cc];
                if (q == 0) f1[4 * g4 + cc] = gq;
                else if (q == 1) r16[4 * g4 + cc] = f1[4 * g4 + cc] + gq;
                else r16[4 * g4 + cc] = act_bwd(ACT, prelu, hv[g4][cc], gq);
            }
        if (q >= 1) piece_store(dump_of(q == 1 ? S_DFEAT : (q == 2 ? S_DH2 : S_DH1), blk), jt, L.lane, r16);
        barrier_lds();                                     // every wave is through with the image and the exchange slot
        WSUB_MARK(55);
    }
    __syncthreads();
    WSUB_MARK(56);
}

// ---- team backward, per-parameter half for layer q (0 Wv1, 1 Wa1, 2 W3, 3 W2): the [sample][unit] images of the upstream gradient
// (bufB, waves 0-3) and of the layer input (bufA, waves 4-7) of ALL four blocks from the dumps, the 16 gradient tiles, the bias ----
template <int SHAPE> static __device__ __noinline__ void wct_wgrad_layer(const WcCtx *ctx_, int q_)
{
    WCT_PROLOGUE;
    const int q = uni(q_);
    const int oWt = q == 0 ? oWv1t : (q == 1 ? oWa1t : (q == 2 ? oW3t : oW2t));
    const int ob = q == 0 ? obv1 : (q == 1 ? oba1 : (q == 2 ? ob3 : ob2));
    float r[64];
    WSUB_DECL;
    if (q < 2) for (int i = tid; i < 4 * W + 4; i += NT) sm_wh[i] = online[oWh + i];
    __syncthreads();
    if (wave < 4) {
        const int blk = wave;
        if (q < 2) {
            const gf4 *hd = (const gf4 *)dump_of(q == 0 ? D_V1 : D_A1, blk) + L.lane;
            const int i = 32 * blk + L.li;
            const float dqi = dq_l[i];
            float da[A];
#pragma unroll
            for (int aa = 0; aa < A; ++aa) da[aa] = dAdv_l[i * A + aa];
#pragma unroll
            for (int pc = 0; pc < 16; ++pc) {
                const f32x4 hv = hd[pc * 64];
                const lfloat *whp = sm_wh + (32 * (pc >> 2) + 8 * (pc & 3) + 4 * L.h) * 4;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const f32x4 wk = *(const lf4 *)(whp + 4 * cc);
                    float up;
                    if (q == 0) up = fma32(dqi, wk[0], 0.0f);
                    else {
                        up = 0.0f;
#pragma unroll
                        for (int aa = 0; aa < A; ++aa) up = fma32(da[aa], wk[1 + aa], up);
                    }
                    r[4 * pc + cc] = act_bwd(ACT, prelu, hv[cc], up);
                }
            }
        } else dump_load(dump_of(q == 2 ? S_DFEAT : S_DH2, blk), L, r);
        tile_to_image(bufB, blk, L, r);
    } else {
        const int blk = wave - 4;
        dump_load(dump_of(q < 2 ? D_FEAT : (q == 2 ? D_H2 : D_H1), blk), L, r);
        tile_to_image(bufA, blk, L, r);
    }
    __syncthreads();
    WSUB_MARK(57);
    L.refresh();
#ifdef WCT_FUSED_OPT
    if constexpr (SP.kind != 0) {
        // the optimizer step as the epilogue of the gradient tiles (t3v_tile_adam: the TD3 team path's routine): the two tiles' parameter /
        // target / Adam state is requested in front of the 128 matrix instructions, the accumulators go through a 4 KB LDS tile per wave
        // (the images are free by then) so that every arena access is 16 bytes per lane; no gradient round trip, no separate optimizer pass
        volatile lfloat *ctrl = (volatile lfloat *)uni_ptr(c->ctrl);
        const AdamConsts ac{ ctrl[10], ctrl[11], unif(c->w1), unif(c->w2), unif(c->beta2), unif(c->adam_eps) };
        const float tau = unif(c->tau), omt = unif(c->omt);
        const int kt = wave >> 1, jt0 = (wave & 1) << 1;
        const int off0 = oWt + (32 * kt) * W + 32 * jt0, off1 = off0 + 32;
        T3vTileState st0, st1;
        t3v_tile_state(L, online + off0, adam_m + off0, adam_v + off0, target + off0, 32, st0);
        t3v_tile_state(L, online + off1, adam_m + off1, adam_v + off1, target + off1, 32, st1);
        f32x16 acc0, acc1;
#pragma unroll
        for (int v = 0; v < 16; ++v) { acc0[v] = 0.0f; acc1[v] = 0.0f; }
        wgrad_accum(bufA, bufB, B, L, acc0, acc1);
        WSUB_MARK(58);
        const float sb = tid < W ? image_colsum(bufB, tid, B) : 0.0f;
        __syncthreads();                                   // every wave is through with the images: their room is the tiles' scratch
        t3v_tile_adam(acc0, st0, bufA + wave * 1024, nullptr, L, online + off0, adam_m + off0, adam_v + off0, target + off0, 32, nullptr, ac, tau, omt);
        // (both tiles' states are in flight behind the matrix instructions: ~190 VGPRs; until the file was built with
        // -fno-optimize-sibling-calls that meant 70 callee-saved registers through scratch per call and the second state was fetched here)
        t3v_tile_adam(acc1, st1, bufA + wave * 1024, nullptr, L, online + off1, adam_m + off1, adam_v + off1, target + off1, 32, nullptr, ac, tau, omt);
        if (tid < W) t3v_adam1(sb, online, adam_m, adam_v, target, ob + tid, ac, tau, omt);
        __syncthreads();
        WSUB_MARK(59);
        WSUB_MARK8(18);
        return;
    }
#endif
    wgrad_tiles(bufA, bufB, B, L, grad + oWt);
    WSUB_MARK(58);
    if (tid < W) grad[ob + tid] = image_colsum(bufB, tid, B);
    __syncthreads();
    WSUB_MARK(59);
    WSUB_MARK8(18);                                        // (member 1: the whole routine)
}

// head output layer (gWh, gbh) from the row-major copies of v1 / a1 -- member 0; layer 1 (gW1t, gb1) from the S_DH1 dumps -- member 1
template <int SHAPE> static __device__ __noinline__ void wct_wgrad_ends(const WcCtx *ctx_, int which_)
{
    WCT_PROLOGUE;
    const int which = uni(which_);
    WSUB_DECL;
#ifdef WCT_FUSED_OPT
    volatile lfloat *ctrl = (volatile lfloat *)uni_ptr(c->ctrl);
    const AdamConsts ac{ ctrl[10], ctrl[11], unif(c->w1), unif(c->w2), unif(c->beta2), unif(c->adam_eps) };
    const float tau = unif(c->tau), omt = unif(c->omt);
    constexpr bool FUSED = SP.kind != 0;
#define WCT_GRAD_OUT(off, val) do { if (FUSED) t3v_adam1((val), online, adam_m, adam_v, target, (off), ac, tau, omt); else grad[(off)] = (val); } while (0)
#else
#define WCT_GRAD_OUT(off, val) grad[(off)] = (val)
#endif
    if (which == 0) {
        constexpr bool PL = SP.kind == 0;                // plain DQN: the output layer sits in the advantage columns, no value column
        const int k = tid & 127, col = tid >> 7;
        if (col <= A && !(PL && col == 0)) {
            const gfloat *rm = (const gfloat *)dump_of(col == 0 ? R_V1 : R_A1, 0) + k;
            float s_ = 0.0f;
            {   // the whole column in flight at once: one memory round trip (two batches of 64 measured 26 k cycles for this routine)
                float hv[B];
#pragma unroll
                for (int u = 0; u < B; ++u) hv[u] = rm[u * W];
                const lfloat *up = col == 0 ? dq_l : dAdv_l + (col - 1);
                const int ups = col == 0 ? 1 : A;
#pragma unroll
                for (int u = 0; u < B; ++u) s_ = fma32(up[u * ups], hv[u], s_);
            }
            WCT_GRAD_OUT(oWh + k * 4 + col, s_);
        }
        if (tid >= 256 && tid < 256 + 1 + A && !(PL && tid == 256)) {
            const int col2 = tid - 256;
            float s_ = 0.0f;
            for (int i = 0; i < B; ++i) s_ = s_ + (col2 == 0 ? dq_l[i] : dAdv_l[i * A + col2 - 1]);
            WCT_GRAD_OUT(obh + col2, s_);
        }
        __syncthreads();
        WSUB_MARK(16);
        return;
    }
    float r[64];
    if (wave < 4) {
        dump_load(dump_of(S_DH1, wave), L, r);
        tile_to_image(bufB, wave, L, r);
    }
    for (int e = tid; e < B * S; e += NT) qv_l[e] = xs[e];                  // the minibatch states (qv is free after the TD step)
    __syncthreads();
    {
        const int j = tid & 127, kq = tid >> 7;
        const lfloat *img = (const lfloat *)bufB;
        for (int k = kq; k < S; k += 4) {
            float s_ = 0.0f;
            for (int i0 = 0; i0 < B; i0 += 8) {
                float dv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) dv[u] = img[(i0 + u) * W + (j ^ (u << 2))];
#pragma unroll
                for (int u = 0; u < 8; ++u) s_ = fma32(dv[u], qv_l[(i0 + u) * S + k], s_);
            }
            WCT_GRAD_OUT(oW1t + k * W + j, s_);
        }
        if (tid >= 384) { const float sb_ = image_colsum(bufB, j, B); WCT_GRAD_OUT(ob1 + j, sb_); }
    }
    __syncthreads();
    WSUB_MARK8(17);
}

template <int SHAPE>
__global__ __launch_bounds__(NT) void dueling_wavechain_kernel(const WcArgs a)
{
    using namespace wcp;
    extern __shared__ __align__(16) float lds[];
    constexpr WcShape SP = kWcShapes[SHAPE];
    constexpr int S = SP.S, A = SP.A, K = S + A, Hse = SP.Hse, T = SP.T, B = WC_B;
    constexpr bool PL = SP.kind == 0;                    // plain DQN (Critic_DQN S-128-128-A)
    static_assert(S % 2 == 0 && S <= 8 && A <= 3 && T <= 32 && Hse <= 128, "shape limits of the wave-chain kernel");
    const lenv_ddqn_cfg &cfg = a.cfg;
    Lane L;
    L.init();
    const int tid = L.tid;
    // a chain on a team of G workgroups (1 or 2): block x + 8 k is member k % G of chain 8 (k / G) + x, so the members share an XCD
    const int G = a.G;
    const int g = G == 1 ? 0 : (int)((blockIdx.x >> 3) % G);
    const int64_t chain = G == 1 ? (int64_t)blockIdx.x : (int64_t)8 * ((blockIdx.x >> 3) / G) + (blockIdx.x & 7);
    if (chain >= a.chains) return;
    // the chain's status word starts at 0 (ok); written here rather than by a memset node in front of the launch (a captured
    // generation replayed under rocprofv3 did not run the memset)
    if (threadIdx.x == 0 && g == 0 && a.out.status) a.out.status[chain] = 0;
    const float prelu = cfg.q_prelu;

    // ---- LDS carve-up ----
    float *bufA = lds, *bufB = bufA + IMG;
    float *sm_w1t = bufB + IMG;                           // [8][128]
    float *sm_bias = sm_w1t + 8 * W;                      // [5][128]: b1 b2 b3 bv1 ba1
    float *sm_wh = sm_bias + 5 * W;                       // [128][4]
    float *sm_bh = sm_wh + 4 * W;                         // [4] (+4 pad)
    float *se_wout = sm_bh + 8;                           // [S+2][Hse]
    float *se_bout = se_wout + (S + 2) * Hse;             // [16]
    float *se_h = se_bout + 16;                           // [3][Hse]
    constexpr int RBH = B > T ? B : T;
    float *qv = se_h + 3 * Hse;                           // [3][B][A]
    float *Vb = qv + 3 * RBH * A;                         // [3][RBH]
    float *Advb = Vb + 3 * RBH;                           // [3][RBH][A]
    float *dq = Advb + 3 * RBH * A;                       // [B]
    float *dAdv = dq + B;                                 // [B][A]
    float *misc = dAdv + B * A;                           // [64]
    // (alignment by index arithmetic on the 16-byte aligned LDS base: an integer round trip hides the address space and turns every access
    // to what is carved out behind it into a FLAT instruction, see td3_wavechain.hip)
    double *dstate = reinterpret_cast<double *>(lds + (((int)(misc + 64 - lds) + 1) & ~1));   // [T][4]
    double *ret = dstate + 4 * T;                         // [T]
    float *ep_rew = reinterpret_cast<float *>(ret + T);   // [T]
    int *alive = reinterpret_cast<int *>(ep_rew + T);     // [T]
    float *state = reinterpret_cast<float *>(alive + T);  // [8]
    float *newrow = state + 8;                            // [16]
    int *tlen = reinterpret_cast<int *>(newrow + 16);     // [T]
    WcCtx *ctx = reinterpret_cast<WcCtx *>(lds + (((int)(reinterpret_cast<float *>(tlen + T) - lds) + 3) & ~3));
    volatile __attribute__((address_space(3))) float *ctrl = (volatile __attribute__((address_space(3))) float *)misc;
    volatile __attribute__((address_space(3))) int *ictrl = (volatile __attribute__((address_space(3))) int *)(misc + 32);      // (explicitly LDS, see td3_wavechain.hip)

    float *arena = a.arena + chain * a.arena_stride;
    float *online = arena + a.a_par, *target = online + PW, *adam_m = target + PW, *adam_v = adam_m + PW, *grad = adam_v + PW;
    float *xs = arena + a.a_xs, *xs2 = arena + a.a_xs2, *dumps = arena + a.a_dump, *rb = arena + a.a_replay;
    float *se_w0T = arena + a.a_se, *se_b0 = se_w0T + 3 * K * Hse;        // [3][K][Hse], [3][Hse]
    double *meter = reinterpret_cast<double *>(arena + a.a_meter);
    float *gva = arena + a.a_gx;                                              // team exchange: Vb [3][B] | Advb [3][B][A]
    unsigned *team_bar = reinterpret_cast<unsigned *>(arena + a.a_bar);
    float *xscr = G == 1 ? xs2 : arena + a.a_xtm + (int64_t)g * RBH * S;      // rows of the one-row / lock-step forwards: a member's own in a team
    const int RS = a.RS;

    // ---- stage the perturbed SE (GTN_worker.py:165-175): first layers transposed into the arena, output layers into LDS ----
    {
        const float sg = a.eps ? a.sign[chain] : 0.0f;
        const float *e = a.eps ? a.eps + (int64_t)a.worker[chain] * a.P_se : nullptr;
        for (int i = tid; i < a.P_se; i += NT) {
            const float w = e ? fma32(sg, e[i], a.theta[i]) : a.theta[i];
            int net = 0, r = i;
            if (r >= a.se_net_size[0]) { r -= a.se_net_size[0]; net = 1; if (r >= a.se_net_size[1]) { r -= a.se_net_size[1]; net = 2; } }
            const int orow = net == 0 ? 0 : (net == 1 ? S : S + 1);
            if (r < Hse * K) { int j = r / K, k = r - j * K; se_w0T[(net * K + k) * Hse + j] = w; }
            else if ((r -= Hse * K) < Hse) se_b0[net * Hse + r] = w;
            else {
                r -= Hse;
                const int n_out = net == 0 ? S : 1;
                if (r < n_out * Hse) { int o = r / Hse, j = r - o * Hse; se_wout[(orow + o) * Hse + j] = w; }
                else se_bout[orow + (r - n_out * Hse)] = w;
            }
        }
    }
    // ---- fresh agent (DuelingDDQN.py:31-36): arena layout, Adam state and gradient (incl. the zero rows of W1t) cleared ----
    if (g == 0) {                                          // the arena is shared by the team: its first member fills it
        for (int p = tid; p < PW; p += NT) { online[p] = 0.0f; target[p] = 0.0f; adam_m[p] = 0.0f; adam_v[p] = 0.0f; grad[p] = 0.0f; }
        __syncthreads();
        for (int p = tid; p < a.P; p += NT) {
            const float w = a.agent_init[chain * a.P + p];
            const int q = PL ? wc_sd_to_arena_plain(p, S, A) : wc_sd_to_arena(p, S, A);
            online[q] = w; target[q] = w;
        }
    }
    if (tid < 64) misc[tid] = 0.0f;
    if (tid == 0) {
        WcCtx cx{ bufA, bufB, sm_w1t, sm_bias, sm_wh, sm_bh, qv, Vb, Advb, dq, dAdv, online, target, grad, xs, xs2, dumps, adam_m, adam_v, (volatile float *)misc, prelu,
                  (float)(1.0 - cfg.adam_beta1), (float)(1.0 - cfg.adam_beta2), (float)cfg.adam_beta2, (float)cfg.adam_eps, (float)cfg.tau,
                  (float)(1.0 - cfg.tau), g, G, gva, dstate, ep_rew, alive, tlen, ret, reinterpret_cast<int *>(misc + 32), cfg.max_steps };
        *ctx = cx;
    }
    __syncthreads();

    const uint64_t key = a.rng_keys[chain];
    constexpr int env_id = SP.env;
    int status = 0;
    WPT_DECL;
    int train_steps = 0, n_act = 0, learn_it = 0, n_test_ep = 0, test_steps = 0, episodes_run = 0;
    double eps_g = cfg.eps_init, b1pow = 1.0, b2pow = 1.0;
    const int rb_cap = (int)a.rb_cap;
    // ---- team barrier (G = 2; wc::team_barrier): the chain's counter and the launch's give-up word are zeroed by wct_team_reset_kernel
    TeamSync tsync{ team_bar, reinterpret_cast<unsigned *>(a.arena + a.a_bar) + 8, ictrl + 5, 0u, G, false, false };
    bool team_dead = false;
    auto team_barrier = [&]() {
        if (G == 1) return;
        wc::team_barrier<false>(tsync, tid);
        if (tsync.dead) { team_dead = true; status = -10; }
    };
    if (G > 1 && tid == 0) reinterpret_cast<unsigned *>(gva)[g] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;   // HW_REG_XCC_ID[3:0]
    team_barrier();                                        // the arena is initialised, every member's XCD id is posted
    if (G > 1) {
        bool same = true;
        const unsigned x0 = reinterpret_cast<unsigned *>(gva)[0];
        for (int m = 1; m < G; ++m) same = same && reinterpret_cast<unsigned *>(gva)[m] == x0;
        team_barrier();                                    // everybody has read the ids before the exchange rows are reused
        tsync.same_xcd = same;
    }
    if (team_dead) {                                       // not all members became resident in time: nothing was computed
        if (a.out.status) atomicMin(&a.out.status[chain], -10);
        return;
    }

    // q_out[I][A] from the head outputs of `slot` (models/actor_critic.py:117-122; learn: mean over ALL I*A advantages)
    auto finish_q = [&](int slot, int I, float *q_out, bool global_mean) {
        const float *Vs = Vb + slot * RBH, *As = Advb + slot * RBH * A;
        if constexpr (PL) {                                // Critic_DQN: the output layer's rows are the Q values
            for (int e = tid; e < I * A; e += NT) q_out[e] = As[e];
        } else
        if (global_mean) {
            if (tid == 0) {
                float sum = 0.0f;
                for (int e = 0; e < I * A; ++e) sum = sum + As[e];
                ctrl[8] = sum / (float)(I * A);
            }
            __syncthreads();
            const float mean = ctrl[8];
            for (int e = tid; e < I * A; e += NT) q_out[e] = Vs[e / A] + (As[e] - mean);
        } else {
            for (int i = tid; i < I; i += NT) {
                float sum = 0.0f;
                for (int aa = 0; aa < A; ++aa) sum = sum + As[i * A + aa];
                const float mean = sum / (float)A;
                for (int aa = 0; aa < A; ++aa) q_out[i * A + aa] = Vs[i] + (As[i * A + aa] - mean);
            }
        }
        __syncthreads();
    };

    auto forward_thin = [&](const float *X, int I) {      // I <= 32 rows through the ONLINE net -> qv (per-row advantage mean)
        wc_forward_thin_layers<SHAPE>(ctx, X, I);
        finish_q(0, I, qv, false);
    };

    // ---- real-env test phase: T episodes in lock-step (BaseAgent.test, agents/base_agent.py:155-227) ----
    auto test_phase = [&]() {
        if constexpr (T <= 16) {
            __syncthreads();
            wc_test_steps<SHAPE>(ctx, (uint32_t)key, (uint32_t)(key >> 32), n_test_ep);      // (ends with a barrier; ictrl[0] = steps taken)
        } else {
            if (tid < T) {
                double st[4];
                real_env_reset_draw(env_id, key, STREAM_TEST_RESET, (int64_t)n_test_ep + tid, st);
                for (int i = 0; i < 4; ++i) dstate[tid * 4 + i] = st[i];
                ep_rew[tid] = 0.0f; alive[tid] = 1; tlen[tid] = 0;
            }
            if (tid == 0) ictrl[0] = 0;
            __syncthreads();
            float *xt = xscr;
            for (int t = 0; t < cfg.max_steps; ++t) {
                if (tid < T) { float obs[8]; real_env_obs(env_id, dstate + tid * 4, obs); for (int i = 0; i < S; ++i) xt[tid * S + i] = obs[i]; }
                __syncthreads();
                forward_thin(xt, T);
                if (tid < T && alive[tid]) {
                    int am = 0; float best = qv[tid * A];
                    for (int aa = 1; aa < A; ++aa) { const float v = qv[tid * A + aa]; if (v > best) { best = v; am = aa; } }
                    double st[4] = { dstate[tid * 4], dstate[tid * 4 + 1], dstate[tid * 4 + 2], dstate[tid * 4 + 3] };
                    double rew; int dn;
                    real_env_step(env_id, st, am, rew, dn);
                    for (int i = 0; i < 4; ++i) dstate[tid * 4 + i] = st[i];
                    ep_rew[tid] = ep_rew[tid] + (float)rew;
                    tlen[tid] = tlen[tid] + 1;
                    atomicAdd(reinterpret_cast<int *>(misc + 32), 1);      // (= ictrl[0])
                    if (dn) alive[tid] = 0;
                }
                __syncthreads();
                int any = 0;
                for (int e = 0; e < T; ++e) any |= alive[e];
                if (!any) break;
            }
            if (tid < T) ret[tid] = (double)ep_rew[tid];
            __syncthreads();
        }
        n_test_ep += T;
        test_steps += ictrl[0];
        __syncthreads();
    };

    const bool budgeted = cfg.step_budget > 0;
    int timed_out_at = -1;
    for (int episode = 0; episode < cfg.train_episodes; ++episode) {
        if (budgeted && (int64_t)train_steps + test_steps > cfg.step_budget) { timed_out_at = episode; break; }
        if (episode == 0) eps_g = cfg.eps_init;
        else { eps_g *= cfg.eps_decay; if (eps_g < cfg.eps_min) eps_g = cfg.eps_min; }
        const bool learning = episode >= cfg.init_episodes;
        if (tid == 0) {
            double st0[4];
            real_env_reset_draw(env_id, key, STREAM_TRAIN_RESET, episode, st0);
            float obs[8];
            real_env_obs(env_id, st0, obs);
            for (int i = 0; i < S; ++i) state[i] = obs[i];
        }
        __syncthreads();
        int ep_len = 0;
        for (int t = 0; t < cfg.max_steps; ++t) {
            WPT_MARK(9);
            const int size_after = train_steps + 1 < rb_cap ? train_steps + 1 : rb_cap;
            const int new_pos = train_steps % rb_cap;
            if (tid == 0) {                                // select_train_action (DuelingDDQN.py:96-103)
                const double u = u64_to_unit(rng_u64(key, STREAM_EPS, (uint64_t)train_steps));
                int explored = u < eps_g, action = -1;
                if (explored) action = (int)u64_to_below(rng_u64(key, STREAM_ACTION, (uint64_t)n_act), (uint32_t)A);
                ictrl[1] = explored; ictrl[2] = action;
            }
            __syncthreads();
            const int explored = ictrl[1];
            if (explored) ++n_act;
            if (!explored) {
                for (int i = tid; i < S; i += NT) xscr[i] = state[i];
                __syncthreads();
                forward_thin(xscr, 1);
                if (tid == 0) {
                    int am = 0; float best = qv[0];
                    for (int aa = 1; aa < A; ++aa) if (qv[aa] > best) { best = qv[aa]; am = aa; }
                    ictrl[2] = am;
                }
                __syncthreads();
            }
            const int action = ictrl[2];
            WPT_MARK(0);
            // ---- EnvWrapper.step -> VirtualEnv.step: x = [onehot(action), state] ----
            for (int uu = tid; uu < 3 * Hse; uu += NT) {
                const int net = uu / Hse, j = uu - net * Hse;
                const float *w = se_w0T + net * K * Hse + j;
                float wk[K];
#pragma unroll
                for (int k = 0; k < K; ++k) wk[k] = w[k * Hse];
                float z = 0.0f;
#pragma unroll
                for (int k = 0; k < K; ++k) z = fma32(k < A ? (k == action ? 1.0f : 0.0f) : state[k - A], wk[k], z);
                z = z + se_b0[uu];
                se_h[uu] = act_fwd(SP.se_act, cfg.se_prelu, z);
            }
            __syncthreads();
            if (tid < S + 2) {
                const int net = tid < S ? 0 : (tid == S ? 1 : 2);
                const float *h = se_h + net * Hse, *w = se_wout + tid * Hse;
                float acc = 0.0f;
                for (int j = 0; j < Hse; ++j) acc = fma32(h[j], w[j], acc);
                acc = acc + se_bout[tid];
                if (tid < S) newrow[S + 1 + tid] = acc; else newrow[2 * S + 1 + (tid - S)] = acc;
            }
            if (tid >= 64 && tid < 64 + S) newrow[tid - 64] = state[tid - 64];
            if (tid == 128) newrow[S] = (float)action;
            __syncthreads();
            if (tid < 2 * S + 3) rb[(int64_t)new_pos * RS + tid] = newrow[tid];
            const float done_now = newrow[2 * S + 2];
            __syncthreads();
            if (tid < S) state[tid] = newrow[S + 1 + tid];
            ++ep_len; ++train_steps;
            __syncthreads();
            WPT_MARK(1);

            if (learning) {
                // ================= DuelingDDQN.learn (DuelingDDQN.py:59-94) =================
                for (int b = tid; b < B; b += NT) {
                    const int64_t n = (int64_t)learn_it * B + b;
                    const int idx = (int)rng_replay_below(key, (uint64_t)n, (uint32_t)size_after);
                    const float *row = rb + (int64_t)idx * RS;
                    // (the whole row into registers first: written element by element the compiler cannot move a load over the store in
                    // front of it -- the arrays may alias as far as it knows -- and the gather was 15 dependent memory round trips)
                    float rv[2 * S + 3];
#pragma unroll
                    for (int i = 0; i < 2 * S + 3; ++i) rv[i] = row[i];
#pragma unroll
                    for (int i = 0; i < S; ++i) { xs[b * S + i] = rv[i]; xs2[b * S + i] = rv[S + 1 + i]; }
                    dAdv[b * A + 0] = rv[S];
                    dAdv[b * A + 1] = rv[2 * S + 1];
                    dq[b] = rv[2 * S + 2];
                }
                __syncthreads();
                WPT_MARK(2);
#ifndef WC_DIAG_NO_FWD
                if (G == 2) {
#pragma unroll 1
                    for (int pass = 0; pass < 2; ++pass) wct_forward<SHAPE>(ctx, pass);
                    WBAR(60);                              // every row's V / Adv is in the exchange arrays
                    for (int e = tid; e < 3 * B; e += NT) Vb[(e / B) * RBH + (e % B)] = gva[e];
                    for (int e = tid; e < 3 * B * A; e += NT) Advb[(e / (B * A)) * RBH * A + (e % (B * A))] = gva[3 * B + e];
                    __syncthreads();
                } else {
#pragma unroll 1
                    for (int pass = 0; pass < 2; ++pass) wc_forward_big<SHAPE>(ctx, pass);  // target net on s'; online net on s (stored) and s'
                }
#endif
                finish_q(1, B, qv + B * A, true);
                finish_q(2, B, qv + 2 * B * A, true);
                finish_q(0, B, qv, true);
                WPT_MARK(3);
                for (int b = tid; b < B; b += NT) {        // TD error (DuelingDDQN.py:80-85)
                    const float g32 = (float)cfg.gamma, norm = (float)(2.0 / (double)B);
                    const int ab = (int)dAdv[b * A + 0];
                    const float rr = dAdv[b * A + 1], d = dq[b];
                    int am = 0; float best = qv[(B + b) * A];
                    for (int aa = 1; aa < A; ++aa) { const float v = qv[(B + b) * A + aa]; if (v > best) { best = v; am = aa; } }
                    const float t1 = g32 * qv[(2 * B + b) * A + am];
                    const float t2 = 1.0f - d;
                    const float y = rr + t1 * t2;
                    dq[b] = norm * (qv[b * A + ab] - y);
                    Vb[b] = (float)ab;
                }
                __syncthreads();
                if (tid == 0) {
                    float s_dq = 0.0f;
                    for (int b = 0; b < B; ++b) s_dq = s_dq + dq[b];
                    ctrl[9] = (-s_dq) / (float)(B * A);            // backward of `- advantages.mean()` (dueling only)
                    b1pow *= cfg.adam_beta1; b2pow *= cfg.adam_beta2;
                    ctrl[10] = (float)(-(cfg.lr / (1.0 - b1pow)));
                    ctrl[11] = (float)__builtin_sqrt(1.0 - b2pow);
                }
                __syncthreads();
                {
                    const float mean_grad = ctrl[9];
                    for (int e = tid; e < B * A; e += NT) {
                        const int b = e / A, aa = e - b * A;
                        const float g = aa == (int)Vb[b] ? dq[b] : 0.0f;
                        dAdv[e] = PL ? g : g + mean_grad;          // (plain DQN: dL/dQ, only entry a_b of a row is non-zero)
                    }
                }
                __syncthreads();
                WPT_MARK(4);
#ifndef WC_DIAG_NO_BWD
                if (G == 2) {
                    wct_backward_chain<SHAPE>(ctx);
                    WBAR(61);                              // all four blocks' gradient dumps are there
                    if constexpr (PL) {                    // plain DQN: the output layer and layer 1 | W2
                        if (g == 0) { wct_wgrad_ends<SHAPE>(ctx, 0); wct_wgrad_ends<SHAPE>(ctx, 1); }
                        else wct_wgrad_layer<SHAPE>(ctx, 3);
                    } else {
#ifdef WCT_FUSED_OPT
                        // (the optimizer step rides on the gradient routines: the head output layer LAST -- the stream layers' upstream
                        // gradients read its weights as they were)
                        if (g == 0) { wct_wgrad_layer<SHAPE>(ctx, 0); wct_wgrad_layer<SHAPE>(ctx, 1); wct_wgrad_ends<SHAPE>(ctx, 0); }
                        else { wct_wgrad_layer<SHAPE>(ctx, 2); wct_wgrad_layer<SHAPE>(ctx, 3); wct_wgrad_ends<SHAPE>(ctx, 1); }
#else
                        if (g == 0) { wct_wgrad_ends<SHAPE>(ctx, 0); wct_wgrad_layer<SHAPE>(ctx, 0); wct_wgrad_layer<SHAPE>(ctx, 1); }
                        else { wct_wgrad_layer<SHAPE>(ctx, 2); wct_wgrad_layer<SHAPE>(ctx, 3); wct_wgrad_ends<SHAPE>(ctx, 1); }
#endif
                    }
#ifdef WCT_FUSED_OPT
                    if constexpr (PL)
#endif
                    {
                    WBAR(62);
                    {   // torch.optim.Adam + Polyak, half of the parameter vector per member (ctrl[10], ctrl[11]: this step's bias corrections)
                        const AdamConsts ac{ ctrl[10], ctrl[11], (float)(1.0 - cfg.adam_beta1), (float)(1.0 - cfg.adam_beta2), (float)cfg.adam_beta2,
                                             (float)cfg.adam_eps };
                        if constexpr (PL) {                // the used part of the arena layout: W1t b1 W2t b2 | Wh bh (column 0 stays zero)
                            const int mid = oW2t + IMG / 2;
                            if (g == 0) wg_adam_t(online, adam_m, adam_v, grad, oW1t, mid - oW1t, ac, target, (float)cfg.tau, (float)(1.0 - cfg.tau), tid, NT);
                            else {
                                wg_adam_t(online, adam_m, adam_v, grad, mid, ob2 + W - mid, ac, target, (float)cfg.tau, (float)(1.0 - cfg.tau), tid, NT);
                                wg_adam_t(online, adam_m, adam_v, grad, oWh, PW - oWh, ac, target, (float)cfg.tau, (float)(1.0 - cfg.tau), tid, NT);
                            }
                        } else {
                        const int half = ((PW / 2) + 3) & ~3;
                        const int lo = g == 0 ? 0 : half, n = g == 0 ? half : PW - half;
                        wg_adam_t(online, adam_m, adam_v, grad, lo, n, ac, target, (float)cfg.tau, (float)(1.0 - cfg.tau), tid, NT);
                        }
                    }
                    }
                    WBAR(63);
                } else wc_backward_big<SHAPE>(ctx);
#endif
                WPT_MARK(6);
                ++learn_it;                                // (optimizer step + Polyak update: inside wc_backward_big)
                __syncthreads();
                WPT_MARK(7);
                if (team_dead) break;                      // (uniform in the workgroup) the team gave up: leave, status -10
            }
            if (done_now > 0.5f) break;
        }
        ++episodes_run;
        if (team_dead) break;
        if (tid == 0 && g == 0 && a.out.episode_len) a.out.episode_len[chain * cfg.train_episodes + episode] = ep_len;
        __syncthreads();
        WPT_MARK(9);
        test_phase();
        WPT_MARK(8);
        if (tid == 0) {
            double sm = 0.0;
            for (int i = 0; i < T; ++i) sm += ret[i];
            const double tm = sm / (double)T;
            meter[episode] = tm;
            if (g == 0 && a.out.episode_test_mean) a.out.episode_test_mean[chain * cfg.train_episodes + episode] = tm;
            int brk = 0;
            if (learning) {
                int lo = episode + 1 - cfg.early_out_num; if (lo < 0) lo = 0;
                double s2 = 0.0;
                for (int i = lo; i <= episode; ++i) s2 += meter[i];
                if (s2 / ((double)(episode + 1 - lo) + 1e-9) >= cfg.solved_reward) brk = 1;
            }
            ictrl[3] = brk;
        }
        __syncthreads();
        const int brk = ictrl[3];
        __syncthreads();
        if (brk) break;
    }
    WPT_MARK(9);
    const int64_t remaining = cfg.step_budget - ((int64_t)train_steps + test_steps);
    const int test_before = test_steps;
    if (!team_dead) test_phase();
    if (budgeted) {
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
    WPT_MARK(8);
#ifdef LENV_PHASE_TIMING
    if (tid == 0 && chain == 0) for (int pi = 0; pi < 12; ++pi) g_wc_phase_cycles[pi] = pt_acc[pi];
#endif
    if (tid == 0 && g == 0) {
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
    if (a.out.final_online && g == 0)
        for (int p = tid; p < a.P; p += NT) a.out.final_online[chain * a.P + p] = online[PL ? wc_sd_to_arena_plain(p, S, A) : wc_sd_to_arena(p, S, A)];
    if (a.out.status && status != 0) atomicMin(&a.out.status[chain], status);
}

}  // namespace lenv

using namespace lenv;

// Host side (internal linkage to the library: declared in lenv_wavechain_host.h): called by lenv_dueling_se_inner_loop_icm
// (dueling_se_inner_loop.hip) for launches whose shape and mode the wave-chain kernel covers.  Returns 0 when `cfg` is not one of its shapes.
int lenv_wc_dueling_shape(const lenv_ddqn_cfg *cfg)
{
    for (int s = 1; s < (int)(sizeof(kWcShapes) / sizeof(kWcShapes[0])); ++s) {
        const WcShape &sp = kWcShapes[s];
        if (cfg->agent_kind == sp.kind && cfg->env_id == sp.env && cfg->state_dim == sp.S && cfg->num_actions == sp.A &&
            (sp.kind == 0 || cfg->feature_dim == WC_H) && cfg->q_hidden == WC_H && cfg->q_layers == 2 && cfg->batch_size == WC_B && cfg->se_hidden == sp.Hse && cfg->se_layers == 1 &&
            cfg->test_episodes == sp.T && cfg->q_act == sp.q_act && cfg->se_act == sp.se_act && cfg->synthetic_env_type == 0 && !cfg->icm_enabled &&
            !cfg->q_layer_norm && cfg->test_mode == 0)      // (test_mode 1, the evaluation harness's training call: GEMM-queue kernel)
            return s;
    }
    return 0;
}

static size_t wc_lds_bytes(const WcShape &sp)
{
    const int S = sp.S, A = sp.A, Hse = sp.Hse, T = sp.T, B = WC_B, RBH = B > T ? B : T;
    size_t f = 2 * (size_t)wc::IMG + 8 * wc::W + 5 * wc::W + 4 * wc::W + 8 + (size_t)(S + 2) * Hse + 16 + 3 * (size_t)Hse + 3 * (size_t)RBH * A +
               3 * (size_t)RBH + 3 * (size_t)RBH * A + B + (size_t)B * A + 64 + 2 + 2 * (4 * (size_t)T + T) + 2 * (size_t)T + 8 + 16 + T + 4 + (sizeof(WcCtx) + 3) / 4;
    return f * sizeof(float);
}

// floats of one chain's arena in the wave-chain layout (the workspace query of the generic path takes the maximum of both)
int64_t lenv_wc_dueling_arena_floats(const lenv_ddqn_cfg *cfg, int shape, int64_t rb_cap, int RS, int P_se)
{
    const WcShape &sp = kWcShapes[shape];
    const int S = sp.S, B = WC_B, T = sp.T, K = sp.S + sp.A;
    int64_t off = 0;
    auto take = [&](int64_t n) { int64_t r = off; off += (n + 3) & ~(int64_t)3; return r; };
    take(5 * (int64_t)wcp::PW); take((int64_t)B * S); take((int64_t)(B > T ? B : T) * S); take((int64_t)WC_NDUMP * 4 * wc::BLK);
    take(3 * (int64_t)(K + 1) * sp.Hse); take(rb_cap * RS); take(2 * (int64_t)(cfg->train_episodes > 0 ? cfg->train_episodes : 1));
    take(3 * (int64_t)B * (1 + sp.A)); take(16); take(2 * (int64_t)(B > T ? B : T) * S);      // team exchange, barrier word, per-member scratch rows
    (void)P_se;
    return (off + 63) & ~(int64_t)63;
}

namespace lenv {
__global__ void wct_team_reset_kernel(float *arena, int64_t arena_stride, int64_t a_bar, int64_t chains)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < chains) { unsigned *b = reinterpret_cast<unsigned *>(arena + c * arena_stride + a_bar); b[0] = 0u; b[8] = 0u; }
}
}

// Workgroups per chain: 2 when all 8 * ceil(chains / 8) * 2 workgroups are resident at once (occupancy API: one per CU at this
// kernel's LDS footprint) -- the members wait for each other --, else 1.  cfg->team_size 1 forces one workgroup per chain.
int lenv_wc_dueling_team(const lenv_ddqn_cfg *cfg, int shape, int64_t chains)
{
    if (cfg->team_size == 1 || chains < 1 || shape <= 0) return 1;
    void (*kern)(const WcArgs) = shape == 2 ? dueling_wavechain_kernel<2> : dueling_wavechain_kernel<1>;
    return lenv_team_grid_resident(reinterpret_cast<const void *>(kern), wc::NT, wc_lds_bytes(kWcShapes[shape]), 8 * ((chains + 7) / 8) * 2) ? 2 : 1;
}

int lenv_wc_dueling_launch(int shape, const lenv_ddqn_cfg *cfg, const float *theta, const float *eps, const int32_t *worker,
                                      const float *sign, const float *agent_init, const uint64_t *rng_keys, int64_t chains, float *arena,
                                      int64_t arena_stride, int64_t rb_cap, int RS, int P, int P_se, const int *se_net_size,
                                      const lenv_inner_out *out, hipStream_t stream)
{
    const WcShape &sp = kWcShapes[shape];
    WcArgs a;
    a.cfg = *cfg;
    a.theta = theta; a.eps = eps; a.worker = worker; a.sign = sign; a.agent_init = agent_init; a.rng_keys = rng_keys;
    a.arena = arena; a.arena_stride = arena_stride; a.out = *out; a.rb_cap = rb_cap; a.RS = RS; a.P = P; a.P_se = P_se;
    for (int i = 0; i < 3; ++i) a.se_net_size[i] = se_net_size[i];
    const int S = sp.S, B = WC_B, T = sp.T, K = sp.S + sp.A;
    int64_t off = 0;
    auto take = [&](int64_t n) { int64_t r = off; off += (n + 3) & ~(int64_t)3; return r; };
    a.a_par = take(5 * (int64_t)wcp::PW); a.a_xs = take((int64_t)B * S); a.a_xs2 = take((int64_t)(B > T ? B : T) * S);
    a.a_dump = take((int64_t)WC_NDUMP * 4 * wc::BLK); a.a_se = take(3 * (int64_t)(K + 1) * sp.Hse); a.a_replay = take(rb_cap * RS);
    a.a_meter = take(2 * (int64_t)(cfg->train_episodes > 0 ? cfg->train_episodes : 1));
    a.a_gx = take(3 * (int64_t)B * (1 + sp.A)); a.a_bar = take(16); a.a_xtm = take(2 * (int64_t)(B > T ? B : T) * S);
    if (off > arena_stride) return LENV_ERR_WORKSPACE;
    const size_t lds_bytes = wc_lds_bytes(sp);
    if (lds_bytes > 160 * 1024) return LENV_ERR_UNSUPPORTED;
    void (*kern)(const WcArgs) = shape == 2 ? dueling_wavechain_kernel<2> : dueling_wavechain_kernel<1>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
        return LENV_ERR_LAUNCH;
    a.chains = chains;
    a.G = lenv_wc_dueling_team(cfg, shape, chains);
    unsigned grid = (unsigned)chains;
    if (a.G > 1) {
        grid = (unsigned)(8 * ((chains + 7) / 8) * a.G);
        hipLaunchKernelGGL(wct_team_reset_kernel, dim3((unsigned)((chains + 255) / 256)), dim3(256), 0, stream, arena, arena_stride, a.a_bar, chains);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(wc::NT), lds_bytes, stream, a);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}

#ifdef LENV_PHASE_TIMING
extern "C" int lenv_debug_wc_phase_cycles(unsigned long long *host_out)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lenv::g_wc_phase_cycles), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -4;
}
#endif
