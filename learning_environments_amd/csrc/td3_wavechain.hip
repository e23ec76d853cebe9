// td3_wavechain.hip -- the TD3 inner loop of BASELINE configs[4] (HalfCheetah stand-in RewardEnv + TD3, default_config_halfcheetah_
// reward_env.yaml: actor 17-128-128-6, twin critics 23-128-128-1, relu, batch 192, policy_delay 1, one test episode) on the
// wave-chain primitives of lenv_wavechain.cuh.  Same semantics and the same canonical arithmetic order -- hence the same bits -- as
// td3_rn_inner_kernel (which stays the generic path: other shapes / envs, tapes, traces, TD3_vary, ICM), different
// execution structure: a wave owns a block of 32 minibatch samples (6 blocks; batch 256: 8) through a whole network pass.
// The shape table (kT3wShapes) lists the six published configurations the kernel is instantiated for: three RewardEnv shapes (batch 192,
// policy_delay 1) and three VirtualEnv shapes (batch 256, policy_delay 2; venv_step = EnvWrapper.step -> VirtualEnv.step,
// envs/virtual_env.py:43-54, in the kernel).
//
//   reference                                          here
//   TD3.learn  agents/TD3.py:63-116                     t3_forward x7 (actor_target, 2 target critics, 2 critics; actor, critic_1)
//   Actor_TD3 / Critic_Q  models/actor_critic.py:11-19,64-71   + t3_backward x3 (critic_1, critic_2; critic_1 for d/da and the actor)
//   critic_optimizer / actor_optimizer.step, Polyak     wg_adam / wg_polyak over the arena-layout parameter vectors
//   select_train_action / select_test_action (1 row)    actor_row1: per-thread k-ascending chains, coalesced K-major weights
//   delayed policy update  agents/TD3.py:101            policy_step: the actor half of the learn step and all soft updates every policy_delay-th step
//
// Arena layout of ONE network (actor, critic_1, critic_2 alike; every matrix K-MAJOR):
//   W1t[24][128] (rows >= in_dim zero) b1[128] | W2t[128][128] b2[128] | Wo[128][8] (columns >= out_dim zero) bo[8]
#include "lenv_wavechain.cuh"
#include "lenv_wavechain_host.h"

namespace lenv {

using namespace wc;

namespace t3p {
constexpr int R1 = 24;                                                    // rows of W1t
constexpr int oW1t = 0, ob1 = R1 * W, oW2t = ob1 + W, ob2 = oW2t + IMG, oWo = ob2 + W, obo = oWo + 8 * W, PN = obo + 8;
static_assert(PN % 4 == 0, "float4 passes");
}

// The published shapes this kernel is instantiated for (hidden 128 x 2):
// the real env, the agent's activation, the test episodes per test phase, the hidden layers of the reward net / the synthetic env's nets.
// k_rep: same_action_num (env steps per chosen action; RewardEnv: the repeats stop at done, their shaped rewards are summed as python floats)
// B: minibatch rows (sample blocks of 32: 6 or 8), policy_delay: TD3.py:101, venv: the training env is a VirtualEnv (synthetic_env_type 0:
// three nets on cat(action, state), virtual_env.py:43-54) instead of a RewardEnv over the real env, hrn: width of those nets
struct T3wShape { int env, act, T, rn_layers, k_rep, B, policy_delay, venv, hrn; };
constexpr T3wShape kT3wShapes[] = {
    { -1, 0, 1, 1, 1, 192, 1, 0, 128 },
    { LENV_ENV_CHEETAH_STANDIN, LENV_ACT_RELU, 1, 1, 1, 192, 1, 0, 128 },     // 1: default_config_halfcheetah_reward_env.yaml = BASELINE configs[4]
    { LENV_ENV_PENDULUM, LENV_ACT_LEAKYRELU, 10, 2, 1, 192, 1, 0, 128 },      // 2: default_config_pendulum_reward_env.yaml (actor 3-128-128-1, critics 4-128-128-1)
    { LENV_ENV_CMC, LENV_ACT_LEAKYRELU, 1, 1, 2, 192, 1, 0, 128 },            // 3: default_config_cmc_reward_env.yaml (actor 2-128-128-1, critics 3-128-128-1; the episode ends at the flag)
    { LENV_ENV_CMC, LENV_ACT_RELU, 1, 2, 2, 256, 2, 1, 96 },                  // 4: default_config_cmc.yaml (VirtualEnv with three 3-96-96-x nets, batch 256, policy_delay 2)
    { LENV_ENV_PENDULUM, LENV_ACT_RELU, 10, 2, 1, 256, 2, 1, 32 },            // 5: default_config_pendulum.yaml's td3 section (VirtualEnv 4-32-32-x; agent td3 = td3_vary without vary_hp)
    { LENV_ENV_CHEETAH_STANDIN, LENV_ACT_RELU, 10, 3, 1, 256, 2, 1, 128 },    // 6: default_config_halfcheetah.yaml's td3 section (VirtualEnv 23-128-128-128-x)
};

// dumps: [NBK] blocks of BLK floats each (register order), or [B][128] row-major copies (same size); NBK = B / 32 sample blocks
enum { TD_C1_H1 = 0, TD_C1_H2, TD_C2_H1, TD_C2_H2, TD_A_H1, TD_A_H2, TR_C1_H2, TR_C2_H2, TR_A_H2, TS_DZ2, TR_DZ2, TR_DH1, TR_DZ2B, TR_DH1B, T3W_NDUMP };

struct T3wArgs {
    lenv_td3_cfg cfg;
    const float *theta, *eps; const int32_t *worker; const float *sign;
    const float *agent_init; const uint64_t *rng_keys;
    float *arena; int64_t arena_stride;
    lenv_td3_out out;
    int64_t rb_cap; int RS;
    int P, Pa, Pc, P_rn;
    int64_t a_par, a_xc, a_xn, a_xa, a_th, a_dump, a_replay, a_meter, a_gx, a_bar, a_w2u, a_rn;
    int G;                                   // workgroups per chain (team): 1, 2, 3 or 6 (batch 192); 1, 2, 4 or 8 (batch 256)
    int64_t chains;
};

struct T3wCtx {
    float *bufA, *bufB, *sm_b, *sm_wo, *sm_bo, *qvec, *dzl, *sm_b2, *sm_wo2;
    float *params, *targets, *grad, *dumps;
    float prelu, ma;
    int g, G;                                // this workgroup's place in its chain's team
    // the team path (td3_wavechain_team.cuh): unit-major copies of the three online W2, the LDS control block (Adam's bias corrections of
    // the step at ctrl[20..23]) and the optimizer constants
    float *w2u, *ctrl;
    float w1, w2, beta2, aeps, tau, omt;
    // the lock-step test episodes (t3w_test_steps): env states [T][SD], returns, lengths; exploration noise scale, steps per episode
    double *xt_d, *ret;
    float *ep_rew;
    int *tlen;
    float action_std;
    int max_steps;
    // the team path's learn step (t3v_learn_step, one out-of-line routine per TD3.learn): the chain's arena and the offsets of its
    // minibatch / exchange arrays, the replay ring, the chain's RNG key, the TD target's constants, the team's barrier words
    float *arena;
    int64_t a_xc, a_xn, a_xa, a_th, a_gx, a_replay;
    int RS, same_xcd;
    uint32_t key_lo, key_hi;
    float gamma, policy_std, policy_clip;
    unsigned *team_bar, *launch_dead;
};

// state-dict index inside ONE net (mlp_off order: W0 [128 x in] b0 W1 [128 x 128] b1 Wout [out x 128] bout) -> arena-layout index
__device__ __forceinline__ int t3w_sd_to_arena(int p, int in, int out)
{
    using namespace t3p;
    int o = p;
    if (o < W * in) { const int j = o / in, k = o - j * in; return oW1t + k * W + j; }
    o -= W * in;
    if (o < W) return ob1 + o;
    o -= W;
    if (o < IMG) { const int j = o >> 7, k = o & 127; return oW2t + k * W + j; }
    o -= IMG;
    if (o < W) return ob2 + o;
    o -= W;
    if (o < out * W) { const int c = o >> 7, k = o & 127; return oWo + k * 8 + c; }
    o -= out * W;
    return obo + o;
}

#ifdef LENV_PHASE_TIMING
__device__ unsigned long long g_t3w_phase_cycles[48];
// sub-phase marks inside the out-of-line routines: their own switch (-DLENV_PHASE_TIMING_SUB; together with the kernel-level marks the
// build of this file trips a backend assertion of the ROCm 7.2 compiler)
#ifdef LENV_PHASE_TIMING_SUB
#define TSUB_DECL unsigned long long sp_last = __builtin_readcyclecounter()
#define TSUB_MARK(i) do { unsigned long long sp_now = __builtin_readcyclecounter(); if (blockIdx.x == 0 && threadIdx.x == 0) g_t3w_phase_cycles[i] += sp_now - sp_last; sp_last = sp_now; } while (0)
#else
#define TSUB_DECL
#define TSUB_MARK(i)
#endif
#define TPT_DECL unsigned long long pt_last = __builtin_readcyclecounter(), pt_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define TPT_MARK(i) do { unsigned long long pt_now = __builtin_readcyclecounter(); pt_acc[i] += pt_now - pt_last; pt_last = pt_now; } while (0)
#ifdef LENV_PHASE_TIMING_SUB
#undef TPT_DECL
#undef TPT_MARK
#define TPT_DECL
#define TPT_MARK(i)
#endif
#else
#define TPT_DECL
#define TPT_MARK(i)
#define TSUB_DECL
#define TSUB_MARK(i)
#endif

#ifdef LENV_PHASE_TIMING_SUB
#define KSUB_RESET ksub_last = __builtin_readcyclecounter()
#define KSUB_MARK(i) do { unsigned long long k_now = __builtin_readcyclecounter(); if (blockIdx.x == 0 && threadIdx.x == 0) g_t3w_phase_cycles[i] += k_now - ksub_last; ksub_last = k_now; } while (0)
#else
#define KSUB_RESET
#define KSUB_MARK(i)
#endif

#define T3W_CTX_PROLOGUE                                                                                                                   \
    using namespace t3p;                                                                                                                   \
    Lane L;                                                                                                                                \
    L.init();                                                                                                                              \
    const int tid = L.tid, wave = L.wave;                                                                                                  \
    (void)tid; (void)wave;                                                                                                                 \
    typedef __attribute__((address_space(3))) const T3wCtx LCtx;                                                                           \
    LCtx *c = (LCtx *)uni_ptr(ctx_);                                                                                                       \
    float *bufA = uni_ptr(c->bufA), *bufB = uni_ptr(c->bufB);                                                                              \
    lfloat *sm_b = (lfloat *)uni_ptr(c->sm_b), *sm_wo = (lfloat *)uni_ptr(c->sm_wo), *sm_bo = (lfloat *)uni_ptr(c->sm_bo);                 \
    float *dumps = uni_ptr(c->dumps);                                                                                                      \
    const float prelu = unif(c->prelu), ma = unif(c->ma);                                                                                  \
    const int tg = uni(c->g), TG = uni(c->G);                                                                                              \
    /* slot s of an n-slot phase (sample blocks: NBK, tile / vector waves: 8) runs on wave s of the team's workgroup s * G / n */          \
    auto mine = [&](int s_, int n_) { return TG == 1 || (s_ * TG) / n_ == tg; };                                                          \
    (void)bufA; (void)bufB; (void)sm_b; (void)sm_wo; (void)sm_bo; (void)dumps; (void)prelu; (void)ma;                                      \
    auto dump_of = [&](int which, int blk) { return dumps + ((int64_t)which * NBK + blk) * BLK; };                                          \
    (void)dump_of; (void)mine

// [32 samples x 128 units] register block -> rows 32 blk .. of a plain [sample][unit] array (16-byte stores)
__device__ __forceinline__ void block_to_rowmajor(float *rm_, int blk, const Lane &L, const float (&r)[64])
{
    gfloat *rm = (gfloat *)rm_ + (32 * blk + L.li) * W + 4 * L.h;
#pragma unroll
    for (int pc = 0; pc < 16; ++pc) *(gf4 *)(rm + 32 * (pc >> 2) + 8 * (pc & 3)) = f32x4{r[4 * pc], r[4 * pc + 1], r[4 * pc + 2], r[4 * pc + 3]};
}

// ---- one network pass over the B = 32 NBK minibatch rows X[i][ldx] (waves 0 .. NBK-1 own the sample blocks) ----------------------
// mode 0 (Critic_Q): q_out[i] = net(x)      mode 1 (Actor_TD3): Y[i][ocol + c] = tanh(net(x)) * max_action, th_out[i][c] = tanh
// A team member owns at most three of the six blocks, so a SECOND, independent critic pass (par2 != null: its own parameters, inputs,
// outputs and dumps; image in bufB, small vectors in the second LDS set) runs next to the first one on the waves behind the member's blocks.
template <int ACT, int IN, int OUT, int NBK>
__device__ __noinline__ void t3w_forward(const T3wCtx *ctx_, const float *par_, const float *X_, int ldx_, int mode_, float *q_out_,
                                         float *Y_, int ldy_, int ocol_, float *th_out_, int d_h1_, int d_h2_, int r_h2_,
                                         const float *par2_ = nullptr, const float *X2_ = nullptr, float *q_out2_ = nullptr, int d_h1_2_ = -2, int r_h2_2_ = -1)
{
    T3W_CTX_PROLOGUE;
    const bool dual = uni(d_h1_2_) != -2;                  // (a flag rather than a null test of the generic pointer)
    (void)d_h2_;
    // This wave's jobs (pass, block).  Team member: (pass 0, block = wave) or, in a dual call, (pass 1, block = wave - 6 / G mod 8: the
    // member's blocks are consecutive, so pass 1 sits on the waves right behind them -- other SIMDs than the blocks' own waves).
    // One workgroup per chain: a single pass puts block w on wave w; a dual pass has 2 NBK = 12 jobs (pass j / 6, block j % 6) for 8 waves:
    // wave w runs job w and, for w < 4, job w + 8 afterwards -- three jobs on every SIMD instead of four for two passes in a row
    // (batch 256: 16 jobs, two per wave).
    int njobs = 0, jp0 = 0, jb0 = 0, jp1 = 0, jb1 = 0;
    if (TG > 1) {
        const int wave2 = (wave - NBK / TG) & 7;
        if (wave < NBK && mine(wave, NBK)) { jp0 = 0; jb0 = wave; njobs = 1; }
        else if (dual && wave2 < NBK && mine(wave2, NBK)) { jp0 = 1; jb0 = wave2; njobs = 1; }
    } else if (dual) {
        jp0 = wave / NBK; jb0 = wave % NBK; njobs = 1;
        if (wave < 2 * NBK - 8) { jp1 = (wave + 8) / NBK; jb1 = (wave + 8) % NBK; njobs = 2; }
    } else if (wave < NBK) { jb0 = wave; njobs = 1; }
    float *Y = uni_ptr(Y_), *th_out = uni_ptr(th_out_);
    constexpr int in = IN, out = OUT;
    const int ldx = uni(ldx_), mode = uni(mode_), ldy = uni(ldy_), ocol = uni(ocol_);
    lfloat *sm_b2 = (lfloat *)uni_ptr(c->sm_b2), *sm_wo2 = (lfloat *)uni_ptr(c->sm_wo2);
    {
        const float *p0 = uni_ptr(par_);
        for (int i = tid; i < 2 * W; i += NT) sm_b[i] = p0[(i < W ? ob1 : ob2 - W) + i];
        for (int i = tid; i < 8 * W + 8; i += NT) sm_wo[i] = p0[oWo + i];                // Wo and bo are contiguous in both places
        if (dual) {
            const float *p1 = uni_ptr(par2_);
            for (int i = tid; i < 2 * W; i += NT) sm_b2[i] = p1[(i < W ? ob1 : ob2 - W) + i];
            for (int i = tid; i < 8 * W + 8; i += NT) sm_wo2[i] = p1[oWo + i];
        }
    }
    StageRegs sr;
    TSUB_DECL;
    // the first job's layer-1 operands (A straight from the K-major array, B from the minibatch rows) are requested first: their
    // latency hides behind the staging of the W2 image(s)
    float xb[in >> 1], wa[in >> 1][4];
    auto l1_operands = [&](int jpass, int jblk) {
        const float *par = uni_ptr(jpass ? par2_ : par_), *X = uni_ptr(jpass ? X2_ : X_);
        const int row0 = 32 * jblk + L.li;
        const gfloat *w1 = (const gfloat *)par + oW1t + L.h * W + L.li;
        const gfloat *xr = (const gfloat *)X + row0 * ldx + L.h;
#pragma unroll
        for (int t = 0; t < (in >> 1); ++t) {
            xb[t] = xr[2 * t];
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) wa[t][jt] = w1[2 * t * W + 32 * jt];
        }
    };
    l1_operands(jp0, jb0);
    stage_load_direct(uni_ptr(par_) + oW2t, L, sr);
    stage_store_direct(bufA, L, sr);
    if (dual) {
        stage_load_direct(uni_ptr(par2_) + oW2t, L, sr);
        stage_store_direct(bufB, L, sr);
    }
    barrier_lds();                                         // images + small vectors: LDS only (the layer-1 operand loads stay in flight)
    TSUB_MARK(16);
    auto run_job = [&](int jpass, int blk) {
        const bool p1 = jpass != 0;
        const float *par = uni_ptr(p1 ? par2_ : par_), *X = uni_ptr(p1 ? X2_ : X_);
        lfloat *q_out = (lfloat *)uni_ptr(p1 ? q_out2_ : q_out_);
        const int d_h1 = uni(p1 ? d_h1_2_ : d_h1_), r_h2 = uni(p1 ? r_h2_2_ : r_h2_);
        const lfloat *smb = p1 ? sm_b2 : sm_b, *smwo = p1 ? sm_wo2 : sm_wo, *smbo = smwo + 8 * W;
        const float *img = p1 ? bufB : bufA;
        const int row = 32 * blk + L.li;
        float r[64];
        f32x16 acc[4];
        acc_zero(acc);
        {   // layer 1 (K = in, 17 or 23); an odd last k is one fmaf per output
#pragma unroll
            for (int t = 0; t < (in >> 1); ++t)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[t][jt], xb[t], acc[jt], 0, 0, 0);
            if (in & 1) {
                const float xl = ((const gfloat *)X)[row * ldx + in - 1];
                const gfloat *wl = (const gfloat *)par + oW1t + (in - 1) * W + 4 * L.h;
#pragma unroll
                for (int pc = 0; pc < 16; ++pc) {
                    const f32x4 wv = *(const gf4 *)(wl + 32 * (pc >> 2) + 8 * (pc & 3));
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) acc[pc >> 2][4 * (pc & 3) + cc] = fma32(xl, wv[cc], acc[pc >> 2][4 * (pc & 3) + cc]);
                }
            }
        }
        tile_bias_act<ACT>(acc, (const float *)smb, L, prelu, r);
        if (d_h1 >= 0) dump_store(dump_of(d_h1, blk), L, r);
        tile_to_operand(r);
        TSUB_MARK(17);
        acc_zero(acc);
        chain128(img, L, r, acc);
        TSUB_MARK(18);
        tile_bias_act<ACT>(acc, (const float *)(smb + W), L, prelu, r);
        if (r_h2 >= 0) block_to_rowmajor(dump_of(r_h2, 0), blk, L, r);     // h2 is kept as a plain [sample][unit] array only
        tile_to_operand(r);
        // output layer: rows 0 .. out-1 of one 32x32 tile (row c of lane half h = 4 h + register)
        f32x16 hacc;
#pragma unroll
        for (int v = 0; v < 16; ++v) hacc[v] = 0.0f;
        const lfloat *wo = smwo + L.h * 8 + (L.li < out ? L.li : out - 1);
#pragma unroll
        for (int t = 0; t < 64; ++t) hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(wo[2 * t * 8], r[breg_of(t)], hacc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int cidx = 4 * L.h + v;
            if (cidx < out) {
                const float z = hacc[v] + smbo[cidx];
                if (mode == 0) q_out[row] = z;
                else {
                    const float th = det_tanhf(lenv_tanh_table, z);
                    if (th_out) th_out[row * out + cidx] = th;
                    Y[row * ldy + ocol + cidx] = th * ma;
                }
            }
        }
        TSUB_MARK(19);
    };
    if (njobs >= 1) run_job(jp0, jb0);
    if (njobs >= 2) { L.refresh(); l1_operands(jp1, jb1); run_job(jp1, jb1); }
    __syncthreads();
    TSUB_MARK(20);
}

// ---- backward of one network, first half: the per-sample chain from dOut[i][out] (LDS; the rows of this workgroup's blocks) back to
// dz2 and dh1 (row-major copies r_dz2 / r_dh1 in the arena for the weight gradients) and, for the policy step, the action part of the
// input gradient turned into the actor's output gradient dz (LDS) right away ----
template <int ACT, int IN, int OUT, int NBK>
__device__ __noinline__ void t3w_backward_chain(const T3wCtx *ctx_, const float *par_, const float *dOut_, int d_h1_, int d_h2_, int r_dz2_, int r_dh1_,
                                                int dx_col_, int dx_n_, const float *th_, float *dz_out_,
                                                const float *par2_ = nullptr, const float *dOut2_ = nullptr, int d_h1_2_ = -2, int d_h2_2_ = -1,
                                                int r_dz2_2_ = -1, int r_dh1_2_ = -1)
{
    T3W_CTX_PROLOGUE;
    // a second, independent network's chain (d_h1_2 given) runs next to the first one: jobs as in t3w_forward
    const bool dual = uni(d_h1_2_) != -2;
    int njobs = 0, jp0 = 0, jb0 = 0, jp1 = 0, jb1 = 0;
    if (TG > 1) {
        const int wave2 = (wave - NBK / TG) & 7;
        if (wave < NBK && mine(wave, NBK)) { jp0 = 0; jb0 = wave; njobs = 1; }
        else if (dual && wave2 < NBK && mine(wave2, NBK)) { jp0 = 1; jb0 = wave2; njobs = 1; }
    } else if (dual) {
        jp0 = wave / NBK; jb0 = wave % NBK; njobs = 1;
        if (wave < 2 * NBK - 8) { jp1 = (wave + 8) / NBK; jb1 = (wave + 8) % NBK; njobs = 2; }
    } else if (wave < NBK) { jb0 = wave; njobs = 1; }
    const float *th = uni_ptr(th_);
    lfloat *dz_out = (lfloat *)uni_ptr(dz_out_);
    constexpr int out = OUT;
    const int dx_col = uni(dx_col_), dx_n = uni(dx_n_);
    lfloat *sm_wo2 = (lfloat *)uni_ptr(c->sm_wo2);
    for (int i = tid; i < 8 * W + 8; i += NT) sm_wo[i] = uni_ptr(par_)[oWo + i];
    if (dual) for (int i = tid; i < 8 * W + 8; i += NT) sm_wo2[i] = uni_ptr(par2_)[oWo + i];
    StageRegs sr, sr2;
    TSUB_DECL;
    stage_load_transposed(uni_ptr(par_) + oW2t, L, sr);
    if (dual) stage_load_transposed(uni_ptr(par2_) + oW2t, L, sr2);
    __syncthreads();                                       // sm_wo staged
    TSUB_MARK(24);
    // dz2 = act'(h2) * (sum_c dOut[i][c] Wo[c][unit], c ascending from 0) of every job of this wave: to the row-major copy the weight
    // gradients read, and -- as the chain's B operand -- kept in registers (first job) or re-read from that copy (second job)
    float r[64];
    f32x16 acc[4];
    auto dz2_block = [&](int jpass, int blk) {
        const bool p1 = jpass != 0;
        const lfloat *dOut = (const lfloat *)uni_ptr(p1 ? dOut2_ : dOut_);
        const lfloat *smwo = p1 ? sm_wo2 : sm_wo;
        const int d_h2 = uni(p1 ? d_h2_2_ : d_h2_), r_dz2 = uni(p1 ? r_dz2_2_ : r_dz2_);
        const int row = 32 * blk + L.li;
        const gfloat *hd = (const gfloat *)dump_of(d_h2, 0) + row * W + 4 * L.h;        // d_h2: the row-major copy of h2
        float dO[out];
#pragma unroll
        for (int cc = 0; cc < out; ++cc) dO[cc] = dOut[row * out + cc];
#pragma unroll
        for (int pc = 0; pc < 16; ++pc) {
            const f32x4 hv = *(const gf4 *)(hd + 32 * (pc >> 2) + 8 * (pc & 3));
            const lfloat *wp = smwo + (32 * (pc >> 2) + 8 * (pc & 3) + 4 * L.h) * 8;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                float wv[8];
                const f32x4 w0 = *(const lf4 *)(wp + 8 * cc);
                wv[0] = w0[0]; wv[1] = w0[1]; wv[2] = w0[2]; wv[3] = w0[3];
                if (out > 4) { const f32x4 w1 = *(const lf4 *)(wp + 8 * cc + 4); wv[4] = w1[0]; wv[5] = w1[1]; wv[6] = w1[2]; wv[7] = w1[3]; }
                float up = 0.0f;
#pragma unroll
                for (int o = 0; o < out; ++o) up = fma32(dO[o], wv[o], up);
                r[4 * pc + cc] = act_bwd(ACT, prelu, hv[cc], up);
            }
        }
        block_to_rowmajor(dump_of(r_dz2, 0), blk, L, r);
        tile_to_operand(r);
    };
    if (njobs > 0) dz2_block(jp0, jb0);
    stage_store_transposed(bufA, L, sr);
    if (dual) stage_store_transposed(bufB, L, sr2);
    barrier_lds();                                         // the chain reads the image only; the dz2 copy in the arena is read by the weight gradients
    TSUB_MARK(25);
    L.refresh();
    auto run_job = [&](int jpass, int blk) {
        const bool p1 = jpass != 0;
        const float *par = uni_ptr(p1 ? par2_ : par_);
        const int d_h1 = uni(p1 ? d_h1_2_ : d_h1_), r_dh1 = uni(p1 ? r_dh1_2_ : r_dh1_);
        const float *img = p1 ? bufB : bufA;
        const int row = 32 * blk + L.li;
        const gf4 *src = (const gf4 *)dump_of(d_h1, blk) + L.lane;
        f32x4 hv[16];
#pragma unroll
        for (int pc = 0; pc < 16; ++pc) hv[pc] = src[pc * 64];
        acc_zero(acc);
        chain128(img, L, r, acc);
#pragma unroll
        for (int pc = 0; pc < 16; ++pc)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) r[4 * pc + cc] = act_bwd(ACT, prelu, hv[pc][cc], acc[pc >> 2][4 * (pc & 3) + cc]);
        if (r_dh1 >= 0) block_to_rowmajor(dump_of(r_dh1, 0), blk, L, r);
        if (dx_n > 0) {
            // dX[i][dx_col + c] = sum_u dh1[i][u] W1[u][dx_col + c] (u ascending), c < dx_n, then dz = (dX * max_action) * (1 - th^2)
            tile_to_operand(r);
            f32x16 hacc;
#pragma unroll
            for (int v = 0; v < 16; ++v) hacc[v] = 0.0f;
            const gfloat *w1 = (const gfloat *)par + oW1t + (dx_col + (L.li < dx_n ? L.li : dx_n - 1)) * W + L.h;
#pragma unroll 16
            for (int t = 0; t < 64; ++t) hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(w1[2 * t], r[breg_of(t)], hacc, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int cidx = 4 * L.h + v;
                if (cidx < dx_n) {
                    const float t_ = th[row * dx_n + cidx];
                    dz_out[row * dx_n + cidx] = (hacc[v] * ma) * fma32(-t_, t_, 1.0f);
                }
            }
        }
    };
    if (njobs >= 1) run_job(jp0, jb0);
    if (njobs >= 2) { L.refresh(); dz2_block(jp1, jb1); run_job(jp1, jb1); }
    __syncthreads();
    TSUB_MARK(26);
}

// ---- backward of one network, second half: the parameter gradients from the row-major copies of ALL B samples (dOut in LDS, h2 /
// dz2 / dh1 in the arena, the h1 register dumps).  The work is cut by wave -- output-layer gradients (VALU), two W2 tiles per wave,
// one W1 column tile or one bias vector per wave -- and wave w's share runs in the team's workgroup w * G / 8 ----
template <int ACT, int IN, int OUT, int NBK>
__device__ __noinline__ void t3w_backward_wgrad(const T3wCtx *ctx_, float *gpar_, const float *X_, int ldx_, const float *dOut_, int d_h1_, int r_h2_,
                                                int r_dz2_, int r_dh1_, int slot0_, int slots_)
{
    T3W_CTX_PROLOGUE;
    // the team deals `slots` wave slots (8 per network of this phase: the two critics' gradients run side by side in different
    // members); a member without a slot of this network skips the call, image staging included
    const int slot0 = uni(slot0_), slots = uni(slots_);
    if (TG > 1 && ((slot0 + NW - 1) * TG) / slots < tg) return;
    if (TG > 1 && (slot0 * TG) / slots > tg) return;
    const float *X = uni_ptr(X_);
    float *gpar = uni_ptr(gpar_);
    const lfloat *dOut = (const lfloat *)uni_ptr(dOut_);
    constexpr int in = IN, out = OUT;
    const int ldx = uni(ldx_), d_h1 = uni(d_h1_), r_h2 = uni(r_h2_), r_dz2 = uni(r_dz2_), r_dh1 = uni(r_dh1_);
    constexpr int B = 32 * NBK, HB = B / 2, NST = HB * 32 / NT;     // rows, rows of a half-batch, 16-byte pieces per thread of a half-batch image
    static_assert(HB <= W && HB * 32 % NT == 0, "a half-batch fills at most one image");
    const bool my_wave = mine(slot0 + wave, slots);
    TSUB_DECL;
    if (my_wave) {
        // output layer: gWo[k][c] = sum_i dOut[i][c] h2[i][k] (i ascending) from the row-major copy of h2; gbo[c] = sum_i dOut[i][c]
        const gfloat *rm = (const gfloat *)dump_of(r_h2, 0);
        for (int e = tid; e < W * out; e += NT) {
            const int k = e & 127, cidx = e >> 7;
            float s = 0.0f;
            for (int i0 = 0; i0 < B; i0 += 64) {
                float hv[64];
#pragma unroll
                for (int u = 0; u < 64; ++u) hv[u] = rm[(i0 + u) * W + k];
#pragma unroll
                for (int u = 0; u < 64; ++u) s = fma32(dOut[(i0 + u) * out + cidx], hv[u], s);
            }
            gpar[oWo + k * 8 + cidx] = s;
        }
        if (tid >= NT - 64 && tid < NT - 64 + out) {
            const int cidx = tid - (NT - 64);
            float s = 0.0f;
            for (int i = 0; i < B; ++i) s = s + dOut[i * out + cidx];
            gpar[obo + cidx] = s;
        }
    }
    TSUB_MARK(24);
    float r[64];
    // gW2t[k][j] = sum_i h1[i][k] dz2[i][j] over the B samples: two half-batches of B / 2 (96 / 128) rows through the two images
    f32x16 acc0, acc1;
#pragma unroll
    for (int v = 0; v < 16; ++v) { acc0[v] = 0.0f; acc1[v] = 0.0f; }
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        L.refresh();
        {   // dz2 rows HB half .. + HB - 1: straight copy of the row-major array into the swizzled image (bufB)
            const gf4 *src = (const gf4 *)dump_of(r_dz2, 0) + half * HB * 32 + tid;
            lfloat *img = (lfloat *)bufB;
            f32x4 v[NST];
#pragma unroll
            for (int u = 0; u < NST; ++u) v[u] = src[u * NT];
#pragma unroll
            for (int u = 0; u < NST; ++u) {
                const int p = tid + u * NT, rr_ = p >> 5, cc_ = (p & 31) << 2;
                *(lf4 *)(img + rr_ * W + (cc_ ^ ((rr_ & 7) << 2))) = v[u];
            }
        }
        if (wave < NBK / 2) {                              // h1 blocks (NBK / 2) half .. from the register dumps -> bufA
            dump_load(dump_of(d_h1, (NBK / 2) * half + wave), L, r);
            tile_to_image(bufA, wave, L, r);
        }
        barrier_lds();
        L.refresh();
        if (my_wave) wgrad_accum(bufA, bufB, HB, L, acc0, acc1);
        barrier_lds();
    }
    if (my_wave) wgrad_store(L, gpar + oW2t, acc0, acc1);
    TSUB_MARK(27);
    // layer 1: gW1t[k][j] = sum_i x[i][k] dh1[i][j] (k < in) on the matrix cores too: A = the minibatch inputs straight from the arena
    // (lane = input column k, clamped), B = an image of dh1 built from its row-major copy, again in two half-batches; waves 0-3 own
    // one 32-unit column tile each, waves 4-5 walk the same images for gb1[j] = sum_i dh1[i][j], waves 6-7 an image of dz2 for gb2
    f32x16 accw;
#pragma unroll
    for (int v = 0; v < 16; ++v) accw[v] = 0.0f;
    float sb = 0.0f;
    bool need_dh1 = false, need_dz2 = false;               // uniform per workgroup
    for (int w_ = 0; w_ < NW; ++w_) if (mine(slot0 + w_, slots)) { if (w_ < 6) need_dh1 = true; else need_dz2 = true; }
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        L.refresh();
        {   // half-batch images of dh1 (bufB: waves 0-5 read it) and dz2 (bufA: waves 6-7) from their row-major copies; a team member
            // stages only what its waves of this phase read
            const gf4 *src = (const gf4 *)dump_of(r_dh1, 0) + half * HB * 32 + tid, *src2 = (const gf4 *)dump_of(r_dz2, 0) + half * HB * 32 + tid;
            lfloat *img = (lfloat *)bufB, *img2 = (lfloat *)bufA;
            f32x4 v[NST], v2[NST];
            if (need_dh1) {
#pragma unroll
                for (int u = 0; u < NST; ++u) v[u] = src[u * NT];
            }
            if (need_dz2) {
#pragma unroll
                for (int u = 0; u < NST; ++u) v2[u] = src2[u * NT];
            }
#pragma unroll
            for (int u = 0; u < NST; ++u) {
                const int p = tid + u * NT, rr_ = p >> 5, cc_ = (p & 31) << 2;
                if (need_dh1) *(lf4 *)(img + rr_ * W + (cc_ ^ ((rr_ & 7) << 2))) = v[u];
                if (need_dz2) *(lf4 *)(img2 + rr_ * W + (cc_ ^ ((rr_ & 7) << 2))) = v2[u];
            }
        }
        barrier_lds();
        L.refresh();
        if (!my_wave) { /* another workgroup of the team runs this wave's tile / vector */ }
        else if (wave < 4) {
            const lfloat *img = (const lfloat *)bufB;
            const lfloat *pb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) pb[q] = img + L.h * W + 32 * wave + L.colsw[q];
            const gfloat *xa_ = (const gfloat *)X + (HB * half + L.h) * ldx + (L.li < in ? L.li : in - 1);
#pragma unroll 2
            for (int t4 = 0; t4 < HB / 2; t4 += 4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) accw = __builtin_amdgcn_mfma_f32_32x32x2f32(xa_[2 * (t4 + q) * ldx], pb[q][2 * (t4 + q) * W], accw, 0, 0, 0);
            }
        } else {                                       // waves 4,5: gb1 from the dh1 image; waves 6,7: gb2 from the dz2 image
            const lfloat *img = (const lfloat *)(wave < 6 ? bufB : bufA);
            const int j = tid & 127;
            for (int i0 = 0; i0 < HB; i0 += 32) {
                float x[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) x[u] = img[(i0 + u) * W + (j ^ ((u & 7) << 2))];
#pragma unroll
                for (int u = 0; u < 32; ++u) sb = sb + x[u];
            }
        }
        barrier_lds();
    }
    if (!my_wave) { }
    else if (wave < 4) {
        gfloat *out_ = (gfloat *)gpar + oW1t + (4 * L.h) * W + 32 * wave + L.li;      // rows k = 8 (v / 4) + 4 h + v % 4 of the K-major gradient
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int k = 8 * (v >> 2) + 4 * L.h + (v & 3);
            if (k < in) out_[(8 * (v >> 2) + (v & 3)) * W] = accw[v];
        }
    } else gpar[(wave < 6 ? ob1 : ob2) + (tid & 127)] = sb;
    __syncthreads();
    TSUB_MARK(28);
}

}  // namespace lenv
#include "td3_wavechain_team.cuh"
namespace lenv {

// ---- the test episodes of one test phase in LOCK-STEP (shapes with T > 1 episodes per phase; BaseAgent.test, base_agent.py:155-227, with
// TD3.select_test_action TD3.py:126-129): row e < T = episode e.  The actor does not change during a test phase, so each wave keeps its
// tile of the 128x128 layer in registers for all max_steps forwards (the thin 16-sample product of lenv_wavechain.cuh: k-ascending chains,
// the bits of the one-row actor); layer 1 and the output layer are per-(unit, row) / per-(row, output) fmaf chains.  The envs of this
// kernel never terminate: every episode takes max_steps steps; reset rows and noise rows have the fixed indices of td3_rn_inner_kernel's
// lock-step rollouts.  Uses bufA (the team path reloads its LDS actor afterwards).
template <int SHAPE> __device__ __noinline__ void t3w_test_steps(const T3wCtx *ctx_, uint32_t key_lo_, uint32_t key_hi_, int first_episode_, uint32_t noise_lo_, uint32_t noise_hi_)
{
    using namespace t3p;
    constexpr T3wShape SP = kT3wShapes[SHAPE];
    using EnvT = ContEnv<SP.env>;
    constexpr int S = EnvT::S, A = EnvT::A, SD = EnvT::SD, ACT = SP.act, T = SP.T, IW = 16, NB = 32;      // NB: steps per noise batch
    constexpr bool CHEETAH = SP.env == LENV_ENV_CHEETAH_STANDIN;
    static_assert(T <= 16 && T * SD <= NT, "one 16-sample tile");
    Lane L;
    L.init();
    const int tid = L.tid, wave = L.wave;
    typedef __attribute__((address_space(3))) const T3wCtx LCtx;
    typedef __attribute__((address_space(3))) double ldouble;
    typedef __attribute__((address_space(3))) int lint;
    LCtx *c = (LCtx *)uni_ptr(ctx_);
    float *bufA = uni_ptr(c->bufA);
    const float *par = uni_ptr(c->params);                 // the actor = net 0 of the arena
    const float prelu = unif(c->prelu), ma = unif(c->ma), astd = unif(c->action_std);
    ldouble *xt_d = (ldouble *)uni_ptr(c->xt_d), *ret = (ldouble *)uni_ptr(c->ret);
    lfloat *ep_rew = (lfloat *)uni_ptr(c->ep_rew);
    lint *tlen = (lint *)uni_ptr(c->tlen);
    const int max_steps = uni(c->max_steps);
    const uint64_t key = ((uint64_t)uni((int)key_hi_) << 32) | (uint32_t)uni((int)key_lo_);
    const int first_episode = uni(first_episode_);
    const int64_t noise_base = (int64_t)(((uint64_t)uni((int)noise_hi_) << 32) | (uint32_t)uni((int)noise_lo_));
    float *imgX = bufA, *imgY = bufA + IW * W;
    lfloat *Xl = (lfloat *)(bufA + 2 * IW * W);           // observation rows [T][S]
    lfloat *at = Xl + 16 * 20;                             // actions [T][A]
    lfloat *nzb = at + 16 * 8;                             // noise batch [NB][T][A]
    lfloat *whl = nzb + NB * 16 * 8;                       // output layer [128][8] + bias [8]
    for (int e = tid; e < T * SD; e += NT) { const int te = e / SD; xt_d[e] = EnvT::reset_word(key, STREAM_TEST_RESET, (int64_t)first_episode + te, e - te * SD); }
    if (tid < T) ep_rew[tid] = 0.0f;
    float a2[32];
    thin_load16(par + oW2t, wave, L, a2);
    const int j = tid & (W - 1);
    float w[S];
#pragma unroll
    for (int k = 0; k < S; ++k) w[k] = ((const gfloat *)par)[oW1t + k * W + j];
    const float bj = ((const gfloat *)par)[ob1 + j];
    const f32x4 bv2 = thin_bias16(par + ob2, wave, L);
    for (int e = tid; e < 8 * W + 8; e += NT) whl[e] = ((const gfloat *)par)[oWo + e];      // (Wo and bo are contiguous)
    // the stand-in's dynamics constants (A [17][17], B [17][6], c [17] doubles) in LDS instead of 23 scattered 8-byte global loads per word
    ldouble *chA = (ldouble *)(whl + 8 * W + 8), *chB = chA + 17 * 17, *chC = chB + 17 * 6;
    if constexpr (CHEETAH) {
        for (int i = tid; i < 17 * 17; i += NT) chA[i] = lenv_cheetah_A[i];
        if (tid < 17 * 6) chB[tid] = lenv_cheetah_B[tid];
        if (tid < 17) chC[tid] = lenv_cheetah_c[tid];
    }
    // word i of x' = clip(c + A x + B a, -10, 10): EnvT::step_word with the constants from LDS (same operations in the same order)
    auto step_word = [&](int i, const ldouble *x, const lfloat *av) -> double {
        if constexpr (CHEETAH) {
            double acc = chC[i];
#pragma unroll
            for (int jj = 0; jj < 17; ++jj) acc = acc + chA[i * 17 + jj] * x[jj];
#pragma unroll
            for (int k = 0; k < 6; ++k) acc = acc + chB[i * 6 + k] * (double)av[k];
            return acc < -10.0 ? -10.0 : (acc > 10.0 ? 10.0 : acc);
        } else return EnvT::step_word(i, (const double *)x, (const float *)av);
    };
    __syncthreads();
    if (tid < T * S) { const int te = tid / S; Xl[tid] = EnvT::obs(tid - te * S, (const double *)(xt_d + te * SD)); }
    __syncthreads();
    for (int ai = 0; ai < max_steps; ++ai) {
        if ((ai & (NB - 1)) == 0) {                        // exploration noise of the next NB steps: (episode te, step ai) has row noise_base + te * max_steps + ai
            for (int e = tid; e < NB * T * A; e += NT) {
                const int st = e / (T * A), r = e - st * (T * A), te = r / A, k = r - te * A;
                const int64_t n = (noise_base + (int64_t)te * max_steps + ai + st) * A + k;
                nzb[e] = (ai + st < max_steps) ? (float)det_normal(key, STREAM_TD3_TEST_NOISE, (uint64_t)n) : 0.0f;
            }
        }
        for (int i = tid >> 7; i < T; i += NT >> 7) {    // layer 1 (K = S): one thread per (unit, row)
            float z = 0.0f;
#pragma unroll
            for (int k = 0; k < S; ++k) z = fma32(Xl[i * S + k], w[k], z);
            ((lfloat *)imgX)[j * IW + i] = act_fwd(ACT, prelu, z + bj);
        }
        __syncthreads();
        thin_layer16v<ACT, 1>(a2, bv2, imgY, a2, bv2, nullptr, imgX, wave, L, prelu);
        __syncthreads();
        if (tid < T * A) {                                 // output layer + tanh, exploration noise, clamp
            const int i = tid / A, o = tid - i * A;
            const lfloat *img = (const lfloat *)imgY + i;
            float acc = 0.0f;
#pragma unroll 16
            for (int k = 0; k < W; ++k) acc = fma32(img[k * IW], whl[k * 8 + o], acc);
            const float av = det_tanhf(lenv_tanh_table, acc + whl[8 * W + o]) * ma;
            const float v = av + (nzb[(ai & (NB - 1)) * T * A + tid] * astd) * ma;
            at[tid] = v < -ma ? -ma : (v > ma ? ma : v);
        }
        __syncthreads();
        double nx = 0.0, pre = 0.0;
        const int wte = tid / SD;
        if (tid < T * SD) nx = step_word(tid - wte * SD, xt_d + wte * SD, at + wte * A);
        if (tid < T) pre = EnvT::reward_pre((const double *)(xt_d + tid * SD), (const float *)(at + tid * A));
        __syncthreads();
        if (tid < T * SD) xt_d[tid] = nx;
        __syncthreads();
        if (tid < T) ep_rew[tid] = ep_rew[tid] + (float)(0.0 + EnvT::reward_post((const double *)(xt_d + tid * SD), pre));
        if (tid < T * S) { const int te = tid / S; Xl[tid] = EnvT::obs(tid - te * S, (const double *)(xt_d + te * SD)); }
        __syncthreads();
    }
    if (tid < T) { ret[tid] = (double)ep_rew[tid]; tlen[tid] = max_steps; }
    __syncthreads();
}

// ---- TD3.learn (agents/TD3.py:63-116) of a team member with one or two sample blocks (G >= 3), as ONE out-of-line routine ----
// The kernel body used to make these twelve calls itself: everything it keeps across an episode (counters, RNG state, the env's LDS
// pointers, 64-bit arena pointers) was live across every one of them, and since the phase routines clobber all but eight VGPRs that
// state went to scratch memory and came back piecemeal after each call.  Here nothing is live across a call except what is re-read
// from the context record in LDS (two dozen ds_read per call site), and the kernel body has ONE call boundary per learn step.
// Same operations in the same order as the inline version of rounds 3-4: bit-identical results.
template <int SHAPE> __device__ __noinline__ void t3v_learn_step(const T3wCtx *ctx_, uint32_t learn_lo_, uint32_t learn_hi_, int size_after_)
{
    using namespace t3p;
    constexpr T3wShape SP = kT3wShapes[SHAPE];
    using EnvT = ContEnv<SP.env>;
    constexpr int S = EnvT::S, A = EnvT::A, SA = S + A, B = SP.B, NBK = B / 32, ACT = SP.act, PD = SP.policy_delay;
    typedef __attribute__((address_space(3))) const T3wCtx LCtx;
    typedef __attribute__((address_space(3))) int lint;
    const int tid = threadIdx.x;
    const int64_t learn_it = (int64_t)(((uint64_t)uni((int)learn_hi_) << 32) | (uint32_t)uni((int)learn_lo_));
    // the delayed policy update (TD3.py:101: total_it % policy_delay == 0, total_it = this step's index + 1): actor step + the soft
    // updates of all three targets on those steps only (uniform over the team: every member counts the same learn steps)
    const bool policy_step = PD == 1 || (learn_it + 1) % PD == 0;
    const int size_after = uni(size_after_);
    // everything else comes from the context record, freshly at every use site (`cx()` hides the pointer from the optimiser so that no
    // value read through it is carried across a call)
    auto cx = [&]() -> LCtx * { const T3wCtx *p_ = ctx_; asm volatile("" : "+v"(p_)); return (LCtx *)uni_ptr(p_); };
    struct Ptrs { float *params, *targets, *w2u; gfloat *xc, *xn, *xa, *thb, *gdq, *gdz; lfloat *q1, *q2, *tq1, *tq2, *rr, *dd, *dq1, *dq2, *dzl; };
    auto ptrs = [&]() -> Ptrs {
        LCtx *c = cx();
        gfloat *ar = (gfloat *)uni_ptr(c->arena);      // (explicitly global: through a pointer read from LDS the accesses would be FLAT)
        Ptrs r;
        r.params = uni_ptr(c->params); r.targets = uni_ptr(c->targets); r.w2u = uni_ptr(c->w2u);
        r.xc = ar + c->a_xc; r.xn = ar + c->a_xn; r.xa = ar + c->a_xa; r.thb = ar + c->a_th; r.gdq = ar + c->a_gx; r.gdz = r.gdq + 2 * B;
        lfloat *qv = (lfloat *)uni_ptr(c->qvec);
        r.q1 = qv; r.q2 = qv + B; r.tq1 = qv + 2 * B; r.tq2 = qv + 3 * B; r.rr = qv + 4 * B; r.dd = qv + 5 * B; r.dq2 = qv + 6 * B; r.dq1 = qv - B; r.dzl = qv;
        return r;
    };
    auto team_barrier = [&]() {
        LCtx *c = cx();
        TeamSync ts{ uni_ptr(c->team_bar), uni_ptr(c->launch_dead), (volatile lint *)((lfloat *)uni_ptr(c->ctrl) + 32) + 5, 0u, uni(c->G), false, uni(c->same_xcd) != 0 };
        wc::team_barrier<true>(ts, tid);
    };
    const int G = uni(cx()->G), g = uni(cx()->g);
    const int gb0 = 32 * (NBK / G) * g, gbn = 32 * (NBK / G);
    const uint64_t key = ((uint64_t)uni((int)cx()->key_hi) << 32) | (uint32_t)uni((int)cx()->key_lo);
    {
        // ReplayBuffer.sample: one (sample, row element) pair per thread; a team member gathers the rows of its own blocks
        LCtx *c = cx();
        const Ptrs P = ptrs();
        const gfloat *rb = (const gfloat *)(uni_ptr(c->arena) + c->a_replay);
        const int RS = uni(c->RS);
#pragma unroll 4
        for (int e = tid; e < gbn * (2 * S + A + 2); e += NT) {
            const int b = gb0 + e / (2 * S + A + 2), i = e - (b - gb0) * (2 * S + A + 2);
            const int64_t n = learn_it * B + b;
            const int idx = (int)rng_replay_below(key, (uint64_t)n, (uint32_t)size_after);
            const float v = rb[(int64_t)idx * RS + i];
            if (i < SA) P.xc[b * SA + i] = v;                           // [s, a]
            else if (i < SA + S) P.xn[b * SA + (i - SA)] = v;           // s' (the action part is filled by actor_target)
            else if (i == SA + S) P.rr[b] = v;
            else P.dd[b] = v;
        }
        __syncthreads();
    }
    // next_actions = (actor_target(s') + clamp(randn * policy_std)).clamp(-max, max) -- and, on the second quad, the policy step's
    // actor(states) (TD3.py:97: the actor is not touched by the critic update, so its forward runs here, next to the target actor's)
    {
        const Ptrs P = ptrs();
        for (int e = tid; e < gbn * S; e += NT) { const int b = gb0 + e / S, i = e - (b - gb0) * S; P.xa[b * SA + i] = P.xc[b * SA + i]; }
        t3v_forward<ACT, S, A, NBK>(ctx_, policy_step ? 2 : 1, SA, 1, P.targets, (float *)P.xn, nullptr, -1, -1, P.params, (float *)P.xc, nullptr, TD_A_H1, TR_A_H2, (float *)P.xn, nullptr, (float *)P.xa, (float *)P.thb, SA, S);
    }
    {
        LCtx *c = cx();
        const Ptrs P = ptrs();
        const float pstd = unif(c->policy_std), clipv = unif(c->policy_clip), ma = unif(c->ma);
        for (int e = tid; e < gbn * A; e += NT) {
            const int b = gb0 + e / A, k = e - (b - gb0) * A;
            const int64_t n = (learn_it * B + b) * A + k;
            const float zn = (float)det_normal(key, STREAM_TD3_POLICY_NOISE, (uint64_t)n);
            float nz = zn * pstd;
            nz = nz < -clipv ? -clipv : (nz > clipv ? clipv : nz);
            const float v = P.xn[b * SA + S + k] + nz;
            P.xn[b * SA + S + k] = v < -ma ? -ma : (v > ma ? ma : v);
        }
        __syncthreads();
        // the twin target critics side by side on the two quads, then the twin critics
        t3v_forward<ACT, SA, 1, NBK>(ctx_, 2, SA, 0, P.targets + PN, (float *)P.xn, (float *)P.tq1, -1, -1, P.targets + 2 * PN, (float *)P.xn, (float *)P.tq2, -1, -1, nullptr, nullptr, nullptr, nullptr, 0, 0);
    }
    {
        const Ptrs P = ptrs();
        t3v_forward<ACT, SA, 1, NBK>(ctx_, 2, SA, 0, P.params + PN, (float *)P.xc, (float *)P.q1, TD_C1_H1, TR_C1_H2, P.params + 2 * PN, (float *)P.xc, (float *)P.q2, TD_C2_H1, TR_C2_H2, nullptr, nullptr,
                                nullptr, nullptr, 0, 0);
    }
    {
        LCtx *c = cx();
        const Ptrs P = ptrs();
        const float g32 = unif(c->gamma);
        const float norm = (float)(2.0 / (double)B);
        for (int b = gb0 + tid; b < gb0 + gbn; b += NT) {
            const float tq = P.tq1[b] < P.tq2[b] ? P.tq1[b] : P.tq2[b];
            const float y = P.rr[b] + ((1.0f - P.dd[b]) * g32) * tq;
            P.dq1[b] = norm * (P.q1[b] - y);
            P.dq2[b] = norm * (P.q2[b] - y);
            P.gdq[b] = P.dq1[b]; P.gdq[B + b] = P.dq2[b];      // the weight gradients need every row's value
        }
        __syncthreads();
        // per-sample halves of the two critic backwards, side by side; then -- once the whole team is there -- the parameter gradients + the
        // critic optimizer step as wave jobs over the team
        t3v_backward<ACT, SA, 1, NBK>(ctx_, 2, P.params + PN, P.w2u + IMG, (float *)P.dq1, TD_C1_H1, TR_C1_H2, TR_DZ2, TR_DH1,
                                 P.params + 2 * PN, P.w2u + 2 * IMG, (float *)P.dq2, TD_C2_H1, TR_C2_H2, TR_DZ2B, TR_DH1B, 0, 0, nullptr, nullptr);
    }
    team_barrier();
    {
        const Ptrs P = ptrs();
        for (int b = tid; b < B; b += NT) { P.dq1[b] = P.gdq[b]; P.dq2[b] = P.gdq[B + b]; }
        __syncthreads();
        const T3vNet n1{ P.params + PN, P.w2u + IMG, (float *)P.dq1, TD_C1_H1, TR_C1_H2, TR_DZ2, TR_DH1 };
        const T3vNet n2{ P.params + 2 * PN, P.w2u + 2 * IMG, (float *)P.dq2, TD_C2_H1, TR_C2_H2, TR_DZ2B, TR_DH1B };
        t3v_wgrad<ACT, SA, 1, SA, NBK>(ctx_, 2, n1, n2, (float *)P.xc, SA, 20, policy_step ? 1 : 0);      // critic_optimizer (+ Polyak of the two critics)
    }
    // actor_loss = (-critic_1(states, actor(states))).mean() with the updated critic_1; actor(states) is in xa since the start of the step
    team_barrier();
    if (!policy_step) return;
    {
        const Ptrs P = ptrs();
        t3v_forward<ACT, SA, 1, NBK>(ctx_, 1, SA, 0, P.params + PN, (float *)P.xa, (float *)P.dq2, TD_C1_H1, TR_C1_H2, P.params + PN, (float *)P.xa, (float *)P.dq2, -1, -1, nullptr, nullptr, nullptr, nullptr,
                                0, 0);
    }
    {
        const Ptrs P = ptrs();
        const float dqa = -(1.0f / (float)B);
        for (int b = tid; b < B; b += NT) P.dq1[b] = dqa;
        __syncthreads();
        t3v_backward<ACT, SA, 1, NBK>(ctx_, 1, P.params + PN, P.w2u + IMG, (float *)P.dq1, TD_C1_H1, TR_C1_H2, -1, -1,
                                 P.params + PN, P.w2u + IMG, (float *)P.dq1, TD_C1_H1, TR_C1_H2, -1, -1, S, A, (float *)P.thb, (float *)P.dzl);
    }
    {
        const Ptrs P = ptrs();
        for (int e = tid; e < gbn * A; e += NT) P.gdz[gb0 * A + e] = P.dzl[gb0 * A + e];
        t3v_backward<ACT, S, A, NBK>(ctx_, 1, P.params, P.w2u, (float *)P.dzl, TD_A_H1, TR_A_H2, TR_DZ2B, TR_DH1B,
                                P.params, P.w2u, (float *)P.dzl, TD_A_H1, TR_A_H2, TR_DZ2B, TR_DH1B, 0, 0, nullptr, nullptr);
    }
    team_barrier();
    {
        const Ptrs P = ptrs();
        for (int e = tid; e < B * A; e += NT) P.dzl[e] = P.gdz[e];
        __syncthreads();
        const T3vNet na{ P.params, P.w2u, (float *)P.dzl, TD_A_H1, TR_A_H2, TR_DZ2B, TR_DH1B };
        t3v_wgrad<ACT, S, A, SA, NBK>(ctx_, 1, na, na, (float *)P.xc, SA, 22, 1);          // actor_optimizer (+ Polyak of the actor)
    }
    team_barrier();
}

template <int SHAPE>
__global__ __launch_bounds__(NT) void td3_wavechain_kernel(const T3wArgs a)
{
    using namespace t3p;
    constexpr T3wShape SP = kT3wShapes[SHAPE];
    using EnvT = ContEnv<SP.env>;
    extern __shared__ __align__(16) float lds[];
    constexpr int S = EnvT::S, A = EnvT::A, SA = S + A, SD = EnvT::SD, B = SP.B, NBK = B / 32, Hrn = SP.hrn, ACT = SP.act, T = SP.T, RNL = SP.rn_layers;
    constexpr int PD = SP.policy_delay;
    constexpr bool VENV = SP.venv != 0;
    static_assert(!VENV || ((RNL == 2 || RNL == 3) && Hrn % 32 == 0 && Hrn <= 128), "the synthetic env's nets: two or three hidden layers of at most 128 units");
    constexpr bool CHEETAH = SP.env == LENV_ENV_CHEETAH_STANDIN;
    constexpr int KREP = SP.k_rep;
    static_assert(S <= 17 && A <= 6 && SD <= 18, "sized for the stand-in");
    static_assert(!(EnvT::TERMINATES && T > 1) && !(KREP > 1 && T > 1), "the lock-step test routine: full-length episodes, one env step per action");
    const lenv_td3_cfg &cfg = a.cfg;
    const int tid = threadIdx.x;
    // A chain is run by a TEAM of G workgroups (G = 1: the plain one-workgroup-per-chain launch).  Workgroups are dealt to the eight
    // XCDs round-robin by index, so the members of a team sit at indices that agree mod 8 and share one L2: block x + 8 k is member
    // k % G of the chain 8 (k / G) + x.
    const int G = a.G;
    const int xcd_ = blockIdx.x & 7, slot_ = blockIdx.x >> 3;
    const int g = G == 1 ? 0 : slot_ % G;
    const int64_t chain = G == 1 ? (int64_t)blockIdx.x : (int64_t)8 * (slot_ / G) + xcd_;
    if (chain >= a.chains) return;
    // the chain's status word starts at 0 (ok); written here rather than by a memset node in front of the launch (a captured
    // generation replayed under rocprofv3 did not run the memset)
    if (threadIdx.x == 0 && g == 0 && a.out.status) a.out.status[chain] = 0;
    const float prelu = cfg.prelu, ma = (float)cfg.max_action;
    const int RS = a.RS, rn_act = cfg.rn_act, rtype = cfg.reward_env_type;

    // ---- LDS carve-up ----
    float *bufA = lds, *bufB = bufA + IMG;
    float *sm_b = bufB + IMG;                             // [2][128] b1 b2
    float *sm_wo = sm_b + 2 * W;                          // [128][8]
    float *sm_bo = sm_wo + 8 * W;                         // [8]
    float *sm_b2 = sm_bo + 8;                             // [2][128], [128][8] + [8]: the second pass of a dual call
    float *sm_wo2 = sm_b2 + 2 * W;
    float *rn_w = sm_wo2 + 8 * W + 8;                     // reward net: W0 [Hrn][S] | b0 [Hrn] | Wout [Hrn] | bout (two hidden layers: in the arena)
    float *rn_h = rn_w + (VENV ? 6 * 128 : (RNL == 1 ? ((a.P_rn + 3) & ~3) : Hrn));   // [Hrn] (two hidden layers: rn_w is the second hidden row; a VirtualEnv:
                                                          // rn_w = the hidden rows [3 nets][2][128], rn_h = its input row [32] | output row [32])
    float *dq1 = rn_h + (VENV ? 128 : Hrn);               // [B]
    float *q1 = dq1 + B;                                  // [B] ... six vectors; from q1 on they double as dz [B][A] in the policy step
    float *q2 = q1 + B, *tq1 = q2 + B, *tq2 = tq1 + B, *rr = tq2 + B, *dd = rr + B;
    float *dq2 = dd + B;                                  // [B]
    float *h_row = dq2 + B;                               // [2][128] hidden rows of the one-row actor
    float *misc = h_row + 2 * W;                          // [64]
    // (alignment by INDEX arithmetic on the LDS base, which is 16-byte aligned: a round trip through an integer would hide from the compiler
    // that everything carved out behind it is LDS, and every access to it would become a FLAT instruction that waits for all loads in flight)
    double *xs_d = reinterpret_cast<double *>(lds + (((int)(misc + 64 - lds) + 1) & ~1));   // [20] train env state
    double *xt_d = xs_d + 20;                             // [17] (the test episodes of a phase run one after the other)
    double *ret = xt_d + 17 * T;                          // [T]
    float *ep_rew = reinterpret_cast<float *>(ret + T);   // [T]
    int *tlen = reinterpret_cast<int *>(ep_rew + T);      // [T]
    float *state = reinterpret_cast<float *>(tlen + T + 1);   // [20]
    float *action = state + 20;                           // [8]
    float *newrow = action + 8;                           // [56]
    T3wCtx *ctx = reinterpret_cast<T3wCtx *>(lds + (((int)(newrow + 56 - lds) + 3) & ~3));
    // (explicitly LDS: address-space inference leaves volatile accesses alone, and a volatile generic access is a FLAT instruction with
    // both cache-bypass bits that waits for every load in flight)
    volatile __attribute__((address_space(3))) float *ctrl = (volatile __attribute__((address_space(3))) float *)misc;
    volatile __attribute__((address_space(3))) int *ictrl = (volatile __attribute__((address_space(3))) int *)(misc + 32);
    float *dzl = q1;

    float *arena = a.arena + chain * a.arena_stride;
    float *params = arena + a.a_par, *targets = params + 3 * PN, *adam_m = targets + 3 * PN, *adam_v = adam_m + 3 * PN, *grad = adam_v + 3 * PN;
    float *rb = arena + a.a_replay, *xc = arena + a.a_xc, *xn = arena + a.a_xn, *xa = arena + a.a_xa, *thb = arena + a.a_th, *dumps = arena + a.a_dump;
    double *meter = reinterpret_cast<double *>(arena + a.a_meter);
    float *gdq = arena + a.a_gx, *gdz = gdq + 2 * B;       // team exchange: dq1 | dq2 [B] each, dz [B][A]
    unsigned *team_bar = reinterpret_cast<unsigned *>(arena + a.a_bar);

    // ---- stage the perturbed reward network (GTN_worker.py:165-175) and the fresh agent (TD3.py:31-39) ----
    {
        const float sg = a.eps ? a.sign[chain] : 0.0f;
        const float *e = a.eps ? a.eps + (int64_t)a.worker[chain] * a.P_rn : nullptr;
        if constexpr (RNL == 1) { for (int i = tid; i < a.P_rn; i += NT) rn_w[i] = e ? fma32(sg, e[i], a.theta[i]) : a.theta[i]; }
        else if (g == 0) { for (int i = tid; i < a.P_rn; i += NT) arena[a.a_rn + i] = e ? fma32(sg, e[i], a.theta[i]) : a.theta[i]; }
    }
    if (g == 0) {                                          // the arena is shared by the team: its first member fills it
        for (int p = tid; p < 3 * PN; p += NT) { params[p] = 0.0f; targets[p] = 0.0f; adam_m[p] = 0.0f; adam_v[p] = 0.0f; grad[p] = 0.0f; }
        __syncthreads();
        for (int p = tid; p < a.P; p += NT) {
            const float w = a.agent_init[chain * a.P + p];
            int q;
            if (p < a.Pa) q = t3w_sd_to_arena(p, S, A);
            else { const int n = (p - a.Pa) / a.Pc; q = (1 + n) * PN + t3w_sd_to_arena(p - a.Pa - n * a.Pc, SA, 1); }
            params[q] = w; targets[q] = w;
            if (G >= 3) {                                  // team path: the second layers also in unit-major order (W2u[j][k] = W2[j][k])
                const int net = q / PN, qi = q - net * PN;
                if (qi >= oW2t && qi < oW2t + IMG) { const int k = (qi - oW2t) >> 7, j = (qi - oW2t) & 127; arena[a.a_w2u + net * IMG + j * W + k] = w; }
            }
        }
    }
    if (tid < 64) misc[tid] = 0.0f;
    // Team path: the stand-in's dynamics constants (A [17][17], B [17][6], c [17] doubles) in LDS -- the small-vector areas of the image
    // passes are free there -- instead of 23 scattered 8-byte global loads per state word and env step
    typedef __attribute__((address_space(3))) double ldouble;
    ldouble *chA = (ldouble *)sm_b, *chB = chA + 17 * 17, *chC = chB + 17 * 6;
    if (CHEETAH && G >= 3) {
        for (int i = tid; i < 17 * 17; i += NT) chA[i] = lenv_cheetah_A[i];
        if (tid < 17 * 6) chB[tid] = lenv_cheetah_B[tid];
        if (tid < 17) chC[tid] = lenv_cheetah_c[tid];
    }
    // word i of x' = clip(c + A x + B a, -10, 10): EnvT::step_word with the constants from LDS (same operations in the same order)
    auto env_step_word = [&](int i, const double *x_, const float *a_) -> double {
        if (!CHEETAH || G < 3) return EnvT::step_word(i, x_, a_);
        const ldouble *x = (const ldouble *)x_;
        const lfloat *av = (const lfloat *)a_;
        double acc = chC[i];
#pragma unroll
        for (int j = 0; j < 17; ++j) acc = acc + chA[i * 17 + j] * x[j];
#pragma unroll
        for (int k = 0; k < 6; ++k) acc = acc + chB[i * 6 + k] * (double)av[k];
        return acc < -10.0 ? -10.0 : (acc > 10.0 ? 10.0 : acc);
    };
    float *w2u = arena + a.a_w2u;
    if (tid == 0) {
        T3wCtx cx{ bufA, bufB, sm_b, sm_wo, sm_bo, q1, dzl, sm_b2, sm_wo2, params, targets, grad, dumps, prelu, ma, g, G, w2u, misc,
                   (float)(1.0 - cfg.adam_beta1), (float)(1.0 - cfg.adam_beta2), (float)cfg.adam_beta2, (float)cfg.adam_eps, (float)cfg.tau,
                   (float)(1.0 - cfg.tau), xt_d, ret, ep_rew, tlen, (float)cfg.action_std, cfg.max_steps,
                   arena, a.a_xc, a.a_xn, a.a_xa, a.a_th, a.a_gx, a.a_replay, a.RS, 0, (uint32_t)a.rng_keys[chain], (uint32_t)(a.rng_keys[chain] >> 32),
                   (float)cfg.gamma, (float)cfg.policy_std, (float)cfg.policy_std_clip, team_bar, reinterpret_cast<unsigned *>(a.arena + a.a_bar) + 8 };
        *ctx = cx;
    }
    __syncthreads();

    const uint64_t key = a.rng_keys[chain];
    int status = 0;
    // ---- team barrier (G > 1, wc::team_barrier): every member has finished its share of a phase and its arena writes are visible to the
    // others.  The chain's counter and the launch's give-up word are zeroed by t3w_team_reset_kernel in front of the launch.
    TeamSync tsync{ team_bar, reinterpret_cast<unsigned *>(a.arena + a.a_bar) + 8, ictrl + 5, 0u, G, false, false };
    // (no copies of the give-up flag or counters in kernel-lifetime variables: what lives across the calls of the learn step ends up in
    // scratch memory and costs a round trip at every use; tsync.dead is read where it matters)
#ifdef LENV_PHASE_TIMING
    unsigned long long bar_cycles = 0;
#endif
    auto team_barrier = [&]() {
        if (G == 1) return;
#ifdef LENV_PHASE_TIMING
        const unsigned long long bt0 = __builtin_readcyclecounter();
#endif
        wc::team_barrier<true>(tsync, tid);
#ifdef LENV_PHASE_TIMING
        bar_cycles += __builtin_readcyclecounter() - bt0;
#endif
    };
#define team_dead (wc::team_is_dead(tsync))
    // sample block of row b -> does it belong to this member (blocks are dealt like the waves that own them)
    auto my_row = [&](int b) { return G == 1 || ((b >> 5) * G) / NBK == g; };
    if (G > 1 && tid == 0) reinterpret_cast<unsigned *>(gdz)[g] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;   // HW_REG_XCC_ID[3:0]
#ifdef LENV_DIAG_TEAM_TIMES
    // diagnostic build: when did this member start / pass the first barrier / leave (ms since the reset kernel), into final_params
    const unsigned long long dg_t0 = *reinterpret_cast<const unsigned long long *>(a.arena + a.a_bar + 10);
    auto dg_stamp = [&](int i) { if (tid == 0 && a.out.final_params) a.out.final_params[chain * a.P + 8 * g + i] = (float)((double)(__builtin_amdgcn_s_memrealtime() - dg_t0) * 1e-5); };
    dg_stamp(0);
    if (tid == 0 && a.out.final_params) a.out.final_params[chain * a.P + 8 * g + 3] = (float)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u);
#endif
    team_barrier();                                        // the arena is initialised, every member's XCD id is posted
#ifdef LENV_DIAG_TEAM_TIMES
    dg_stamp(1);
#endif
    if (G > 1) {
        // are all members on one XCD (the block-to-XCD round robin the index mapping above counts on)?  The same answer in every member.
        bool same = true;
        const unsigned x0 = reinterpret_cast<unsigned *>(gdz)[0];
        for (int m = 1; m < G; ++m) same = same && reinterpret_cast<unsigned *>(gdz)[m] == x0;
        team_barrier();                                    // everybody has read the ids before the exchange rows are reused
        tsync.same_xcd = same;
        if (tid == 0) ctx->same_xcd = same ? 1 : 0;        // (the learn-step routine builds its barrier record from the context)
        __syncthreads();
    }
    if (team_dead) {                                       // not all members became resident in time: nothing was computed
        if (a.out.status) atomicMin(&a.out.status[chain], -10);
#ifdef LENV_DIAG_TEAM_TIMES
        dg_stamp(2);
#endif
        return;
    }
    TPT_DECL;
    int64_t n_rand = 0, n_actn = 0, n_testn = 0, n_test_ep = 0, learn_it = 0;
    int train_steps = 0, test_steps = 0, episodes_run = 0;
    double pows[4] = { 1.0, 1.0, 1.0, 1.0 };
    const int rb_cap = (int)a.rb_cap;
    const float g32 = (float)cfg.gamma;

    // ---- the actor on ONE row (TD3.select_train_action / select_test_action): thread j owns unit j, k-ascending fmaf chains over the
    // K-major arrays (coalesced across the threads) ----
    // Team path: between two learn steps a member's big LDS buffers are idle, so the actor's weights live there (W2t [128][128] in bufB,
    // W1t / b1 / b2 / Wo / bo at the head of bufA): act_lds_load() after every actor update, and the one-row forward of the env and test steps
    // -- every member repeats it -- reads LDS instead of fetching 73 KB through the L2 three dependent batches deep (21 k -> 6 k
    // cycles per test step).  Same k-ascending chains.
    // LDS layout: W1t [S][128] | b1 | b2 | Wo^T [8][132] | bo at the head of bufA; W2 UNIT-major [128][132] (row = unit, k contiguous, rows
    // padded to 132 floats: a lane reads four terms per 16-byte LDS instruction) in the last 512 floats of bufA + all of bufB
#ifdef LENV_PHASE_TIMING_SUB
    unsigned long long ksub_last = 0;
#endif
    constexpr int aL_b1 = S * W, aL_b2 = aL_b1 + W, aL_wo = aL_b2 + W, aL_bo = aL_wo + 8 * 132, aL_w2 = IMG - 512, aLD = 132;
    auto act_lds_load = [&]() {
        const gf4 *s2 = (const gf4 *)w2u;                  // the actor's unit-major copy W2u[j][k] (kept by the optimizer epilogue)
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = s2[tid + u * NT];
        const gf4 *s1 = (const gf4 *)(params + oW1t), *sb1 = (const gf4 *)(params + ob1), *sb2 = (const gf4 *)(params + ob2);
        lf4 *d = (lf4 *)bufA;
        lfloat *df = (lfloat *)bufA;
        for (int i = tid; i < S * W / 4; i += NT) d[i] = s1[i];
        if (tid < W / 4) { d[aL_b1 / 4 + tid] = sb1[tid]; d[aL_b2 / 4 + tid] = sb2[tid]; }
        for (int i = tid; i < 8 * W; i += NT) df[aL_wo + (i & 7) * aLD + (i >> 3)] = ((const gfloat *)params)[oWo + i];      // Wo[k][c] -> [c][k]
        if (tid < 8) df[aL_bo + tid] = ((const gfloat *)params)[obo + tid];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = tid + u * NT, j = idx >> 5, k4 = (idx & 31) << 2;
            *(lf4 *)(df + aL_w2 + j * aLD + k4) = v[u];
        }
        __syncthreads();
    };
    // Exploration noise of the env / test steps: A standard normals per step (det_normal: ~4 k cycles of fp64 polynomials, on six lanes
    // when drawn step by step).  The team path draws them 64 steps at a time with all 512 threads (one value each) into an LDS batch
    // behind the dynamics constants: same values, a sixty-fourth of the time on the serial path.
    lfloat *nzb = (lfloat *)(sm_b + 832);                  // [64][A]
    uint32_t nz_cur = 0;
    int64_t nz_step0 = 0;
    auto step_noise = [&](uint32_t stream, int64_t step) -> float {      // value of lane tid < A; every thread calls (uniform refill)
        if (G < 3) return tid < A ? (float)det_normal(key, stream, (uint64_t)(step * A + tid)) : 0.0f;
        if (nz_cur != stream || step < nz_step0 || step >= nz_step0 + 64) {
            __syncthreads();
            for (int e = tid; e < 64 * A; e += NT) nzb[e] = (float)det_normal(key, stream, (uint64_t)(step * A + e));
            nz_cur = stream; nz_step0 = step;
            __syncthreads();
        }
        return tid < A ? nzb[(int)(step - nz_step0) * A + tid] : 0.0f;
    };
    // sum_k h[k] * w[k], k = 0..127 ascending in one fmaf chain from 0; both vectors in LDS; the 16-byte reads of the next 32 terms are
    // in flight while the current 32 are chained (the chain itself is ~8 cycles per term)
    auto row_chain = [&](const lfloat *wrow, const lfloat *hv) -> float {
        f32x4 w4[8], h4[8], w4n[8], h4n[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { w4[u] = *(const lf4 *)(wrow + 4 * u); h4[u] = *(const lf4 *)(hv + 4 * u); }
        float z = 0.0f;
#pragma unroll
        for (int k0 = 0; k0 < W; k0 += 32) {
            if (k0 + 32 < W) {
#pragma unroll
                for (int u = 0; u < 8; ++u) { w4n[u] = *(const lf4 *)(wrow + k0 + 32 + 4 * u); h4n[u] = *(const lf4 *)(hv + k0 + 32 + 4 * u); }
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) z = fma32(h4[u >> 2][u & 3], w4[u >> 2][u & 3], z);
#pragma unroll
            for (int u = 0; u < 8; ++u) { w4[u] = w4n[u]; h4[u] = h4n[u]; }
        }
        return z;
    };
    auto actor_row1_lds = [&](const float *x_, float *out_) {
        const lfloat *x = (const lfloat *)x_, *A1 = (const lfloat *)bufA;
        lfloat *hr = (lfloat *)h_row, *out = (lfloat *)out_;
        // (all the LDS reads of a 32-term piece are requested before its fmaf chain starts: written term by term the compiler issues
        // read, wait, fmaf -- one LDS round trip per term, 15 k cycles per call)
        if (tid < W) {
            float w[S], xv[S];
#pragma unroll
            for (int k = 0; k < S; ++k) { w[k] = A1[k * W + tid]; xv[k] = x[k]; }
            float z = 0.0f;
#pragma unroll
            for (int k = 0; k < S; ++k) z = fma32(xv[k], w[k], z);
            hr[tid] = act_fwd(ACT, prelu, z + A1[aL_b1 + tid]);
        }
        KSUB_MARK(24);
        __syncthreads();
        KSUB_MARK(25);
        if (tid < W) {
            const lfloat *wrow = A1 + aL_w2 + tid * aLD;
            hr[W + tid] = act_fwd(ACT, prelu, row_chain(wrow, hr) + A1[aL_b2 + tid]);
        }
        KSUB_MARK(26);
        __syncthreads();
        KSUB_MARK(27);
        if (tid < A) {
            const lfloat *wrow = A1 + aL_wo + tid * aLD;
            out[tid] = det_tanhf(lenv_tanh_table, row_chain(wrow, hr + W) + A1[aL_bo + tid]) * ma;
        }
        KSUB_MARK(28);
        __syncthreads();
        KSUB_MARK(29);
    };
    auto actor_row1 = [&](const float *x, float *out) {
        if (G >= 3) { actor_row1_lds(x, out); return; }
        const gfloat *par = (const gfloat *)params;
        if (tid < W) {
            float w[S];
#pragma unroll
            for (int k = 0; k < S; ++k) w[k] = par[oW1t + k * W + tid];
            float z = 0.0f;
#pragma unroll
            for (int k = 0; k < S; ++k) z = fma32(x[k], w[k], z);
            h_row[tid] = act_fwd(ACT, prelu, z + par[ob1 + tid]);
        }
        __syncthreads();
        if (tid < W) {
            float z = 0.0f;
            for (int k0 = 0; k0 < W; k0 += 64) {           // 64 coalesced weight reads in flight, then the ordered chain
                float w[64];
#pragma unroll
                for (int u = 0; u < 64; ++u) w[u] = par[oW2t + (k0 + u) * W + tid];
#pragma unroll
                for (int u = 0; u < 64; ++u) z = fma32(h_row[k0 + u], w[u], z);
            }
            h_row[W + tid] = act_fwd(ACT, prelu, z + par[ob2 + tid]);
        }
        __syncthreads();
        if (tid < A) {
            float z = 0.0f;
            for (int k0 = 0; k0 < W; k0 += 64) {
                float w[64];
#pragma unroll
                for (int u = 0; u < 64; ++u) w[u] = par[oWo + (k0 + u) * 8 + tid];
#pragma unroll
                for (int u = 0; u < 64; ++u) z = fma32(h_row[W + k0 + u], w[u], z);
            }
            out[tid] = det_tanhf(lenv_tanh_table, z + par[obo + tid]) * ma;
        }
        __syncthreads();
    };

    auto adam = [&](int p0, int n, int pi, bool polyak) {
        if (tid == 0) {
            pows[pi] *= cfg.adam_beta1; pows[pi + 1] *= cfg.adam_beta2;
            ctrl[10] = (float)(-(cfg.lr / (1.0 - pows[pi])));
            ctrl[11] = (float)__builtin_sqrt(1.0 - pows[pi + 1]);
        }
        __syncthreads();
        const float neg_step = ctrl[10], bc2_sqrt = ctrl[11];
        const float w1 = (float)(1.0 - cfg.adam_beta1), w2 = (float)(1.0 - cfg.adam_beta2), beta2 = (float)cfg.adam_beta2, aeps = (float)cfg.adam_eps;
        const AdamConsts ac{ neg_step, bc2_sqrt, w1, w2, beta2, aeps };
        // the Polyak update of the same parameters rides in the pass (TD3.py:104-116 runs it after both optimizer steps of a policy
        // step; nothing between the critic step and the soft update reads a target net: element for element the same
        // tau * w + (1 - tau) * t on the same operands); polyak false: a learn step without the delayed policy update
        // a team cuts the range into G pieces of whole float4s
        const int piece = G == 1 ? n : (((n + G - 1) / G + 3) & ~3);
        const int lo = p0 + piece * g, hi = lo + piece < p0 + n ? lo + piece : p0 + n;
        if (hi > lo) wg_adam(params, adam_m, adam_v, grad, lo, hi - lo, ac, polyak ? targets : nullptr, (float)cfg.tau, (float)(1.0 - cfg.tau));
        __syncthreads();
    };

    // phi = reward_net(obs) -> ctrl[slot]   (one hidden layer, types 1 / 2 / 5 / 6: no info inputs on this path)
    auto rn_eval = [&](const float *obs, int slot) {
        if (rtype == 0) { if (tid == 0) ctrl[slot] = 0.0f; __syncthreads(); return; }
        if constexpr (RNL == 2) {
            static_assert(RNL != 2 || (Hrn * S) % 4 == 0, "16-byte rows of the second layer");
            // build_nn_from_config (model_utils.py:16-29): Linear(S, H) | Linear(H, H) | Linear(H, 1) in Module.parameters() order, from the
            // chain's arena (td3_rn_inner_kernel's deeper reward nets: thread j = unit j, k ascending)
            const gfloat *rnp = (const gfloat *)(arena + a.a_rn);
            const gfloat *W0g = rnp, *b0g = rnp + Hrn * S, *W1g = b0g + Hrn, *b1g = W1g + Hrn * Hrn, *Wog = b1g + Hrn, *bog = Wog + Hrn;
            lfloat *h1 = (lfloat *)rn_h, *h2 = (lfloat *)rn_w;
            if (tid < Hrn) {
                float z = 0.0f;
#pragma unroll
                for (int k = 0; k < S; ++k) z = fma32(obs[k], W0g[tid * S + k], z);
                h1[tid] = act_fwd(rn_act, cfg.rn_prelu, z + b0g[tid]);
            }
            __syncthreads();
            if (tid < Hrn) {
                const gf4 *wr = (const gf4 *)(W1g + tid * Hrn);
                float z = 0.0f;
#pragma unroll 1
                for (int k0 = 0; k0 < Hrn; k0 += 32) {
                    f32x4 w4[8], h4[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { w4[u] = wr[(k0 >> 2) + u]; h4[u] = *(const lf4 *)(h1 + k0 + 4 * u); }
#pragma unroll
                    for (int u = 0; u < 32; ++u) z = fma32(h4[u >> 2][u & 3], w4[u >> 2][u & 3], z);
                }
                h2[tid] = act_fwd(rn_act, cfg.rn_prelu, z + b1g[tid]);
            }
            __syncthreads();
            if (tid == 0) {
                float acc = 0.0f;
#pragma unroll 1
                for (int j0 = 0; j0 < Hrn; j0 += 32) {
                    f32x4 h4[8], w4[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) { h4[u] = *(const lf4 *)(h2 + j0 + 4 * u); w4[u] = *(const gf4 *)(Wog + j0 + 4 * u); }
#pragma unroll
                    for (int u = 0; u < 32; ++u) acc = fma32(h4[u >> 2][u & 3], w4[u >> 2][u & 3], acc);
                }
                ctrl[slot] = acc + bog[0];
            }
            __syncthreads();
            return;
        }
        const float *W0 = rn_w, *b0 = rn_w + Hrn * S;
        for (int j = tid; j < Hrn; j += NT) {
            float z = 0.0f;
            for (int k = 0; k < S; ++k) z = fma32(obs[k], W0[j * S + k], z);
            rn_h[j] = act_fwd(rn_act, cfg.rn_prelu, z + b0[j]);
        }
        __syncthreads();
        const float *Wo = b0 + Hrn, *bo = Wo + Hrn;
        if (tid == 0) {
            // (the reads of a 32-term piece are all requested before its fmaf chain starts; term by term they cost one LDS round trip each)
            const lfloat *hl = (const lfloat *)rn_h, *wl = (const lfloat *)Wo;
            float acc = 0.0f;
#pragma unroll 1
            for (int j0 = 0; j0 < Hrn; j0 += 32) {
                f32x4 h4[8], w4[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { h4[u] = *(const lf4 *)(hl + j0 + 4 * u); w4[u] = *(const lf4 *)(wl + j0 + 4 * u); }
#pragma unroll
                for (int u = 0; u < 32; ++u) acc = fma32(h4[u >> 2][u & 3], w4[u >> 2][u & 3], acc);
            }
            ctrl[slot] = acc + bo[0];
        }
        __syncthreads();
    };

    // ---- EnvWrapper.step -> VirtualEnv.step (virtual_env.py:43-54) of the training loop: the three nets of the synthetic env (state,
    // reward, done; build_nn_from_config: Linear(S + A, H) | Linear(H, H) | Linear(H, out) each, Module.parameters() order, the perturbed
    // copy in the chain's arena) on the row cat(action, state).  The nets run side by side on thread groups of 128; thread j owns unit j of
    // its net's layer and runs td3_rn_inner_kernel's k-ascending chain (then + bias, activation).  EnvWrapper.step repeats the step
    // same_action_num times whatever the done output says and sums the fp32 rewards (env_wrapper.py:24-29); reward and done see the
    // pre-transition state.  Leaves the replay row in newrow (the caller's barrier follows).
    auto venv_step = [&]() {
        static_assert(!VENV || (SA <= 32 && S + 2 <= 32 && Hrn % 32 == 0 && Hrn <= 128), "input / output rows of 32 floats, 32-term pieces, rows of 128");
        lfloat *xse = (lfloat *)rn_h, *nse = xse + 32, *hrow = (lfloat *)rn_w;      // hrow: [3 nets][2][128], a net's hidden layers alternate between its two rows
        constexpr int P_HID = Hrn * SA + Hrn + (RNL - 1) * (Hrn * Hrn + Hrn), P_STATE = P_HID + S * Hrn + S, P_ONE = P_HID + Hrn + 1;
        const int net = tid >> 7, j = tid & 127;
        const bool unit = net < 3 && j < Hrn;
        const gfloat *np = (const gfloat *)(arena + a.a_rn) + (net == 0 ? 0 : (net == 1 ? P_STATE : P_STATE + P_ONE));
        const int n_out = net == 0 ? S : 1, ocol = net == 0 ? 0 : S + net - 1;
        lfloat *hr = hrow + (2 * (net < 3 ? net : 0)) * 128;
        if (tid < A) xse[tid] = action[tid];
        if (tid >= 64 && tid < 64 + S) xse[A + tid - 64] = state[tid - 64];
        if (tid >= 128 && tid < 128 + S) newrow[tid - 128] = state[tid - 128];
        if (tid >= 192 && tid < 192 + A) newrow[S + tid - 192] = action[tid - 192];
        __syncthreads();
        // sum_k h[k] * w[k] over a row of Hrn weights (16-byte loads, alignment of the row: 4 bytes) and Hrn LDS activations, k ascending from 0
        auto row_dot = [&](const gfloat *wrow, const lfloat *hv) -> float {
            const gf4 *wr = (const gf4 *)wrow;
            float z = 0.0f;
#pragma unroll 1
            for (int k0 = 0; k0 < Hrn; k0 += 32) {
                f32x4 w4[8], h4[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { w4[u] = wr[(k0 >> 2) + u]; h4[u] = *(const lf4 *)(hv + k0 + 4 * u); }
#pragma unroll
                for (int u = 0; u < 32; ++u) z = fma32(h4[u >> 2][u & 3], w4[u >> 2][u & 3], z);
            }
            return z;
        };
#pragma unroll 1
        for (int r_ = 0; r_ < KREP; ++r_) {
            if (unit) {
                float z = 0.0f;
#pragma unroll
                for (int k = 0; k < SA; ++k) z = fma32(xse[k], np[j * SA + k], z);
                hr[j] = act_fwd(rn_act, cfg.rn_prelu, z + np[Hrn * SA + j]);
            }
            __syncthreads();
#pragma unroll
            for (int l = 1; l < RNL; ++l) {
                const gfloat *wl = np + Hrn * SA + Hrn + (l - 1) * (Hrn * Hrn + Hrn);
                if (unit) hr[(l & 1) * 128 + j] = act_fwd(rn_act, cfg.rn_prelu, row_dot(wl + j * Hrn, hr + ((l - 1) & 1) * 128) + wl[Hrn * Hrn + j]);
                __syncthreads();
            }
            if (net < 3 && j < n_out) nse[ocol + j] = row_dot(np + P_HID + j * Hrn, hr + ((RNL - 1) & 1) * 128) + np[P_HID + n_out * Hrn + j];
            __syncthreads();
            if (tid < S) { newrow[S + A + tid] = nse[tid]; xse[A + tid] = nse[tid]; }
            if (tid == 64) { newrow[2 * S + A] = r_ == 0 ? nse[S] : newrow[2 * S + A] + nse[S]; newrow[2 * S + A + 1] = nse[S + 1]; }
            if (r_ + 1 < KREP) __syncthreads();
        }
    };

    // ---- real-env test phase (BaseAgent.test): T episodes one after the other (the envs of this kernel never terminate: every episode
    // takes max_steps steps, and the reset row / noise rows of an episode have fixed indices -- td3_rn_inner_kernel's lock-step rollouts
    // draw the same ones), actions from the one-row actor + exploration noise (TD3.py:126-129) ----
    auto test_phase = [&]() {
        const int64_t nstride = cfg.max_steps;
        float *xt = newrow;                                // [S] observation
        float *at = newrow + 24;                           // [A] action
        if constexpr (T > 1) {                              // (measured for the one-episode shape too: 9 % slower there than the LDS-resident one-row actor)
            __syncthreads();
            t3w_test_steps<SHAPE>(ctx, (uint32_t)key, (uint32_t)(key >> 32), (int)n_test_ep, (uint32_t)n_testn, (uint32_t)((uint64_t)n_testn >> 32));      // (ends with a barrier)
            test_steps += T * cfg.max_steps;
            if (G >= 3) act_lds_load();                    // the routine used bufA
        } else
#pragma unroll 1
        for (int te = 0; te < T; ++te) {
            for (int e = tid; e < SD; e += NT) xt_d[e] = EnvT::reset_word(key, STREAM_TEST_RESET, n_test_ep + te, e);
            if (tid == 0) ep_rew[te] = 0.0f;
            __syncthreads();
            int my_el = 0;
            bool ep_alive = true;                          // (uniform: the episode's state lives in LDS)
            const int nag = (cfg.max_steps + KREP - 1) / KREP;     // agent steps of a full-length episode
            for (int ai = 0; ai < nag && ep_alive; ++ai) {
                KSUB_RESET;
                if (tid < S) xt[tid] = EnvT::obs(tid, xt_d);
                __syncthreads();
                KSUB_MARK(12);
                actor_row1(xt, at);
                KSUB_MARK(13);
                const float zn = step_noise(STREAM_TD3_TEST_NOISE, n_testn + (int64_t)te * nstride + ai);
                if (tid < A) {
                    const float v = at[tid] + (zn * (float)cfg.action_std) * ma;
                    at[tid] = v < -ma ? -ma : (v > ma ? ma : v);
                }
                __syncthreads();
                KSUB_MARK(14);
                // the action is applied same_action_num times or until the episode ends (the env's own flag / TimeLimit); the rewards of the
                // repeats are summed as python floats (env_wrapper.py:56-61)
                double rsum = 0.0;
#pragma unroll 1
                for (int r_ = 0; r_ < KREP && ep_alive; ++r_) {
                    double nx = 0.0, pre = 0.0;
                    if (tid < SD) nx = env_step_word(tid, xt_d, at);
                    if (tid == 0) pre = EnvT::reward_pre(xt_d, at);
                    __syncthreads();
                    if (tid < SD) xt_d[tid] = nx;
                    __syncthreads();
                    if (tid == 0) rsum = rsum + EnvT::reward_post(xt_d, pre);
                    ++my_el;
                    if (EnvT::done(xt_d) || my_el >= cfg.max_steps) ep_alive = false;
                    __syncthreads();
                }
                if (tid == 0) ep_rew[te] = ep_rew[te] + (float)rsum;
                KSUB_MARK(15);
            }
            if (tid == 0) { ret[te] = (double)ep_rew[te]; tlen[te] = my_el; }
            test_steps += my_el;
        }
        n_test_ep += T;
        n_testn += (int64_t)T * nstride;
        __syncthreads();
    };

    const bool budgeted = cfg.step_budget > 0;
    int timed_out_at = -1;
    if (G >= 3) act_lds_load();                            // the fresh actor
    for (int episode = 0; episode < cfg.train_episodes; ++episode) {
        if (budgeted && (int64_t)train_steps + test_steps > cfg.step_budget) { timed_out_at = episode; break; }
        const bool learning = episode >= cfg.init_episodes;
        for (int i = tid; i < SD; i += NT) xs_d[i] = EnvT::reset_word(key, STREAM_TRAIN_RESET, (int64_t)episode, i);
        __syncthreads();
        if (tid < S) state[tid] = EnvT::obs(tid, xs_d);
        __syncthreads();
        if (!VENV && (rtype == 1 || rtype == 2)) rn_eval(state, 12);
        int ep_len = 0, env_steps = 0;
        for (int t = 0; t < cfg.max_steps; t += VENV ? KREP : 1) {      // (base_agent.py:104 range(0, max_steps, same_action_num); a RewardEnv's TimeLimit ends its loop)
            const int size_after = train_steps + 1 < rb_cap ? train_steps + 1 : rb_cap;
            const int new_pos = train_steps % rb_cap;
            if (!learning) {
                if (tid < A) action[tid] = (float)(-(double)ma + (2.0 * (double)ma) * u64_to_unit(rng_u64(key, STREAM_TD3_RAND_ACTION, (uint64_t)(n_rand * A + tid))));
                ++n_rand;
                __syncthreads();
            } else {
                actor_row1(state, action);
                const float zn = step_noise(STREAM_TD3_ACT_NOISE, n_actn);
                if (tid < A) {
                    const float v = action[tid] + (zn * (float)cfg.action_std) * ma;
                    action[tid] = v < -ma ? -ma : (v > ma ? ma : v);
                }
                ++n_actn;
                __syncthreads();
            }
            if constexpr (VENV) venv_step();
            else {
            // ---- EnvWrapper.step -> RewardEnv.step -> real_env.step + TimeLimit ----
            if (tid < S) newrow[tid] = state[tid];
            if (tid >= 64 && tid < 64 + A) newrow[S + tid - 64] = action[tid - 64];
            double rsum = 0.0;                             // (thread 0) python-float sum of the repeats' shaped rewards (env_wrapper.py:56-61)
#pragma unroll 1
            for (int r_ = 0; r_ < KREP; ++r_) {
                double nx = 0.0, pre = 0.0;
                if (tid < SD) nx = env_step_word(tid, xs_d, action);
                if (tid == 64) pre = EnvT::reward_pre(xs_d, action);
                __syncthreads();
                if (tid < SD) xs_d[tid] = nx;
                if (tid == 64) xs_d[18] = pre;
                __syncthreads();
                ++env_steps;
                const bool dn = EnvT::done(xs_d) || env_steps >= cfg.max_steps;
                if (tid < S) newrow[S + A + tid] = EnvT::obs(tid, xs_d);
                __syncthreads();
                rn_eval(newrow + S + A, 13);
                if (tid == 0) {
                    const double rew = EnvT::reward_post(xs_d, xs_d[18]);
                    const float r32 = (float)rew, phi_s = ctrl[12], phi_s2 = ctrl[13];
                    float shaped;
                    switch (rtype) {
                    case 0: shaped = r32; break;
                    case 1: shaped = g32 * phi_s2 - phi_s; break;
                    case 2: shaped = (r32 + g32 * phi_s2) - phi_s; break;
                    case 5: shaped = phi_s2; break;
                    default: shaped = r32 + phi_s2; break;
                    }
                    rsum = rsum + (double)shaped;
                    newrow[2 * S + A] = (float)rsum; newrow[2 * S + A + 1] = dn ? 1.0f : 0.0f;
                    ctrl[12] = phi_s2;
                }
                __syncthreads();
                if (tid < S) state[tid] = newrow[S + A + tid];
                if (dn) break;                             // (uniform) the repeats stop at done
                if (r_ + 1 < KREP) __syncthreads();
            }
            }
            __syncthreads();
            if (tid < 2 * S + A + 2) rb[(int64_t)new_pos * RS + tid] = newrow[tid];
            const float done_now = newrow[2 * S + A + 1];
            __syncthreads();
            if (tid < S) state[tid] = newrow[S + A + tid];
            ep_len += KREP; ++train_steps;                 // base_agent.py:122: episode_length += same_action_num
            __syncthreads();
            TPT_MARK(0);
            if (learning) {
                // ================= TD3.learn (TD3.py:63-116) =================
                if (G >= 3) {
                    // ================= a member with one or two sample blocks: the team path (td3_wavechain_team.cuh) =================
                    if (tid == 0) {                        // torch.optim.Adam's bias corrections of this step's two optimizer steps
                        pows[0] *= cfg.adam_beta1; pows[1] *= cfg.adam_beta2;
                        ctrl[20] = (float)(-(cfg.lr / (1.0 - pows[0]))); ctrl[21] = (float)__builtin_sqrt(1.0 - pows[1]);      // (12, 13: the reward net's phi values)
                        if (PD == 1 || (learn_it + 1) % PD == 0) {        // the actor's optimizer steps on the delayed policy updates only
                            pows[2] *= cfg.adam_beta1; pows[3] *= cfg.adam_beta2;
                            ctrl[22] = (float)(-(cfg.lr / (1.0 - pows[2]))); ctrl[23] = (float)__builtin_sqrt(1.0 - pows[3]);
                        }
                    }
                    // replay gather, five forward calls, three backward chains, the two optimizer job phases and the four team barriers:
                    // one out-of-line routine, so that this body has ONE call boundary per learn step (t3v_learn_step)
                    t3v_learn_step<SHAPE>(ctx, (uint32_t)learn_it, (uint32_t)((uint64_t)learn_it >> 32), size_after);
                    TPT_MARK(7);
                    ++learn_it;
                    act_lds_load();                        // the updated actor, for the next env steps / the test episode
                    TPT_MARK(8);
                } else {
                const int gb0 = G == 1 ? 0 : 32 * (NBK / G) * g, gbn = G == 1 ? B : 32 * (NBK / G);
#pragma unroll 4
                for (int e = tid; e < gbn * (2 * S + A + 2); e += NT) {
                    const int b = gb0 + e / (2 * S + A + 2), i = e - (b - gb0) * (2 * S + A + 2);
                    const int64_t n = learn_it * B + b;
                    const int idx = (int)rng_replay_below(key, (uint64_t)n, (uint32_t)size_after);
                    const float v = rb[(int64_t)idx * RS + i];
                    if (i < SA) xc[b * SA + i] = v;                           // [s, a]
                    else if (i < SA + S) xn[b * SA + (i - SA)] = v;           // s' (the action part is filled by actor_target)
                    else if (i == SA + S) rr[b] = v;
                    else dd[b] = v;
                }
                __syncthreads();
                TPT_MARK(1);
                // next_actions = (actor_target(s') + clamp(randn * policy_std)).clamp(-max, max)
                t3w_forward<ACT, S, A, NBK>(ctx, targets, xn, SA, 1, nullptr, xn, SA, S, nullptr, -1, -1, -1);
                for (int e = tid; e < B * A; e += NT) {
                    const int b = e / A, k = e - b * A;
                    if (!my_row(b)) continue;
                    const int64_t n = (learn_it * B + b) * A + k;
                    const float zn = (float)det_normal(key, STREAM_TD3_POLICY_NOISE, (uint64_t)n);
                    float nz = zn * (float)cfg.policy_std;
                    const float clipv = (float)cfg.policy_std_clip;
                    nz = nz < -clipv ? -clipv : (nz > clipv ? clipv : nz);
                    const float v = xn[b * SA + S + k] + nz;
                    xn[b * SA + S + k] = v < -ma ? -ma : (v > ma ? ma : v);
                }
                __syncthreads();
                TPT_MARK(2);
                {
                    // three blocks per member (G = 2): twin critics side by side on six waves; one workgroup per chain: the twelve
                    // block-passes of the twin critics on eight waves, three per SIMD
                    t3w_forward<ACT, SA, 1, NBK>(ctx, targets + PN, xn, SA, 0, tq1, nullptr, 0, 0, nullptr, -1, -1, -1, targets + 2 * PN, xn, tq2, -1, -1);
                    t3w_forward<ACT, SA, 1, NBK>(ctx, params + PN, xc, SA, 0, q1, nullptr, 0, 0, nullptr, TD_C1_H1, TD_C1_H2, TR_C1_H2, params + 2 * PN, xc, q2,
                                            TD_C2_H1, TR_C2_H2);
                }
                TPT_MARK(3);
                {
                    const float norm = (float)(2.0 / (double)B);
                    for (int b = tid; b < B; b += NT) {
                        if (!my_row(b)) continue;
                        const float tq = tq1[b] < tq2[b] ? tq1[b] : tq2[b];
                        const float y = rr[b] + ((1.0f - dd[b]) * g32) * tq;
                        dq1[b] = norm * (q1[b] - y);
                        dq2[b] = norm * (q2[b] - y);
                        if (G > 1) { gdq[b] = dq1[b]; gdq[B + b] = dq2[b]; }      // the weight gradients need every row's value
                    }
                }
                __syncthreads();
                TPT_MARK(4);
                // per-sample halves of the two critic backwards (this member's blocks), then -- once the whole team is there -- the
                // parameter gradients, cut by wave over the team
                // G = 2 and one workgroup per chain: the twin critics' chains side by side
                t3w_backward_chain<ACT, SA, 1, NBK>(ctx, params + PN, dq1, TD_C1_H1, TR_C1_H2, TR_DZ2, TR_DH1, 0, 0, nullptr, nullptr,
                                                   params + 2 * PN, dq2, TD_C2_H1, TR_C2_H2, TR_DZ2B, TR_DH1B);
                team_barrier();
                if (G > 1) {
                    for (int b = tid; b < B; b += NT) { dq1[b] = gdq[b]; dq2[b] = gdq[B + b]; }
                    __syncthreads();
                }
                t3w_backward_wgrad<ACT, SA, 1, NBK>(ctx, grad + PN, xc, SA, dq1, TD_C1_H1, TR_C1_H2, TR_DZ2, TR_DH1, 0, 2 * NW);
                t3w_backward_wgrad<ACT, SA, 1, NBK>(ctx, grad + 2 * PN, xc, SA, dq2, TD_C2_H1, TR_C2_H2, TR_DZ2B, TR_DH1B, NW, 2 * NW);
                TPT_MARK(5);
                team_barrier();
                const bool policy_step = PD == 1 || (learn_it + 1) % PD == 0;      // TD3.py:101 (total_it = learn_it + 1)
                adam(PN, 2 * PN, 0, policy_step);          // critic_optimizer (+ the critics' soft update on a policy step)
                team_barrier();
                TPT_MARK(6);
                ++learn_it;
                if (policy_step) {
                    // actor_loss = (-critic_1(states, actor(states))).mean() with the updated critic_1
                    for (int e = tid; e < gbn * S; e += NT) { const int b = gb0 + e / S, i = e - (b - gb0) * S; xa[b * SA + i] = xc[b * SA + i]; }
                    __syncthreads();
                    t3w_forward<ACT, S, A, NBK>(ctx, params, xc, SA, 1, nullptr, xa, SA, S, thb, TD_A_H1, TD_A_H2, TR_A_H2);
                    t3w_forward<ACT, SA, 1, NBK>(ctx, params + PN, xa, SA, 0, dq2, nullptr, 0, 0, nullptr, TD_C1_H1, -1, TR_C1_H2);
                    const float dqa = -(1.0f / (float)B);
                    for (int b = tid; b < B; b += NT) dq1[b] = dqa;
                    __syncthreads();
                    t3w_backward_chain<ACT, SA, 1, NBK>(ctx, params + PN, dq1, TD_C1_H1, TR_C1_H2, TR_DZ2, -1, S, A, thb, dzl);
                    if (G > 1) {
                        for (int e = tid; e < B * A; e += NT) if (my_row(e / A)) gdz[e] = dzl[e];
                    }
                    t3w_backward_chain<ACT, S, A, NBK>(ctx, params, dzl, TD_A_H1, TR_A_H2, TR_DZ2B, TR_DH1B, 0, 0, nullptr, nullptr);
                    team_barrier();
                    if (G > 1) {
                        for (int e = tid; e < B * A; e += NT) dzl[e] = gdz[e];
                        __syncthreads();
                    }
                    t3w_backward_wgrad<ACT, S, A, NBK>(ctx, grad, xc, SA, dzl, TD_A_H1, TR_A_H2, TR_DZ2B, TR_DH1B, 0, NW);
                    TPT_MARK(7);
                    team_barrier();
                    adam(0, PN, 2, true);
                    team_barrier();
                    TPT_MARK(8);
                }
                }
                if (team_dead) break;                      // (uniform in the workgroup) the team gave up: leave, status -10
            }
            if (done_now > 0.5f) break;
        }
        ++episodes_run;
        if (team_dead) break;
        if (tid == 0 && g == 0 && a.out.episode_len) a.out.episode_len[chain * cfg.train_episodes + episode] = ep_len;
        __syncthreads();
        TPT_MARK(10);
        test_phase();
        TPT_MARK(9);
        if (tid == 0) {
            double sm_ = 0.0;
            for (int i = 0; i < T; ++i) sm_ += ret[i];
            const double tm = sm_ / (double)T;
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
    TPT_MARK(10);
    const int64_t remaining = cfg.step_budget - ((int64_t)train_steps + test_steps);
    const int test_before = test_steps;
    if (!team_dead) test_phase();
    if (budgeted) {
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
    TPT_MARK(9);
#if defined(LENV_PHASE_TIMING) && !defined(LENV_PHASE_TIMING_SUB)
    if (tid == 0 && chain == 0) { for (int pi = 0; pi < 12; ++pi) g_t3w_phase_cycles[pi] = pt_acc[pi]; g_t3w_phase_cycles[11] = bar_cycles; }
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
    if (a.out.final_params && g == 0) {
        for (int p = tid; p < a.P; p += NT) {
            int q;
            if (p < a.Pa) q = t3w_sd_to_arena(p, S, A);
            else { const int n = (p - a.Pa) / a.Pc; q = (1 + n) * PN + t3w_sd_to_arena(p - a.Pa - n * a.Pc, SA, 1); }
            a.out.final_params[chain * a.P + p] = params[q];
        }
    }
    if (a.out.status && (status != 0 || team_dead)) atomicMin(&a.out.status[chain], team_dead ? -10 : status);
#undef team_dead
}

}  // namespace lenv

using namespace lenv;

// the published cfg-5 shape in production form (checked by the caller: counter RNG, no trace, no hp, no ICM)
int lenv_wc_td3_shape(const lenv_td3_cfg *cfg)
{
    const int k_rep = cfg->same_action_num > 1 ? cfg->same_action_num : 1;
    if (cfg->test_mode != 0) return 0;      // BaseAgent.train without a test env (the evaluation harness's call): GEMM-queue kernel
    // default_config_cmc.yaml: TD3 on a VirtualEnv (three 3-96-96-x nets), batch 256, delayed policy updates
    if (cfg->hidden == 128 && cfg->layers == 2 && cfg->batch_size == 256 && cfg->virtual_env && cfg->rn_hidden == 96 && cfg->rn_layers == 2 && cfg->policy_delay == 2 &&
        !cfg->icm_enabled && !cfg->use_layer_norm && !cfg->rn_layer_norm && cfg->env_id == LENV_ENV_CMC && cfg->state_dim == 2 && cfg->action_dim == 1 &&
        cfg->test_episodes == 1 && cfg->act == LENV_ACT_RELU && k_rep == 2)
        return 4;
    // default_config_pendulum.yaml / default_config_halfcheetah.yaml with a fixed-shape TD3 (their td3 sections: batch 256, policy_delay 2, ten test episodes)
    if (cfg->hidden == 128 && cfg->layers == 2 && cfg->batch_size == 256 && cfg->virtual_env && cfg->policy_delay == 2 && !cfg->icm_enabled && !cfg->use_layer_norm &&
        !cfg->rn_layer_norm && cfg->test_episodes == 10 && cfg->act == LENV_ACT_RELU && k_rep == 1) {
        if (cfg->env_id == LENV_ENV_PENDULUM && cfg->state_dim == 3 && cfg->action_dim == 1 && cfg->rn_hidden == 32 && cfg->rn_layers == 2) return 5;
        if (cfg->env_id == LENV_ENV_CHEETAH_STANDIN && cfg->state_dim == 17 && cfg->action_dim == 6 && cfg->rn_hidden == 128 && cfg->rn_layers == 3) return 6;
    }
    if (!(cfg->hidden == 128 && cfg->layers == 2 && cfg->batch_size == 192 && cfg->rn_hidden == 128 && !cfg->virtual_env &&
          cfg->policy_delay == 1 && !cfg->icm_enabled && !cfg->use_layer_norm && !(cfg->rn_layer_norm && cfg->rn_layers >= 2) &&
          (cfg->reward_env_type == 0 || cfg->reward_env_type == 1 || cfg->reward_env_type == 2 || cfg->reward_env_type == 5 || cfg->reward_env_type == 6)))
        return 0;
    if (cfg->env_id == LENV_ENV_CHEETAH_STANDIN && cfg->state_dim == 17 && cfg->action_dim == 6 && cfg->test_episodes == 1 && cfg->rn_layers == 1 &&
        cfg->act == LENV_ACT_RELU && k_rep == 1)
        return 1;
    if (cfg->env_id == LENV_ENV_PENDULUM && cfg->state_dim == 3 && cfg->action_dim == 1 && cfg->test_episodes == 10 && cfg->rn_layers == 2 &&
        cfg->act == LENV_ACT_LEAKYRELU && k_rep == 1)
        return 2;
    if (cfg->env_id == LENV_ENV_CMC && cfg->state_dim == 2 && cfg->action_dim == 1 && cfg->test_episodes == 1 && cfg->rn_layers == 1 &&
        cfg->act == LENV_ACT_LEAKYRELU && k_rep == 2)
        return 3;
    return 0;
}

// parameters of the training env's nets that live in the chain's arena: a reward net with several hidden layers, or the three nets of a VirtualEnv
static int64_t t3w_env_arena_params(const lenv_td3_cfg *cfg)
{
    if (cfg->virtual_env) {
        const int64_t H = cfg->rn_hidden, in = cfg->state_dim + cfg->action_dim;
        int64_t hid = H * in + H;
        for (int l = 1; l < cfg->rn_layers; ++l) hid += H * H + H;
        return 3 * hid + (cfg->state_dim * H + cfg->state_dim) + 2 * (H + 1);
    }
    return cfg->rn_layers > 1 ? lenv_rn_num_params(cfg->reward_env_type, cfg->state_dim, cfg->info_dim, cfg->rn_hidden, cfg->rn_layers) : 0;
}

static void t3w_offsets(const lenv_td3_cfg *cfg, int64_t rb_cap, int RS, T3wArgs &a, int64_t *total)
{
    const int shape = lenv_wc_td3_shape(cfg);
    const int SA = 23, B = kT3wShapes[shape].B, NBK = B / 32;
    int64_t off = 0;
    auto take = [&](int64_t n) { int64_t r = off; off += (n + 3) & ~(int64_t)3; return r; };
    a.a_par = take(5 * 3 * (int64_t)t3p::PN); a.a_xc = take((int64_t)B * SA); a.a_xn = take((int64_t)B * SA); a.a_xa = take((int64_t)B * SA);
    a.a_th = take((int64_t)B * 6); a.a_dump = take((int64_t)T3W_NDUMP * NBK * wc::BLK); a.a_replay = take(rb_cap * RS);
    a.a_meter = take(2 * (int64_t)(cfg->train_episodes > 0 ? cfg->train_episodes : 1));
    a.a_gx = take(2 * (int64_t)B + (int64_t)B * 6);             // team exchange: dq1 | dq2 | dz
    a.a_bar = take(16);                                          // team barrier counter (one cache line of its own would be 32 floats; the slot is padded below)
    a.a_w2u = take(3 * (int64_t)wc::IMG);                        // team path: unit-major copies W2u[j][k] of the three online second layers
    a.a_rn = take(t3w_env_arena_params(cfg));                   // a deeper reward net / the VirtualEnv's nets
    *total = (off + 63) & ~(int64_t)63;
}

namespace lenv {
// the team barrier counters start at 0: a kernel rather than a memset node (see the status word in the fused kernels)
__global__ void t3w_team_reset_kernel(float *arena, int64_t arena_stride, int64_t a_bar, int64_t chains)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < chains) { unsigned *b = reinterpret_cast<unsigned *>(arena + c * arena_stride + a_bar); b[0] = 0u; b[8] = 0u; }
#ifdef LENV_DIAG_TEAM_TIMES
    if (c == 0) *reinterpret_cast<unsigned long long *>(arena + a_bar + 10) = __builtin_amdgcn_s_memrealtime();
#endif
}
}

static size_t t3w_lds_bytes(int shape, int P_rn)
{
    const int B = kT3wShapes[shape].B, T = kT3wShapes[shape].T;
    // the reward net itself, or its second hidden row, or the hidden rows of a VirtualEnv's three nets
    const size_t rn_floats = kT3wShapes[shape].venv ? 6 * 128 : (kT3wShapes[shape].rn_layers == 1 ? (size_t)((P_rn + 3) & ~3) : 128);
    const size_t lds_floats = 2 * (size_t)wc::IMG + 2 * (2 * wc::W + 8 * wc::W + 8) + rn_floats + 128 + 8 * (size_t)B + 2 * wc::W + 64 + 2 + 2 * (20 + 17 * T + T) + 2 * T + 1 +
                              20 + 8 + 56 + 4 + (sizeof(T3wCtx) + 3) / 4;
    return lds_floats * sizeof(float);
}
static void (*t3w_kernel(int shape))(const T3wArgs)
{
    switch (shape) {
    case 6: return td3_wavechain_kernel<6>;
    case 5: return td3_wavechain_kernel<5>;
    case 4: return td3_wavechain_kernel<4>;
    case 3: return td3_wavechain_kernel<3>;
    case 2: return td3_wavechain_kernel<2>;
    default: return td3_wavechain_kernel<1>;
    }
}

// Workgroups per chain.  A team only works when every workgroup of the launch is resident at the same time (its members wait for each
// other): 8 * ceil(chains / 8) * G workgroups must fit the device (occupancy API: one per CU at this kernel's LDS footprint).  192
// minibatch rows = 6 sample blocks: G = 6, 3, 2 or 1; 256 rows = 8 blocks: G = 8, 4, 2 or 1; cfg->team_size caps the choice (0 =
// automatic, 1 = the plain launch).
static int t3w_pick_team(const lenv_td3_cfg *cfg, int64_t chains)
{
    const int want = cfg->team_size > 0 ? cfg->team_size : 8;
    if (want == 1 || chains < 1) return 1;
    const int P_rn = cfg->virtual_env ? 0 : (int)lenv_rn_num_params(cfg->reward_env_type, cfg->state_dim, cfg->info_dim, cfg->rn_hidden, cfg->rn_layers);
    const int shape = lenv_wc_td3_shape(cfg);
    if (!shape) return 1;
    void (*kern)(const T3wArgs) = t3w_kernel(shape);
    const int64_t padded = 8 * ((chains + 7) / 8);
    const int NBK = kT3wShapes[shape].B / 32;
    for (int G : { 8, 6, 4, 3, 2 })
        if (NBK % G == 0 && G <= want && lenv_team_grid_resident(reinterpret_cast<const void *>(kern), wc::NT, t3w_lds_bytes(shape, P_rn), padded * G)) return G;
    return 1;
}

int lenv_wc_td3_team(const lenv_td3_cfg *cfg, int64_t chains) { return t3w_pick_team(cfg, chains); }

int64_t lenv_wc_td3_arena_floats(const lenv_td3_cfg *cfg, int64_t rb_cap, int RS)
{
    T3wArgs a;
    int64_t total;
    t3w_offsets(cfg, rb_cap, RS, a, &total);
    return total;
}

int lenv_wc_td3_launch(const lenv_td3_cfg *cfg, const float *theta, const float *eps, const int32_t *worker, const float *sign, const float *agent_init,
                       const uint64_t *rng_keys, int64_t chains, float *arena, int64_t arena_stride, int64_t rb_cap, int RS, int P, int Pa, int Pc, int P_rn,
                       const lenv_td3_out *out, hipStream_t stream)
{
    T3wArgs a;
    a.cfg = *cfg;
    a.theta = theta; a.eps = eps; a.worker = worker; a.sign = sign; a.agent_init = agent_init; a.rng_keys = rng_keys;
    a.arena = arena; a.arena_stride = arena_stride; a.out = *out; a.rb_cap = rb_cap; a.RS = RS; a.P = P; a.Pa = Pa; a.Pc = Pc; a.P_rn = P_rn;
    int64_t total;
    t3w_offsets(cfg, rb_cap, RS, a, &total);
    if (total > arena_stride) return LENV_ERR_WORKSPACE;
    const int shape = lenv_wc_td3_shape(cfg);
    if (!shape) return LENV_ERR_UNSUPPORTED;
    const size_t lds_bytes = t3w_lds_bytes(shape, P_rn);
    if (lds_bytes > 160 * 1024) return LENV_ERR_UNSUPPORTED;
    void (*kern)(const T3wArgs) = t3w_kernel(shape);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return LENV_ERR_LAUNCH;
    a.chains = chains;
    a.G = t3w_pick_team(cfg, chains);
    unsigned grid = (unsigned)chains;
    if (a.G > 1) {
        grid = (unsigned)(8 * ((chains + 7) / 8) * a.G);
        hipLaunchKernelGGL(t3w_team_reset_kernel, dim3((unsigned)((chains + 255) / 256)), dim3(256), 0, stream, arena, arena_stride, a.a_bar, chains);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(wc::NT), lds_bytes, stream, a);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}

#ifdef LENV_PHASE_TIMING_SUB
extern "C" int lenv_debug_t3v_jobs(unsigned long long *host_out)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lenv::g_t3v_jobs), sizeof(unsigned long long) * 20) == hipSuccess ? 0 : -4;
}
#endif
#ifdef LENV_PHASE_TIMING
extern "C" int lenv_debug_t3w_phase_cycles(unsigned long long *host_out)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lenv::g_t3w_phase_cycles), sizeof(unsigned long long) * 48) == hipSuccess ? 0 : -4;
}
#endif
