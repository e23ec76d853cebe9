// ddqn_se_inner_loop.hip -- fused NES inner loop for gfx950: one workgroup (16 wave64) per chain.
//
// Replaces, for `chains` independent (theta +/- eps) perturbations at once, the reference's
//   GTN_Worker.calc_score                     agents/GTN_worker.py:187-221
//     select_agent -> DDQN()                  agents/agent_utils.py:15-66, agents/DDQN.py:14-38
//     BaseAgent.train(env=SE, test_env=real)  agents/base_agent.py:64-153
//        DDQN.select_train_action             agents/DDQN.py:97-104
//        EnvWrapper.step -> VirtualEnv.step   envs/env_wrapper.py:16-47, envs/virtual_env.py:43-54
//        ReplayBuffer.add / sample            utils.py:24-45
//        DDQN.learn (+Adam, Polyak)           agents/DDQN.py:60-94
//        BaseAgent.test on the real env       agents/base_agent.py:155-227 (after every train episode)
//        early-out                            agents/base_agent.py:49-62,141-148
//     final BaseAgent.test, statistics.mean   agents/GTN_worker.py:199-209
//
// Design: the whole chain runs inside one launch of 12 waves.  SE weights
// (theta + sign*eps), the Q-net, its target, and all minibatch activations live in LDS; Adam state and
// master copies of the Q parameters live in the owning thread's registers; only the replay buffer is in
// HBM/L2.  Per training step: the last wave acts + steps the SE + appends the transition while the other waves
// prefetch their minibatch rows; then the 3*B (sample, pass) forward items run one per thread (pass-major, dense),
// one thread per sample forms the TD error, up to 12 waves back-propagate + reduce the batch gradient in
// micro-chunks, one thread per parameter applies Adam + Polyak.
// Arithmetic order is the oracle's canonical order (oracle/lenv_oracle.h) => results are bit-identical.
#include "lenv_device.cuh"

#include <type_traits>

namespace lenv {

constexpr int NT = 768;           // threads per chain: 12 wave64 = 3 per SIMD, 168 VGPRs each
constexpr int NW = NT / 64;       // waves per chain
constexpr int ENV_WAVE = NW - 1;  // wave that plays the environment/actor role during a training step
constexpr int FWD_THREADS = NT - 64;   // threads available for (sample, pass) forward items while the env wave acts
constexpr int MAX_B = FWD_THREADS / 3; // minibatch samples (3 forward items per sample, one item per thread)
constexpr int MAX_PPT = 2;        // Q-net parameters owned per thread (P_agent <= MAX_PPT*NT)

#ifndef LENV_DDQN_TEAM_GRP
#define LENV_DDQN_TEAM_GRP 4
#endif
#ifndef LENV_DDQN_WIDE_GRP
#define LENV_DDQN_WIDE_GRP 4                              // samples per group of a TWIDE gradient wave (all LDS reads of a group in flight)
#endif
#ifndef LENV_DDQN_WIDE_NPF
#define LENV_DDQN_WIDE_NPF 4                              // granule pairs per thread in flight in the TWIDE exchange
#endif
#ifndef LENV_DDQN_WIDE_CB
#define LENV_DDQN_WIDE_CB 10                              // hidden-unit pairs per block of the WIDE output-layer chains (all reads of a block in flight)
#endif

// Diagnostic build only (-DLENV_PHASE_TIMING): per-phase shader-clock totals of chain 0, never in the shipped library.
#ifdef LENV_PHASE_TIMING
__device__ unsigned long long g_phase_cycles[64];
#define PT_DECL unsigned long long pt_last = __builtin_readcyclecounter(), pt_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt_own[4] = {0, 0, 0, 0}
#define PT_MARK(i) do { unsigned long long pt_now = __builtin_readcyclecounter(); pt_acc[i] += pt_now - pt_last; pt_last = pt_now; } while (0)
// own-work stamp: time since the last PT_MARK at which THIS wave reached the coming barrier (no reset)
#define PT_OWN(i) do { pt_own[i] += __builtin_readcyclecounter() - pt_last; } while (0)
#define PT_FLUSH(first, n) do { if (chain == 0) for (int pi = 0; pi < (n); ++pi) g_phase_cycles[(first) + pi] = pt_acc[pi]; } while (0)
#else
#define PT_DECL
#define PT_MARK(i)
#define PT_OWN(i)
#define PT_FLUSH(first, n)
#endif

// LDS carve-up and derived sizes of one (S, A, Hq, Hse, B, T, grad_chunk) shape.  constexpr so that the kernel instantiated for the
// published CartPole configuration (SHAPE 1 below) gets every offset, trip count and tail condition as a literal, while every
// other shape reads the same struct from the kernel arguments.
struct InnerLayout {
    int P_q, P_se, se_net_size[3];
    int HP, chunk, n_chunks, n_chunks4, tanh16;
    int split_L, split_D;                             // forward items beyond the first nw-4 waves and the parts each is cut into (0 = no split)
    int o_se_w0T, o_se_b0, o_se_wout, o_se_bout, o_se_h, o_q_onl, o_q_tgt, o_q_w2, o_wscr, o_hB, o_sB, o_rda,
        o_dqB, o_qres, o_part, o_newrow, o_ctrl, o_ret, o_tanh, o_cand, o_cur_state, o_hX, lds_floats;
    int rc;                                           // LENV_OK or the reason the shape does not fit
};
constexpr int layout_mlp_params(int in, int H, int out) { return in * H + H + H * out + out; }
constexpr InnerLayout make_inner_layout(int S, int A, int Hq, int Hse, int B, int T, int grad_chunk, int nt, int nw, int max_ppt, int max_b)
{
    InnerLayout a{};
    a.P_q = layout_mlp_params(S, Hq, A);
    a.se_net_size[0] = layout_mlp_params(S + A, Hse, S);
    a.se_net_size[1] = a.se_net_size[2] = layout_mlp_params(S + A, Hse, 1);
    a.P_se = a.se_net_size[0] + a.se_net_size[1] + a.se_net_size[2];
    a.rc = LENV_ERR_UNSUPPORTED;
    if (a.P_q > max_ppt * nt) return a;
    a.HP = (Hq + 1) & ~1;                       // even (8-byte pair stores of h) ...
    if (((a.HP >> 1) & 1) == 0) a.HP += 2;      // ... with HP/2 odd: lanes = samples write pairs without bank conflicts
    a.chunk = grad_chunk > 0 ? grad_chunk : (B + nw - 1) / nw;
    a.n_chunks = (B + a.chunk - 1) / a.chunk;
    if (a.n_chunks > nw) return a;
    a.n_chunks4 = (a.n_chunks + 3) & ~3;
    const int K = S + A, HqPad = (Hq + 63) & ~63;
    // Items of the third forward wave of SIMDs 0 / 1 (see the kernel's "split" path): L of them, each cut into D parts over the
    // hidden-unit pairs so that the four third waves share their activations; the parts' h rows go through o_hX
    const int L_all = 3 * B - (nw - 4) * 64;
    const int split_D = (L_all > 0 && L_all <= 128 && (Hq + 1) / 2 >= 8) ? (L_all <= 64 ? 4 : (L_all <= 85 ? 3 : 2)) : 0;
    // the tanh image is carved first (offset 0, see det_tanh_lds_off): 16 bank-private copies when they fit, else one;
    // the split is dropped before the copies are
    for (int variant = 0; variant < 3; ++variant) {
        const int sixteen = variant < 2 ? 1 : 0;
        const int D = variant == 0 ? split_D : 0;
        if (variant == 1 && split_D == 0) continue;
        a.split_D = D; a.split_L = D ? L_all : 0;
        int o = 0;
#define LENV_TAKE(n) ([&]() { int r_ = o; o += ((n) + 3) & ~3; return r_; }())
        a.tanh16 = sixteen;
        a.o_tanh = LENV_TAKE(sixteen ? LENV_TANH16_FLOATS : LENV_TANH1_FLOATS);
        a.o_se_w0T = LENV_TAKE(3 * K * Hse); a.o_se_b0 = LENV_TAKE(3 * Hse); a.o_se_wout = LENV_TAKE((S + 2) * ((Hse + 3) & ~3)); a.o_se_bout = LENV_TAKE(S + 2);
        a.o_se_h = LENV_TAKE(2 * 3 * ((Hse + 3) & ~3));
        const int PRh = ((2 * S + 2 + 3) & ~3) + ((2 * A + 3) & ~3);      // floats per pair record (rec_pr<S, A>())
        a.o_q_onl = LENV_TAKE(((Hq + 1) / 2) * PRh + A); a.o_q_tgt = LENV_TAKE(((Hq + 1) / 2) * PRh + A); a.o_q_w2 = LENV_TAKE(A * ((Hq + 3) & ~3));
        a.o_wscr = LENV_TAKE(nw * HqPad);
        a.o_hB = LENV_TAKE(B * a.HP);
        a.o_sB = LENV_TAKE(B * ((S + 3) & ~3)); a.o_rda = LENV_TAKE(B * 4); a.o_dqB = LENV_TAKE(4 * B);
        a.o_qres = LENV_TAKE(3 * max_b * A);
        a.o_part = LENV_TAKE(a.n_chunks4 * a.P_q);
        a.o_newrow = LENV_TAKE(16); a.o_ctrl = LENV_TAKE(8 + nw);
        a.o_ret = LENV_TAKE(3 * T + 2);
        a.o_cand = LENV_TAKE(16 * A); a.o_cur_state = LENV_TAKE(16);
        a.o_hX = LENV_TAKE(a.split_L * a.HP);
#undef LENV_TAKE
        a.lds_floats = o;
        if ((long long)o * 4 <= 160 * 1024) { a.rc = LENV_OK; return a; }
    }
    return a;
}
// The published one-hidden-layer DDQN configurations of the reference, each with its own shape-specialised instantiation (SHAPE
// template parameter of the kernel): env, S, A, Q-net width / activation, batch, SE width / activation, test episodes, max_steps.
// chunk is the package's pick_grad_chunk value for the shape (the launch must carry the same one).
struct ShapeSpec { int env, S, A, Hq, q_act, B, Hse, se_act, T, max_steps, chunk; };
constexpr ShapeSpec kShapes[] = {
    { -1, 4, 2, 1, 0, 1, 1, 0, 1, 1, 1 },                                                                            // 0: generic (unused entry)
    { LENV_ENV_CARTPOLE, 4, 2, 57, LENV_ACT_TANH, 199, 83, LENV_ACT_LEAKYRELU, 10, 200, 17 },    // 1: default_config_cartpole_syn_env.yaml = BASELINE configs[1]
    { LENV_ENV_CARTPOLE, 4, 2, 64, LENV_ACT_RELU, 32, 128, LENV_ACT_LEAKYRELU, 1, 200, 3 },      // 2: default_config_cartpole.yaml
    { LENV_ENV_ACROBOT, 6, 3, 112, LENV_ACT_LEAKYRELU, 149, 167, LENV_ACT_PRELU, 10, 500, 38 },  // 3: default_config_acrobot_syn_env.yaml
};
constexpr int kNumShapes = sizeof(kShapes) / sizeof(kShapes[0]);

struct InnerArgs {
    lenv_ddqn_cfg cfg;
    const float *theta, *eps; const int32_t *worker; const float *sign;
    const float *agent_init; const uint64_t *rng_keys;
    lenv_tapes tapes; int has_tapes;
    float *replay; double *meter; int64_t rb_cap; int row_stride;
    lenv_inner_out out;
    InnerLayout L;
    // fp32 constants of the learn step, converted ONCE on the host exactly as the kernel used to ((float) of the double
    // expression): keeps ten double-precision config fields (20 SGPRs) and their conversions out of the step loop
    float f_gamma, f_norm, f_w1, f_w2, f_beta2, f_adam_eps, f_tau, f_omt;
    // torch.optim.Adam bias corrections per learn step t = 1, 2, ...: {-(lr / (1 - beta1^t)), sqrt(1 - beta2^t)} as fp32,
    // computed in double on the host with the same operation sequence the kernel's thread NT-1 used to run every step
    const float2 *adam_sched;
    // TEAM instantiation (a chain on team_G co-resident workgroups, see the kernel): per chain team_stride floats at team_ws --
    // 16 words (barrier counter, the members' XCD ids, give-up flag), then two copies (learn-step parity) of the chunk partials
    // [n_chunks4][P_q] as 8-byte {value, learn step + 1} granules
    float *team_ws;
    int64_t team_stride, chains;
    int team_G;
};

// Workgroup-internal flag in LDS: the env wave publishes "phase A of step `tag` is done" while the other waves are already
// in the minibatch forward; only waves that need its results poll.  LDS operations of one wave execute in order, so every
// LDS write the env wave issued before the flag store is visible to a wave that has read the new flag value.
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ void lds_flag_store(float *p, int tag)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    *(volatile lds_int *)((lds_int *)p) = tag;
}
__device__ __forceinline__ void lds_flag_wait(const float *p, int tag)
{
    while (*(volatile lds_int *)((lds_int *)const_cast<float *>(p)) != tag) __builtin_amdgcn_s_sleep(2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Q-net parameters live in LDS as PAIR records: hidden units (2q, 2q+1) share one record of PR = 2*RP floats
//   [W1[2q][0], W1[2q+1][0], ..., W1[2q][S-1], W1[2q+1][S-1] | b1[2q], b1[2q+1] | pad to 16 B || W2[0..A-1][2q] | W2[0..A-1][2q+1] | pad ]
// so one thread can evaluate two hidden units per iteration with packed fp32 math on adjacent register pairs (layer 1:
// the two units of the pair; output layer: the actions of one unit).  The layer-1 part [0, OW2) and the output part
// [OW2, PR) are separate 16-byte-aligned pieces: the forward loop reads them at different times (software pipeline).
// The output biases b2 follow the last record.  Canonical parameter index p -> LDS offset:
template <int S> constexpr int rec_ow2() { return (2 * S + 2 + 3) & ~3; }
template <int S, int A> constexpr int rec_pr() { return rec_ow2<S>() + ((2 * A + 3) & ~3); }
template <int S, int A>
__device__ __forceinline__ int packed_off(int p, int Hq, int PR)
{
    const int npairs = (Hq + 1) >> 1;
    const int nW1 = Hq * S;
    if (p < nW1) { int j = p / S, i = p - j * S; return (j >> 1) * PR + 2 * i + (j & 1); }
    p -= nW1;
    if (p < Hq) return (p >> 1) * PR + 2 * S + (p & 1);
    p -= Hq;
    if (p < A * Hq) { int aa = p / Hq; int j = p - aa * Hq; return (j >> 1) * PR + rec_ow2<S>() + (j & 1) * A + aa; }
    p -= A * Hq;
    return npairs * PR + p;
}

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// tanh_tab = LDS byte address of the workgroup's image of the canonical tanh table, tl = its addressing (lenv_device.cuh)
template <int ACT>
__device__ __forceinline__ float act_fwd_t(uint32_t tanh_tab, const TanhLds &tl, float prelu, float z)
{
    if constexpr (ACT == LENV_ACT_RELU) return z > 0.0f ? z : 0.0f;
    else if constexpr (ACT == LENV_ACT_LEAKYRELU) return z > 0.0f ? z : z * 0.01f;
    else if constexpr (ACT == LENV_ACT_TANH) return det_tanhf_lds(tanh_tab, tl, z);
    else if constexpr (ACT == LENV_ACT_PRELU) return z > 0.0f ? z : prelu * z;
    else return z;
}

template <int ACT>
__device__ __forceinline__ float act_bwd_t(float prelu, float a, float g)
{
    if constexpr (ACT == LENV_ACT_RELU) return a > 0.0f ? g : 0.0f;
    else if constexpr (ACT == LENV_ACT_LEAKYRELU) return a > 0.0f ? g : g * 0.01f;
    else if constexpr (ACT == LENV_ACT_TANH) return g * fma32(-a, a, 1.0f);
    else if constexpr (ACT == LENV_ACT_PRELU) return a > 0.0f ? g : prelu * g;
    else return g;
}

// Two-stage activation of a hidden-unit PAIR, so that the table gathers of the canonical tanh (lenv_device.cuh, v4) can
// be issued ahead of their use.  The pair form keeps the two-at-a-time steps (t + 2^19, the grid point, d) on packed fp32
// instructions; everything is the same operation sequence as det_tanhf, element by element.
template <int ACT>
struct ActPipe2 {
    v2f z, d;
    float4 k0, k1;
    __device__ __forceinline__ void issue(uint32_t tanh_tab, const TanhLds &tl, v2f zz)
    {
        z = zz;
        if constexpr (ACT == LENV_ACT_TANH) {
            const float a0 = __builtin_fabsf(zz.x), a1 = __builtin_fabsf(zz.y);
            const v2f t = {a0 < LENV_TANH_TMAX ? a0 : LENV_TANH_TMAX, a1 < LENV_TANH_TMAX ? a1 : LENV_TANH_TMAX};
            const v2f u = t + (v2f){LENV_TANH_MAGIC, LENV_TANH_MAGIC};
            d = t - (u - (v2f){LENV_TANH_MAGIC, LENV_TANH_MAGIC});
            k0 = det_tanh_lds_gather(tanh_tab, tl.off(__float_as_uint(u.x)));
            k1 = det_tanh_lds_gather(tanh_tab, tl.off(__float_as_uint(u.y)));
        }
    }
    __device__ __forceinline__ v2f finish(float prelu) const
    {
        if constexpr (ACT == LENV_ACT_TANH) return (v2f){det_tanh_poly(k0, d.x, z.x), det_tanh_poly(k1, d.y, z.y)};
        else return (v2f){act_fwd_t<ACT>(0u, TanhLds{}, prelu, z.x), act_fwd_t<ACT>(0u, TanhLds{}, prelu, z.y)};
    }
};

// acc = fmaf chain over j = 0..n-1 of h[j]*w[j] in index order (the canonical order); the loads are 16-byte and
// issued eight at a time so the chain is bound by fma latency, not by one LDS round trip per term.
// h and w must be 16-byte aligned.
__device__ __forceinline__ float seq_dot_lds(const float *h, const float *w, int n)
{
    float acc = 0.0f;
    const float4 *h4 = reinterpret_cast<const float4 *>(h), *w4 = reinterpret_cast<const float4 *>(w);
    const int n4 = n >> 2;
    int q = 0;
    for (; q + 4 <= n4; q += 4) {
        const float4 a0 = h4[q], a1 = h4[q + 1], a2 = h4[q + 2], a3 = h4[q + 3];
        const float4 b0 = w4[q], b1 = w4[q + 1], b2 = w4[q + 2], b3 = w4[q + 3];
        acc = fma32(a0.x, b0.x, acc); acc = fma32(a0.y, b0.y, acc); acc = fma32(a0.z, b0.z, acc); acc = fma32(a0.w, b0.w, acc);
        acc = fma32(a1.x, b1.x, acc); acc = fma32(a1.y, b1.y, acc); acc = fma32(a1.z, b1.z, acc); acc = fma32(a1.w, b1.w, acc);
        acc = fma32(a2.x, b2.x, acc); acc = fma32(a2.y, b2.y, acc); acc = fma32(a2.z, b2.z, acc); acc = fma32(a2.w, b2.w, acc);
        acc = fma32(a3.x, b3.x, acc); acc = fma32(a3.y, b3.y, acc); acc = fma32(a3.z, b3.z, acc); acc = fma32(a3.w, b3.w, acc);
    }
    for (; q < n4; ++q) {
        const float4 a0 = h4[q], b0 = w4[q];
        acc = fma32(a0.x, b0.x, acc); acc = fma32(a0.y, b0.y, acc); acc = fma32(a0.z, b0.z, acc); acc = fma32(a0.w, b0.w, acc);
    }
    for (int j = n4 << 2; j < n; ++j) acc = fma32(h[j], w[j], acc);
    return acc;
}

// Batch-1 greedy action of a Critic_DQN held in LDS (pair records W + row-major output rows W2rows):
// one wave, lane = hidden unit.  Returns argmax_a Q(obs) (first maximum), identical in every lane.
template <int S, int A, int PR, int QACT>
__device__ __forceinline__ int wave_q_argmax(const float *W, const float *W2rows, int HqP, const float (&obs)[S], float *scratch,
                                             int Hq, float prelu, int lane, uint32_t tanh_tab, const TanhLds &tl)
{
    for (int j = lane; j < Hq; j += 64) {
        const float *rec = W + (j >> 1) * PR + (j & 1);
        float z = 0.0f;
#pragma unroll
        for (int i = 0; i < S; ++i) z = fma32(obs[i], rec[2 * i], z);
        z = z + rec[2 * S];
        scratch[j] = act_fwd_t<QACT>(tanh_tab, tl, prelu, z);
    }
    wave_sync();
    float q = 0.0f;
    if (lane < A) q = seq_dot_lds(scratch, W2rows + lane * HqP, Hq) + W[((Hq + 1) >> 1) * PR + lane];
    wave_sync();
    float best = __shfl(q, 0);
    int arg = 0;
#pragma unroll
    for (int a = 1; a < A; ++a) {
        float v = __shfl(q, a);
        if (v > best) { best = v; arg = a; }
    }
    return arg;
}

template <int ENV>
__device__ __forceinline__ void real_env_step(double (&st)[4], int action, double &rew, int &done)
{
    if constexpr (ENV == LENV_ENV_CARTPOLE) cartpole_step(st, action, rew, done);
    else acrobot_step(st, action, rew, done);
}

template <int ENV, int S>
__device__ __forceinline__ void real_env_obs(const double (&st)[4], float (&obs)[S])
{
    if constexpr (ENV == LENV_ENV_CARTPOLE) {
#pragma unroll
        for (int i = 0; i < 4; ++i) obs[i] = (float)st[i];
    } else {
        obs[0] = (float)det_cos(st[0]); obs[1] = (float)det_sin(st[0]);
        obs[2] = (float)det_cos(st[1]); obs[3] = (float)det_sin(st[1]);
        obs[4] = (float)st[2]; obs[5] = (float)st[3];
    }
}

// SHAPE > 0 = a published configuration (kShapes above; 1 = BASELINE configs[1]): network widths, batch size, chunking and the whole LDS layout are
// literals, which frees the scalar registers that otherwise carry them (the generic build spills ~350 SGPR values to VGPR lanes
// and reloads ~100 of them per learn step with v_readlane, a VALU slot each) and removes the tail code of the pair loops.
//
// TEAM = a chain runs on G co-resident workgroups of one XCD (launches that leave most of the GPU idle: the shards of a population
// spread over several GPUs).  Every member keeps the whole chain state and repeats the cheap serial parts (acting, the SE step, the
// replay append -- identical rows to the same addresses --, the test episodes); the minibatch is dealt over the members by whole
// gradient micro-chunks: a member runs the forward items, TD errors and chunk gradients of ITS samples only, leaves its chunk
// partials in the chain's exchange buffer, the members meet at ONE agent-scope barrier per learn step, read the other members'
// partials, and each applies the same Adam step to its own copy of the parameters.  The partials are summed in chunk order by
// every member, so the bits are those of the one-workgroup launch for every G.
// TWIDE = the TEAM instantiation for teams of three and more (every forward item cut over the idle lanes, see `wide` below): a separate
// instantiation, so that neither carries the other's forward code -- the kernel sits at the 168-VGPR cap of a 12-wave workgroup, and
// every extra code path showed up as spill reloads in all of them.
// RENV = the agent trains on a RewardEnv over the REAL env (gtn.synthetic_env_type 1, envs/reward_env.py:61-133; default_config_cartpole_reward_env.yaml)
// or -- reward_env_type 0 -- on the real env itself (experiments/syn_env_run_vary_hp.py:47-54, mode 0): the env wave steps the real env's physics
// (fp64 state, TimeLimit at max_steps) and shapes the reward with the perturbed reward network theta (S -> se_hidden -> 1), whose phi(s') it keeps
// for the next step; there is nothing to speculate on.  Its own instantiations (one workgroup per chain), so that the VirtualEnv builds carry none
// of it.  Oracle: ddqn_se_chain_impl's reward_env branch.
template <int ENV, int S, int A, int QACT, int PPT, int SHAPE = 0, bool TEAM = false, bool TWIDE = false, bool RENV = false>
__global__ __launch_bounds__(NT) void ddqn_se_inner_kernel(const InnerArgs a)
{
    static_assert(!TWIDE || TEAM, "TWIDE is a TEAM layout");
    static_assert(!RENV || (SHAPE == 0 && !TEAM), "RewardEnv / real-env training: generic one-workgroup instantiations only");
    extern __shared__ __align__(16) float lds[];
    const lenv_ddqn_cfg &cfg = a.cfg;
    constexpr bool FIXED = SHAPE != 0;
    constexpr ShapeSpec SPEC = kShapes[SHAPE];
    constexpr int FIX_HQ = SPEC.Hq, FIX_HSE = SPEC.Hse, FIX_B = SPEC.B, FIX_T = SPEC.T, FIX_CHUNK = SPEC.chunk, FIX_MAX_STEPS = SPEC.max_steps;
    static_assert(!FIXED || (SPEC.env == ENV && SPEC.S == S && SPEC.A == A && SPEC.q_act == QACT), "shape table entry vs instantiation");
    constexpr InnerLayout LC = make_inner_layout(S, A, FIX_HQ, FIX_HSE, FIX_B, FIX_T, FIX_CHUNK, NT, NW, MAX_PPT, MAX_B);
    static_assert(!FIXED || LC.rc == LENV_OK, "the fixed shape must fit");
#define LV(f) (FIXED ? LC.f : a.L.f)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // TEAM: block x + 8 k is member k % G of chain 8 (k / G) + x -- consecutive blocks go to consecutive XCDs, so a chain's members share one
    int64_t chain_ = blockIdx.x;
    int g_ = 0, G_ = 1;
    if constexpr (TEAM) {
        G_ = a.team_G;
        const int64_t xb = blockIdx.x, k = xb >> 3;
        chain_ = 8 * (k / G_) + (xb & 7);
        g_ = (int)(k % G_);
        if (chain_ >= a.chains) return;
    }
    const int64_t chain = chain_;
    const int g = g_, G = G_;
    constexpr int K = S + A;
    constexpr int OW2 = rec_ow2<S>();                 // offset of the output-layer part inside a pair record
    constexpr int PR = rec_pr<S, A>();                // floats per pair record
    constexpr int RS = (2 * S + 3 + 3) & ~3;          // replay row stride (floats)
    constexpr int SP = (S + 3) & ~3;                  // minibatch-state row stride in LDS (16-byte rows)
    // Literal in the SHAPE 1 build: SE width, batch size, test episodes, parameter count, chunking and every LDS offset.  The Q-net
    // width stays a run-time value outside the minibatch forward: as a literal it makes the compiler keep more per-lane state
    // live and spill ~70 VGPRs to scratch inside the forward and gradient loops (measured: 11.4 us per learn step instead of 8.8).
    const int Hq = cfg.q_hidden, Hse = FIXED ? FIX_HSE : cfg.se_hidden, B = FIXED ? FIX_B : cfg.batch_size;
    const int HP = a.L.HP, P = LV(P_q), T_EP = FIXED ? FIX_T : cfg.test_episodes, MAX_STEPS = FIXED ? FIX_MAX_STEPS : cfg.max_steps;
    const int HqP = (Hq + 3) & ~3, HseP = (Hse + 3) & ~3;

    float *se_w0T = lds + LV(o_se_w0T), *se_b0 = lds + LV(o_se_b0), *se_wout = lds + LV(o_se_wout), *se_bout = lds + LV(o_se_bout);
    float *se_h = lds + LV(o_se_h), *q_onl = lds + LV(o_q_onl), *q_tgt = lds + LV(o_q_tgt), *q_w2 = lds + LV(o_q_w2);
    float *wscr = lds + LV(o_wscr) + wave * ((Hq + 63) & ~63);
    float *se_hw = se_h + (wave & 1) * 3 * HseP;      // per-wave SE hidden scratch (waves NW-2 / NW-1)
    float *cand = lds + LV(o_cand), *cur_state = lds + LV(o_cur_state);
    float *hB = lds + LV(o_hB), *sB = lds + LV(o_sB), *rda = lds + LV(o_rda), *dqB = lds + LV(o_dqB);
    float *qres = lds + LV(o_qres), *part = lds + LV(o_part), *newrow = lds + LV(o_newrow);
    float *ctrl = lds + LV(o_ctrl);            // [0..1] done (double buffered by step parity), [2] break, [8..] wave step counts (every use is barrier-separated)
    double *ret = reinterpret_cast<double *>(lds + LV(o_ret));   // [test_episodes] returns
    int *tlen = reinterpret_cast<int *>(lds + LV(o_ret) + 2 * T_EP);   // [test_episodes] lengths of the last test phase
    // LDS image of the canonical tanh table at the START of the workgroup's LDS (its base folds into the DS immediate
    // offset): 16 bank-private copies (32 KB, conflict-free gathers) when the shapes leave room, one copy (2 KB) otherwise
    det_tanh_lds_stage(lds, LV(tanh16) != 0, tid, NT);
    // LDS byte address of the image: the literal 0 (this kernel has no static LDS, so its dynamic LDS starts at address
    // 0; a literal lets the gather address be the bare v_and_or result).  Verified below, never assumed silently.
    constexpr uint32_t tanh_tab = 0u;
    if (lds_addr_of(lds) != 0u) { if (tid == 0 && a.out.status) a.out.status[chain] = -7; return; }
    const TanhLds tl = TanhLds::make(LV(tanh16) != 0, lane);

    // ---------------- stage the perturbed SE: W = theta + sign*eps[worker]  (GTN_worker.py:165-175) ----------------
    if constexpr (RENV) {
        // the reward network (RewardEnv.build_reward_net, reward_env.py:29-53: Drn -> Hse -> 1 with Drn = S, or the 1-input dummy of type 0 that
        // is never evaluated), flat in Module.parameters() order W0 [Hse][Drn] | b0 | Wout [Hse] | bout, into the SE's rows: first layer
        // transposed (lane = hidden unit reads consecutive words), output row 0
        const float sg = a.eps ? a.sign[chain] : 0.0f;
        const int Drn = cfg.reward_env_type == 0 ? 1 : S, P_rn = Drn * Hse + 2 * Hse + 1;
        const float *e = a.eps ? a.eps + (int64_t)a.worker[chain] * P_rn : nullptr;
        for (int i = tid; i < P_rn; i += NT) {
            const float w = e ? fma32(sg, e[i], a.theta[i]) : a.theta[i];
            int r = i;
            if (r < Hse * Drn) { const int j = r / Drn, k = r - j * Drn; se_w0T[k * Hse + j] = w; }
            else if ((r -= Hse * Drn) < Hse) se_b0[r] = w;
            else if ((r -= Hse) < Hse) se_wout[r] = w;
            else se_bout[0] = w;
        }
    } else {
        const float sg = a.eps ? a.sign[chain] : 0.0f;
        const int P_se = LV(P_se), sn0 = LV(se_net_size[0]), sn1 = LV(se_net_size[1]);
        const float *e = a.eps ? a.eps + (int64_t)a.worker[chain] * P_se : nullptr;
        for (int i = tid; i < P_se; i += NT) {
            float w = e ? fma32(sg, e[i], a.theta[i]) : a.theta[i];
            int net = 0, r = i;
            if (r >= sn0) { r -= sn0; net = 1; if (r >= sn1) { r -= sn1; net = 2; } }
            const int orow = net == 0 ? 0 : (net == 1 ? S : S + 1);
            if (r < Hse * K) { int j = r / K, k = r - j * K; se_w0T[(net * K + k) * Hse + j] = w; }
            else if ((r -= Hse * K) < Hse) se_b0[net * Hse + r] = w;
            else {
                r -= Hse;
                const int n_out = net == 0 ? S : 1;
                if (r < n_out * Hse) { int o = r / Hse, j = r - o * Hse; se_wout[(orow + o) * HseP + j] = w; }
                else se_bout[orow + (r - n_out * Hse)] = w;
            }
        }
    }
    // ---------------- fresh DDQN agent: online = target = agent_init (DDQN.py:33-35), Adam state 0 ----------------
    // thread tid owns parameters tid and tid+NT (master copy, target, Adam m/v in registers)
    float p_onl[PPT], p_tgt[PPT], p_m[PPT], p_v[PPT];
    int my_off[PPT], my_w2[PPT];
    {
        // pad slots and the missing unit of an odd last pair are read (and masked) by the forward loop: keep them finite
        const int rec_floats = ((Hq + 1) >> 1) * PR + A;
        for (int i = tid; i < rec_floats; i += NT) { q_onl[i] = 0.0f; q_tgt[i] = 0.0f; }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = tid + k * NT;
        p_onl[k] = p_tgt[k] = p_m[k] = p_v[k] = 0.0f;
        my_off[k] = 0; my_w2[k] = -1;
        if (p < P) {
            p_onl[k] = a.agent_init[chain * P + p];
            p_tgt[k] = p_onl[k];
            my_off[k] = packed_off<S, A>(p, Hq, PR);
            q_onl[my_off[k]] = p_onl[k];
            q_tgt[my_off[k]] = p_tgt[k];
            const int r2 = p - (Hq * S + Hq);          // row-major copy of the output layer for the batch-1 chains
            my_w2[k] = (r2 >= 0 && r2 < A * Hq) ? (r2 / Hq) * HqP + (r2 % Hq) : -1;
            if (my_w2[k] >= 0) q_w2[my_w2[k]] = p_onl[k];
        }
    }
    if (tid < 8 + NW) ctrl[tid] = 0.0f;
    // GB2_LANE (the published CartPole shape: 57 hidden units in rows of 58 words): the output-bias gradient gb2[a] = sum_b dm_b[a] is the
    // output-weight gradient of a virtual hidden unit whose activation is 1.0 -- fma(dm, 1.0, acc) and acc + dm round the same sum once.  The
    // idle lane 57 of every gradient wave plays that unit (the pad word of every h row holds 1.0 from here on; the forward never writes it:
    // the last pair stores its first word only), and the per-sample packed add that every lane spent on gb2 is gone from the loop.
    constexpr bool GB2_LANE = SHAPE == 1 && !TWIDE && A == 2 && (FIX_HQ & 1) == 1 && FIX_HQ < 64;
    if constexpr (GB2_LANE) { for (int b = tid; b < B; b += NT) hB[b * HP + FIX_HQ] = 1.0f; }
    for (int i = tid; i < LV(n_chunks4) * P; i += NT) part[i] = 0.0f;  // padding chunk slots stay +0 (see the Adam phase)
    __syncthreads();

    const uint64_t key = a.rng_keys ? a.rng_keys[chain] : 0;
    // the shape-specialised build serves production launches only: counter RNG, no step trace (tests with tapes / traces take the generic one)
    const bool tape = !FIXED && cfg.rng_mode == LENV_RNG_TAPE;
    int status = 0;
    // counters (all uniform): one eps-uniform draw and one replay append per train step
    int train_steps = 0, n_act = 0, learn_it = 0, n_test_ep = 0, test_steps = 0, wr_pos = 0;
    double eps_g = cfg.eps_init;
    float *rb = a.replay + chain * a.rb_cap * RS;
    const int rb_cap = (int)a.rb_cap;
    double *meter = a.meter + chain * cfg.train_episodes;
    const double reset_lim = ENV == LENV_ENV_CARTPOLE ? 0.05 : 0.1;
    int episodes_run = 0;
    float state[S];
#pragma unroll
    for (int i = 0; i < S; ++i) state[i] = 0.0f;

    // forward work item of this thread: item = pass*B + b (pass-major, dense over the first 3*B threads)
    //
    // SPLIT (layout split_D > 0): 3 B items fill waves 0 .. NW-5 and spill L = 3 B - 64 (NW-4) <= 128 items into the third wave of
    // SIMD 0 (and 1), while the third waves of SIMDs 2 / 3 (speculation, env) have little to do in the forward interval: the
    // phase would end when SIMDs 0 / 1 have issued three waves' worth of vector work.  So the L spilled items (all of pass 2:
    // target net on the next states) are cut into D parts over the hidden-unit pairs, and lane u < L D of waves NW-4 .. NW-1
    // computes the ACTIVATIONS of part u / L of item u % L into an LDS row; the output layer's accumulation -- the one
    // sequential piece, k ascending over the hidden units -- is then run per item by the thread that owned it, from those rows.
    // Same operations on the same values in the same order: the bits do not change.
    // TEAM: this member's share of the minibatch = whole micro-chunks [mc0, mc1) = samples [mb0, mb0 + Bm)
#ifdef LENV_DDQN_UNEVEN_MC
    // EXPERIMENT build only (tools/uneven_team_ab.sh, docs/notebook_r06.md section 1): teams of two dealt UNEVENLY -- member 0 takes chunks
    // [0, LENV_DDQN_UNEVEN_MC), member 1 the rest -- to time a primary's learn step when a helper workgroup takes a quarter of its minibatch
    const int mc0 = TEAM ? (G == 2 ? (g ? LENV_DDQN_UNEVEN_MC : 0) : g * LV(n_chunks) / G) : 0;
    const int mc1 = TEAM ? (G == 2 ? (g ? LV(n_chunks) : LENV_DDQN_UNEVEN_MC) : (g + 1) * LV(n_chunks) / G) : LV(n_chunks);
#else
    const int mc0 = TEAM ? g * LV(n_chunks) / G : 0, mc1 = TEAM ? (g + 1) * LV(n_chunks) / G : LV(n_chunks);
#endif
    const int mb0 = mc0 * LV(chunk), Bm = (mc1 * LV(chunk) < B ? mc1 * LV(chunk) : B) - mb0;
    // TEAM: a member of a team of two has 3 Bm = ~300 items = four full waves (one per SIMD) and a fifth, part-filled one that would
    // make SIMD 0 carry two; those spilled items are cut four ways over waves 4 .. 7, which have nothing else to do in the interval
    // (the same machinery, one SIMD-round earlier; bigger teams have at most one forward wave per SIMD anyway)
    const bool tsplit = TEAM && !TWIDE && 3 * Bm > 256 && 3 * Bm - 256 <= 64 && 3 * Bm - 256 <= LV(split_L) && 2 * Bm <= 256 && LV(split_D) > 0;
    // TEAM, three and more members (WIDE): a member has at most 204 items -- two to four waves, each ALONE on its SIMD, and an item's 29
    // pair stages are a chain of dependent LDS round trips (7.3 k cycles per learn step at G = 6, tools/phase_timing.py) while eight waves
    // idle.  So EVERY item is cut WIDE_D ways over the hidden-unit pairs: lane u < 3 Bm WIDE_D computes the activations of part u / (3 Bm)
    // of item u % (3 Bm) into an LDS row with a branch-free pipelined loop of the same trip count in every lane (the parts overlap by a
    // pair or two where the pair count does not divide: identical values to identical words); ONE workgroup barrier; then lane t < 3 Bm
    // runs item t's output layer -- the sequential piece, k ascending -- from its row.  Rows: pass 0 in the sample's own h row (the backward
    // needs it there), passes 1 / 2 in h rows of samples this member does not own (3 Bm <= B) or, pass 2, in the split layout's rows.
    // Same operations on the same values in the same order: the bits do not change.
    const int WIDE_L = 3 * Bm;
    // parts per item: six when that fits the eleven waves next to the env wave (teams of six: 102 items, 612 lanes), else three on at most
    // eight waves (teams of four: 153 items, 459 lanes -- four parts on ten waves contend for the SIMDs and lose to whole items, measured)
    const int WIDE_D = TWIDE ? (6 * WIDE_L <= (NW - 1) * 64 ? 6 : 3) : 0;
    const bool wide = TWIDE;                                // (the host launches TWIDE exactly when ddqn_team_wide_ok() holds for every member)
    const int SPLIT_D = TEAM ? (tsplit ? 4 : 0) : LV(split_D), SPLIT_L = TEAM ? (tsplit ? 3 * Bm - 256 : 0) : LV(split_L);
    const bool split = SPLIT_D > 0;
    constexpr int SPLIT_W0 = TEAM ? 4 : NW - 4;          // first of the four waves that share the spilled items' activations
    constexpr int SPLIT_T0 = SPLIT_W0 * 64;
    const int split_u = tid - SPLIT_T0;                  // lane number inside them
    const bool h_lane = wide ? tid < WIDE_L * WIDE_D : (split && split_u >= 0 && split_u < SPLIT_L * SPLIT_D);
    const int wide_item = wide && h_lane ? tid % WIDE_L : 0;
    const int split_li = wide ? wide_item % Bm : (h_lane ? split_u % SPLIT_L : 0), split_part = wide ? (h_lane ? tid / WIDE_L : 0) : (h_lane ? split_u / SPLIT_L : 0);
    const bool chain_lane = !wide && split && split_u >= 0 && tid < 3 * Bm;
    const bool split_wave = !wide && split && wave >= SPLIT_W0 && wave < SPLIT_W0 + 4;
    const int fwd_pass = wide ? wide_item / Bm : ((split && split_u >= 0) ? 2 : (tid < Bm ? 0 : (tid < 2 * Bm ? 1 : 2)));
    const int fwd_b = wide ? mb0 + split_li : ((split && split_u >= 0) ? mb0 + SPLIT_T0 + split_li - 2 * Bm : mb0 + tid - fwd_pass * Bm);
    const bool full_item = !wide && (split ? tid < SPLIT_T0 : tid < 3 * Bm);  // runs all pairs of its item, output layer included
    const bool fwd_active = full_item || h_lane;          // samples a replay row
    // WIDE: the LDS row that holds the activations of (pass, sample index bi inside the member's share)
    auto wide_row = [&](int pass, int bi) -> float * {
        if (pass == 0) return hB + (mb0 + bi) * HP;
        if (pass == 2 && 3 * Bm > B) return lds + LV(o_hX) + bi * HP;
        int r = mb0 + Bm + (pass == 1 ? bi : Bm + bi);   // the (pass - 1) Bm + bi-th row behind the member's own, wrapping
        if (r >= B) r -= B;
        return hB + r * HP;
    };

    // speculation layout: waves without forward items (at most the last two) evaluate the SE for every action of the
    // NEXT step while the other waves run the minibatch forwards; the env wave then only has to pick a candidate.
    const int n_fwd_waves = ((wide ? WIDE_L * WIDE_D : 3 * Bm) + 63) >> 6;
    const int first_spec = n_fwd_waves > NW - 2 ? n_fwd_waves : NW - 2;
    const int n_spec = NW - first_spec;                 // 0, 1 or 2 (uniform)
    bool spec_valid = false;

    // EnvWrapper.step -> VirtualEnv.step for ONE wave: x = [onehot(action), st]; returns output `lane` (< S+2) of
    // [next_state(S), reward, done] in lane `lane`.  Lane = hidden unit; outputs are sequential fmaf chains.
    auto se_eval = [&](float *hbuf, const float (&st)[S], int action) -> float {
        float x[K];
#pragma unroll
        for (int k = 0; k < A; ++k) x[k] = (k == action) ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < S; ++i) x[A + i] = st[i];
        auto se_hidden = [&](auto act_tag) {
            constexpr int SEACT = decltype(act_tag)::value;
            for (int uu = lane; uu < 3 * Hse; uu += 64) {
                const int net = (uu >= Hse ? 1 : 0) + (uu >= 2 * Hse ? 1 : 0), j = uu - net * Hse;   // no division
                const float *w = se_w0T + net * (K * Hse) + j;
                float wv[K];
#pragma unroll
                for (int k = 0; k < K; ++k) { wv[k] = *w; w += Hse; }      // every weight read in flight before the chain
                const float bias = se_b0[uu];
                float z = 0.0f;
#pragma unroll
                for (int k = 0; k < K; ++k) z = fma32(x[k], wv[k], z);
                z = z + bias;
                hbuf[net * HseP + j] = act_fwd_t<SEACT>(tanh_tab, tl, cfg.se_prelu, z);
            }
        };
        if constexpr (FIXED) se_hidden(std::integral_constant<int, SPEC.se_act>{});             // the published SE's activation
        else switch (cfg.se_act) {
        case LENV_ACT_RELU: se_hidden(std::integral_constant<int, LENV_ACT_RELU>{}); break;
        case LENV_ACT_LEAKYRELU: se_hidden(std::integral_constant<int, LENV_ACT_LEAKYRELU>{}); break;
        case LENV_ACT_TANH: se_hidden(std::integral_constant<int, LENV_ACT_TANH>{}); break;
        case LENV_ACT_PRELU: se_hidden(std::integral_constant<int, LENV_ACT_PRELU>{}); break;
        default: se_hidden(std::integral_constant<int, LENV_ACT_IDENTITY>{}); break;
        }
        wave_sync();
        float acc = 0.0f;
        if (lane < S + 2) {
            const int net = lane < S ? 0 : (lane == S ? 1 : 2);
            acc = seq_dot_lds(hbuf + net * HseP, se_wout + lane * HseP, Hse) + se_bout[lane];
        }
        wave_sync();
        return acc;
    };

    // RENV: phi(ob) of the reward network for ONE wave (lane = hidden unit, the output a sequential fmaf chain), the same value in every lane
    [[maybe_unused]] auto rn_eval = [&](float *hbuf, const float (&ob)[S]) -> float {
        auto rn_hidden = [&](auto act_tag) {
            constexpr int SEACT = decltype(act_tag)::value;
            for (int j = lane; j < Hse; j += 64) {
                const float *w = se_w0T + j;
                float wv[S];
#pragma unroll
                for (int k = 0; k < S; ++k) { wv[k] = *w; w += Hse; }
                const float bias = se_b0[j];
                float z = 0.0f;
#pragma unroll
                for (int k = 0; k < S; ++k) z = fma32(ob[k], wv[k], z);
                z = z + bias;
                hbuf[j] = act_fwd_t<SEACT>(tanh_tab, tl, cfg.se_prelu, z);
            }
        };
        switch (cfg.se_act) {
        case LENV_ACT_RELU: rn_hidden(std::integral_constant<int, LENV_ACT_RELU>{}); break;
        case LENV_ACT_LEAKYRELU: rn_hidden(std::integral_constant<int, LENV_ACT_LEAKYRELU>{}); break;
        case LENV_ACT_TANH: rn_hidden(std::integral_constant<int, LENV_ACT_TANH>{}); break;
        case LENV_ACT_PRELU: rn_hidden(std::integral_constant<int, LENV_ACT_PRELU>{}); break;
        default: rn_hidden(std::integral_constant<int, LENV_ACT_IDENTITY>{}); break;
        }
        wave_sync();
        float acc = 0.0f;
        if (lane == 0) acc = seq_dot_lds(hbuf, se_wout, Hse) + se_bout[0];
        wave_sync();
        return __shfl(acc, 0);
    };
    // RENV: the real training env's own state (fp64, RewardEnv.real_env), the steps of its episode (TimeLimit), phi(s) of the state it is in
    [[maybe_unused]] double st_d[4] = {0.0, 0.0, 0.0, 0.0};
    [[maybe_unused]] int env_steps = 0;
    [[maybe_unused]] float phi_s = 0.0f;
    [[maybe_unused]] bool have_phi = false;
    // lenv_ddqn_cfg::test_mode 1 = BaseAgent.train(env, test_env=None) (base_agent.py:134-148): no per-episode tests, the training env's own
    // episode reward feeds the meter (the env wave sums it); generic instantiations only
    const bool no_test_env = FIXED ? false : cfg.test_mode == 1;
    [[maybe_unused]] float tr_reward = 0.0f;

    // one real-env test phase (BaseAgent.test + DDQN.select_test_action): wave w plays test episodes w, w+NW, ...
    auto test_phase = [&]() {
        int my_steps = 0;
        for (int te = wave; te < T_EP; te += NW) {
            double st[4];
            const int64_t row = (int64_t)n_test_ep + te;
            if (tape) {
                if (row >= a.tapes.test_reset_stride) { status = -5; for (int i = 0; i < 4; ++i) st[i] = 0.0; }
                else for (int i = 0; i < 4; ++i) st[i] = a.tapes.test_reset[(chain * a.tapes.test_reset_stride + row) * 4 + i];
            } else {
                for (int i = 0; i < 4; ++i)
                    st[i] = -reset_lim + (2 * reset_lim) * u64_to_unit(rng_u64(key, STREAM_TEST_RESET, (uint64_t)(row * 4 + i)));
            }
            float ep_reward = 0.0f;
            int ep_steps = 0;
            for (int t = 0; t < MAX_STEPS; ++t) {
                float obs[S];
                real_env_obs<ENV, S>(st, obs);
                const int act = wave_q_argmax<S, A, PR, QACT>(q_onl, q_w2, HqP, obs, wscr, Hq, cfg.q_prelu, lane, tanh_tab, tl);
                double rew; int done;
                real_env_step<ENV>(st, act, rew, done);
                ep_reward = ep_reward + (float)rew;
                ++my_steps; ++ep_steps;
                if (done) break;
            }
            if (lane == 0) { ret[te] = (double)ep_reward; tlen[te] = ep_steps; }
        }
        if (lane == 0) ctrl[8 + wave] = __int_as_float(my_steps);
        n_test_ep += T_EP;
        __syncthreads();
        for (int w = 0; w < NW; ++w) test_steps += __float_as_int(ctrl[8 + w]);
    };

    // ReplayBuffer.sample (utils.py:34-45) for forward item (fwd_b): draw the index of learn step `lit` and start loading the
    // row.  Issued one step AHEAD (right after the TD error, so the HBM/L2 latency hides behind the gradient and Adam
    // phases); a row equal to the slot the env wave is about to write is patched from LDS at use time.
    float4 rowv[RS / 4];                                // kept as whole vectors: nothing touches them until the forward
    int my_idx = -1, pf_status = 0;
    bool pf_valid = false;
    // index draw (status through st) and row load, separately: the prefetch of the next step's row has a third case in between
    auto draw_row = [&](int lit, int size_after, int &st) -> int {
        const int64_t n = (int64_t)lit * B + fwd_b;
        int idx;
        st = 0;
        if (tape) {
            if (n >= a.tapes.replay_idx_stride) { st = -4; idx = 0; }
            else idx = a.tapes.replay_idx[chain * a.tapes.replay_idx_stride + n];
            if (idx < 0 || idx >= size_after) { if (!st) st = -6; idx = 0; }
        } else idx = (int)rng_replay_below(key, (uint64_t)n, (uint32_t)size_after);
        return idx;
    };
    auto load_row = [&](int new_pos) {
        if (my_idx != new_pos) {
            const float4 *src = reinterpret_cast<const float4 *>(rb + (int64_t)my_idx * RS);
#pragma unroll
            for (int v = 0; v < RS / 4; ++v) rowv[v] = src[v];
        }
    };
    auto fetch_row = [&](int lit, int size_after, int new_pos) {
        my_idx = draw_row(lit, size_after, pf_status);
        load_row(new_pos);
    };
    // the env wave draws the exploration decision of the NEXT step while the other waves run the minibatch forwards
    int nx_action = 0, nx_explored = 0;
    bool nx_valid = false;
    auto draw_action = [&](int step_no) {
        // ---- select_train_action, random branch (DDQN.py:97-101): u = random.random(); u < eps -> random action ----
        double u;
        if (tape) {
            if (step_no >= a.tapes.eps_uniform_stride) { status = -2; u = 1.0; }
            else u = a.tapes.eps_uniform[chain * a.tapes.eps_uniform_stride + step_no];
        } else u = u64_to_unit(rng_u64(key, STREAM_EPS, (uint64_t)step_no));
        nx_explored = 0; nx_action = 0;
        if (u < eps_g) {
            nx_explored = 1;
            if (tape) {
                if (n_act >= a.tapes.rand_action_stride) { status = -3; nx_action = 0; }
                else nx_action = a.tapes.rand_action[chain * a.tapes.rand_action_stride + n_act];
            } else nx_action = (int)u64_to_below(rng_u64(key, STREAM_ACTION, (uint64_t)n_act), (uint32_t)A);
            ++n_act;
        }
    };

    // Deterministic time-out (lenv_ddqn_cfg::step_budget, base_agent.py:30-47): elapsed = env steps taken so far
    const bool budgeted = cfg.step_budget > 0;
    int timed_out_at = -1;
    // ---- TEAM barrier (as in td3_wavechain.hip): one monotonically increasing counter per chain, zeroed by a kernel in front of the
    // launch; thread 0 releases, arrives, waits for the epoch's count, acquires.  Members on one XCD share its L2 (the vector L1
    // writes through): release = the stores have left the CU, acquire = this CU's L1 lines dropped; otherwise the agent-scope fences.
    // A member that waits longer than LENV_TEAM_GIVEUP_TICKS for the team to assemble (0.25 s: a foreign kernel holds CUs; 5 s once it has) gives up for good (status -10)
    // instead of hanging the device; the caller repeats the launch with one workgroup per chain.
    unsigned team_epoch = 0;
    bool team_dead = false, team_same_xcd = false;
    unsigned *team_bar = TEAM ? reinterpret_cast<unsigned *>(a.team_ws + chain * a.team_stride) : nullptr;
    auto team_barrier = [&]() {
        if constexpr (TEAM) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's own stores are acknowledged (__syncthreads() is s_barrier without a vmcnt wait)
            __syncthreads();
            ++team_epoch;
            if (tid == 0 && !team_dead) {
                if (team_same_xcd) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_fetch_add(team_bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target_ = team_epoch * (unsigned)G;
                const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();      // constant 100 MHz
                unsigned spins = 0;
                bool gave_up = false;
                while (__hip_atomic_load(team_bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target_) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 63u) == 0u && __builtin_amdgcn_s_memrealtime() - w0 > (team_epoch <= 1u ? LENV_TEAM_GIVEUP_TICKS : LENV_TEAM_GIVEUP_TICKS_RUN)) { gave_up = true; break; }
                }
                // (the invalidate has to be complete before the barrier below releases the other waves: vmcnt counts it)
                if (team_same_xcd) asm volatile("buffer_inv sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
                else { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                if (gave_up) team_bar[15] = 1u;
            }
            __syncthreads();
            if (!team_dead && __hip_atomic_load(team_bar + 15, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { team_dead = true; status = -10; }
        }
    };
    if constexpr (TEAM) {
        if (tid == 0) team_bar[1 + g] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u;       // HW_REG_XCC_ID[3:0]
        team_barrier();                                        // every member's XCD id is posted
        bool same = true;
        const unsigned x0 = __hip_atomic_load(team_bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int m = 1; m < G; ++m) same = same && __hip_atomic_load(team_bar + 1 + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == x0;
        team_same_xcd = same;
    }
    PT_DECL;
    for (int episode = 0; episode < cfg.train_episodes; ++episode) {
        if (budgeted && (int64_t)train_steps + test_steps > cfg.step_budget) { timed_out_at = episode; break; }   // uniform
        // DDQN.update_parameters_per_episode (DDQN.py:112-117)
        if (episode == 0) eps_g = cfg.eps_init;
        else { eps_g *= cfg.eps_decay; if (eps_g < cfg.eps_min) eps_g = cfg.eps_min; }
        const bool learning = episode >= cfg.init_episodes;

        // env.reset(): VirtualEnv.reset -> real-env reset state as fp32 (virtual_env.py:35-41)
        if (wave == ENV_WAVE) {
            double st0[4];
            if (tape) {
                if (episode >= a.tapes.train_reset_stride) { status = -5; for (int i = 0; i < 4; ++i) st0[i] = 0.0; }
                else for (int i = 0; i < 4; ++i) st0[i] = a.tapes.train_reset[(chain * a.tapes.train_reset_stride + episode) * 4 + i];
            } else {
                for (int i = 0; i < 4; ++i)
                    st0[i] = -reset_lim + (2 * reset_lim) * u64_to_unit(rng_u64(key, STREAM_TRAIN_RESET, (uint64_t)(episode * 4 + i)));
            }
            real_env_obs<ENV, S>(st0, state);
            if constexpr (RENV) {                          // RewardEnv.reset -> real_env.reset(): the env's own state; no phi(s) yet
#pragma unroll
                for (int i = 0; i < 4; ++i) st_d[i] = st0[i];
                env_steps = 0; have_phi = false;
            }
            if constexpr (!FIXED) tr_reward = 0.0f;
        }

        int ep_len = 0;
        spec_valid = false;                               // a reset state has no precomputed candidates
        for (int t = 0; t < MAX_STEPS; ++t) {
            const int size_after = train_steps + 1 < rb_cap ? train_steps + 1 : rb_cap;     // ReplayBuffer.size after this add
            const int new_pos = wr_pos;                                                     // ReplayBuffer.ptr before this add (== train_steps % rb_cap)
            // ================= phase A =================
            // In a learning episode there is NO barrier after phase A: the other waves go straight from the previous step's
            // Adam barrier into this step's minibatch forward (their rows were prefetched, the weights are final) while the env
            // wave acts, steps the SE and appends; it then publishes step_tag in ctrl[5].  Only a wave that sampled the very
            // row being appended, and the speculation wave (needs the new state), wait for that flag.
            const int step_tag = train_steps + 1;
            PT_MARK(9);
            // split layout: what the four third waves do up to the spilled items' output layers is the critical path of the forward
            // interval (the SIMDs' arbiters otherwise serve the older waves first and these rows arrive last)
            if (split_wave && learning) __builtin_amdgcn_s_setprio(3);
            if (wave == ENV_WAVE) {
                // ---- select_train_action (DDQN.py:97-104) ----
                if (!nx_valid) draw_action(train_steps);
                nx_valid = false;
                int action = nx_action;
                const int explored = nx_explored;
                if (!explored)
                    action = wave_q_argmax<S, A, PR, QACT>(q_onl, q_w2, HqP, state, wscr, Hq, cfg.q_prelu, lane, tanh_tab, tl);
                PT_MARK(0);   // (env wave) act
                // ---- EnvWrapper.step -> VirtualEnv.step (envs/env_wrapper.py:16-47, envs/virtual_env.py:43-54) ----
                float next_state[S], reward, done;
                if constexpr (RENV) {
                    // ---- EnvWrapper.step -> RewardEnv.step (reward_env.py:61-66): the real transition (TimeLimit: done at max_steps), reward =
                    // _calc_reward(state, next_state, reward) with the perturbed reward net (:68-133); oracle: rn_shape_one ----
                    double rew; int dn;
                    real_env_step<ENV>(st_d, action, rew, dn);
                    ++env_steps;
                    if (env_steps >= MAX_STEPS) dn = 1;
                    real_env_obs<ENV, S>(st_d, next_state);
                    const int rtype = cfg.reward_env_type;
                    const float r32 = (float)rew;
                    float shaped = r32;                                            // type 0: the real reward passes through
                    if (rtype != 0) {
                        if ((rtype == 1 || rtype == 2) && !have_phi) phi_s = rn_eval(se_hw, state);      // phi(s): carried over after the first step
                        const float phi_s2 = rn_eval(se_hw, next_state);
                        const float g32 = a.f_gamma;
                        switch (rtype) {
                        case 1: shaped = g32 * phi_s2 - phi_s; break;
                        case 2: shaped = (r32 + g32 * phi_s2) - phi_s; break;
                        case 5: shaped = phi_s2; break;
                        default: shaped = r32 + phi_s2; break;                    // 6
                        }
                        phi_s = phi_s2; have_phi = true;
                    }
                    reward = shaped; done = dn ? 1.0f : 0.0f;
                } else if (spec_valid) {
                    // evaluated for every action during the previous step's forward interval
#pragma unroll
                    for (int i = 0; i < S; ++i) next_state[i] = cand[action * 16 + i];
                    reward = cand[action * 16 + S]; done = cand[action * 16 + S + 1];
                } else {
                    const float acc = se_eval(se_hw, state, action);
#pragma unroll
                    for (int i = 0; i < S; ++i) next_state[i] = __shfl(acc, i);
                    reward = __shfl(acc, S); done = __shfl(acc, S + 1);
                }
                if (lane < S) {
                    float v = next_state[0];
#pragma unroll
                    for (int i = 1; i < S; ++i) v = (lane == i) ? next_state[i] : v;
                    cur_state[lane] = v;
                }
                // ---- ReplayBuffer.add (utils.py:24-32) ----
                {
                    float val = done;
#pragma unroll
                    for (int i = 0; i < S; ++i) {
                        if (lane == i) val = state[i];
                        if (lane == S + 1 + i) val = next_state[i];
                    }
                    if (lane == S) val = (float)action;
                    if (lane == 2 * S + 1) val = reward;
                    if (lane < 2 * S + 3) { rb[(int64_t)new_pos * RS + lane] = val; newrow[lane] = val; }
                }
                if (lane == 0) {
                    ctrl[t & 1] = done;
                    if (!FIXED && a.out.trace_action && train_steps < a.out.trace_cap) {
                        const int64_t k = chain * a.out.trace_cap + train_steps;
                        a.out.trace_action[k] = action | (explored << 16);
                        for (int i = 0; i < S; ++i) { a.out.trace_state[k * S + i] = state[i]; a.out.trace_next_state[k * S + i] = next_state[i]; }
                        a.out.trace_reward_done[k * 2] = reward; a.out.trace_reward_done[k * 2 + 1] = done;
                    }
                }
#pragma unroll
                for (int i = 0; i < S; ++i) state[i] = next_state[i];
                if constexpr (!FIXED) tr_reward = tr_reward + reward;             // base_agent.py:121 episode_reward += reward (fp32 tensors)
                if (learning && lane == 0) lds_flag_store(ctrl + 5, step_tag);
                PT_MARK(1);   // (env wave) SE step + append
            }
            if (learning && fwd_active) {                          // (the env wave has such lanes in the split layout only)
                if (!pf_valid) fetch_row(learn_it, size_after, new_pos);     // first learn step: nothing was prefetched
                pf_valid = false;
                if (pf_status) status = pf_status;
                if (wave != ENV_WAVE && __builtin_amdgcn_ballot_w64(my_idx == new_pos) != 0) lds_flag_wait(ctrl + 5, step_tag);   // needs newrow
            }
            ++ep_len; ++train_steps;
            wr_pos = wr_pos + 1 == rb_cap ? 0 : wr_pos + 1;    // ReplayBuffer.ptr = (ptr + 1) % max_size, without the division
            if (!learning) __syncthreads();                    // B1 (init episodes only: nothing else to overlap with)
            PT_MARK(2);

            if (learning) {
                // ================= learn: DDQN.learn (DDQN.py:60-94) =================
                if (fwd_active) {
                    float row[RS];
                    if (my_idx == new_pos) {
#pragma unroll
                        for (int v = 0; v < RS; ++v) row[v] = newrow[v];
                    } else {
#pragma unroll
                        for (int v = 0; v < RS / 4; ++v) { row[v * 4] = rowv[v].x; row[v * 4 + 1] = rowv[v].y; row[v * 4 + 2] = rowv[v].z; row[v * 4 + 3] = rowv[v].w; }
                    }
                    float x[S];
#pragma unroll
                    for (int i = 0; i < S; ++i) x[i] = fwd_pass == 0 ? row[i] : row[S + 1 + i];
                    if ((full_item || (wide && split_part == 0)) && fwd_pass == 0) {
                        // the sample's state / reward / done / action for the TD error and the backward: stored BEFORE the pair loop, so that
                        // the row's registers are dead inside it
#pragma unroll
                        for (int i = 0; i < S; ++i) sB[fwd_b * SP + i] = row[i];
                        rda[fwd_b * 4 + 0] = row[2 * S + 1]; rda[fwd_b * 4 + 1] = row[2 * S + 2]; rda[fwd_b * 4 + 2] = row[S];
                    }
                    const float *W = fwd_pass == 2 ? q_tgt : q_onl;
                    float q[A];
#pragma unroll
                    for (int aa = 0; aa < A; ++aa) q[aa] = 0.0f;
                    // Two hidden units per iteration (pair record, packed fp32 math); the output accumulators stay sequential in
                    // j (canonical order).  SOFTWARE PIPELINE, one pair of skew: while the tanh-table gathers of pair jp are in
                    // flight the thread already runs layer 1 of pair jp+1 and issues ITS gathers, then finishes pair jp
                    // (polynomial, output accumulation, h store).  With 3 waves per SIMD the LDS latency of a gather (~100+
                    // cycles) is otherwise exposed once per pair.
                    constexpr int N1 = OW2 / 4, N2 = (PR - OW2) / 4, PR4 = PR / 4;
                    const float4 *W4 = reinterpret_cast<const float4 *>(W);
                    const int Hqf = FIXED ? FIX_HQ : Hq;                            // literal only here: pair count and tail of this loop
                    const int npairs = (Hqf + 1) >> 1;
                    float *hrow = wide ? wide_row(fwd_pass, split_li) : (full_item ? hB + fwd_b * HP : lds + LV(o_hX) + split_li * HP);
                    float4 r1[N1], r2[N2];
                    auto load1 = [&](int jp) {
#pragma unroll
                        for (int v = 0; v < N1; ++v) r1[v] = W4[jp * PR4 + v];
                    };
                    auto load2 = [&](int jp) {
#pragma unroll
                        for (int v = 0; v < N2; ++v) r2[v] = W4[jp * PR4 + N1 + v];
                    };
                    auto layer1 = [&]() -> v2f {
                        float rec[OW2];
#pragma unroll
                        for (int v = 0; v < N1; ++v) { rec[4 * v] = r1[v].x; rec[4 * v + 1] = r1[v].y; rec[4 * v + 2] = r1[v].z; rec[4 * v + 3] = r1[v].w; }
                        v2f z = {0.0f, 0.0f};
#pragma unroll
                        for (int i = 0; i < S; ++i) z = fma2((v2f){x[i], x[i]}, (v2f){rec[2 * i], rec[2 * i + 1]}, z);
                        return z + (v2f){rec[2 * S], rec[2 * S + 1]};
                    };
                    // finish pair jp (its activations were issued one stage earlier) with the output part in r2; an h-only lane
                    // of the split layout stops at the activations and leaves them in its item's LDS row
                    auto finish = [&](const ActPipe2<QACT> &ap, int jp, bool two, auto honly_tag) {
                        if constexpr (decltype(honly_tag)::value) {
                            const v2f hh = ap.finish(cfg.q_prelu);
                            if (two) *reinterpret_cast<v2f *>(hrow + 2 * jp) = hh;
                            else hrow[2 * jp] = hh.x;
                            return;
                        }
                        float w2[PR - OW2];
#pragma unroll
                        for (int v = 0; v < N2; ++v) { w2[4 * v] = r2[v].x; w2[4 * v + 1] = r2[v].y; w2[4 * v + 2] = r2[v].z; w2[4 * v + 3] = r2[v].w; }
                        const v2f hh = ap.finish(cfg.q_prelu);
                        const float h0 = hh.x, h1 = hh.y;
                        // output layer: unit 2jp then unit 2jp+1, packed over actions
                        if constexpr (A == 2) {
                            v2f qq = {q[0], q[1]};
                            qq = fma2((v2f){h0, h0}, (v2f){w2[0], w2[1]}, qq);
                            if (two) qq = fma2((v2f){h1, h1}, (v2f){w2[2], w2[3]}, qq);
                            q[0] = qq.x; q[1] = qq.y;
                        } else {
#pragma unroll
                            for (int aa = 0; aa < A; ++aa) q[aa] = fma32(h0, w2[aa], q[aa]);
                            if (two) {
#pragma unroll
                                for (int aa = 0; aa < A; ++aa) q[aa] = fma32(h1, w2[A + aa], q[aa]);
                            }
                        }
                        if (fwd_pass == 0) {                       // HP is even: 8-byte aligned pair store
                            if (two) *reinterpret_cast<v2f *>(hrow + 2 * jp) = hh;
                            else hrow[2 * jp] = h0;
                        }
                    };
                    // one pipeline stage: start pair jp+1 into `nxt`, finish pair jp from `cur`.  FULL = steady state
                    // (pair jp+2 exists, pair jp has both units): no conditions, so the stage is straight-line code and the
                    // compiler can count outstanding LDS reads instead of draining them
                    auto stage = [&](ActPipe2<QACT> &cur, ActPipe2<QACT> &nxt, int jp, int pend, auto full_tag, auto honly_tag) {
                        constexpr bool FULL = decltype(full_tag)::value, HONLY = decltype(honly_tag)::value;
                        if (FULL || jp + 1 < pend) {
                            nxt.issue(tanh_tab, tl, layer1());      // r1 holds pair jp+1
                            if (FULL || jp + 2 < pend) load1(jp + 2);
                        }
                        finish(cur, jp, FULL || 2 * jp + 1 < Hqf, honly_tag);  // r2 holds pair jp
                        if (!HONLY && (FULL || jp + 1 < pend)) load2(jp + 1);
                    };
                    using T = std::true_type;
                    using F = std::false_type;
                    ActPipe2<QACT> pa, pb;
                    if (full_item) {
                        load1(0);
                        pa.issue(tanh_tab, tl, layer1());
                        if (npairs > 1) load1(1);
                        load2(0);
                        int jp = 0;
#pragma unroll 1
                        for (; jp + 4 <= npairs; jp += 2) {             // jp+3 < npairs: both stages are steady-state
                            stage(pa, pb, jp, npairs, T{}, F{});
                            stage(pb, pa, jp + 1, npairs, T{}, F{});
                        }
                        for (; jp + 2 <= npairs; jp += 2) {             // at most one more double stage, with the tail conditions
                            stage(pa, pb, jp, npairs, F{}, F{});
                            stage(pb, pa, jp + 1, npairs, F{}, F{});
                        }
                        if (jp < npairs) finish(pa, jp, 2 * jp + 1 < Hqf, F{});
#pragma unroll
                        for (int aa = 0; aa < A; ++aa) qres[(fwd_pass * MAX_B + fwd_b) * A + aa] = q[aa] + W[npairs * PR + aa];
                    } else if (wide) {
                        // WIDE h-only lane: `len` pairs from p0 on, the same trip count in every lane and no condition in the loop (a pair is
                        // always stored whole: for an odd width the second word of the last pair is the row's pad word; loads past the last
                        // record are clamped, the activation issued past the part's end is never finished)
                        const int len = (npairs + WIDE_D - 1) / WIDE_D, last = npairs - 1;
                        int jp = split_part * len < npairs - len ? split_part * len : npairs - len;
                        auto wstage = [&](ActPipe2<QACT> &cur, ActPipe2<QACT> &nxt) {
                            nxt.issue(tanh_tab, tl, layer1());                                  // r1 holds pair jp + 1 (clamped)
                            load1(jp + 2 < last ? jp + 2 : last);
                            *reinterpret_cast<v2f *>(hrow + 2 * jp) = cur.finish(cfg.q_prelu);
                            ++jp;
                        };
                        load1(jp);
                        pa.issue(tanh_tab, tl, layer1());
                        load1(jp + 1 < last ? jp + 1 : last);
#pragma unroll 1
                        for (int i = 0; i + 2 <= len; i += 2) { wstage(pa, pb); wstage(pb, pa); }
                        if (len & 1) *reinterpret_cast<v2f *>(hrow + 2 * jp) = pa.finish(cfg.q_prelu);
                    } else {
                        // split layout, h-only lane: the activations of pairs [p0, p1) of its item
                        const int sd = SPLIT_D > 0 ? SPLIT_D : 1, p0 = split_part * npairs / sd, p1 = (split_part + 1) * npairs / sd;
                        load1(p0);
                        pa.issue(tanh_tab, tl, layer1());
                        if (p0 + 1 < p1) load1(p0 + 1);
                        int jp = p0;
#pragma unroll 1
                        for (; jp + 2 <= p1; jp += 2) {
                            stage(pa, pb, jp, p1, F{}, T{});
                            stage(pb, pa, jp + 1, p1, F{}, T{});
                        }
                        if (jp < p1) finish(pa, jp, 2 * jp + 1 < Hqf, T{});
                    }
                    // The replay row of the NEXT learn step, requested as soon as this thread's forward work is over (its row registers are
                    // free: the sample's side data went to LDS before the pair loop): the waves that finish their items early wait for the
                    // interval's barrier anyway, and the draw (four quarter-rate multiplies) no longer opens every wave's gradient interval.
                    // next step: ReplayBuffer.size = min(train_steps + 1, cap), write slot = train_steps % cap
                    const int nsz = train_steps + 1 < rb_cap ? train_steps + 1 : rb_cap;
                    my_idx = draw_row(learn_it + 1, nsz, pf_status);
                    // (the row the env wave appended in THIS step may still be on its way to memory: that one is taken from its LDS copy
                    // behind the interval's barrier, below)
                    if (my_idx != new_pos) load_row(wr_pos);
                    pf_valid = true;
                }
                if (split_wave) {
                    // split layout: this wave's activation rows are written -- tell the two waves that run the spilled items' output layers
                    const int k = wave - SPLIT_W0;
                    if (lane == 0) lds_flag_store(ctrl + 3 + k + (k >> 1), step_tag);          // ctrl[3], [4], [6], [7]
                    if (wave >= first_spec) PT_OWN(1);
                }
                if (split_wave && wave < first_spec) {
                    lds_flag_wait(ctrl + 3, step_tag); lds_flag_wait(ctrl + 4, step_tag);
                    lds_flag_wait(ctrl + 6, step_tag); lds_flag_wait(ctrl + 7, step_tag);
                    PT_OWN(1);
                    if (chain_lane) {
                        // output layer of spilled item tid (pass 2, sample tid - 2 B): the canonical chain over all hidden units, h from the rows
                        constexpr int N1 = OW2 / 4, N2 = (PR - OW2) / 4, PR4 = PR / 4;
                        const float4 *W4 = reinterpret_cast<const float4 *>(q_tgt);
                        const int Hqf = FIXED ? FIX_HQ : Hq;
                        const int npairs = (Hqf + 1) >> 1;
                        const float *hx = lds + LV(o_hX) + split_u * HP;
                        float q[A];
#pragma unroll
                        for (int aa = 0; aa < A; ++aa) q[aa] = 0.0f;
                        // blocks of eight pairs: every LDS read of a block is in flight before its first use (one round trip per
                        // block instead of one per pair -- this chain runs alone on its wave)
                        constexpr int CB = 8;
#pragma unroll 1
                        for (int j0 = 0; j0 < npairs; j0 += CB) {
                            v2f hv[CB];
                            float4 wv[CB][N2];
#pragma unroll
                            for (int u = 0; u < CB; ++u) {
                                const int jp = j0 + u < npairs ? j0 + u : npairs - 1;            // (clamped reads, unused)
                                hv[u] = *reinterpret_cast<const v2f *>(hx + 2 * jp);             // (the odd tail's second word is never used)
#pragma unroll
                                for (int v = 0; v < N2; ++v) wv[u][v] = W4[jp * PR4 + N1 + v];
                            }
#pragma unroll
                            for (int u = 0; u < CB; ++u) {
                                const int jp = j0 + u;
                                if (jp < npairs) {
                                    const bool two = 2 * jp + 1 < Hqf;
                                    float w2[PR - OW2];
#pragma unroll
                                    for (int v = 0; v < N2; ++v) { w2[4 * v] = wv[u][v].x; w2[4 * v + 1] = wv[u][v].y; w2[4 * v + 2] = wv[u][v].z; w2[4 * v + 3] = wv[u][v].w; }
                                    if constexpr (A == 2) {
                                        v2f qq = {q[0], q[1]};
                                        qq = fma2((v2f){hv[u].x, hv[u].x}, (v2f){w2[0], w2[1]}, qq);
                                        if (two) qq = fma2((v2f){hv[u].y, hv[u].y}, (v2f){w2[2], w2[3]}, qq);
                                        q[0] = qq.x; q[1] = qq.y;
                                    } else {
#pragma unroll
                                        for (int aa = 0; aa < A; ++aa) q[aa] = fma32(hv[u].x, w2[aa], q[aa]);
                                        if (two) {
#pragma unroll
                                            for (int aa = 0; aa < A; ++aa) q[aa] = fma32(hv[u].y, w2[A + aa], q[aa]);
                                        }
                                    }
                                }
                            }
                        }
#pragma unroll
                        for (int aa = 0; aa < A; ++aa) qres[(2 * MAX_B + (mb0 + tid - 2 * Bm)) * A + aa] = q[aa] + q_tgt[npairs * PR + aa];
                    }
                }
                if (split_wave) __builtin_amdgcn_s_setprio(0);
                if (wide) {
                    // every wave with h-only lanes counts itself in (LDS atomic behind a release fence: its rows are written); the chain
                    // waves wait for this step's count.  No workgroup barrier: the env and speculation waves go their own way.
                    const int n_hw = (WIDE_L * WIDE_D + 63) >> 6;
                    if (wave < n_hw && lane == 0) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __hip_atomic_fetch_add((lds_int *)(ctrl + 3), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    if (wave < ((WIDE_L + 63) >> 6)) {
                        const int want_cnt = n_hw * (learn_it + 1);
                        while (*(volatile lds_int *)((lds_int *)(ctrl + 3)) < want_cnt) __builtin_amdgcn_s_sleep(1);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    }
                    if (tid < WIDE_L) {
                        // output layer of item tid = (pass, sample): the canonical chain over all hidden units, h from the item's row, in
                        // blocks of eight pairs with every LDS read of a block in flight before its first use
                        constexpr int N1 = OW2 / 4, N2 = (PR - OW2) / 4, PR4 = PR / 4;
                        const int cpass = tid / Bm, cbi = tid - cpass * Bm;
                        const float *Wc = cpass == 2 ? q_tgt : q_onl;
                        const float4 *W4 = reinterpret_cast<const float4 *>(Wc);
                        const int Hqf = FIXED ? FIX_HQ : Hq;
                        const int npairs = (Hqf + 1) >> 1;
                        const float *hx = wide_row(cpass, cbi);
                        float q[A];
#pragma unroll
                        for (int aa = 0; aa < A; ++aa) q[aa] = 0.0f;
                        constexpr int CB = LENV_DDQN_WIDE_CB;
#pragma unroll 1
                        for (int j0 = 0; j0 < npairs; j0 += CB) {
                            v2f hv[CB];
                            float4 wv[CB][N2];
#pragma unroll
                            for (int u = 0; u < CB; ++u) {
                                const int jp = j0 + u < npairs ? j0 + u : npairs - 1;            // (clamped reads, unused)
                                hv[u] = *reinterpret_cast<const v2f *>(hx + 2 * jp);             // (the odd tail's second word is never used)
#pragma unroll
                                for (int v = 0; v < N2; ++v) wv[u][v] = W4[jp * PR4 + N1 + v];
                            }
#pragma unroll
                            for (int u = 0; u < CB; ++u) {
                                const int jp = j0 + u;
                                if (jp < npairs) {
                                    const bool two = 2 * jp + 1 < Hqf;
                                    float w2[PR - OW2];
#pragma unroll
                                    for (int v = 0; v < N2; ++v) { w2[4 * v] = wv[u][v].x; w2[4 * v + 1] = wv[u][v].y; w2[4 * v + 2] = wv[u][v].z; w2[4 * v + 3] = wv[u][v].w; }
                                    if constexpr (A == 2) {
                                        v2f qq = {q[0], q[1]};
                                        qq = fma2((v2f){hv[u].x, hv[u].x}, (v2f){w2[0], w2[1]}, qq);
                                        if (two) qq = fma2((v2f){hv[u].y, hv[u].y}, (v2f){w2[2], w2[3]}, qq);
                                        q[0] = qq.x; q[1] = qq.y;
                                    } else {
#pragma unroll
                                        for (int aa = 0; aa < A; ++aa) q[aa] = fma32(hv[u].x, w2[aa], q[aa]);
                                        if (two) {
#pragma unroll
                                            for (int aa = 0; aa < A; ++aa) q[aa] = fma32(hv[u].y, w2[A + aa], q[aa]);
                                        }
                                    }
                                }
                            }
                        }
#pragma unroll
                        for (int aa = 0; aa < A; ++aa) qres[(cpass * MAX_B + mb0 + cbi) * A + aa] = q[aa] + Wc[npairs * PR + aa];
                    }
                }
                if (wave >= first_spec) {
                    if (wave != ENV_WAVE) lds_flag_wait(ctrl + 5, step_tag);       // cur_state / done of this step
                  if (ctrl[t & 1] <= 0.5f) {
                    // ---- speculative SE step of the NEXT env step for every action (this wave's share) ----
                    float st[S];
#pragma unroll
                    for (int i = 0; i < S; ++i) st[i] = cur_state[i];
                    if constexpr (!RENV) {
                    for (int act = wave - first_spec; act < A; act += n_spec) {
                        const float acc = se_eval(se_hw, st, act);
                        if (lane < S + 2) cand[act * 16 + lane] = acc;
                    }
                    }
                    if (wave == ENV_WAVE && t + 1 < MAX_STEPS) {
                        draw_action(train_steps);              // train_steps already counts this step: index of the next one
                        nx_valid = true;
                    }
                  }
                }
                PT_OWN(0);
                __syncthreads();                               // B2
                PT_MARK(3);
                if (fwd_active && my_idx == new_pos) {             // next step samples the row appended in this one: the LDS copy is still there
#pragma unroll
                    for (int v = 0; v < RS / 4; ++v) rowv[v] = *reinterpret_cast<const float4 *>(newrow + 4 * v);
                }
                // ---- batch gradient in micro-chunks: wave c reduces samples [c*chunk, (c+1)*chunk) ----
                // lane = hidden unit j.  Per sample (canonical order, oracle orc_ddqn_learn): da = dq*W2[a_b][j] (as a sum over
                // the action masks, exactly one term non-zero), dz = act'(h)*da, gW1[j][:] += dz*s, gb1[j] += dz,
                // gW2[a][j] += dqm[a]*h, gb2[a] += dqm[a] -- adding an exact zero leaves the other actions' sums untouched.
                if (wave < mc1 - mc0) {
                    const int cg = mc0 + wave;                         // (TEAM: this member's chunks; else mc0 = 0, mc1 = n_chunks)
                    const int b0 = cg * LV(chunk), b1 = (b0 + LV(chunk) < B) ? b0 + LV(chunk) : B;
                    float *pc = part + cg * P;
                    // TEAM: the chunk's partials also go to the chain's exchange buffer (copy of this learn step's parity) for the other members
                    // as 8-byte granules {value, learn step + 1}: a granule is stored whole, so a reader that sees this step's tag has this
                    // step's value -- the hand-off needs no fence and no separate flag
                    float2 *xg = TEAM ? reinterpret_cast<float2 *>(a.team_ws + chain * a.team_stride + 16) + ((int64_t)(learn_it & 1) * LV(n_chunks4) + cg) * P : nullptr;
                    const float xtag = __int_as_float(learn_it + 1);
                    // TD target and dLoss/dQ(s,a) of the chunk's own samples (DDQN.py:82-86; mse_loss backward = 2/B * diff), one
                    // lane per sample: nobody else reads these rows, so the chunk goes on behind a wave-level fence -- no
                    // workgroup barrier between the TD error and the gradient
                    for (int b = b0 + lane; b < b1; b += 64) {
                        const float g32 = a.f_gamma, norm = a.f_norm;
                        // every LDS word the sample needs is requested before the first is used: one round trip, not three
                        // (reward/done/action -> argmax row -> the two selected entries); the selections are register moves
                        float q0[A], q1[A], q2[A];
#pragma unroll
                        for (int aa = 0; aa < A; ++aa) { q0[aa] = qres[(0 * MAX_B + b) * A + aa]; q1[aa] = qres[(1 * MAX_B + b) * A + aa]; q2[aa] = qres[(2 * MAX_B + b) * A + aa]; }
                        const float r = rda[b * 4], d = rda[b * 4 + 1];
                        const int ab = (int)rda[b * 4 + 2];
                        float best = q1[0], qt = q2[0], qa = q0[0];                 // argmax: first maximum, as torch.max
#pragma unroll
                        for (int aa = 1; aa < A; ++aa) { if (q1[aa] > best) { best = q1[aa]; qt = q2[aa]; } if (ab == aa) qa = q0[aa]; }
                        const float t1 = g32 * qt;
                        const float t2 = 1.0f - d;
                        const float y = r + t1 * t2;
                        const float diff = qa - y;
                        const float dq = norm * diff;
                        // dL/dQ(s_b, .) as a row: only entry a_b is non-zero (DDQN.py:84 gather) -- the backward pass multiplies
                        // by these masks instead of branching on the action
                        float4 dm;
                        dm.x = ab == 0 ? dq : 0.0f; dm.y = ab == 1 ? dq : 0.0f; dm.z = ab == 2 ? dq : 0.0f; dm.w = 0.0f;
                        *reinterpret_cast<float4 *>(dqB + 4 * b) = dm;
                    }
                    wave_sync();
                    for (int j = lane; j < ((Hq + 63) & ~63); j += 64) {
                        const bool jv = j < Hq;
                        float gW1[S], gW2[A], gb2[A], w2j[A], gb1 = 0.0f;
#pragma unroll
                        for (int i = 0; i < S; ++i) gW1[i] = 0.0f;
#pragma unroll
                        for (int aa = 0; aa < A; ++aa) { gW2[aa] = 0.0f; gb2[aa] = 0.0f; w2j[aa] = jv ? q_onl[(j >> 1) * PR + OW2 + (j & 1) * A + aa] : 0.0f; }
                        // one sample: everything in canonical order (oracle orc_ddqn_learn)
                        auto accumulate = [&](float h, const float4 dm4, const float (&sv)[SP]) {
                            const float dm[3] = {dm4.x, dm4.y, dm4.z};
                            float da = dm[0] * w2j[0];
#pragma unroll
                            for (int aa = 1; aa < A; ++aa) da = fma32(dm[aa], w2j[aa], da);
                            const float dz = act_bwd_t<QACT>(cfg.q_prelu, h, da);
                            if constexpr ((S & 1) == 0) {
#pragma unroll
                                for (int i = 0; i < S; i += 2) {
                                    const v2f g = fma2((v2f){dz, dz}, (v2f){sv[i], sv[i + 1]}, (v2f){gW1[i], gW1[i + 1]});
                                    gW1[i] = g.x; gW1[i + 1] = g.y;
                                }
                            } else {
#pragma unroll
                                for (int i = 0; i < S; ++i) gW1[i] = fma32(dz, sv[i], gW1[i]);
                            }
                            gb1 = gb1 + dz;
                            if constexpr (A == 2) {
                                const v2f g2 = fma2((v2f){dm[0], dm[1]}, (v2f){h, h}, (v2f){gW2[0], gW2[1]});
                                gW2[0] = g2.x; gW2[1] = g2.y;
                                if constexpr (!GB2_LANE) {
                                    const v2f gb = (v2f){gb2[0], gb2[1]} + (v2f){dm[0], dm[1]};
                                    gb2[0] = gb.x; gb2[1] = gb.y;
                                }
                            } else {
#pragma unroll
                                for (int aa = 0; aa < A; ++aa) { gW2[aa] = fma32(dm[aa], h, gW2[aa]); gb2[aa] = gb2[aa] + dm[aa]; }
                            }
                        };
                        auto load_s = [&](const float *sp, float (&sv)[SP]) {
#pragma unroll
                            for (int v = 0; v < SP / 4; ++v) {
                                const float4 f = *reinterpret_cast<const float4 *>(sp + 4 * v);
                                sv[4 * v] = f.x; sv[4 * v + 1] = f.y; sv[4 * v + 2] = f.z; sv[4 * v + 3] = f.w;
                            }
                        };
                        // running pointers (one add per round each): h column of this lane, dq rows and state rows of the chunk
                        // (vzero: a zero the compiler cannot see through -- it keeps the two wave-uniform row pointers in VGPRs, so a
                        // group's reads differ in their immediate offsets only instead of each getting its own s_add + v_mov)
                        int vzero;
                        asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
                        const float *hp = hB + ((jv || (GB2_LANE && j == Hq)) ? j : 0) + b0 * HP, *dqp = dqB + 4 * b0 + vzero, *sp = sB + SP * b0 + vzero;
                        int bq = b0;
                        // whole groups of GRP samples: every LDS read of a group is issued before the first use.  (A TEAM member's gradient
                        // wave is alone on its SIMD and its 17-sample loop is a chain of five LDS round trips, 3.4 k cycles; groups of six /
                        // eight for it -- LENV_DDQN_TEAM_GRP -- were measured SLOWER, 25.9 -> 26.7 / 29.5 ms at 24 chains: the kernel sits at the
                        // 168-VGPR cap of a 12-wave workgroup and the extra live registers come back as spill reloads in the forward.)
                        constexpr int GRP = TWIDE ? LENV_DDQN_WIDE_GRP : (TEAM ? LENV_DDQN_TEAM_GRP : 4);
#pragma unroll 1
                        for (; bq + GRP <= b1; bq += GRP, hp += GRP * HP, dqp += 4 * GRP, sp += GRP * SP) {
                            float hv[GRP], sv[GRP][SP];
                            float4 dmv[GRP];
#pragma unroll
                            for (int u = 0; u < GRP; ++u) {
                                hv[u] = hp[u * HP];
                                dmv[u] = *reinterpret_cast<const float4 *>(dqp + 4 * u);
                                load_s(sp + SP * u, sv[u]);
                            }
#pragma unroll
                            for (int u = 0; u < GRP; ++u) accumulate(hv[u], dmv[u], sv[u]);
                        }
                        for (; bq < b1; ++bq, hp += HP, dqp += 4, sp += SP) {       // chunk tail
                            float sv[SP];
                            const float h = hp[0];
                            const float4 dm4 = *reinterpret_cast<const float4 *>(dqp);
                            load_s(sp, sv);
                            accumulate(h, dm4, sv);
                        }
                        if (jv) {
#pragma unroll
                            for (int i = 0; i < S; ++i) pc[j * S + i] = gW1[i];
                            pc[Hq * S + j] = gb1;
#pragma unroll
                            for (int aa = 0; aa < A; ++aa) pc[Hq * S + Hq + aa * Hq + j] = gW2[aa];
                            if constexpr (TEAM) {
#pragma unroll
                                for (int i = 0; i < S; ++i) xg[j * S + i] = make_float2(gW1[i], xtag);
                                xg[Hq * S + j] = make_float2(gb1, xtag);
#pragma unroll
                                for (int aa = 0; aa < A; ++aa) xg[Hq * S + Hq + aa * Hq + j] = make_float2(gW2[aa], xtag);
                            }
                        }
                        if (GB2_LANE ? j == Hq : j == 0) {
#pragma unroll
                            for (int aa = 0; aa < A; ++aa) {
                                const float gbv = GB2_LANE ? gW2[aa] : gb2[aa];
                                pc[Hq * S + Hq + A * Hq + aa] = gbv;
                                if constexpr (TEAM) xg[Hq * S + Hq + A * Hq + aa] = make_float2(gbv, xtag);
                            }
                        }
                    }
                }
                PT_OWN(2);
                if constexpr (TEAM) {
                    // The other members' chunks, into the LDS rows the Adam phase sums in chunk order: every thread polls ITS granules
                    // (loads that go to the XCD's L2, which the members share and their stores write through to) until they carry this step's
                    // tag -- one L2 round trip when the other members are done, and the only synchronisation of the learn step.  A buffer is
                    // written again two learn steps later; by then every member has consumed it (a member cannot finish step t+1 without
                    // the partials the others produce AFTER consuming step t's).  Members that do not share an XCD meet at the barrier first.
                    if (!team_same_xcd) team_barrier();
                    const unsigned long long *xs_ = reinterpret_cast<const unsigned long long *>(a.team_ws + chain * a.team_stride + 16)
                                                    + (int64_t)(learn_it & 1) * LV(n_chunks4) * P;
                    const unsigned want_tag = (unsigned)(learn_it + 1);
                    const int lo = mc0 * P, hi = mc1 * P, tot = LV(n_chunks) * P;
                    // Granules are fetched in PAIRS (one 16-byte load: n_chunks4 * P is even and the area is 64-byte aligned, so pair q = granules
                    // 2q, 2q+1 never straddles the two parity copies): half the dependent L2 round trips of the granule-by-granule loop of
                    // rounds 3-4, which cost 3.5 k of the 6.9 k cycles of a team member's gradient interval (tools/phase_timing.py, 24 chains).
                    // A pair may contain one of this member's own granules (written by its own gradient waves a moment ago): it is polled like
                    // any other and not copied.  (All of a parameter's foreign partials in flight at once -- twelve 8-byte loads per thread --
                    // was built twice, inline and out of line: the two dozen registers put spill reloads into the forward's pair loop, 26.6 ->
                    // 32.3 ms per generation at 24 chains.)
                    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
                    const int npairs_x = (tot + 1) >> 1;
                    if constexpr (!TWIDE) {
                    for (int q = tid; q < npairs_x; q += NT) {
                        const int e0 = 2 * q, e1 = e0 + 1;
                        const bool need0 = e0 < lo || e0 >= hi, need1 = e1 < tot && (e1 < lo || e1 >= hi);
                        if (!need0 && !need1) continue;
                        u32x4 v = {0u, 0u, 0u, 0u};
                        unsigned spins = 0;
                        unsigned long long w0 = 0;
                        const u32x4 *src = reinterpret_cast<const u32x4 *>(xs_ + e0);
                        while (!team_dead) {
                            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src) : "memory");
                            if ((!need0 || v.y == want_tag) && (!need1 || v.w == want_tag)) break;
                            __builtin_amdgcn_s_sleep(1);
                            // give up after LENV_TEAM_GIVEUP_TICKS_RUN (the team has assembled by now; or as soon as another member of the chain
                            // has), for good: a thread that gave up never polls again
                            if ((++spins & 63u) == 0u) {
                                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                                if (w0 == 0) w0 = now;
                                if (now - w0 > LENV_TEAM_GIVEUP_TICKS_RUN || __hip_atomic_load(team_bar + 15, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                                    team_bar[15] = 1u; status = -10; team_dead = true; break;
                                }
                            }
                        }
                        if (need0) part[e0] = __uint_as_float(v.x);
                        if (need1) part[e1] = __uint_as_float(v.z);
                    }
                                    } else {
                    // NPF pairs per thread in flight at once: one in the TEAM instantiation of teams of two (at the 168-VGPR cap more live
                    // registers come back as spill reloads in its forward), LENV_DDQN_WIDE_NPF in the TWIDE instantiation
                    constexpr int NPF = TWIDE ? LENV_DDQN_WIDE_NPF : 1;
                    for (int q0 = tid; q0 < npairs_x; q0 += NT * NPF) {
                        u32x4 v[NPF];
                        bool need0[NPF], need1[NPF];
                        const u32x4 *src[NPF];
#pragma unroll
                        for (int u = 0; u < NPF; ++u) {
                            const int q = q0 + u * NT, e0 = 2 * q, e1 = e0 + 1;
                            need0[u] = q < npairs_x && (e0 < lo || e0 >= hi);
                            need1[u] = q < npairs_x && e1 < tot && (e1 < lo || e1 >= hi);
                            src[u] = reinterpret_cast<const u32x4 *>(xs_ + (q < npairs_x ? e0 : 0));
                            v[u] = u32x4{0u, 0u, 0u, 0u};
                        }
                        unsigned spins = 0;
                        unsigned long long w0 = 0;
                        while (!team_dead) {
#pragma unroll
                            for (int u = 0; u < NPF; ++u)
                                if (need0[u] || need1[u]) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "+v"(v[u]) : "v"(src[u]) : "memory");
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            bool all_in = true;
#pragma unroll
                            for (int u = 0; u < NPF; ++u) {
                                asm volatile("" : "+v"(v[u]));         // (the loads' results are defined only behind the wait)
                                all_in = all_in && (!need0[u] || v[u].y == want_tag) && (!need1[u] || v[u].w == want_tag);
                            }
                            if (all_in) break;
                            __builtin_amdgcn_s_sleep(1);
                            // give up after LENV_TEAM_GIVEUP_TICKS_RUN (the team has assembled by now; or as soon as another member of the chain
                            // has), for good: a thread that gave up never polls again
                            if ((++spins & 63u) == 0u) {
                                const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                                if (w0 == 0) w0 = now;
                                if (now - w0 > LENV_TEAM_GIVEUP_TICKS_RUN || __hip_atomic_load(team_bar + 15, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                                    team_bar[15] = 1u; status = -10; team_dead = true; break;
                                }
                            }
                        }
#pragma unroll
                        for (int u = 0; u < NPF; ++u) {
                            const int e0 = 2 * (q0 + u * NT);
                            if (need0[u]) part[e0] = __uint_as_float(v[u].x);
                            if (need1[u]) part[e0 + 1] = __uint_as_float(v[u].z);
                        }
                    }
                                    }
                }
                __syncthreads();                               // B4
                PT_MARK(5);
                // ---- torch.optim.Adam single-tensor step + Polyak (DDQN.py:88-93), one thread per parameter ----
                {
                    // Adam bias corrections of this step (torch.optim.Adam: step_size = lr/(1-beta1^t), sqrt(1-beta2^t))
                    const float2 sched = a.adam_sched[learn_it];
                    const float neg_step = sched.x, bc2_sqrt = sched.y;
                    const float w1 = a.f_w1, w2 = a.f_w2, beta2 = a.f_beta2;
                    const float adam_eps = a.f_adam_eps, tau = a.f_tau, omt = a.f_omt;
#pragma unroll
                    for (int k = 0; k < PPT; ++k) {
                        const int p = tid + k * NT;
                        if (p < P) {
                            // chunk partials in order; the slots n_chunks .. n_chunks4-1 (n_chunks rounded up to 4) hold +0 for
                            // the whole run, and g + 0 == g, so whole groups of four are summed without per-slot conditions
                            float g;
                            {
                                float pv[4];
#pragma unroll
                                for (int c = 0; c < 4; ++c) pv[c] = part[c * P + p];
                                g = pv[0]; g = g + pv[1]; g = g + pv[2]; g = g + pv[3];
                            }
                            for (int c0 = 4; c0 < LV(n_chunks4); c0 += 4) {          // uniform trip count, four slots per round
                                float pv[4];
#pragma unroll
                                for (int c = 0; c < 4; ++c) pv[c] = part[(c0 + c) * P + p];
#pragma unroll
                                for (int c = 0; c < 4; ++c) g = g + pv[c];
                            }
                            p_m[k] = fma32(w1, g - p_m[k], p_m[k]);
                            p_v[k] = p_v[k] * beta2;
                            p_v[k] = fma32(w2 * g, g, p_v[k]);
                            const float denom = __builtin_sqrtf(p_v[k]) / bc2_sqrt + adam_eps;
                            p_onl[k] = p_onl[k] + (neg_step * p_m[k]) / denom;
                            p_tgt[k] = tau * p_onl[k] + omt * p_tgt[k];
                            q_onl[my_off[k]] = p_onl[k];
                            q_tgt[my_off[k]] = p_tgt[k];
                            if (my_w2[k] >= 0) q_w2[my_w2[k]] = p_onl[k];
                        }
                    }
                }
                ++learn_it;
                PT_OWN(3);
                __syncthreads();                               // B5
                PT_MARK(6);
            }
            spec_valid = !RENV && learning && n_spec > 0;
            if (ctrl[t & 1] > 0.5f) break;                     // base_agent.py:128 (slot written in this step's phase A, behind B1/B2)
        }
        ++episodes_run;
        if (tid == 0 && a.out.episode_len) a.out.episode_len[chain * cfg.train_episodes + episode] = ep_len;
        if constexpr (!FIXED) { if (no_test_env && wave == ENV_WAVE && lane == 0) ctrl[8] = tr_reward; }      // (ctrl[8..]: free outside the test phase)
        __syncthreads();

        // ---- per-episode test on the real env (base_agent.py:134-136); none without a test env (test_mode 1) ----
        PT_MARK(9);
        if (!no_test_env) test_phase();
        PT_MARK(7);
        int brk = 0;
        if (tid == 0) {
            double tm;
            if (no_test_env) tm = (double)ctrl[8];       // avg_meter_reward.update(episode_reward) (base_agent.py:138)
            else {
                double sm = 0.0;
                for (int i = 0; i < T_EP; ++i) sm += ret[i];
                tm = sm / (double)T_EP;
            }
            meter[episode] = tm;
            if (a.out.episode_test_mean) a.out.episode_test_mean[chain * cfg.train_episodes + episode] = tm;
            // early out (base_agent.py:49-62,141-148; AverageMeter._mean utils.py:103-105): on the real env the test env's rule; without a
            // test env break_env = the training env -- the virtual rule on a VirtualEnv, the real rule on a RewardEnv / the real env
            if (learning) brk = meter_env_solved_inl(meter, episode + 1, cfg.early_out_num, no_test_env && !RENV, cfg.solved_reward,
                                                     cfg.early_out_virtual_diff, episode, cfg.init_episodes);
            ctrl[2] = (float)brk;
        }
        __syncthreads();
        brk = ctrl[2] > 0.5f;
        __syncthreads();
        if (brk) break;
    }

    // ---- final test (GTN_worker.py:199) and score = statistics.mean(reward_list_test) ----
    const int64_t remaining = cfg.step_budget - ((int64_t)train_steps + test_steps);     // time_remaining - elapsed
    const int test_before = test_steps;
    test_phase();
    if (budgeted) {
        // BaseAgent.test under the time-out: episode e starts only while the earlier episodes of this test used <= remaining
        // steps; the rest of the list is padded with the minimum so far (-1e9 if empty).  Episodes are independent, so the
        // list rolled out above is cut here.
        if (tid == 0) {
            int64_t used = 0;
            int stop = T_EP;
            for (int te = 0; te < T_EP; ++te) {
                if (used > remaining) { stop = te; break; }
                used += tlen[te];
            }
            double mn = -1e9;
            if (stop > 0) { mn = ret[0]; for (int i = 1; i < stop; ++i) if (ret[i] < mn) mn = ret[i]; }
            for (int te = stop; te < T_EP; ++te) ret[te] = mn;
            ctrl[6] = __int_as_float((int)used);
        }
        __syncthreads();
        test_steps = test_before + __float_as_int(ctrl[6]);
    }
#ifdef LENV_PHASE_TIMING
    if (tid == 0) PT_FLUSH(0, 10);
    if (tid == ENV_WAVE * 64) { if (chain == 0) { g_phase_cycles[10] = pt_acc[0]; g_phase_cycles[11] = pt_acc[1]; } }
    if (chain == 0 && lane == 0) {       // own-work stamps (forward, TD, gradient, Adam) of every wave
        for (int pi = 0; pi < 4; ++pi) g_phase_cycles[12 + 4 * wave + pi] = pt_own[pi];
    }
#endif
    if (tid == 0) {
        double sm = 0.0;
        for (int i = 0; i < T_EP; ++i) sm += ret[i];
        a.out.score[chain] = sm / (double)T_EP;
        if (a.out.final_returns) for (int i = 0; i < T_EP; ++i) a.out.final_returns[chain * T_EP + i] = ret[i];
        if (a.out.stats) {
            a.out.stats[chain * 4 + 0] = episodes_run; a.out.stats[chain * 4 + 1] = train_steps;
            a.out.stats[chain * 4 + 2] = learn_it; a.out.stats[chain * 4 + 3] = test_steps;
        }
        // episodes that never ran: NaN / 0, or -- after a time-out -- time_is_up's padding (base_agent.py:33-44): rewards
        // with the minimum so far (-1e9 if none), lengths with the maximum so far (1e9 if none)
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
    if (a.out.final_online) {
#pragma unroll
        for (int k = 0; k < PPT; ++k)
            if (tid + k * NT < P) a.out.final_online[chain * P + tid + k * NT] = p_onl[k];
    }
    if (a.out.status) {
        // any thread that saw a tape underrun reports it
        if (status != 0) atomicMin(&a.out.status[chain], status);
    }
}
#undef LV

}  // namespace lenv

using namespace lenv;

static int64_t inner_rb_cap(const lenv_ddqn_cfg *cfg)
{
    int64_t cap = (int64_t)cfg->train_episodes * cfg->max_steps;
    if (cap > cfg->rb_size) cap = cfg->rb_size;
    return cap < 1 ? 1 : cap;
}

static int inner_row_stride(const lenv_ddqn_cfg *cfg) { return (2 * cfg->state_dim + 3 + 3) & ~3; }

// Adam bias-correction schedule (see InnerArgs::adam_sched): entry t = { -(lr / (1 - beta1^(t+1))), sqrt(1 - beta2^(t+1)) } with the
// powers as the running double products torch keeps (oracle: ddqn_learn).  The table lives in the CALLER's workspace and is
// filled on the caller's stream by this prologue kernel (include/lenv_hip.h: an entry never allocates, never synchronises and
// keeps no global state -- the launch is graph-capturable on its first call).  Two lanes carry the two serial product chains
// 256 entries at a time; the conversions (IEEE double divide / sqrt) run one entry per thread.
__global__ __launch_bounds__(256) void adam_schedule_kernel(double lr, double beta1, double beta2, int64_t n, float2 *sched, int32_t *status, int64_t chains)
{
    if (status) for (int64_t c = threadIdx.x; c < chains; c += 256) status[c] = 0;      // the chains' status words start at 0 (ok)
    __shared__ double pw[2][256];
    __shared__ double carry[2];
    const int tid = threadIdx.x;
    if (tid < 2) carry[tid] = 1.0;
    __syncthreads();
    for (int64_t t0 = 0; t0 < n; t0 += 256) {
        if (tid == 0 || tid == 64) {
            const int c = tid >> 6;
            const double beta = c ? beta2 : beta1;
            double p = carry[c];
            const int m = n - t0 < 256 ? (int)(n - t0) : 256;
            for (int i = 0; i < m; ++i) { p *= beta; pw[c][i] = p; }
            carry[c] = p;
        }
        __syncthreads();
        if (t0 + tid < n) {
            float2 e;
            e.x = (float)(-(lr / (1.0 - pw[0][tid])));
            e.y = (float)__builtin_sqrt(1.0 - pw[1][tid]);
            sched[t0 + tid] = e;
        }
        __syncthreads();
    }
}
// TEAM launches: the chains' barrier words (counter, XCD ids, give-up flag) and every granule's tag start at zero
__global__ __launch_bounds__(256) void ddqn_team_reset_kernel(float *team_ws, int64_t n_floats)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_floats; i += (int64_t)gridDim.x * 256) team_ws[i] = 0.0f;   // tags included
}
static int64_t adam_schedule_len(const lenv_ddqn_cfg *cfg) { return (int64_t)(cfg->train_episodes > 0 ? cfg->train_episodes : 0) * cfg->max_steps + 1; }

// LDS carve-up of one chain's workgroup; returns LENV_ERR_UNSUPPORTED when the shapes do not fit 160 KiB
static int inner_layout(const lenv_ddqn_cfg *cfg, InnerArgs &a)
{
    a.L = make_inner_layout(cfg->state_dim, cfg->num_actions, cfg->q_hidden, cfg->se_hidden, cfg->batch_size, cfg->test_episodes,
                            cfg->grad_chunk, NT, NW, MAX_PPT, MAX_B);
    return a.L.rc;
}

// diagnostic switch: kernel_variant GENERIC runs the published shape through the generic instantiation (A/B timing)
// (the shape-specialised builds are VirtualEnv + calc_score builds: a RewardEnv / real-env launch and test_mode 1 take the generic ones)
static bool cfg_disables_fixed_shape(const lenv_ddqn_cfg *cfg)
{
    return (cfg->kernel_variant & LENV_VARIANT_GENERIC) != 0 || cfg->synthetic_env_type != 0 || cfg->test_mode != 0;
}

// index into kShapes of the published shape this launch has exactly (0 = none: generic instantiation)
template <int I> static bool shape_matches(const lenv_ddqn_cfg *cfg, const InnerLayout &L)
{
    constexpr ShapeSpec sp = kShapes[I];
    constexpr InnerLayout LC = make_inner_layout(sp.S, sp.A, sp.Hq, sp.Hse, sp.B, sp.T, sp.chunk, NT, NW, MAX_PPT, MAX_B);
    return cfg->env_id == sp.env && cfg->q_act == sp.q_act && cfg->q_hidden == sp.Hq && cfg->se_hidden == sp.Hse && cfg->batch_size == sp.B &&
           cfg->test_episodes == sp.T && cfg->max_steps == sp.max_steps && cfg->se_act == sp.se_act && L.chunk == sp.chunk &&
           L.lds_floats == LC.lds_floats && L.tanh16 == LC.tanh16 && L.split_D == LC.split_D && L.split_L == LC.split_L && L.P_q <= NT;
}
static int published_shape(const lenv_ddqn_cfg *cfg, const InnerLayout &L)
{
    static_assert(kNumShapes == 4, "extend the dispatch below together with kShapes");
    if (shape_matches<1>(cfg, L)) return 1;
    if (shape_matches<2>(cfg, L)) return 2;
    if (shape_matches<3>(cfg, L)) return 3;
    return 0;
}

static int inner_check(const lenv_ddqn_cfg *cfg)
{
    const int S = cfg->state_dim, A = cfg->num_actions, Hq = cfg->q_hidden, Hse = cfg->se_hidden, B = cfg->batch_size;
    if (cfg->agent_kind != 0) return LENV_ERR_INVALID;                             // DuelingDDQN: lenv_dueling_se_inner_loop
    if (cfg->q_layers != 1 || cfg->se_layers != 1) return LENV_ERR_UNSUPPORTED;   // multi-layer Q-nets: GEMM-tiled kernel (plain-DQN mode)
    if (cfg->icm_enabled) return LENV_ERR_UNSUPPORTED;                             // ICM agents: lenv_dueling_se_inner_loop_icm
    if (cfg->synthetic_env_type == 1) {
        // RewardEnv over the real env / the real env itself (the RENV instantiations): an explicit micro-chunk (grad_chunk 0 = ONE sequential
        // batch gradient = the GEMM-tiled kernel's order, what a cfg built for that kernel carries), a reward net without LayerNorm, the
        // reward types the real CartPole / Acrobot step can serve (no info vector)
        const int t = cfg->reward_env_type;
        if (cfg->grad_chunk <= 0 || cfg->se_layer_norm || !(t == 0 || t == 1 || t == 2 || t == 5 || t == 6)) return LENV_ERR_UNSUPPORTED;
    } else if (cfg->synthetic_env_type != 0) return LENV_ERR_INVALID;
    if (cfg->same_action_num > 1) return LENV_ERR_UNSUPPORTED;                     // several env steps per action: GEMM-tiled kernel
    if (cfg->test_mode < 0 || cfg->test_mode > 1) return LENV_ERR_INVALID;         // 1 = BaseAgent.train without a test env (the evaluation harness)
    // the Q-net's shared nn.PReLU slope is a trained parameter in the reference (DDQN.py:36 Adam over model.parameters());
    // refusing beats silently training with a frozen 0.25
    if (cfg->q_act == LENV_ACT_PRELU) return LENV_ERR_UNSUPPORTED;
    if (B < 1 || B > MAX_B || Hq < 1 || Hse < 1 || cfg->test_episodes < 1 || cfg->train_episodes < 0 || cfg->max_steps < 1)
        return LENV_ERR_UNSUPPORTED;
    if (!((cfg->env_id == LENV_ENV_CARTPOLE && S == 4 && A == 2) || (cfg->env_id == LENV_ENV_ACROBOT && S == 6 && A == 3)))
        return LENV_ERR_UNSUPPORTED;
    return LENV_OK;
}

#ifdef LENV_PHASE_TIMING
extern "C" int lenv_debug_phase_cycles(unsigned long long *host_out)
{
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(lenv::g_phase_cycles), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -4;
}
#endif

extern "C" int64_t lenv_ddqn_se_lds_bytes(const lenv_ddqn_cfg *cfg)
{
    if (!cfg) return LENV_ERR_INVALID;
    int rc = inner_check(cfg);
    if (rc != LENV_OK) return rc;
    InnerArgs a;
    rc = inner_layout(cfg, a);
    if (rc != LENV_OK) return rc;
    return (int64_t)a.L.lds_floats * (int64_t)sizeof(float);
}

extern "C" int lenv_ddqn_se_forward_split(const lenv_ddqn_cfg *cfg, int32_t *items, int32_t *parts)
{
    if (!cfg || !items || !parts) return LENV_ERR_INVALID;
    int rc = inner_check(cfg);
    if (rc != LENV_OK) return rc;
    InnerArgs a;
    rc = inner_layout(cfg, a);
    if (rc != LENV_OK) return rc;
    *items = a.L.split_L; *parts = a.L.split_D;
    return LENV_OK;
}

// floats per chain of the team exchange area: 16 barrier words + two copies (learn-step parity) of the chunk partials as 8-byte
// {value, tag} granules (0 when the layout is refused)
static int64_t inner_team_stride(const lenv_ddqn_cfg *cfg)
{
    InnerArgs t;
    if (inner_check(cfg) != LENV_OK || inner_layout(cfg, t) != LENV_OK) return 0;
    return (16 + 4 * (int64_t)t.L.n_chunks4 * t.L.P_q + 63) & ~(int64_t)63;
}

// the TEAM instantiation that runs this cfg (a chain on a team of workgroups: the generic instantiation with the exchange code, or the
// published CartPole shape's)
typedef void (*InnerKern)(const InnerArgs);
static InnerKern ddqn_team_kernel(const lenv_ddqn_cfg *cfg, const InnerLayout &L)
{
    InnerKern kern = nullptr;
#define LENV_PICK2(ENVID, SS, AA, PP)                                                                              \
    switch (cfg->q_act) {                                                                                      \
    case LENV_ACT_RELU: kern = ddqn_se_inner_kernel<ENVID, SS, AA, LENV_ACT_RELU, PP, 0, true>; break;            \
    case LENV_ACT_LEAKYRELU: kern = ddqn_se_inner_kernel<ENVID, SS, AA, LENV_ACT_LEAKYRELU, PP, 0, true>; break;  \
    case LENV_ACT_TANH: kern = ddqn_se_inner_kernel<ENVID, SS, AA, LENV_ACT_TANH, PP, 0, true>; break;            \
    default: kern = ddqn_se_inner_kernel<ENVID, SS, AA, LENV_ACT_IDENTITY, PP, 0, true>; break;                   \
    }
    if (cfg->env_id == LENV_ENV_CARTPOLE) { if (L.P_q <= NT) { LENV_PICK2(LENV_ENV_CARTPOLE, 4, 2, 1) } else { LENV_PICK2(LENV_ENV_CARTPOLE, 4, 2, 2) } }
    else { if (L.P_q <= NT) { LENV_PICK2(LENV_ENV_ACROBOT, 6, 3, 1) } else { LENV_PICK2(LENV_ENV_ACROBOT, 6, 3, 2) } }
#undef LENV_PICK2
    if (!cfg_disables_fixed_shape(cfg) && published_shape(cfg, L) == 1) kern = ddqn_se_inner_kernel<LENV_ENV_CARTPOLE, 4, 2, LENV_ACT_TANH, 1, 1, true>;
    return kern;
}

// TWIDE launches (teams of four and six at B = 199): every member's items cut six or three ways must fit the waves next to the env wave,
// the nets must have at least eight hidden-unit pairs, and the rows of passes 1 / 2 must find room (h rows of samples the member does
// not own, or the split layout's rows for pass 2).  Bm = the largest share of a member.
static bool ddqn_team_wide_ok(const lenv_ddqn_cfg *cfg, const InnerLayout &L, int G)
{
    if (G < 3 || (cfg->kernel_variant & LENV_VARIANT_TEAM_NARROW)) return false;
    const int B = cfg->batch_size, per = (L.n_chunks + G - 1) / G;
    int Bm = per * L.chunk;
    if (Bm > B) Bm = B;
    if (((cfg->q_hidden + 1) >> 1) < 8) return false;
    if (6 * (3 * Bm) > (NW - 1) * 64 && 3 * (3 * Bm) > 8 * 64) return false;       // six parts on eleven waves or three parts on eight (see the kernel's WIDE_D)         // at least five parts per item (measured: four parts at G = 4 lose to whole items, 27.2 vs 26.2 ms)
    if (!(3 * Bm <= B || Bm <= L.split_L)) return false;
    return 2 * Bm <= B - Bm;
}
static InnerKern ddqn_team_wide_kernel(const lenv_ddqn_cfg *cfg, const InnerLayout &L)
{
    InnerKern kern = nullptr;
#define LENV_PICK2(ENVID, SS, AA, PP)                                                                                      \
    switch (cfg->q_act) {                                                                                                  \
    case LENV_ACT_RELU: kern = ddqn_se_inner_kernel<ENVID, SS, AA, LENV_ACT_RELU, PP, 0, true, true>; break;            \
    case LENV_ACT_LEAKYRELU: kern = ddqn_se_inner_kernel<ENVID, SS, AA, LENV_ACT_LEAKYRELU, PP, 0, true, true>; break;  \
    case LENV_ACT_TANH: kern = ddqn_se_inner_kernel<ENVID, SS, AA, LENV_ACT_TANH, PP, 0, true, true>; break;            \
    default: kern = ddqn_se_inner_kernel<ENVID, SS, AA, LENV_ACT_IDENTITY, PP, 0, true, true>; break;                   \
    }
    if (cfg->env_id == LENV_ENV_CARTPOLE) { if (L.P_q <= NT) { LENV_PICK2(LENV_ENV_CARTPOLE, 4, 2, 1) } else { LENV_PICK2(LENV_ENV_CARTPOLE, 4, 2, 2) } }
    else { if (L.P_q <= NT) { LENV_PICK2(LENV_ENV_ACROBOT, 6, 3, 1) } else { LENV_PICK2(LENV_ENV_ACROBOT, 6, 3, 2) } }
#undef LENV_PICK2
    if (!cfg_disables_fixed_shape(cfg) && published_shape(cfg, L) == 1) kern = ddqn_se_inner_kernel<LENV_ENV_CARTPOLE, 4, 2, LENV_ACT_TANH, 1, 1, true, true>;
    return kern;
}

// Workgroups per chain of a launch with `chains` chains: G > 1 when the chains leave enough of the GPU idle for every member of every
// chain to be resident at once (occupancy API x CU count for the TEAM instantiation at this cfg's LDS footprint; blocks are dealt to the
// XCDs round-robin, so a team is 8 blocks apart) and the minibatch has at least G micro-chunks to deal.  cfg->team_size: 0 = automatic
// (launches of fewer than 16 chains keep one workgroup per chain), 1 = never, G = at most G (also below 16 chains); tape-mode launches
// and launches with a step trace are never teamed.
// Launches of more chains than this never run on teams (one workgroup per chain already fills half the device): ONE predicate for the team
// picker and for the size of the exchange area in the workspace (the two had drifted apart)
constexpr int64_t DDQN_TEAM_MAX_CHAINS = 128;

static int ddqn_pick_team(const lenv_ddqn_cfg *cfg, int64_t chains, bool production)
{
    const int want = cfg->team_size > 0 ? cfg->team_size : 0;      // 0 = automatic
    if (!production || want == 1 || chains < 1) return 1;
    if (want == 0 && chains < 16) return 1;
    if (chains > DDQN_TEAM_MAX_CHAINS) return 1;                  // (no exchange area is carved for such launches: lenv_ddqn_se_workspace_bytes)
    if (cfg->synthetic_env_type != 0) return 1;                   // RewardEnv / real-env training: one workgroup per chain (no TEAM instantiation)
    InnerArgs t;
    if (inner_check(cfg) != LENV_OK || inner_layout(cfg, t) != LENV_OK) return 1;
    const size_t lds_bytes = (size_t)t.L.lds_floats * sizeof(float);
    const int64_t slots = 8 * ((chains + 7) / 8);
    int best = 1;
    for (int G : { 2, 3, 4, 6 }) {
        const void *kern = reinterpret_cast<const void *>(ddqn_team_wide_ok(cfg, t.L, G) ? ddqn_team_wide_kernel(cfg, t.L) : ddqn_team_kernel(cfg, t.L));
        if (G <= t.L.n_chunks && (want == 0 || G <= want) && lenv_team_grid_resident(kern, NT, lds_bytes, slots * G)) best = G;
    }
    return best;
}

extern "C" int lenv_ddqn_se_team_size(const lenv_ddqn_cfg *cfg, int64_t chains)
{
    if (!cfg) return 1;
    return ddqn_pick_team(cfg, chains, cfg->rng_mode == LENV_RNG_COUNTER);
}

extern "C" size_t lenv_ddqn_se_workspace_bytes(const lenv_ddqn_cfg *cfg, int64_t chains)
{
    if (!cfg || chains < 0) return 0;
    size_t replay = (size_t)chains * inner_rb_cap(cfg) * inner_row_stride(cfg) * sizeof(float);
    size_t meter = (size_t)chains * (cfg->train_episodes > 0 ? cfg->train_episodes : 1) * sizeof(double);
    size_t sched = (((size_t)adam_schedule_len(cfg) * sizeof(float2)) + 255) & ~(size_t)255;
    // the team exchange area: present whenever a launch of this many chains COULD be teamed (counter mode, at most half the device's
    // CUs' worth of chains), whatever cfg->team_size says today -- an inner loop sizes its workspace once and team_size may be changed
    // on it later (A/B tooling, the -10 fall-back and back) -- and without asking the occupancy API on every query
    size_t team = (cfg->rng_mode == LENV_RNG_COUNTER && chains >= 1 && chains <= DDQN_TEAM_MAX_CHAINS) ? (size_t)chains * (size_t)inner_team_stride(cfg) * sizeof(float) : 0;
    return ((replay + 255) & ~(size_t)255) + ((meter + 255) & ~(size_t)255) + sched + team + 256;
}

extern "C" int lenv_ddqn_se_inner_loop(const lenv_ddqn_cfg *cfg, const float *theta, const float *eps,
                                       const int32_t *worker, const float *sign, const float *agent_init,
                                       const uint64_t *rng_keys, const lenv_tapes *tapes, int64_t chains,
                                       void *workspace, size_t workspace_bytes, const lenv_inner_out *out, void *stream)
{
    if (!cfg || !theta || !agent_init || !out || !out->score || !workspace || chains < 0) return LENV_ERR_INVALID;
    if (eps && (!worker || !sign)) return LENV_ERR_INVALID;
    if (cfg->rng_mode == LENV_RNG_TAPE && !tapes) return LENV_ERR_INVALID;
    if (cfg->rng_mode == LENV_RNG_COUNTER && !rng_keys) return LENV_ERR_INVALID;
    if (chains == 0) return LENV_OK;
    const int crc = inner_check(cfg);
    if (crc != LENV_OK) return crc;
    if (workspace_bytes < lenv_ddqn_se_workspace_bytes(cfg, chains)) return LENV_ERR_WORKSPACE;

    InnerArgs a;
    a.cfg = *cfg;
    a.theta = theta; a.eps = eps; a.worker = worker; a.sign = sign; a.agent_init = agent_init; a.rng_keys = rng_keys;
    a.has_tapes = tapes != nullptr;
    if (tapes) a.tapes = *tapes; else a.tapes = lenv_tapes{};
    a.rb_cap = inner_rb_cap(cfg);
    a.row_stride = inner_row_stride(cfg);
    a.replay = static_cast<float *>(workspace);
    size_t replay_bytes = ((size_t)chains * a.rb_cap * a.row_stride * sizeof(float) + 255) & ~(size_t)255;
    a.meter = reinterpret_cast<double *>(static_cast<char *>(workspace) + replay_bytes);
    a.out = *out;
    const int lrc = inner_layout(cfg, a);
    if (lrc != LENV_OK) return lrc;
    const size_t lds_bytes = (size_t)a.L.lds_floats * sizeof(float);
    a.f_gamma = (float)cfg->gamma; a.f_norm = (float)(2.0 / (double)cfg->batch_size);
    a.f_w1 = (float)(1.0 - cfg->adam_beta1); a.f_w2 = (float)(1.0 - cfg->adam_beta2); a.f_beta2 = (float)cfg->adam_beta2;
    a.f_adam_eps = (float)cfg->adam_eps; a.f_tau = (float)cfg->tau; a.f_omt = (float)(1.0 - cfg->tau);
    const size_t meter_bytes = ((size_t)chains * (cfg->train_episodes > 0 ? cfg->train_episodes : 1) * sizeof(double) + 255) & ~(size_t)255;
    float2 *sched = reinterpret_cast<float2 *>(static_cast<char *>(workspace) + replay_bytes + meter_bytes);
    a.adam_sched = sched;
    const size_t sched_bytes = (((size_t)adam_schedule_len(cfg) * sizeof(float2)) + 255) & ~(size_t)255;
    a.team_ws = reinterpret_cast<float *>(static_cast<char *>(workspace) + replay_bytes + meter_bytes + sched_bytes);
    a.team_stride = inner_team_stride(cfg);
    a.chains = chains;
    a.team_G = ddqn_pick_team(cfg, chains, cfg->rng_mode == LENV_RNG_COUNTER && !out->trace_action);
    if (a.team_G > 1 && workspace_bytes < replay_bytes + meter_bytes + sched_bytes + (size_t)chains * (size_t)a.team_stride * sizeof(float)) return LENV_ERR_WORKSPACE;

    void (*kern)(const InnerArgs) = nullptr;
#define LENV_PICK2(ENVID, SS, AA, PP)                                                                              \
    switch (cfg->q_act) {                                                                                          \
    case LENV_ACT_RELU: kern = ddqn_se_inner_kernel<ENVID, SS, AA, LENV_ACT_RELU, PP>; break;                      \
    case LENV_ACT_LEAKYRELU: kern = ddqn_se_inner_kernel<ENVID, SS, AA, LENV_ACT_LEAKYRELU, PP>; break;            \
    case LENV_ACT_TANH: kern = ddqn_se_inner_kernel<ENVID, SS, AA, LENV_ACT_TANH, PP>; break;                      \
    case LENV_ACT_PRELU: kern = ddqn_se_inner_kernel<ENVID, SS, AA, LENV_ACT_PRELU, PP>; break;                    \
    default: kern = ddqn_se_inner_kernel<ENVID, SS, AA, LENV_ACT_IDENTITY, PP>; break;                             \
    }
#define LENV_PICK(ENVID, SS, AA) if (a.L.P_q <= NT) { LENV_PICK2(ENVID, SS, AA, 1) } else { LENV_PICK2(ENVID, SS, AA, 2) }
    if (cfg->env_id == LENV_ENV_CARTPOLE) { LENV_PICK(LENV_ENV_CARTPOLE, 4, 2) }
    else { LENV_PICK(LENV_ENV_ACROBOT, 6, 3) }
#undef LENV_PICK2
    if (cfg->synthetic_env_type == 1) {
        // the RENV instantiations (RewardEnv over the real env / the real env itself as the training env)
#define LENV_PICK2(ENVID, SS, AA, PP)                                                                                                   \
        switch (cfg->q_act) {                                                                                                           \
        case LENV_ACT_RELU: kern = ddqn_se_inner_kernel<ENVID, SS, AA, LENV_ACT_RELU, PP, 0, false, false, true>; break;                \
        case LENV_ACT_LEAKYRELU: kern = ddqn_se_inner_kernel<ENVID, SS, AA, LENV_ACT_LEAKYRELU, PP, 0, false, false, true>; break;      \
        case LENV_ACT_TANH: kern = ddqn_se_inner_kernel<ENVID, SS, AA, LENV_ACT_TANH, PP, 0, false, false, true>; break;                \
        default: kern = ddqn_se_inner_kernel<ENVID, SS, AA, LENV_ACT_IDENTITY, PP, 0, false, false, true>; break;                       \
        }
        if (cfg->env_id == LENV_ENV_CARTPOLE) { LENV_PICK(LENV_ENV_CARTPOLE, 4, 2) }
        else { LENV_PICK(LENV_ENV_ACROBOT, 6, 3) }
#undef LENV_PICK2
    }
#undef LENV_PICK
    if (cfg->rng_mode == LENV_RNG_COUNTER && !out->trace_action && !cfg_disables_fixed_shape(cfg)) {
        switch (published_shape(cfg, a.L)) {
        case 1: kern = ddqn_se_inner_kernel<LENV_ENV_CARTPOLE, 4, 2, LENV_ACT_TANH, 1, 1>; break;
        case 2: kern = ddqn_se_inner_kernel<LENV_ENV_CARTPOLE, 4, 2, LENV_ACT_RELU, 1, 2>; break;
        case 3: kern = ddqn_se_inner_kernel<LENV_ENV_ACROBOT, 6, 3, LENV_ACT_LEAKYRELU, 2, 3>; break;
        default: break;
        }
    }
    unsigned grid = (unsigned)chains;
    if (a.team_G > 1) {
        kern = ddqn_team_wide_ok(cfg, a.L, a.team_G) ? ddqn_team_wide_kernel(cfg, a.L) : ddqn_team_kernel(cfg, a.L);
        grid = (unsigned)(8 * ((chains + 7) / 8) * a.team_G);
        hipLaunchKernelGGL(ddqn_team_reset_kernel, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream), a.team_ws, chains * a.team_stride);
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return LENV_ERR_LAUNCH;
    hipLaunchKernelGGL(adam_schedule_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), cfg->lr, cfg->adam_beta1, cfg->adam_beta2,
                       adam_schedule_len(cfg), sched, out->status, chains);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds_bytes, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}
