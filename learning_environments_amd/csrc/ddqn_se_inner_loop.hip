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
// Design (DESIGN.md "Kernel K-inner"): the whole chain runs inside one launch.  SE weights
// (theta + sign*eps), the Q-net, its target, and all minibatch activations live in LDS; Adam state and
// master copies of the Q parameters live in the owning thread's registers; only the replay buffer is in
// HBM/L2.  Per training step: wave 12 acts + steps the SE + appends the transition while waves 0-11
// prefetch their minibatch rows; then 12 waves run the three Q forwards (thread = sample x pass), one thread
// per sample forms the TD error, up to 16 waves back-propagate + reduce the batch gradient in micro-chunks, 1 thread/parameter applies Adam.
// Arithmetic order is the oracle's canonical order (oracle/lenv_oracle.h) => results are bit-identical.
#include "lenv_device.cuh"

namespace lenv {

constexpr int NT = 1024;          // threads per chain
constexpr int NW = NT / 64;       // waves per chain
constexpr int ENV_WAVE = 12;      // wave that plays the environment/actor role during a training step
constexpr int MAX_B = 256;        // minibatch samples (one per thread of a 256-thread pass group)
constexpr int PPT = 2;            // Q-net parameters owned per thread (P_agent <= PPT*NT)

struct InnerArgs {
    lenv_ddqn_cfg cfg;
    const float *theta, *eps; const int32_t *worker; const float *sign;
    const float *agent_init; const uint64_t *rng_keys;
    lenv_tapes tapes; int has_tapes;
    float *replay; double *meter; int64_t rb_cap; int row_stride;
    lenv_inner_out out;
    int P_q, P_se, se_net_size[3];
    int RP, HP, chunk, n_chunks;
    // LDS offsets (floats)
    int o_se_w0T, o_se_b0, o_se_wout, o_se_bout, o_se_h, o_q_onl, o_q_tgt, o_wscr, o_hB, o_sB, o_rda,
        o_dqB, o_qres, o_part, o_newrow, o_ctrl, o_ret, lds_floats;
};

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int S, int A>
__device__ __forceinline__ int packed_off(int p, int Hq, int RP)
{
    const int nW1 = Hq * S;
    if (p < nW1) { int j = p / S; return j * RP + (p - j * S); }
    p -= nW1;
    if (p < Hq) return p * RP + S;
    p -= Hq;
    if (p < A * Hq) { int aa = p / Hq; int j = p - aa * Hq; return j * RP + S + 1 + aa; }
    p -= A * Hq;
    return Hq * RP + p;
}

// Batch-1 greedy action of a Critic_DQN held as packed records in LDS: one wave, lane = hidden unit.
// Returns argmax_a Q(obs) (first maximum), identical in every lane.
template <int S, int A>
__device__ __forceinline__ int wave_q_argmax(const float *W, const float (&obs)[S], float *scratch, int Hq, int RP,
                                             int act, float prelu, int lane)
{
    for (int j = lane; j < Hq; j += 64) {
        const float *rec = W + j * RP;
        float z = 0.0f;
#pragma unroll
        for (int i = 0; i < S; ++i) z = fma32(obs[i], rec[i], z);
        z = z + rec[S];
        scratch[j] = act_fwd(act, prelu, z);
    }
    wave_sync();
    float q = 0.0f;
    if (lane < A) {
        for (int j = 0; j < Hq; ++j) q = fma32(scratch[j], W[j * RP + S + 1 + lane], q);
        q = q + W[Hq * RP + lane];
    }
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

template <int ENV, int S, int A>
__global__ __launch_bounds__(NT) void ddqn_se_inner_kernel(const InnerArgs a)
{
    extern __shared__ __align__(16) float lds[];
    const lenv_ddqn_cfg &cfg = a.cfg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t chain = blockIdx.x;
    constexpr int K = S + A;
    const int Hq = cfg.q_hidden, Hse = cfg.se_hidden, B = cfg.batch_size, RP = a.RP, HP = a.HP, P = a.P_q;
    const int RS = a.row_stride;

    float *se_w0T = lds + a.o_se_w0T, *se_b0 = lds + a.o_se_b0, *se_wout = lds + a.o_se_wout, *se_bout = lds + a.o_se_bout;
    float *se_h = lds + a.o_se_h, *q_onl = lds + a.o_q_onl, *q_tgt = lds + a.o_q_tgt, *wscr = lds + a.o_wscr + wave * ((Hq + 63) & ~63);
    float *hB = lds + a.o_hB, *sB = lds + a.o_sB, *rda = lds + a.o_rda, *dqB = lds + a.o_dqB;
    float *qres = lds + a.o_qres, *part = lds + a.o_part, *newrow = lds + a.o_newrow;
    volatile float *ctrl = lds + a.o_ctrl;             // [0..1] done (double buffered by step parity), [2] break flag, [4..] wave step counts
    double *ret = reinterpret_cast<double *>(lds + a.o_ret);   // [test_episodes] returns

    // ---------------- stage the perturbed SE: W = theta + sign*eps[worker]  (GTN_worker.py:165-175) ----------------
    {
        const float sg = a.eps ? a.sign[chain] : 0.0f;
        const float *e = a.eps ? a.eps + (int64_t)a.worker[chain] * a.P_se : nullptr;
        for (int i = tid; i < a.P_se; i += NT) {
            float w = e ? fma32(sg, e[i], a.theta[i]) : a.theta[i];
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
    // ---------------- fresh DDQN agent: online = target = agent_init (DDQN.py:33-35), Adam state 0 ----------------
    // thread tid owns parameters tid and tid+NT (master copy, target, Adam m/v in registers)
    float p_onl[PPT], p_tgt[PPT], p_m[PPT], p_v[PPT];
    int my_off[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int p = tid + k * NT;
        p_onl[k] = p_tgt[k] = p_m[k] = p_v[k] = 0.0f;
        my_off[k] = 0;
        if (p < P) {
            p_onl[k] = a.agent_init[chain * P + p];
            p_tgt[k] = p_onl[k];
            my_off[k] = packed_off<S, A>(p, Hq, RP);
            q_onl[my_off[k]] = p_onl[k];
            q_tgt[my_off[k]] = p_tgt[k];
        }
    }
    if (tid < 8) ctrl[tid] = 0.0f;
    __syncthreads();

    const uint64_t key = a.rng_keys ? a.rng_keys[chain] : 0;
    const bool tape = cfg.rng_mode == LENV_RNG_TAPE;
    int status = 0;
    int64_t n_eps = 0, n_act = 0, n_test_ep = 0, learn_it = 0, rb_ptr = 0, rb_size = 0, n_trace = 0;
    int64_t train_steps = 0, test_steps = 0;
    double b1pow = 1.0, b2pow = 1.0, eps_g = cfg.eps_init;
    float *rb = a.replay + chain * a.rb_cap * RS;
    double *meter = a.meter + chain * cfg.train_episodes;
    const double reset_lim = ENV == LENV_ENV_CARTPOLE ? 0.05 : 0.1;
    const float g32 = (float)cfg.gamma, norm = (float)(2.0 / (double)B);
    const float w1 = (float)(1.0 - cfg.adam_beta1), w2 = (float)(1.0 - cfg.adam_beta2), beta2 = (float)cfg.adam_beta2;
    const float adam_eps = (float)cfg.adam_eps, tau = (float)cfg.tau, omt = (float)(1.0 - cfg.tau);
    int episodes_run = 0, n_meter = 0;
    float state[S];
#pragma unroll
    for (int i = 0; i < S; ++i) state[i] = 0.0f;

    // one real-env test phase (BaseAgent.test + DDQN.select_test_action): wave w plays test episode w (+16 per round)
    auto test_phase = [&]() {
        int my_steps = 0;
        for (int te = wave; te < cfg.test_episodes; te += NW) {
            double st[4];
            const int64_t row = n_test_ep + te;
            if (tape) {
                if (row >= a.tapes.test_reset_stride) { status = -5; for (int i = 0; i < 4; ++i) st[i] = 0.0; }
                else for (int i = 0; i < 4; ++i) st[i] = a.tapes.test_reset[(chain * a.tapes.test_reset_stride + row) * 4 + i];
            } else {
                for (int i = 0; i < 4; ++i)
                    st[i] = -reset_lim + (2 * reset_lim) * u64_to_unit(rng_u64(key, STREAM_TEST_RESET, (uint64_t)(row * 4 + i)));
            }
            float ep_reward = 0.0f;
            for (int t = 0; t < cfg.max_steps; ++t) {
                float obs[S];
                real_env_obs<ENV, S>(st, obs);
                const int act = wave_q_argmax<S, A>(q_onl, obs, wscr, Hq, RP, cfg.q_act, cfg.q_prelu, lane);
                double rew; int done;
                real_env_step<ENV>(st, act, rew, done);
                ep_reward = ep_reward + (float)rew;
                ++my_steps;
                if (done) break;
            }
            if (lane == 0) ret[te] = (double)ep_reward;
        }
        if (lane == 0) ctrl[4 + wave] = __int_as_float(my_steps);
        n_test_ep += cfg.test_episodes;
        __syncthreads();
        for (int w = 0; w < NW; ++w) test_steps += __float_as_int(ctrl[4 + w]);
    };

    for (int episode = 0; episode < cfg.train_episodes; ++episode) {
        // DDQN.update_parameters_per_episode (DDQN.py:112-117)
        if (episode == 0) eps_g = cfg.eps_init;
        else { eps_g *= cfg.eps_decay; if (eps_g < cfg.eps_min) eps_g = cfg.eps_min; }
        const bool learning = episode >= cfg.init_episodes;

        // env.reset(): VirtualEnv.reset -> real-env reset state as fp32 (virtual_env.py:35-41)
        {
            double st0[4];
            if (tape) {
                if (episode >= a.tapes.train_reset_stride) { status = -5; for (int i = 0; i < 4; ++i) st0[i] = 0.0; }
                else for (int i = 0; i < 4; ++i) st0[i] = a.tapes.train_reset[(chain * a.tapes.train_reset_stride + episode) * 4 + i];
            } else {
                for (int i = 0; i < 4; ++i)
                    st0[i] = -reset_lim + (2 * reset_lim) * u64_to_unit(rng_u64(key, STREAM_TRAIN_RESET, (uint64_t)(episode * 4 + i)));
            }
            real_env_obs<ENV, S>(st0, state);
        }

        int ep_len = 0;
        for (int t = 0; t < cfg.max_steps; ++t) {
            const int64_t size_after = rb_size + 1 < a.rb_cap ? rb_size + 1 : a.rb_cap;
            const int64_t new_pos = rb_ptr;
            // ================= phase A =================
            float row[16];
            int my_idx = -1;
            const int b = tid & (MAX_B - 1), pass = tid >> 8;
            if (wave == ENV_WAVE) {
                // ---- select_train_action (DDQN.py:97-104) ----
                double u;
                if (tape) {
                    if (n_eps >= a.tapes.eps_uniform_stride) { status = -2; u = 1.0; }
                    else u = a.tapes.eps_uniform[chain * a.tapes.eps_uniform_stride + n_eps];
                } else u = u64_to_unit(rng_u64(key, STREAM_EPS, (uint64_t)n_eps));
                ++n_eps;
                int action, explored = 0;
                if (u < eps_g) {
                    explored = 1;
                    if (tape) {
                        if (n_act >= a.tapes.rand_action_stride) { status = -3; action = 0; }
                        else action = a.tapes.rand_action[chain * a.tapes.rand_action_stride + n_act];
                    } else action = (int)u64_to_below(rng_u64(key, STREAM_ACTION, (uint64_t)n_act), (uint32_t)A);
                    ++n_act;
                } else {
                    action = wave_q_argmax<S, A>(q_onl, state, wscr, Hq, RP, cfg.q_act, cfg.q_prelu, lane);
                }
                // ---- EnvWrapper.step -> VirtualEnv.step: x = [onehot(action), state] ----
                float x[K];
#pragma unroll
                for (int k = 0; k < A; ++k) x[k] = (k == action) ? 1.0f : 0.0f;
#pragma unroll
                for (int i = 0; i < S; ++i) x[A + i] = state[i];
                for (int uu = lane; uu < 3 * Hse; uu += 64) {
                    const int net = uu / Hse, j = uu - net * Hse;
                    const float *w = se_w0T + net * K * Hse + j;
                    float z = 0.0f;
#pragma unroll
                    for (int k = 0; k < K; ++k) z = fma32(x[k], w[k * Hse], z);
                    z = z + se_b0[uu];
                    se_h[uu] = act_fwd(cfg.se_act, cfg.se_prelu, z);
                }
                wave_sync();
                float acc = 0.0f;
                if (lane < S + 2) {
                    const int net = lane < S ? 0 : (lane == S ? 1 : 2);
                    const float *h = se_h + net * Hse, *w = se_wout + lane * Hse;
                    for (int j = 0; j < Hse; ++j) acc = fma32(h[j], w[j], acc);
                    acc = acc + se_bout[lane];
                }
                wave_sync();
                float next_state[S];
#pragma unroll
                for (int i = 0; i < S; ++i) next_state[i] = __shfl(acc, i);
                const float reward = __shfl(acc, S), done = __shfl(acc, S + 1);
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
                    if (lane < 2 * S + 3) { rb[new_pos * RS + lane] = val; newrow[lane] = val; }
                }
                if (lane == 0) {
                    ctrl[t & 1] = done;
                    if (a.out.trace_action && n_trace < a.out.trace_cap) {
                        const int64_t k = chain * a.out.trace_cap + n_trace;
                        a.out.trace_action[k] = action | (explored << 16);
                        for (int i = 0; i < S; ++i) { a.out.trace_state[k * S + i] = state[i]; a.out.trace_next_state[k * S + i] = next_state[i]; }
                        a.out.trace_reward_done[k * 2] = reward; a.out.trace_reward_done[k * 2 + 1] = done;
                    }
                }
                ++n_trace;
#pragma unroll
                for (int i = 0; i < S; ++i) state[i] = next_state[i];
            } else if (learning && pass < 3 && b < B) {
                // ---- ReplayBuffer.sample (utils.py:34-45): prefetch this step's minibatch row ----
                const int64_t n = learn_it * B + b;
                if (tape) {
                    if (n >= a.tapes.replay_idx_stride) { status = -4; my_idx = 0; }
                    else my_idx = a.tapes.replay_idx[chain * a.tapes.replay_idx_stride + n];
                    if (my_idx < 0 || my_idx >= size_after) { status = -6; my_idx = 0; }
                } else my_idx = (int)u64_to_below(rng_u64(key, STREAM_REPLAY, (uint64_t)n), (uint32_t)size_after);
                if (my_idx != new_pos) {
                    const float4 *src = reinterpret_cast<const float4 *>(rb + (int64_t)my_idx * RS);
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (v * 4 < RS) { float4 f = src[v]; row[v * 4] = f.x; row[v * 4 + 1] = f.y; row[v * 4 + 2] = f.z; row[v * 4 + 3] = f.w; }
                }
            }
            rb_ptr = rb_ptr + 1 == a.rb_cap ? 0 : rb_ptr + 1;
            rb_size = size_after;
            ++ep_len; ++train_steps;
            __syncthreads();                                   // B1
            const float done_now = ctrl[t & 1];

            if (learning) {
                // ================= learn: DDQN.learn (DDQN.py:60-94) =================
                if (pass < 3 && b < B) {
                    if (my_idx == (int)new_pos) {
#pragma unroll
                        for (int v = 0; v < 16; ++v) if (v < RS) row[v] = newrow[v];
                    }
                    float x[S];
#pragma unroll
                    for (int i = 0; i < S; ++i) x[i] = pass == 0 ? row[i] : row[S + 1 + i];
                    const float *W = pass == 2 ? q_tgt : q_onl;
                    float q[A];
#pragma unroll
                    for (int aa = 0; aa < A; ++aa) q[aa] = 0.0f;
                    for (int j = 0; j < Hq; ++j) {
                        const float *rec = W + j * RP;
                        float z = 0.0f;
#pragma unroll
                        for (int i = 0; i < S; ++i) z = fma32(x[i], rec[i], z);
                        z = z + rec[S];
                        const float h = act_fwd(cfg.q_act, cfg.q_prelu, z);
#pragma unroll
                        for (int aa = 0; aa < A; ++aa) q[aa] = fma32(h, rec[S + 1 + aa], q[aa]);
                        if (pass == 0) hB[b * HP + j] = h;
                    }
#pragma unroll
                    for (int aa = 0; aa < A; ++aa) qres[(pass * MAX_B + b) * A + aa] = q[aa] + W[Hq * RP + aa];
                    if (pass == 0) {
#pragma unroll
                        for (int i = 0; i < S; ++i) sB[b * S + i] = row[i];
                        rda[b * 4 + 0] = row[2 * S + 1]; rda[b * 4 + 1] = row[2 * S + 2]; rda[b * 4 + 2] = row[S];
                    }
                }
                __syncthreads();                               // B2
                if (tid < B) {
                    // TD target and dLoss/dQ(s,a) per sample (DDQN.py:82-86; mse_loss backward = 2/B * diff)
                    const float r = rda[b * 4], d = rda[b * 4 + 1];
                    const int ab = (int)rda[b * 4 + 2];
                    int am = 0;
                    float best = qres[(1 * MAX_B + b) * A];
#pragma unroll
                    for (int aa = 1; aa < A; ++aa) { float v = qres[(1 * MAX_B + b) * A + aa]; if (v > best) { best = v; am = aa; } }
                    const float t1 = g32 * qres[(2 * MAX_B + b) * A + am];
                    const float t2 = 1.0f - d;
                    const float y = r + t1 * t2;
                    const float diff = qres[(0 * MAX_B + b) * A + ab] - y;
                    dqB[b] = norm * diff;
                }
                __syncthreads();                               // B3
                // ---- batch gradient in micro-chunks: wave c reduces samples [c*chunk, (c+1)*chunk) ----
                if (wave < a.n_chunks) {
                    const int b0 = wave * a.chunk, b1 = (b0 + a.chunk < B) ? b0 + a.chunk : B;
                    float *pc = part + wave * P;
                    for (int j = lane; j < ((Hq + 63) & ~63); j += 64) {
                        const bool jv = j < Hq;
                        float gW1[S], gW2[A], gb2[A], gb1 = 0.0f;
#pragma unroll
                        for (int i = 0; i < S; ++i) gW1[i] = 0.0f;
#pragma unroll
                        for (int aa = 0; aa < A; ++aa) { gW2[aa] = 0.0f; gb2[aa] = 0.0f; }
                        for (int bb = b0; bb < b1; ++bb) {
                            const float h = jv ? hB[bb * HP + j] : 0.0f;
                            const float dq = dqB[bb];
                            const int ab = (int)rda[bb * 4 + 2];
                            // dL/dz through the output layer and the activation (only row a_b of dQ is non-zero)
                            const float da = dq * (jv ? q_onl[j * RP + S + 1 + ab] : 0.0f);
                            const float dz = act_bwd(cfg.q_act, cfg.q_prelu, h, da);
#pragma unroll
                            for (int i = 0; i < S; ++i) gW1[i] = fma32(dz, sB[bb * S + i], gW1[i]);
                            gb1 = gb1 + dz;
#pragma unroll
                            for (int aa = 0; aa < A; ++aa)
                                if (ab == aa) { gW2[aa] = fma32(dq, h, gW2[aa]); gb2[aa] = gb2[aa] + dq; }
                        }
                        if (jv) {
#pragma unroll
                            for (int i = 0; i < S; ++i) pc[j * S + i] = gW1[i];
                            pc[Hq * S + j] = gb1;
#pragma unroll
                            for (int aa = 0; aa < A; ++aa) pc[Hq * S + Hq + aa * Hq + j] = gW2[aa];
                        }
                        if (j == 0) {
#pragma unroll
                            for (int aa = 0; aa < A; ++aa) pc[Hq * S + Hq + A * Hq + aa] = gb2[aa];
                        }
                    }
                }
                __syncthreads();                               // B4
                // ---- torch.optim.Adam single-tensor step + Polyak (DDQN.py:88-93), one thread per parameter ----
                b1pow *= cfg.adam_beta1;
                b2pow *= cfg.adam_beta2;
                {
                    const double bc1 = 1.0 - b1pow, bc2 = 1.0 - b2pow;
                    const float neg_step = (float)(-(cfg.lr / bc1));
                    const float bc2_sqrt = (float)__builtin_sqrt(bc2);
#pragma unroll
                    for (int k = 0; k < PPT; ++k) {
                        const int p = tid + k * NT;
                        if (p < P) {
                            float g = part[p];
                            for (int c = 1; c < a.n_chunks; ++c) g = g + part[c * P + p];
                            p_m[k] = fma32(w1, g - p_m[k], p_m[k]);
                            p_v[k] = p_v[k] * beta2;
                            p_v[k] = fma32(w2 * g, g, p_v[k]);
                            const float denom = __builtin_sqrtf(p_v[k]) / bc2_sqrt + adam_eps;
                            p_onl[k] = p_onl[k] + (neg_step * p_m[k]) / denom;
                            p_tgt[k] = tau * p_onl[k] + omt * p_tgt[k];
                            q_onl[my_off[k]] = p_onl[k];
                            q_tgt[my_off[k]] = p_tgt[k];
                        }
                    }
                }
                ++learn_it;
                __syncthreads();                               // B5
            }
            if (done_now > 0.5f) break;                        // base_agent.py:128
        }
        ++episodes_run;
        if (tid == 0 && a.out.episode_len) a.out.episode_len[chain * cfg.train_episodes + episode] = ep_len;
        __syncthreads();

        // ---- per-episode test on the real env (base_agent.py:134-136) ----
        test_phase();
        int brk = 0;
        if (tid == 0) {
            double sm = 0.0;
            for (int i = 0; i < cfg.test_episodes; ++i) sm += ret[i];
            const double tm = sm / (double)cfg.test_episodes;
            meter[n_meter] = tm;
            if (a.out.episode_test_mean) a.out.episode_test_mean[chain * cfg.train_episodes + episode] = tm;
            // early out on the real env (base_agent.py:49-62,141-148; AverageMeter._mean utils.py:103-105)
            if (learning) {
                int lo = n_meter + 1 - cfg.early_out_num; if (lo < 0) lo = 0;
                double s2 = 0.0;
                for (int i = lo; i <= n_meter; ++i) s2 += meter[i];
                const double avg = s2 / ((double)(n_meter + 1 - lo) + 1e-9);
                if (avg >= cfg.solved_reward) brk = 1;
            }
            ctrl[2] = (float)brk;
        }
        ++n_meter;
        __syncthreads();
        brk = ctrl[2] > 0.5f;
        __syncthreads();
        if (brk) break;
    }

    // ---- final test (GTN_worker.py:199) and score = statistics.mean(reward_list_test) ----
    test_phase();
    if (tid == 0) {
        double sm = 0.0;
        for (int i = 0; i < cfg.test_episodes; ++i) sm += ret[i];
        a.out.score[chain] = sm / (double)cfg.test_episodes;
        if (a.out.final_returns) for (int i = 0; i < cfg.test_episodes; ++i) a.out.final_returns[chain * cfg.test_episodes + i] = ret[i];
        if (a.out.stats) {
            a.out.stats[chain * 4 + 0] = episodes_run; a.out.stats[chain * 4 + 1] = train_steps;
            a.out.stats[chain * 4 + 2] = learn_it; a.out.stats[chain * 4 + 3] = test_steps;
        }
        const double nan = __builtin_nan("");
        for (int e = episodes_run; e < cfg.train_episodes; ++e) {
            if (a.out.episode_test_mean) a.out.episode_test_mean[chain * cfg.train_episodes + e] = nan;
            if (a.out.episode_len) a.out.episode_len[chain * cfg.train_episodes + e] = 0;
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

}  // namespace lenv

using namespace lenv;

static inline int64_t mlp_params(int in, int H, int L, int out)
{
    return (int64_t)in * H + H + (int64_t)(L - 1) * ((int64_t)H * H + H) + (int64_t)H * out + out;
}

static int64_t inner_rb_cap(const lenv_ddqn_cfg *cfg)
{
    int64_t cap = (int64_t)cfg->train_episodes * cfg->max_steps;
    if (cap > cfg->rb_size) cap = cfg->rb_size;
    return cap < 1 ? 1 : cap;
}

static int inner_row_stride(const lenv_ddqn_cfg *cfg) { return (2 * cfg->state_dim + 3 + 3) & ~3; }

// LDS carve-up of one chain's workgroup; returns LENV_ERR_UNSUPPORTED when the shapes do not fit 160 KiB
static int inner_layout(const lenv_ddqn_cfg *cfg, InnerArgs &a)
{
    const int S = cfg->state_dim, A = cfg->num_actions, Hq = cfg->q_hidden, Hse = cfg->se_hidden, B = cfg->batch_size;
    a.P_q = (int)mlp_params(S, Hq, 1, A);
    a.se_net_size[0] = (int)mlp_params(S + A, Hse, 1, S);
    a.se_net_size[1] = a.se_net_size[2] = (int)mlp_params(S + A, Hse, 1, 1);
    a.P_se = a.se_net_size[0] + a.se_net_size[1] + a.se_net_size[2];
    if (a.P_q > PPT * NT) return LENV_ERR_UNSUPPORTED;
    a.RP = (S + 1 + A + 3) & ~3;
    a.HP = Hq | 1;
    a.chunk = cfg->grad_chunk > 0 ? cfg->grad_chunk : (B + NW - 1) / NW;
    a.n_chunks = (B + a.chunk - 1) / a.chunk;
    if (a.n_chunks > NW) return LENV_ERR_UNSUPPORTED;
    const int K = S + A, HqPad = (Hq + 63) & ~63;
    int o = 0;
    auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
    a.o_se_w0T = take(3 * K * Hse); a.o_se_b0 = take(3 * Hse); a.o_se_wout = take((S + 2) * Hse); a.o_se_bout = take(S + 2);
    a.o_se_h = take(3 * Hse);
    a.o_q_onl = take(Hq * a.RP + A); a.o_q_tgt = take(Hq * a.RP + A);
    a.o_wscr = take(NW * HqPad);
    a.o_hB = take(B * a.HP);
    a.o_sB = take(B * S); a.o_rda = take(B * 4); a.o_dqB = take(B);
    a.o_qres = take(3 * MAX_B * A);
    a.o_part = take(a.n_chunks * a.P_q);
    a.o_newrow = take(16); a.o_ctrl = take(4 + NW);
    a.o_ret = take(2 * cfg->test_episodes + 2);
    a.lds_floats = o;
    if ((size_t)o * sizeof(float) > 160 * 1024) return LENV_ERR_UNSUPPORTED;
    return LENV_OK;
}

static int inner_check(const lenv_ddqn_cfg *cfg)
{
    const int S = cfg->state_dim, A = cfg->num_actions, Hq = cfg->q_hidden, Hse = cfg->se_hidden, B = cfg->batch_size;
    if (cfg->q_layers != 1 || cfg->se_layers != 1) return LENV_ERR_UNSUPPORTED;   // hidden_layer > 1: next round
    if (B < 1 || B > MAX_B || Hq < 1 || Hse < 1 || cfg->test_episodes < 1 || cfg->train_episodes < 0 || cfg->max_steps < 1)
        return LENV_ERR_UNSUPPORTED;
    if (!((cfg->env_id == LENV_ENV_CARTPOLE && S == 4 && A == 2) || (cfg->env_id == LENV_ENV_ACROBOT && S == 6 && A == 3)))
        return LENV_ERR_UNSUPPORTED;
    return LENV_OK;
}

extern "C" int64_t lenv_ddqn_se_lds_bytes(const lenv_ddqn_cfg *cfg)
{
    if (!cfg) return LENV_ERR_INVALID;
    int rc = inner_check(cfg);
    if (rc != LENV_OK) return rc;
    InnerArgs a;
    rc = inner_layout(cfg, a);
    if (rc != LENV_OK) return rc;
    return (int64_t)a.lds_floats * (int64_t)sizeof(float);
}

extern "C" size_t lenv_ddqn_se_workspace_bytes(const lenv_ddqn_cfg *cfg, int64_t chains)
{
    if (!cfg || chains < 0) return 0;
    size_t replay = (size_t)chains * inner_rb_cap(cfg) * inner_row_stride(cfg) * sizeof(float);
    size_t meter = (size_t)chains * (cfg->train_episodes > 0 ? cfg->train_episodes : 1) * sizeof(double);
    return ((replay + 255) & ~(size_t)255) + meter + 256;
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
    const size_t lds_bytes = (size_t)a.lds_floats * sizeof(float);

    void (*kern)(const InnerArgs) = nullptr;
    if (cfg->env_id == LENV_ENV_CARTPOLE) kern = ddqn_se_inner_kernel<LENV_ENV_CARTPOLE, 4, 2>;
    else kern = ddqn_se_inner_kernel<LENV_ENV_ACROBOT, 6, 3>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return LENV_ERR_LAUNCH;
    if (out->status) {
        e = hipMemsetAsync(out->status, 0, sizeof(int32_t) * chains, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return LENV_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)chains), dim3(NT), lds_bytes, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}
