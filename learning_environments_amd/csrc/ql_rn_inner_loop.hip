// ql_rn_inner_loop.hip -- fused NES inner loop for config 4: tabular Q-learning on a potential-shaped RewardEnv over a
// grid MDP, one wave64 per chain.
//
// Replaces, for `chains` (theta +/- eps) perturbations of the reward network at once, the reference's
//   GTN_Worker.calc_score                       agents/GTN_worker.py:187-221
//     QL() / select_train_action / learn        agents/QL.py:13-106 (python-float64 Q-table, argmax on the fp32 cast row)
//     BaseAgent.train / test                    agents/base_agent.py:64-227
//     EnvWrapper.step -> RewardEnv.step         envs/env_wrapper.py:49-70, envs/reward_env.py:61-133
//     GridworldEnv.step + TimeLimit             envs/gridworld.py:38-110 (given as transition tables)
//
// The reward network only ever sees one-hot states, so phi(s) for all N states and the shaped reward of every (s,a) are
// evaluated ONCE per chain (lane = state, coalesced reads of theta/eps) and kept in LDS; the training itself is a
// strictly sequential scalar walk (<= train_episodes*max_steps steps) executed by lane 0 on the LDS-resident fp64 Q-table.
// Integer/fp64 work is exact by construction; phi uses the canonical sequential-fmaf order of oracle/lenv_oracle.h.
#include "lenv_device.cuh"

namespace lenv {

struct QlArgs {
    lenv_ql_cfg cfg;
    const float *theta, *eps; const int32_t *worker; const float *sign;
    const float *shaped_override;
    const int32_t *next_state; const double *reward; const uint8_t *done;
    const uint64_t *rng_keys;
    lenv_tapes tapes;
    lenv_ql_out out;
    int64_t P;
    int draw_cap;          // exploration draws precomputed per episode (a speed-up: draws past it are computed inline); clamped to what fits the LDS
};

__device__ __forceinline__ int ql_argmax_f32(const double *row, int n)
{
    int best = 0;
    float bv = (float)row[0];
    for (int i = 1; i < n; ++i) { const float v = (float)row[i]; if (v > bv) { bv = v; best = i; } }
    return best;
}

// phi(s) = reward_net(one_hot(s)) for all states with W = theta + sign*eps[worker] (reward_env.py:74-76,
// GTN_worker.py:165-175), then RewardEnv._calc_reward for every (s,a) (reward_env.py:81-110) in fp32, left to right.
// One wave; phi/shaped are LDS (or global) arrays of the calling wave.  Ends with a barrier.
__device__ __forceinline__ void rn_phi_and_shaped(const lenv_ql_cfg &cfg, const float *theta, const float *eps_all,
                                                  const int32_t *worker, const float *sign, const float *shaped_override,
                                                  const int32_t *next_state, const double *reward, int64_t P, int64_t chain,
                                                  int lane, float *phi, float *shaped, float *shaped_out, float *phi_out, float *hbuf)
{
    const int N = cfg.n_states, A = cfg.n_actions, H = cfg.rn_hidden;
    const int t = cfg.reward_env_type;
    if (t != 0 && !shaped_override && cfg.rn_layers > 1) {
        // reward nets with several hidden layers (default_config_gridworld_reward_env.yaml: HoleRoomLarge ships hidden_layer 2;
        // build_nn_from_config, models/model_utils.py:16-29): Linear(N, H) | [Linear(H, H)] x (layers - 1) | Linear(H, 1), one
        // shared activation.  Lane = state; the hidden rows of a lane live in LDS (hbuf [2][H][64], lane-minor: conflict-free);
        // every dot product is the k-ascending fmaf chain from 0 with the bias added last (oracle: mlp_forward_one)
        const float sg = eps_all ? sign[chain] : 0.0f;
        const float *th = theta, *e = eps_all ? eps_all + (int64_t)worker[chain] * P : nullptr;
        auto Wp = [&](int64_t i) { return e ? fma32(sg, e[i], th[i]) : th[i]; };
        float *h0 = hbuf + lane, *h1 = hbuf + (int64_t)H * 64 + lane;
        for (int s0 = 0; s0 < N; s0 += 64) {
            const int s = s0 + lane < N ? s0 + lane : N - 1;                      // idle lanes redo the last state (uniform loops)
            const int64_t off_b0 = (int64_t)H * N;
            for (int j = 0; j < H; ++j) h0[j * 64] = act_fwd(cfg.rn_act, cfg.rn_prelu, Wp((int64_t)j * N + s) + Wp(off_b0 + j));
            float *hin = h0, *hout = h1;
            int64_t off = off_b0 + H;
            for (int l = 1; l < cfg.rn_layers; ++l) {
                for (int j = 0; j < H; ++j) {
                    float z = 0.0f;
                    for (int k = 0; k < H; ++k) z = fma32(hin[k * 64], Wp(off + (int64_t)j * H + k), z);
                    z = z + Wp(off + (int64_t)H * H + j);
                    hout[j * 64] = cfg.rn_layer_norm ? z : act_fwd(cfg.rn_act, cfg.rn_prelu, z);
                }
                if (cfg.rn_layer_norm) {
                    // the reward net's own LayerNorm (envs section use_layer_norm; model_utils.py:22-37): NES perturbs nn.Linear modules only, so
                    // weight 1 / bias 0; the lane owns the row: sequential mean / variance as the oracle's mlp_forward_one_ex
                    float sm = 0.0f, sv = 0.0f;
                    for (int j = 0; j < H; ++j) sm = sm + hout[j * 64];
                    const float mean = sm / (float)H;
                    for (int j = 0; j < H; ++j) { const float dj = hout[j * 64] - mean; sv = fma32(dj, dj, sv); }
                    const float r = 1.0f / __builtin_sqrtf(sv / (float)H + 1e-5f);
                    for (int j = 0; j < H; ++j) hout[j * 64] = act_fwd(cfg.rn_act, cfg.rn_prelu, fma32((hout[j * 64] - mean) * r, 1.0f, 0.0f));
                }
                off += (int64_t)H * H + H;
                float *tmp = hin; hin = hout; hout = tmp;
            }
            float acc = 0.0f;
            for (int j = 0; j < H; ++j) acc = fma32(hin[j * 64], Wp(off + j), acc);
            if (s0 + lane < N) phi[s] = acc + Wp(off + H);
        }
    } else if (t != 0 && !shaped_override) {
        const float sg = eps_all ? sign[chain] : 0.0f;
        const float *th = theta, *e = eps_all ? eps_all + (int64_t)worker[chain] * P : nullptr;
        auto W = [&](int64_t i) { return e ? fma32(sg, e[i], th[i]) : th[i]; };
        const int64_t off_b0 = (int64_t)H * N, off_wo = off_b0 + H, off_bo = off_wo + H;
        for (int s = lane; s < N; s += 64) {
            float acc = 0.0f;
            for (int j = 0; j < H; ++j) {
                const float z = W((int64_t)j * N + s) + W(off_b0 + j);   // one-hot input: the fmaf chain collapses to W0[j][s]
                const float h = act_fwd(cfg.rn_act, cfg.rn_prelu, z);
                acc = fma32(h, W(off_wo + j), acc);
            }
            phi[s] = acc + W(off_bo);
        }
    } else {
        for (int s = lane; s < N; s += 64) phi[s] = 0.0f;
    }
    __syncthreads();
    const float g32 = (float)cfg.gamma;
    for (int i = lane; i < N * A; i += 64) {
        float v;
        if (shaped_override) v = shaped_override[i];
        else {
            const int s = i / A, s2 = next_state[i];
            const float r32 = (float)reward[i];
            switch (t) {
            case 0: v = r32; break;
            case 1: v = g32 * phi[s2] - phi[s]; break;
            case 2: v = (r32 + g32 * phi[s2]) - phi[s]; break;
            case 5: v = phi[s2]; break;
            default: v = r32 + phi[s2]; break;
            }
        }
        shaped[i] = v;
        if (shaped_out) shaped_out[i] = v;
    }
    if (phi_out) for (int s = lane; s < N; s += 64) phi_out[s] = phi[s];
    __syncthreads();
}

__global__ __launch_bounds__(64) void rn_shape_kernel(const QlArgs a, float *phi_out, float *shaped_out)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int N = a.cfg.n_states, A = a.cfg.n_actions;
    float *phi = reinterpret_cast<float *>(lds_raw), *shaped = phi + N;
    const int64_t chain = blockIdx.x;
    rn_phi_and_shaped(a.cfg, a.theta, a.eps, a.worker, a.sign, nullptr, a.next_state, a.reward, a.P, chain, threadIdx.x, phi,
                      shaped, shaped_out + chain * N * A, phi_out ? phi_out + chain * N : nullptr, shaped + N * A);
}

// KIND: 0 QL, 1 SARSA; CB: count-based exploration bonus (compile-time so that the plain-QL walk carries none of their code)
template <int KIND, bool CB>
__global__ __launch_bounds__(64) void ql_rn_inner_kernel(const QlArgs a)
{
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const lenv_ql_cfg &cfg = a.cfg;
    const int lane = threadIdx.x;
    const int64_t chain = blockIdx.x;
    const int N = cfg.n_states, A = cfg.n_actions;
    double *q = reinterpret_cast<double *>(lds_raw);                 // [N*A] fp64 Q-table
    double *meter = q + N * A;                                       // [train_episodes]
    double *rets = meter + cfg.train_episodes;                       // [test_episodes]
    float *phi = reinterpret_cast<float *>(rets + cfg.test_episodes);    // [N]
    float *shaped = phi + N;                                         // [N*A]
    int *visits = reinterpret_cast<int *>(shaped + N * A);           // [N*A] visitation counts n(s,a) (count-based agents)
    float *hbuf = reinterpret_cast<float *>(visits + N * A);         // [2][H][64] hidden rows of a multi-layer reward net (rn_layers > 1 only)
    // this episode's exploration draws, filled by all 64 lanes before lane 0 walks the episode (counter mode): the draws are a
    // function of (key, stream, index) only, and two 64-bit mixes per draw are a third of a step of the one-lane walk
    const int draw_cap = a.draw_cap;                                 // <= max_steps * (1 + batch_size for SARSA): the host clamps it to the LDS left
    // (through an integer on purpose: it leaves these buffers behind generic pointers, and lane 0's one-lane dependent walk reads them faster
    // as FLAT loads than as ds_reads -- measured 1.21 vs 1.26 us per step with the address space made explicit)
    double *u_buf = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(hbuf + (cfg.rn_layers > 1 ? 2 * 64 * cfg.rn_hidden : 0)) + 7) & ~(uintptr_t)7);   // [draw_cap]
    int *a_buf = reinterpret_cast<int *>(u_buf + draw_cap);          // [draw_cap]
    volatile int *xctl = a_buf + draw_cap;                           // [8] walker -> wave: stop flags, draw counters

    rn_phi_and_shaped(cfg, a.theta, a.eps, a.worker, a.sign, a.shaped_override, a.next_state, a.reward, a.P, chain, lane,
                      phi, shaped, a.out.shaped ? a.out.shaped + chain * N * A : nullptr, nullptr, hbuf);
    for (int i = lane; i < N * A; i += 64) { q[i] = 0.0; if (CB) visits[i] = 0; }   // QL.py:25,31
    __syncthreads();
    // The walk is wave-uniform: all 64 lanes run it on the same values (LDS / table reads broadcast, the Q-table writes hit one address
    // with one value), so its branches are scalar branches; only lane 0 writes the step trace.  (Forcing the indices into
    // SGPRs with v_readfirstlane as well was measured slower: 2.30 vs 2.07 ms.)
    const bool walker = true;

    // ---- the sequential part: lane 0 (the other lanes only refill the draw buffers between episodes) ----
    const uint64_t key = a.rng_keys ? a.rng_keys[chain] : 0;
    const bool tape = cfg.rng_mode == LENV_RNG_TAPE;
    int status = 0, episodes_run = 0;
    int64_t n_eps = 0, n_act = 0, train_steps = 0, learn_steps = 0, test_steps = 0;
    double eps_g = cfg.eps_init;
    const int k_rep = cfg.same_action_num > 1 ? cfg.same_action_num : 1;   // env steps per chosen action

    // BaseAgent.test; with `budgeted` the time_is_up check of base_agent.py:177-184 runs before every episode against the
    // env steps this test has used (`remaining` = budget left when the test starts) and pads like base_agent.py:33-36
    auto test_phase = [&](bool budgeted = false, int64_t remaining = 0) {
        int64_t used = 0;
        for (int te = 0; te < cfg.test_episodes; ++te) {
            if (budgeted && used > remaining) {
                double mn = -1e9;
                if (te > 0) { mn = rets[0]; for (int i = 1; i < te; ++i) if (rets[i] < mn) mn = rets[i]; }
                for (int i = te; i < cfg.test_episodes; ++i) rets[i] = mn;
                break;
            }
            int s = cfg.start_state;
            float ep_reward = 0.0f;                                    // fp32 tensor accumulation, base_agent.py:212
            if (k_rep == 1 && A == 4) {
                // the state's transition rows are requested together with its Q row: one memory round trip per step, not two
                for (int st = 0; st < cfg.max_steps; ++st) {
                    const int4 nx = *reinterpret_cast<const int4 *>(a.next_state + s * 4);
                    const uchar4 dv = *reinterpret_cast<const uchar4 *>(a.done + s * 4);
                    const double r0 = a.reward[s * 4], r1 = a.reward[s * 4 + 1], r2 = a.reward[s * 4 + 2], r3 = a.reward[s * 4 + 3];
                    const int ac = ql_argmax_f32(q + s * 4, 4);
                    const int dn = ac == 0 ? dv.x : (ac == 1 ? dv.y : (ac == 2 ? dv.z : dv.w));
                    ep_reward = ep_reward + (float)(ac == 0 ? r0 : (ac == 1 ? r1 : (ac == 2 ? r2 : r3)));
                    s = ac == 0 ? nx.x : (ac == 1 ? nx.y : (ac == 2 ? nx.z : nx.w));
                    ++test_steps; ++used;
                    if (dn) break;
                }
            } else if (k_rep == 1) {
                for (int st = 0; st < cfg.max_steps; ++st) {
                    const int ac = ql_argmax_f32(q + s * A, A);
                    const int dn = a.done[s * A + ac];
                    ep_reward = ep_reward + (float)a.reward[s * A + ac];
                    s = a.next_state[s * A + ac];
                    ++test_steps; ++used;
                    if (dn) break;
                }
            } else {
                int dn = 0, el = 0;
                for (int st = 0; st < cfg.max_steps && !dn; st += k_rep) {     // base_agent.py:194 range(0, max_steps, same_action_num)
                    const int ac = ql_argmax_f32(q + s * A, A);
                    double rsum = 0.0;                                 // EnvWrapper.step: python-float sum, the repeats stop at done
                    for (int r_ = 0; r_ < k_rep; ++r_) {
                        dn = a.done[s * A + ac];
                        rsum = rsum + a.reward[s * A + ac];
                        s = a.next_state[s * A + ac];
                        ++test_steps; ++used; ++el;
                        if (el >= cfg.max_steps) dn = 1;
                        if (dn) break;
                    }
                    ep_reward = ep_reward + (float)rsum;
                }
            }
            rets[te] = (double)ep_reward;
        }
    };
    auto mean_rets = [&]() { double sm = 0.0; for (int i = 0; i < cfg.test_episodes; ++i) sm += rets[i]; return sm / (double)cfg.test_episodes; };

    int timed_out_at = -1;
    int64_t eps0 = 0, act0 = 0;                                       // draw indices the buffers start at
    for (int episode = 0; episode < cfg.train_episodes; ++episode) {
        // deterministic time-out (lenv_ql_cfg::step_budget, base_agent.py:30-47,90-97): elapsed = env steps taken so far
        if (walker) {
            xctl[0] = (cfg.step_budget > 0 && train_steps + test_steps > cfg.step_budget) ? 1 : 0;
            xctl[1] = (int)(n_eps & 0xffffffff); xctl[2] = (int)(n_eps >> 32); xctl[3] = (int)(n_act & 0xffffffff); xctl[4] = (int)(n_act >> 32);
        }
        __syncthreads();
        if (xctl[0]) { timed_out_at = episode; break; }
        eps0 = (int64_t)(((uint64_t)(uint32_t)xctl[2] << 32) | (uint32_t)xctl[1]);
        act0 = (int64_t)(((uint64_t)(uint32_t)xctl[4] << 32) | (uint32_t)xctl[3]);
        if (!tape) for (int i = lane; i < draw_cap; i += 64) {
            u_buf[i] = u64_to_unit(rng_u64(key, STREAM_EPS, (uint64_t)(eps0 + i)));
            a_buf[i] = (int)u64_to_below(rng_u64(key, STREAM_ACTION, (uint64_t)(act0 + i)), (uint32_t)A);
        }
        __syncthreads();
        int solved_brk = 0;
        if (walker) {
        if (episode == 0) eps_g = cfg.eps_init;                       // QL.py:101-106
        else { eps_g *= cfg.eps_decay; if (eps_g < cfg.eps_min) eps_g = cfg.eps_min; }
        int s = cfg.start_state, ep_len = 0, env_steps = 0;
        float tr_reward = 0.0f;                                        // base_agent.py:102,121 episode_reward += reward (fp32 tensor)
        for (int st = 0; st < cfg.max_steps; st += k_rep) {            // base_agent.py:104 range(0, max_steps, same_action_num)
            double u;
            if (tape) { if (n_eps >= a.tapes.eps_uniform_stride) { status = -2; u = 1.0; } else u = a.tapes.eps_uniform[chain * a.tapes.eps_uniform_stride + n_eps]; }
            else u = n_eps - eps0 < draw_cap ? u_buf[n_eps - eps0] : u64_to_unit(rng_u64(key, STREAM_EPS, (uint64_t)n_eps));
            ++n_eps;
            int ac, explored = 0;
            if (u < eps_g) {
                explored = 1;
                if (tape) { if (n_act >= a.tapes.rand_action_stride) { status = -3; ac = 0; } else ac = a.tapes.rand_action[chain * a.tapes.rand_action_stride + n_act]; }
                else ac = n_act - act0 < draw_cap ? a_buf[n_act - act0] : (int)u64_to_below(rng_u64(key, STREAM_ACTION, (uint64_t)n_act), (uint32_t)A);
                ++n_act;
            } else ac = ql_argmax_f32(q + s * A, A);
            // EnvWrapper.step (env_wrapper.py:56-61): the action same_action_num times or until done (gym.wrappers.TimeLimit: done after
            // max_steps env steps), the shaped rewards summed as python floats and stored as one fp32 value
            int s2, dn;
            double r;
            if (k_rep == 1) {                                          // (the shipped configurations: kept free of the repeat bookkeeping)
                s2 = a.next_state[s * A + ac];
                dn = a.done[s * A + ac];
                if (st + 1 >= cfg.max_steps) dn = 1;
                r = (double)shaped[s * A + ac];
            } else {
                s2 = s; dn = 0;
                double rsum = 0.0;
                for (int r_ = 0; r_ < k_rep; ++r_) {
                    const int sc = s2;
                    dn = a.done[sc * A + ac];
                    ++env_steps;
                    if (env_steps >= cfg.max_steps) dn = 1;
                    rsum = rsum + (double)shaped[sc * A + ac];
                    s2 = a.next_state[sc * A + ac];
                    if (dn) break;
                }
                r = (double)(float)rsum;
            }
            // QL.learn (QL.py:37-75) / SARSA.learn (SARSA.py:36-60), only once episode >= init_episodes (base_agent.py:127-128)
            if (episode >= cfg.init_episodes) {
                for (int k = 0; k < cfg.batch_size; ++k) {
                    double boot;
                    if (KIND == 1) {                                   // next_action = select_train_action(next_state)
                        double u2;
                        if (tape) { if (n_eps >= a.tapes.eps_uniform_stride) { status = -2; u2 = 1.0; } else u2 = a.tapes.eps_uniform[chain * a.tapes.eps_uniform_stride + n_eps]; }
                        else u2 = n_eps - eps0 < draw_cap ? u_buf[n_eps - eps0] : u64_to_unit(rng_u64(key, STREAM_EPS, (uint64_t)n_eps));
                        ++n_eps;
                        int a2;
                        if (u2 < eps_g) {
                            if (tape) { if (n_act >= a.tapes.rand_action_stride) { status = -3; a2 = 0; } else a2 = a.tapes.rand_action[chain * a.tapes.rand_action_stride + n_act]; }
                            else a2 = n_act - act0 < draw_cap ? a_buf[n_act - act0] : (int)u64_to_below(rng_u64(key, STREAM_ACTION, (uint64_t)n_act), (uint32_t)A);
                            ++n_act;
                        } else a2 = ql_argmax_f32(q + s2 * A, A);
                        boot = q[s2 * A + a2];
                    } else {
                        boot = q[s2 * A];
                        for (int i = 1; i < A; ++i) if (q[s2 * A + i] > boot) boot = q[s2 * A + i];
                    }
                    double rr = r;
                    if (CB) {                                          // QL.py:52-55
                        visits[s * A + ac] += 1;
                        rr += cfg.beta / (__builtin_sqrt((double)visits[s * A + ac]) + 1e-9);
                    }
                    const double delta = rr + cfg.gamma * boot * (dn ? 0.0 : 1.0) - q[s * A + ac];
                    q[s * A + ac] += cfg.alpha * delta;
                }
                ++learn_steps;
            }
            if (lane == 0 && a.out.trace_action && train_steps < a.out.trace_cap) {
                const int64_t k = chain * a.out.trace_cap + train_steps;
                a.out.trace_action[k] = ac | (explored << 16);
                a.out.trace_state[k * 2] = s; a.out.trace_state[k * 2 + 1] = s2;
                a.out.trace_reward_done[k * 2] = (float)r; a.out.trace_reward_done[k * 2 + 1] = dn ? 1.0f : 0.0f;
            }
            s = s2;
            tr_reward = tr_reward + (float)r;
            ep_len += k_rep; ++train_steps;
            if (dn) break;
        }
        ++episodes_run;
        if (a.out.episode_len) a.out.episode_len[chain * cfg.train_episodes + episode] = ep_len;
        // lenv_ql_cfg::test_mode 1 = train(env, test_env=None): no per-episode tests, the RewardEnv's own episode reward feeds the meter
        // (base_agent.py:134-138); a grid RewardEnv is not a VirtualEnv, so the real rule applies either way (:57-60)
        double tm;
        if (cfg.test_mode == 1) tm = (double)tr_reward;
        else { test_phase(); tm = mean_rets(); }
        meter[episode] = tm;
        if (a.out.episode_test_mean) a.out.episode_test_mean[chain * cfg.train_episodes + episode] = tm;
        if (episode >= cfg.init_episodes)                             // base_agent.py:141-148, utils.py:103-105
            solved_brk = meter_env_solved(meter, episode + 1, cfg.early_out_num, false, cfg.solved_reward, 0.0, episode, cfg.init_episodes);
        xctl[5] = solved_brk;
        }
        __syncthreads();
        if (xctl[5]) break;
    }
    if (!walker) return;
    test_phase(cfg.step_budget > 0, cfg.step_budget - (train_steps + test_steps));
    a.out.score[chain] = mean_rets();
    if (a.out.final_returns) for (int i = 0; i < cfg.test_episodes; ++i) a.out.final_returns[chain * cfg.test_episodes + i] = rets[i];
    if (a.out.stats) {
        a.out.stats[chain * 4 + 0] = episodes_run; a.out.stats[chain * 4 + 1] = train_steps;
        a.out.stats[chain * 4 + 2] = learn_steps; a.out.stats[chain * 4 + 3] = test_steps;
    }
    double pad_r = __builtin_nan("");
    int pad_l = 0;
    if (timed_out_at >= 0) {                                          // time_is_up's padding, base_agent.py:33-44
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
    if (a.out.q_table) for (int i = 0; i < N * A; ++i) a.out.q_table[chain * N * A + i] = q[i];
    if (a.out.status) a.out.status[chain] = status;
}

}  // namespace lenv

using namespace lenv;

// parameters of the grid reward net (models/model_utils.py:16-29 on one-hot states): Linear(N, H) | (layers - 1) x Linear(H, H) | Linear(H, 1)
static int64_t ql_rn_params(const lenv_ql_cfg *cfg)
{
    const int64_t H = cfg->rn_hidden;
    return (int64_t)cfg->n_states * H + H + (int64_t)(cfg->rn_layers - 1) * (H * H + H) + H + 1;
}

extern "C" int lenv_ql_rn_inner_loop(const lenv_ql_cfg *cfg, const float *theta, const float *eps, const int32_t *worker,
                                     const float *sign, const float *shaped_override, const int32_t *next_state,
                                     const double *reward, const uint8_t *done, const uint64_t *rng_keys,
                                     const lenv_tapes *tapes, int64_t chains, const lenv_ql_out *out, void *stream)
{
    if (!cfg || !next_state || !reward || !done || !out || !out->score || chains < 0) return LENV_ERR_INVALID;
    if (!theta && !shaped_override) return LENV_ERR_INVALID;
    if (eps && (!worker || !sign)) return LENV_ERR_INVALID;
    if (cfg->rng_mode == LENV_RNG_TAPE && !tapes) return LENV_ERR_INVALID;
    if (cfg->rng_mode == LENV_RNG_COUNTER && !rng_keys) return LENV_ERR_INVALID;
    if (chains == 0) return LENV_OK;
    const int t = cfg->reward_env_type;
    if (!(t == 0 || t == 1 || t == 2 || t == 5 || t == 6)) return LENV_ERR_UNSUPPORTED;   // info-vector types: next round
    if (cfg->rn_layers < 1 || cfg->rn_layers > 4) return LENV_ERR_UNSUPPORTED;
    if (cfg->n_states < 1 || cfg->n_actions < 1 || cfg->n_actions > 16 || cfg->test_episodes < 1 || cfg->train_episodes < 0 ||
        cfg->max_steps < 1 || cfg->batch_size < 1)
        return LENV_ERR_UNSUPPORTED;
    QlArgs a;
    a.cfg = *cfg;
    a.theta = theta; a.eps = eps; a.worker = worker; a.sign = sign; a.shaped_override = shaped_override;
    a.next_state = next_state; a.reward = reward; a.done = done; a.rng_keys = rng_keys;
    if (tapes) a.tapes = *tapes; else a.tapes = lenv_tapes{};
    a.out = *out;
    a.P = ql_rn_params(cfg);
    const size_t NA = (size_t)cfg->n_states * cfg->n_actions;
    // tables first; the per-episode draw buffers get what is left (the kernel computes draws past draw_cap inline: a long episode or a
    // large SARSA batch runs slower, it is not refused)
    const size_t fixed_bytes = sizeof(double) * (NA + cfg->train_episodes + cfg->test_episodes) + sizeof(float) * (cfg->n_states + NA) + sizeof(int) * NA + 16 +
                               (cfg->rn_layers > 1 ? sizeof(float) * 2 * 64 * (size_t)cfg->rn_hidden : 0) + 8 + 64;
    if (fixed_bytes > 160 * 1024) return LENV_ERR_UNSUPPORTED;
    const size_t per_draw = sizeof(double) + sizeof(int);
    size_t draws = (size_t)cfg->max_steps * (cfg->agent_kind == 1 ? 1 + (size_t)cfg->batch_size : 1);
    if (fixed_bytes + per_draw * draws > 160 * 1024) draws = (160 * 1024 - fixed_bytes) / per_draw;
    a.draw_cap = (int)draws;
    const size_t lds_bytes = fixed_bytes + per_draw * draws;
    if (cfg->agent_kind != 0 && cfg->agent_kind != 1) return LENV_ERR_UNSUPPORTED;
    void (*kern)(const QlArgs) = cfg->agent_kind == 1 ? (cfg->count_based ? ql_rn_inner_kernel<1, true> : ql_rn_inner_kernel<1, false>)
                                                      : (cfg->count_based ? ql_rn_inner_kernel<0, true> : ql_rn_inner_kernel<0, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return LENV_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)chains), dim3(64), lds_bytes, static_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}


// RewardEnv shaping for a population (the `rn_shape_population` entry of SURVEY.md §8(b)): phi_out [chains,N] (optional),
// shaped_out [chains,N*A].
extern "C" int lenv_rn_shape_population(const lenv_ql_cfg *cfg, const float *theta, const float *eps, const int32_t *worker,
                                        const float *sign, int64_t chains, const int32_t *next_state, const double *reward,
                                        float *phi_out, float *shaped_out, void *stream)
{
    if (!cfg || !theta || !next_state || !reward || !shaped_out || chains < 0) return LENV_ERR_INVALID;
    if (eps && (!worker || !sign)) return LENV_ERR_INVALID;
    if (chains == 0) return LENV_OK;
    const int t = cfg->reward_env_type;
    if (!(t == 0 || t == 1 || t == 2 || t == 5 || t == 6)) return LENV_ERR_UNSUPPORTED;
    if (cfg->rn_layers < 1 || cfg->rn_layers > 4) return LENV_ERR_UNSUPPORTED;
    QlArgs a;
    a.cfg = *cfg;
    a.theta = theta; a.eps = eps; a.worker = worker; a.sign = sign; a.shaped_override = nullptr;
    a.next_state = next_state; a.reward = reward; a.done = nullptr; a.rng_keys = nullptr;
    a.tapes = lenv_tapes{}; a.out = lenv_ql_out{};
    a.P = ql_rn_params(cfg);
    const size_t lds_bytes = sizeof(float) * ((size_t)cfg->n_states * (1 + cfg->n_actions)) + 16 + (cfg->rn_layers > 1 ? sizeof(float) * 2 * 64 * (size_t)cfg->rn_hidden : 0);
    if (lds_bytes > 160 * 1024) return LENV_ERR_UNSUPPORTED;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(rn_shape_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return LENV_ERR_LAUNCH;
    hipLaunchKernelGGL(rn_shape_kernel, dim3((unsigned)chains), dim3(64), lds_bytes, static_cast<hipStream_t>(stream), a, phi_out, shaped_out);
    return hipGetLastError() == hipSuccess ? LENV_OK : LENV_ERR_LAUNCH;
}
