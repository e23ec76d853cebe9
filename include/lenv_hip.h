/*
 * lenv_hip.h -- C-ABI of liblenv_hip.so: the MI355X (gfx950) implementation of the NES
 * inner-loop hot path of automl/learning_environments.
 *
 * The reference has no FFI; its boundary for this path is the Python object API
 * (envs/env_wrapper.py:16-70 EnvWrapper.step, agents/GTN_worker.py:76-108 GTN_Worker.run,
 * agents/GTN_master.py:81-116 GTN_Master.run).  These entry points are what a ctypes binding
 * added to those Python classes calls (INTEGRATION.md shows the stubs).  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller unless marked HOST;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); calls are asynchronous
 *     on that stream, never allocate, never synchronise, and are graph-capturable;
 *   - return value: 0 = ok, negative = LENV_ERR_* (no exceptions cross the ABI);
 *   - no internal threads, no global mutable state, no environment variables: everything that steers a launch is in its cfg;
 *     re-entrant per stream.
 *
 * Flat parameter layout of an MLP (models/model_utils.py:4-39), identical to the reference's
 * state-dict order with PReLU slopes removed: W0[H,in] b0[H] {W_l[H,H] b_l[H]} Wout[out,H] bout[out].
 * SE theta = state_net | reward_net | done_net (envs/virtual_env.py:23-31).
 */
#ifndef LENV_HIP_H
#define LENV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LENV_ABI_VERSION 7

enum {
    LENV_OK = 0,
    LENV_ERR_INVALID = -1,      /* bad argument / inconsistent sizes */
    LENV_ERR_UNSUPPORTED = -2,  /* shape or option outside what the kernels implement (NotImplementedError) */
    LENV_ERR_WORKSPACE = -3,    /* workspace too small */
    LENV_ERR_LAUNCH = -4,       /* HIP launch failed (hipGetLastError) */
    LENV_ERR_NO_DEVICE = -5
};

enum { LENV_ACT_IDENTITY = 0, LENV_ACT_RELU = 1, LENV_ACT_LEAKYRELU = 2, LENV_ACT_TANH = 3, LENV_ACT_PRELU = 4 };
enum { LENV_ENV_CARTPOLE = 0, LENV_ENV_ACROBOT = 1, LENV_ENV_CHEETAH_STANDIN = 2, LENV_ENV_MOUNTAINCAR = 3, LENV_ENV_PENDULUM = 4, LENV_ENV_CMC = 5 };
enum { LENV_RNG_COUNTER = 0, LENV_RNG_TAPE = 1 };
/* lenv_ddqn_cfg / lenv_td3_cfg `kernel_variant` bits (0 = the fastest kernel that takes the launch; the bits exist for A/B timing and for
 * the parity tests that hold the kernels against each other -- every variant produces the same bits):
 * NO_WAVECHAIN keeps the GEMM-queue kernel where a wave-chain kernel exists, GENERIC skips the shape-specialised instantiations,
 * TEAM_NARROW keeps whole forward items per lane in DDQN teams of three and more (default there: every item cut over the idle lanes). */
enum { LENV_VARIANT_NO_WAVECHAIN = 1, LENV_VARIANT_GENERIC = 2, LENV_VARIANT_TEAM_NARROW = 4,
       LENV_VARIANT_NO_DIRECT = 8 };   /* NO_DIRECT: the GEMM-queue kernels keep narrow nets on the product queue too (A/B timing, kernel-vs-kernel tests) */

/* models/model_utils.py:4-39 */
typedef struct {
    int32_t in_dim, hidden, layers, out_dim, act;
    float prelu;
    /* `use_layer_norm` of models/model_utils.py:22-37: ONE shared nn.LayerNorm(hidden) (eps 1e-5) after every hidden Linear but
     * the first, before the activation; its weight and bias [hidden] follow the second Linear in the flat vector
     * (Module.parameters() order).  Honoured by lenv_mlp_num_params / lenv_mlp_forward / lenv_se_step_population (the three SE nets of
     * a VirtualEnv.step).  The fused loops take the AGENT's LayerNorm through their own cfg fields (lenv_ddqn_cfg::q_layer_norm,
     * lenv_td3_cfg::use_layer_norm, lenv_td3d_cfg::use_layer_norm); the synthetic env's nets inside the DDQN /
     * DuelingDDQN loop: lenv_ddqn_cfg::se_layer_norm; the TD3-family and tabular loops: rn_layer_norm / se_layer_norm of their cfgs). */
    int32_t use_layer_norm;
} lenv_mlp_desc;

/* agents/DDQN.py:15-38, agents/base_agent.py:9-26, envs/env_factory.py:45-59 */
typedef struct {
    int32_t env_id;
    int32_t state_dim;
    int32_t num_actions;
    int32_t max_steps;
    int32_t se_hidden, se_layers, se_act;
    float se_prelu;
    int32_t q_hidden, q_layers, q_act;
    float q_prelu;
    int32_t batch_size, rb_size;
    int32_t train_episodes, test_episodes, init_episodes;
    int32_t early_out_num;
    int32_t grad_chunk;   /* samples per gradient micro-chunk; 0 = ceil(batch/16) */
    int32_t rng_mode;
    int32_t agent_kind;   /* 0 = DDQN (agents/DDQN.py), 1 = DuelingDDQN (agents/DuelingDDQN.py) */
    int32_t feature_dim;  /* DuelingDDQN: width of the feature vector / of the two head hidden layers */
    double solved_reward;
    double gamma, lr, tau;
    double eps_init, eps_min, eps_decay;
    double adam_beta1, adam_beta2, adam_eps;
    /* Deterministic stand-in for the wall-clock time-out of BaseAgent.train/test (agents/base_agent.py:30-47,90-97,177-184):
     * "elapsed" = env steps (train + test) this chain has taken.  <= 0: no budget.  At the start of a training episode with
     * elapsed > step_budget the per-episode reward list is padded with its minimum so far (-1e9 if empty) and training
     * stops; the final test gets the remainder and pads its returns the same way (so a chain that timed out in training
     * scores -1e9, as the reference does).  All four fused inner loops implement it. */
    int64_t step_budget;
    /* Intrinsic Curiosity Module inside learn() (select_agent "ddqn_icm" / "duelingddqn_icm": agents/DDQN.py:40-58,74-76,
     * models/icm_baseline.py:8-172; config section `icm`).  icm_enabled != 0: every learn step first trains the ICM on the
     * minibatch (one Adam step with icm_lr) and adds eta * mean_f (features(s') - forward(s, a))^2 to the minibatch rewards.
     * Only lenv_dueling_se_inner_loop_icm takes such a cfg (GEMM-tiled kernel; fresh ICM parameters per chain). */
    int32_t icm_enabled, icm_feature_dim, icm_hidden;
    /* `use_layer_norm` of the ENV's config section with se_layers >= 2 (models/model_utils.py:22-37): the shared nn.LayerNorm behind hidden
     * Linear 2..L of each of the three SE nets (synthetic_env_type 1: of the reward net).  NES perturbs and updates nn.Linear parameters only (agents/GTN_worker.py:156-175,
     * agents/GTN_master.py:281-296): the module keeps its initial affine (weight 1, bias 0) and theta stays the Linear parameters.
     * GEMM-tiled kernel only (lenv_dueling_se_inner_loop*); nothing to do with one hidden layer. */
    int32_t se_layer_norm;
    double icm_lr, icm_beta, icm_eta;
    /* gtn.synthetic_env_type: 0 = the agent trains on the VirtualEnv (theta = the three SE nets); 1 = on a RewardEnv over the
     * REAL env (envs/reward_env.py:61-133, default_config_cartpole_reward_env.yaml): real transitions, reward through the
     * reward network theta (state_dim -> se_hidden x se_layers -> 1, se_act; a 1-input dummy for type 0), reward_env_type 0, 1, 2, 5 or 6
     * (the real CartPole / Acrobot step carries no info vector).  GEMM-tiled kernel only (lenv_dueling_se_inner_loop*). */
    int32_t synthetic_env_type, reward_env_type;
    /* same_action_num (agents/base_agent.py:20,104,194; envs/env_wrapper.py:24-29,56-61): env steps per chosen action -- a VirtualEnv
     * repeats the step whatever the done flag says and sums the fp32 rewards, a real env's repeats stop at done (python-float sum);
     * 0 and 1 both mean 1.  Values > 1: GEMM-tiled kernel only (lenv_dueling_se_inner_loop*, plain-DQN mode for DDQN). */
    int32_t same_action_num;
    /* Workgroups per chain (a TEAM: the members split the minibatch / the gradient tiles of a learn step and meet at barriers; same
     * bits for every size).  0 = automatic: the largest team for which every workgroup of the launch is resident at once on an
     * otherwise idle device (checked with the occupancy API at launch); 1 = one workgroup per chain; G > 1 = at most G.  A launch
     * whose members cannot all be resident is never made with a team (the entry falls back to 1).  Team launches need the device to
     * themselves: a member that waits longer than ~0.25 s for the others (another kernel holds the CUs) gives up, every chain of the
     * launch reports status -10 within that time, and the caller repeats the launch with team_size 1 (engine.py does). */
    int32_t team_size;
    int32_t kernel_variant;   /* LENV_VARIANT_* bits, 0 = fastest */
    /* `use_layer_norm` of the agent's config section (models/model_utils.py:22-37): ONE shared nn.LayerNorm(hidden) (eps 1e-5) behind every
     * hidden Linear but the first of the Q-net (agent_kind 0) / of the DuelingDDQN's feature stream (its heads have one hidden layer: none
     * there); its weight | bias sit behind the second Linear in the parameter vector (Module.parameters() order), fresh agents start them at
     * 1 | 0.  Nothing to do with one hidden layer.  GEMM-tiled kernel (lenv_dueling_se_inner_loop*); forward and backward in the loop. */
    int32_t q_layer_norm;
    /* ABI 7.  test_mode = which BaseAgent.train call the loop is (agents/base_agent.py:64,134-148):
     *   0 = train(env, test_env=real_env), GTN_Worker.calc_score (agents/GTN_worker.py:195-199): test_episodes real-env test episodes after
     *       every training episode feed the meter; early-out = the real rule on it (mean of the last early_out_num entries >= solved_reward);
     *   1 = train(env, test_env=None), what every downstream evaluation calls (experiments/syn_env_evaluate_cartpole_vary_hp_2.py:38-41):
     *       NO per-episode tests; the meter is fed by the training env's own episode reward (the fp32 sum of the step rewards,
     *       base_agent.py:121,138); break_env = the training env (:141-146): a VirtualEnv stops on
     *       |avg - avg_last| / (|avg_last| + 1e-9) < early_out_virtual_diff once episode >= init_episodes + early_out_num
     *       (base_agent.py:49-56; AverageMeter.get_mean_last, utils.py:97-105), a RewardEnv / the real env (synthetic_env_type 1) on
     *       avg >= solved_reward.  out->episode_test_mean[] then holds the training episode rewards (the function's reward_train list);
     *       the final test and the score are unchanged.  Training on the REAL env (experiments/syn_env_run_vary_hp.py:47-54, mode 0) =
     *       synthetic_env_type 1 with reward_env_type 0 (envs/reward_env.py:80-81: the real reward passes through; theta is not read). */
    int32_t test_mode;
    double early_out_virtual_diff;
} lenv_ddqn_cfg;

/* RNG tapes (parity mode).  Per-chain rows: element [c*stride + n]; all DEVICE pointers. */
typedef struct {
    const double *eps_uniform;  int64_t eps_uniform_stride;   /* random.random()         agents/DDQN.py:98 */
    const int32_t *rand_action; int64_t rand_action_stride;   /* action_space.sample()   envs/env_wrapper.py:88 */
    const int32_t *replay_idx;  int64_t replay_idx_stride;    /* np.random.randint       utils.py:35 */
    const double *train_reset;  int64_t train_reset_stride;   /* rows of 4 doubles       envs/virtual_env.py:36 */
    const double *test_reset;   int64_t test_reset_stride;    /* rows of 4 doubles */
} lenv_tapes;

/* Outputs of the fused inner loop; every pointer may be NULL except `score`. */
typedef struct {
    double *score;              /* [chains] statistics.mean(final test returns)   GTN_worker.py:209 */
    int64_t *stats;             /* [chains,4] episodes_run, train_steps, learn_steps, test_steps */
    int32_t *status;            /* [chains] 0 ok, <0 internal error: -2..-5 tape underrun (eps / action / replay / reset), -6 replay index out of range, -7 unexpected LDS placement, -8 per-chain hyper-parameter outside cfg's maxima */
    double *episode_test_mean;  /* [chains,train_episodes]  reward_list_train (NaN past early-out) */
    int32_t *episode_len;       /* [chains,train_episodes] */
    double *final_returns;      /* [chains,test_episodes] */
    float *final_online;        /* [chains,P_agent] trained Q-net (canonical flat layout) */
    /* optional per-step trace, rows [chains,trace_cap] */
    int64_t trace_cap;
    int32_t *trace_action;      /* action | explored<<16 */
    float *trace_state;         /* [chains,trace_cap,S] */
    float *trace_next_state;    /* [chains,trace_cap,S] */
    float *trace_reward_done;   /* [chains,trace_cap,2] */
} lenv_inner_out;

int lenv_abi_version(void);
/* sizeof of the ABI structs as the library was compiled (a binding checks its mirror against it): which = 0 lenv_mlp_desc, 1 lenv_ddqn_cfg,
 * 2 lenv_ql_cfg, 3 lenv_td3_cfg, 4 lenv_td3d_cfg, 5 lenv_tapes, 6 lenv_inner_out, 7 lenv_ql_out, 8 lenv_td3_tapes, 9 lenv_td3_out,
 * 10 lenv_td3d_tapes, 11 lenv_chain_hp, 12 lenv_icm_io; anything else: LENV_ERR_INVALID.  HOST. */
int64_t lenv_struct_size(int32_t which);
const char *lenv_error_string(int code);
/* number of parameters of an MLP / of the three-net SE */
int64_t lenv_mlp_num_params(const lenv_mlp_desc *d /*HOST*/);

/*
 * Population-batched VirtualEnv step: replaces EnvWrapper.step -> VirtualEnv.step
 * (envs/env_wrapper.py:16-47, envs/virtual_env.py:43-54) for `chains` perturbed SEs at once,
 * W_c = theta + sign[c]*eps[worker[c]] (agents/GTN_worker.py:165-175; eps may be NULL).
 * state [chains,n_per_chain,S], action [chains,n_per_chain] (index), outputs same leading shape.
 */
int lenv_se_step_population(const lenv_mlp_desc *state_net /*HOST*/, const lenv_mlp_desc *reward_net /*HOST*/,
                            const lenv_mlp_desc *done_net /*HOST*/, const float *theta, const float *eps,
                            const int32_t *worker, const float *sign, int64_t chains, int32_t n_per_chain,
                            const float *state, const int32_t *action, float *next_state, float *reward,
                            float *done, void *stream);

/*
 * DDQN TD forward over replay minibatches (agents/DDQN.py:63-85): for every chain, gathers rows
 * idx[c,b] of its replay buffer (row = [s(S), a, s'(S), r, done], stride row_stride floats) and writes
 * q_sa[c,b] = Q(s)[a], y[c,b] = r + gamma * Q_target(s')[argmax Q(s')] * (1 - done).
 * online/target: [chains,P_agent] flat Critic_DQN parameters (models/actor_critic.py:84-91).
 */
int lenv_qnet_td_forward(const lenv_mlp_desc *qnet /*HOST*/, const float *online, const float *target,
                         const float *replay, int64_t replay_cap, int32_t row_stride, const int32_t *idx,
                         int64_t chains, int32_t batch, double gamma, float *q_sa, float *y, void *stream);

/*
 * Fused inner loop = GTN_Worker.calc_score (agents/GTN_worker.py:187-221) for `chains` independent
 * chains, one workgroup per chain: fresh DDQN agent (agent_init [chains,P_agent]), BaseAgent.train on the
 * perturbed SE with per-episode real-env tests and early-out (agents/base_agent.py:64-153), final
 * BaseAgent.test (agents/base_agent.py:155-227).  rng_keys [chains] (counter mode) / tapes (tape mode).
 * The replay buffers, the per-episode meter and the Adam bias-correction table (8 B per possible learn step, filled by a
 * prologue kernel on `stream`) live in `workspace` (lenv_ddqn_se_workspace_bytes): the entry allocates nothing.
 */
size_t lenv_ddqn_se_workspace_bytes(const lenv_ddqn_cfg *cfg /*HOST*/, int64_t chains);
/* LDS bytes one chain needs for this cfg (incl. grad_chunk), or a negative LENV_ERR_* when unsupported / > 160 KiB */
int64_t lenv_ddqn_se_lds_bytes(const lenv_ddqn_cfg *cfg /*HOST*/);
/* How the minibatch forward of this cfg is laid out (diagnostic, host only): *items = forward items (sample, pass) that spill
 * beyond the first eight waves of the workgroup, *parts = the number of pieces each of them is cut into over the hidden-unit pairs
 * (0 = plain layout: nothing spills, too much spills, the net is too narrow to cut, or the shared rows do not fit LDS). */
int lenv_ddqn_se_forward_split(const lenv_ddqn_cfg *cfg /*HOST*/, int32_t *items, int32_t *parts);
/* Workgroups per chain lenv_ddqn_se_inner_loop would use for a counter-mode launch of `chains` chains on the current device: > 1 when
 * the launch leaves enough CUs idle for every member of every chain to be resident (shards of a population spread over several
 * GPUs); the members deal the minibatch by whole gradient micro-chunks and meet once per learn step -- same bits for every team
 * size.  cfg->team_size caps / forces the size (1 = never).  Status -10 = a member gave up waiting (see team_size). */
int lenv_ddqn_se_team_size(const lenv_ddqn_cfg *cfg /*HOST*/, int64_t chains);
int lenv_ddqn_se_inner_loop(const lenv_ddqn_cfg *cfg /*HOST*/, const float *theta, const float *eps,
                            const int32_t *worker, const float *sign, const float *agent_init,
                            const uint64_t *rng_keys, const lenv_tapes *tapes /*HOST struct of device ptrs, may be NULL*/,
                            int64_t chains, void *workspace, size_t workspace_bytes,
                            const lenv_inner_out *out /*HOST struct of device ptrs*/, void *stream);

/*
 * Config 4: fused inner loop for the tabular agents (Q-learning, SARSA, and their count-based variants) on a potential-shaped
 * RewardEnv over a grid MDP (agents/QL.py:13-106, agents/SARSA.py:12-91, envs/reward_env.py:61-133, envs/gridworld.py:38-110), one wave per chain.
 * The MDP is given as transition tables next_state/reward/done [n_states, n_actions] (device pointers, shared by all
 * chains); theta = flat reward_net parameters (PReLU slope excluded), perturbed per chain as theta + sign*eps[worker].
 * shaped_override [n_states*n_actions] (optional) replaces the reward-net evaluation by a given shaped-reward table.
 */
typedef struct {
    int32_t n_states, n_actions, start_state, max_steps;
    int32_t rn_hidden, rn_layers, rn_act;
    float rn_prelu;
    int32_t reward_env_type;            /* 0,1,2,5,6 */
    int32_t train_episodes, test_episodes, init_episodes, early_out_num, batch_size;
    int32_t rng_mode;
    int32_t agent_kind;                 /* 0 QL (agents/QL.py:37-75), 1 SARSA (agents/SARSA.py:36-60: bootstrap on a freshly drawn
                                           eps-greedy next action) */
    int32_t count_based;                /* ql_cb / sarsa_cb (agents/agent_utils.py:57-64): reward += beta / (sqrt(n(s,a)) + 1e-9) */
    double solved_reward, alpha, gamma, eps_init, eps_min, eps_decay, beta;
    int64_t step_budget;                /* env-step stand-in for time_remaining, see lenv_ddqn_cfg::step_budget */
    int32_t same_action_num;            /* env steps per chosen action (base_agent.py:104,194; env_wrapper.py:56-61: stop at done, python-float
                                           reward sum); 0 and 1 both mean 1 */
    int32_t rn_layer_norm;              /* the ENV section's `use_layer_norm` with rn_layers >= 2, as lenv_ddqn_cfg::se_layer_norm (ABI 6; was padding) */
    int32_t test_mode;                  /* ABI 7: as lenv_ddqn_cfg::test_mode (1 = BaseAgent.train without a test env) */
    double early_out_virtual_diff;      /* never read: a grid RewardEnv is not a VirtualEnv (the real rule applies) */
} lenv_ql_cfg;

typedef struct {
    double *score;              /* [chains] */
    int64_t *stats;             /* [chains,4] episodes_run, train_steps, learn_steps, test_steps */
    int32_t *status;            /* [chains] */
    double *episode_test_mean;  /* [chains,train_episodes] */
    int32_t *episode_len;       /* [chains,train_episodes] */
    double *final_returns;      /* [chains,test_episodes] */
    double *q_table;            /* [chains,n_states*n_actions] final fp64 Q-table */
    float *shaped;              /* [chains,n_states*n_actions] shaped reward of every (s,a) */
    int64_t trace_cap;
    int32_t *trace_action;      /* [chains,trace_cap] action | explored<<16 */
    int32_t *trace_state;       /* [chains,trace_cap,2] state, next_state */
    float *trace_reward_done;   /* [chains,trace_cap,2] */
} lenv_ql_out;

int lenv_ql_rn_inner_loop(const lenv_ql_cfg *cfg /*HOST*/, const float *theta, const float *eps, const int32_t *worker,
                          const float *sign, const float *shaped_override, const int32_t *next_state, const double *reward,
                          const uint8_t *done, const uint64_t *rng_keys, const lenv_tapes *tapes /*HOST, may be NULL*/,
                          int64_t chains, const lenv_ql_out *out /*HOST struct of device ptrs*/, void *stream);

/*
 * RewardEnv shaping for a population of perturbed reward networks on a grid MDP (envs/reward_env.py:67-133): phi_out
 * [chains,n_states] (optional) = reward_net(one_hot(s)), shaped_out [chains,n_states*n_actions] = what RewardEnv.step
 * returns as reward for (s,a).  Only n_states/n_actions/rn_* /reward_env_type/gamma of cfg are read.
 */
int lenv_rn_shape_population(const lenv_ql_cfg *cfg /*HOST*/, const float *theta, const float *eps, const int32_t *worker,
                             const float *sign, int64_t chains, const int32_t *next_state, const double *reward,
                             float *phi_out, float *shaped_out, void *stream);

/*
 * Real-environment reset/step for n independent instances (replaces gym==0.17.3 CartPole-v0 / Acrobot-v1
 * reset()/step() + gym.wrappers.TimeLimit behind EnvWrapper.reset/step, envs/env_wrapper.py:49-85).
 * state [n,4] float64 (gym's internal state), elapsed [n] TimeLimit counters, obs [n,S] fp32 observations.
 * Reset states are U(-lim,lim)^4 drawn from the counter RNG: value = f(keys[i], episode[i]).
 */
int lenv_real_env_reset(int32_t env_id, const uint64_t *keys, const int64_t *episode, int64_t n, double *state,
                        float *obs, int32_t *elapsed, void *stream);
int lenv_real_env_step(int32_t env_id, int32_t max_steps, int64_t n, const int32_t *action, double *state,
                       int32_t *elapsed, float *obs, float *reward, float *done, void *stream);

/*
 * Fused inner loop for DuelingDDQN agents on a synthetic environment (cfg.agent_kind == 1; agents/DuelingDDQN.py:59-110,
 * models/actor_critic.py:94-122; BASELINE config 3).  Same contract and outputs as lenv_ddqn_se_inner_loop; the agent's
 * parameters / Adam state / activations live in the workspace (lenv_dueling_se_workspace_bytes), agent_init is
 * [chains, lenv_dueling_num_params(cfg)] in state-dict order feature_stream | value_stream | advantage_stream.
 * cfg.agent_kind == 0 selects the plain-DQN mode: a DDQN (agents/DDQN.py:60-94) whose Critic_DQN (models/actor_critic.py:
 * 84-91) has q_layers 1-2 hidden layers of up to 128 units -- the shapes lenv_ddqn_se_inner_loop refuses, e.g. the
 * 6-128-128-3 net of default_config_acrobot.yaml; cfg.feature_dim is ignored, cfg.grad_chunk must be 0 (one sequential
 * batch gradient), agent_init is [chains, lenv_dueling_num_params(cfg)] in the Critic_DQN state-dict order.
 */
/*
 * Per-chain hyper-parameters of the *_vary agents (agents/DDQN_vary.py:26-59, agents/DuelingDDQN_vary.py:24-69: every agent
 * the worker builds draws its own lr / batch_size / hidden_size / hidden_layer).  Device arrays [chains]; with them the
 * cfg carries the MAXIMA (batch_size, q_hidden, q_layers: workspace, LDS and the row stride of agent_init / final_online
 * are sized from cfg) and chain c runs with the c-th entries -- its agent_init row holds lenv_dueling_num_params at ITS
 * shapes, the rest of the row is ignored.  q_layers = max(1, hidden_layer) as in lenv_ddqn_cfg.  A chain whose values
 * exceed cfg's reports status -8.
 */
typedef struct lenv_chain_hp {
    const double *lr;
    const int32_t *batch_size, *q_hidden, *q_layers;
} lenv_chain_hp;

size_t lenv_dueling_se_workspace_bytes(const lenv_ddqn_cfg *cfg /*HOST*/, int64_t chains);
int64_t lenv_dueling_num_params(const lenv_ddqn_cfg *cfg /*HOST*/);
int lenv_dueling_se_inner_loop(const lenv_ddqn_cfg *cfg /*HOST*/, const float *theta, const float *eps,
                               const int32_t *worker, const float *sign, const float *agent_init,
                               const uint64_t *rng_keys, const lenv_tapes *tapes /*HOST, may be NULL*/, int64_t chains,
                               void *workspace, size_t workspace_bytes, const lenv_inner_out *out /*HOST*/, void *stream);
/* Workgroups per chain a production launch (counter RNG, no trace, no hp, no ICM) of this cfg will use: the wave-chain kernel of the
 * published Acrobot SE + DuelingDDQN shape runs a chain on a TEAM of 2 workgroups when 8 * ceil(chains / 8) * 2 of them are resident at
 * once (one per CU); every other launch: 1.  cfg->team_size 1 forces 1.  Same bits either way. */
int lenv_dueling_team_size(const lenv_ddqn_cfg *cfg /*HOST*/, int64_t chains);
/* Fresh agents (nn.Linear default init, the draw of lenv_nes_draw) for chains with their own shapes: row c of agent_init
 * [chains, lenv_dueling_num_params(cfg)] gets the parameters of a (hp->q_hidden[c], hp->q_layers[c]) network, keyed by
 * rng_keys[c].  hp == NULL: every chain has cfg's shapes (== lenv_nes_draw's agent_init for the same keys). */
int lenv_dueling_agent_init_hp(const lenv_ddqn_cfg *cfg /*HOST*/, const lenv_chain_hp *hp /*HOST struct of device arrays*/,
                               const uint64_t *rng_keys, int64_t chains, float *agent_init, void *stream);
/* ICM agents: icm_init [chains, lenv_icm_num_params(cfg)] = fresh ICMModel parameters per chain in state-dict order
 * (features_model, inverse_model, forward_pre_model, residual_block1..4 {fc1, fc2}, forward_post_model); icm_final (may be
 * NULL) receives them after the last learn step.  hp != NULL: the *_icm_vary agents (per-chain hyper-parameters of the agent;
 * the ICM keeps cfg's shapes and learning rate).  cfg->icm_enabled == 0: identical to lenv_dueling_se_inner_loop_hp. */
typedef struct lenv_icm_io {
    const float *icm_init;
    float *icm_final;
} lenv_icm_io;
int64_t lenv_icm_num_params(const lenv_ddqn_cfg *cfg /*HOST*/);
int lenv_dueling_se_inner_loop_icm(const lenv_ddqn_cfg *cfg /*HOST*/, const lenv_chain_hp *hp /*HOST, may be NULL*/,
                                   const lenv_icm_io *icm /*HOST struct of device arrays*/, const float *theta, const float *eps,
                                   const int32_t *worker, const float *sign, const float *agent_init, const uint64_t *rng_keys,
                                   const lenv_tapes *tapes /*HOST*/, int64_t chains, void *workspace, size_t workspace_bytes,
                                   const lenv_inner_out *out /*HOST*/, void *stream);
/* the same with per-chain hyper-parameters (hp == NULL: identical to lenv_dueling_se_inner_loop) */
int lenv_dueling_se_inner_loop_hp(const lenv_ddqn_cfg *cfg /*HOST*/, const lenv_chain_hp *hp /*HOST struct of device arrays*/,
                                  const float *theta, const float *eps, const int32_t *worker, const float *sign,
                                  const float *agent_init, const uint64_t *rng_keys, const lenv_tapes *tapes /*HOST*/,
                                  int64_t chains, void *workspace, size_t workspace_bytes, const lenv_inner_out *out /*HOST*/,
                                  void *stream);

/*
 * Config 5: fused inner loop for a TD3 agent on a RewardEnv over a continuous-state real env (agents/TD3.py:63-135,
 * envs/reward_env.py:61-133).  Real env = the documented HalfCheetah-v3 STAND-IN (17 obs / 6 actions, see
 * tools/gen_cheetah_standin.py; MuJoCo is unavailable).  theta = flat reward_net parameters (17 -> H -> 1, PReLU slope
 * excluded); agent_init [chains, P] = actor | critic_1 | critic_2 in state-dict order (lenv_td3_num_params).
 */
typedef struct {
    int32_t env_id, state_dim, action_dim, max_steps;
    int32_t rn_hidden, rn_layers, rn_act;
    float rn_prelu;
    int32_t reward_env_type;                 /* 0-8, 101, 102 (envs/reward_env.py:29-59) */
    int32_t info_dim;                        /* length of the real env's info vector (4 for the stand-in; used by types 3,4,7,8,101,102) */
    int32_t hidden, layers, act;             /* actor / critic MLPs (models/actor_critic.py:11-19,64-71) */
    float prelu;
    int32_t batch_size, rb_size, train_episodes, test_episodes, init_episodes, early_out_num, policy_delay, rng_mode;
    double solved_reward, gamma, lr, tau, action_std, policy_std, policy_std_clip, max_action;
    double adam_beta1, adam_beta2, adam_eps;
    int64_t step_budget;                     /* env-step stand-in for time_remaining, see lenv_ddqn_cfg::step_budget */
    /* TD3(icm=True), select_agent "td3_icm" (agents/TD3.py:44-60,68-70): as lenv_ddqn_cfg's icm_* fields; continuous actions,
     * so the action vector is the ICM's input and its inverse loss is an MSE.  Only lenv_td3_rn_inner_loop_icm takes it. */
    int32_t icm_enabled, icm_feature_dim, icm_hidden;
    /* `use_layer_norm` of the td3 section (models/model_utils.py:22-37): ONE shared nn.LayerNorm(hidden) behind every hidden Linear but the
     * first of the actor and of each critic (three modules, one per net), weight | bias behind the net's second Linear */
    int32_t use_layer_norm;
    double icm_lr, icm_beta, icm_eta;
    /* virtual_env != 0 (gtn.synthetic_env_type 0, default_config_halfcheetah.yaml): the agent trains on a VirtualEnv
     * (envs/virtual_env.py:43-54) instead of the RewardEnv: theta = state_net | reward_net | done_net, each
     * (action_dim + state_dim) -> rn_hidden x rn_layers (1-3) -> {state_dim, 1, 1} with rn_act; the learned done flag (> 0.5)
     * ends a training episode.  Tests run on the real (stand-in) env as before. */
    int32_t virtual_env;
    /* same_action_num (agents/base_agent.py:20,104,194; envs/env_wrapper.py:24,57): env steps per chosen action -- the rewards of
     * the repeats are summed, a real env's repeats stop at done; 0 and 1 both mean 1 (MountainCarContinuous configs ship 2) */
    int32_t same_action_num;
    int32_t team_size;        /* workgroups per chain, as lenv_ddqn_cfg::team_size (0 = automatic, 1 = never a team, G = at most G) */
    int32_t kernel_variant;   /* LENV_VARIANT_* bits, 0 = fastest */
    /* ABI 6.  `use_layer_norm` of the ENV's config section with rn_layers >= 2: the reward net (RewardEnv) / the three SE nets (VirtualEnv)
     * carry the shared nn.LayerNorm behind hidden Linear 2..L.  As lenv_ddqn_cfg::se_layer_norm: NES perturbs nn.Linear modules only, theta / eps
     * stay the Linear parameters, the kernel normalises with the constructor's weight 1 / bias 0. */
    int32_t rn_layer_norm;
    int32_t test_mode;                  /* ABI 7: as lenv_ddqn_cfg::test_mode (1 = BaseAgent.train without a test env) */
    double early_out_virtual_diff;
} lenv_td3_cfg;

/* RNG tapes (parity mode); per-chain rows, strides in ROWS (rows of A floats / B ints / S doubles as noted) */
typedef struct {
    const float *rand_action;  int64_t rand_action_stride;    /* rows of A: env.get_random_action()      env_wrapper.py:87-90 */
    const float *act_noise;    int64_t act_noise_stride;      /* rows of A: randn in select_train_action TD3.py:123 */
    const float *test_noise;   int64_t test_noise_stride;     /* rows of A: randn in select_test_action  TD3.py:128, one row per
                                                                 * chosen action in the reference's order (episode by episode) */
    const float *policy_noise; int64_t policy_noise_stride;   /* rows of A: randn_like(actions) in learn TD3.py:75 */
    const int32_t *replay_idx; int64_t replay_idx_stride;     /* elements */
    const double *train_reset; int64_t train_reset_stride;    /* rows of the env's own state: 17 (stand-in), 2 (Pendulum: theta, */
    const double *test_reset;  int64_t test_reset_stride;     /* theta_dot; MountainCarContinuous: position, velocity) */
} lenv_td3_tapes;

typedef struct {
    double *score;              /* [chains] */
    int64_t *stats;             /* [chains,4] episodes_run, train_steps, learn_steps, test_steps */
    int32_t *status;            /* [chains] 0 ok; -3..-8 as lenv_inner_out::status (-7, -8 here also: noise / policy-noise tape underrun); -9 Gumbel tape underrun (lenv_td3d_inner_loop) */
    double *episode_test_mean;  /* [chains,train_episodes] */
    int32_t *episode_len;       /* [chains,train_episodes] */
    double *final_returns;      /* [chains,test_episodes] */
    float *final_params;        /* [chains,P] */
    int64_t trace_cap;
    float *trace_action;        /* [chains,trace_cap,A] */
    float *trace_state;         /* [chains,trace_cap,S] */
    float *trace_next_state;    /* [chains,trace_cap,S] */
    float *trace_reward;        /* [chains,trace_cap] shaped reward */
} lenv_td3_out;

size_t lenv_td3_rn_workspace_bytes(const lenv_td3_cfg *cfg /*HOST*/, int64_t chains);
int64_t lenv_td3_num_params(const lenv_td3_cfg *cfg /*HOST*/, int64_t *actor_params /*HOST out*/, int64_t *critic_params /*HOST out*/);
int lenv_td3_rn_inner_loop(const lenv_td3_cfg *cfg /*HOST*/, const float *theta, const float *eps, const int32_t *worker,
                           const float *sign, const float *agent_init, const uint64_t *rng_keys,
                           const lenv_td3_tapes *tapes /*HOST, may be NULL*/, int64_t chains, void *workspace,
                           size_t workspace_bytes, const lenv_td3_out *out /*HOST*/, void *stream);
/* TD3_vary (agents/TD3_vary.py:24-58): per-chain lr / batch_size / hidden_size / hidden_layer through the lenv_chain_hp arrays: q_hidden ->
 * cfg.hidden, q_layers -> cfg.layers; cfg carries the maxima, agent_init rows hold actor | critic_1 | critic_2 at the chain's
 * own shapes (row stride lenv_td3_num_params(cfg)).  hp == NULL: identical to lenv_td3_rn_inner_loop. */
int lenv_td3_rn_inner_loop_hp(const lenv_td3_cfg *cfg /*HOST*/, const lenv_chain_hp *hp /*HOST struct of device arrays*/,
                              const float *theta, const float *eps, const int32_t *worker, const float *sign,
                              const float *agent_init, const uint64_t *rng_keys, const lenv_td3_tapes *tapes /*HOST*/,
                              int64_t chains, void *workspace, size_t workspace_bytes, const lenv_td3_out *out /*HOST*/, void *stream);
/* TD3 with an ICM: icm as in lenv_dueling_se_inner_loop_icm (icm_init rows of lenv_td3_icm_num_params(cfg) floats);
 * hp may be NULL (td3_icm) or the per-chain hyper-parameters (td3_icm_vary) */
int64_t lenv_td3_icm_num_params(const lenv_td3_cfg *cfg /*HOST*/);
int lenv_td3_rn_inner_loop_icm(const lenv_td3_cfg *cfg /*HOST*/, const lenv_chain_hp *hp /*HOST, may be NULL*/,
                               const lenv_icm_io *icm /*HOST struct of device arrays*/, const float *theta, const float *eps,
                               const int32_t *worker, const float *sign, const float *agent_init, const uint64_t *rng_keys,
                               const lenv_td3_tapes *tapes /*HOST*/, int64_t chains, void *workspace, size_t workspace_bytes,
                               const lenv_td3_out *out /*HOST*/, void *stream);
/* Workgroups per chain a production launch (counter RNG, no trace, no hp, no ICM) of this cfg will use: the wave-chain kernel of the
 * published HalfCheetah RewardEnv + TD3 shape runs a chain on a TEAM of 6, 3 or 2 workgroups when 8 * ceil(chains / 8) * G of them
 * are resident at once (one per CU); every other launch: 1.  cfg->team_size caps it.  Same bits for every G. */
int lenv_td3_rn_team_size(const lenv_td3_cfg *cfg /*HOST*/, int64_t chains);
int lenv_td3_agent_init_hp(const lenv_td3_cfg *cfg /*HOST*/, const lenv_chain_hp *hp, const uint64_t *rng_keys, int64_t chains,
                           float *agent_init, void *stream);

/*
 * TD3_discrete_vary: fused inner loop for TD3 on a DISCRETE action space through a Gumbel-softmax actor, trained on a VirtualEnv and
 * tested on the real env (agents/TD3_discrete_vary.py:16-117,159-177; models/actor_critic.py:22-35 Actor_TD3_discrete;
 * agents/base_agent.py:64-227 with discretize_action: the replay buffer keeps the action VECTOR, the env gets its argmax;
 * envs/env_wrapper.py:16-47 one-hot of that index in front of the three SE nets, envs/virtual_env.py:43-54).
 * use_layer_norm (models/model_utils.py:22-37): ONE nn.LayerNorm(hidden) shared by the hidden Linear layers 2..L of a net, in front
 * of the activation.  Flat parameters of a net in Module.parameters() order: W0 b0 [W1 b1 [LNw LNb] W2 b2 ...] Wout bout;
 * agent_init / final_params rows = actor | critic_1 | critic_2.  theta = state_net | reward_net | done_net as in lenv_ddqn_cfg.
 */
typedef struct {
    int32_t env_id, state_dim, action_dim, max_steps;   /* LENV_ENV_CARTPOLE 4 / 2, LENV_ENV_ACROBOT 6 / 3, LENV_ENV_MOUNTAINCAR 2 / 3 */
    int32_t se_hidden, se_layers, se_act;
    float se_prelu;
    int32_t hidden, layers, act;                         /* actor S -> A and critics (S + A) -> 1 (models/model_utils.py:4-39) */
    float prelu;
    int32_t use_layer_norm;
    int32_t gumbel_hard;                                 /* gumbel_softmax_hard: straight-through one-hot forward (actor_critic.py:31,35) */
    int32_t batch_size, rb_size, train_episodes, test_episodes, init_episodes, early_out_num, policy_delay, rng_mode;
    double solved_reward, gamma, lr, tau, action_std, policy_std, policy_std_clip, max_action;
    double gumbel_temp;                                  /* gumbel_softmax_temp, annealed to 1/20 of it over the first 2000 learn calls (:59-68) */
    double adam_beta1, adam_beta2, adam_eps;
    int64_t step_budget;                                 /* env-step stand-in for time_remaining, see lenv_ddqn_cfg::step_budget */
    int32_t se_layer_norm;                               /* ABI 6: the ENV section's `use_layer_norm`, as lenv_ddqn_cfg::se_layer_norm */
    int32_t test_mode;                  /* ABI 7: as lenv_ddqn_cfg::test_mode (1 = BaseAgent.train without a test env) */
    double early_out_virtual_diff;
} lenv_td3d_cfg;

/* RNG tapes (parity mode); per-chain rows, strides in ROWS (rows of A floats / one int / four doubles as noted) */
typedef struct {
    const int32_t *rand_action; int64_t rand_action_stride;    /* Discrete.sample() of the init episodes (env_wrapper.py:87-92) */
    const float *act_noise;     int64_t act_noise_stride;      /* rows of A: randn(action_dim) in select_train_action :167 */
    const float *test_noise;    int64_t test_noise_stride;     /* rows of A: the same in select_test_action :171 */
    const float *policy_noise;  int64_t policy_noise_stride;   /* rows of A, B per learn call: randn_like(actions) :76 */
    const float *gumbel_act;    int64_t gumbel_act_stride;     /* rows of A: Gumbel(0,1) draws of F.gumbel_softmax in select_train_action */
    const float *gumbel_test;   int64_t gumbel_test_stride;    /* rows of A: ... in select_test_action */
    const float *gumbel_target; int64_t gumbel_target_stride;  /* rows of A, B per learn call: actor_target(next_states) :77 */
    const float *gumbel_actor;  int64_t gumbel_actor_stride;   /* rows of A, B per policy update: actor(states) :101 */
    const int32_t *replay_idx;  int64_t replay_idx_stride;     /* elements, B per learn call */
    const double *train_reset;  int64_t train_reset_stride;    /* rows of 4: the reset env's own state */
    const double *test_reset;   int64_t test_reset_stride;
} lenv_td3d_tapes;

size_t lenv_td3d_workspace_bytes(const lenv_td3d_cfg *cfg /*HOST*/, int64_t chains);
int64_t lenv_td3d_num_params(const lenv_td3d_cfg *cfg /*HOST*/, int64_t *actor_params /*HOST out*/, int64_t *critic_params /*HOST out*/);
int64_t lenv_td3d_se_num_params(const lenv_td3d_cfg *cfg /*HOST*/);
/* hp (may be NULL): per-chain lr / batch_size / hidden_size / hidden_layer of vary_hyperparameters (:119-157); cfg carries the
 * maxima, agent_init rows hold the nets at the chain's own shapes (row stride lenv_td3d_num_params(cfg)).  out: lenv_td3_out
 * (trace_action rows hold the action vectors the replay buffer got). */
int lenv_td3d_inner_loop(const lenv_td3d_cfg *cfg /*HOST*/, const lenv_chain_hp *hp /*HOST struct of device arrays, may be NULL*/,
                         const float *theta, const float *eps, const int32_t *worker, const float *sign, const float *agent_init,
                         const uint64_t *rng_keys, const lenv_td3d_tapes *tapes /*HOST, may be NULL*/, int64_t chains, void *workspace,
                         size_t workspace_bytes, const lenv_td3_out *out /*HOST*/, void *stream);
/* fresh agents: nn.Linear default init from the chain key's counter stream, LayerNorm weight 1 / bias 0 */
int lenv_td3d_agent_init(const lenv_td3d_cfg *cfg /*HOST*/, const lenv_chain_hp *hp /*may be NULL*/, const uint64_t *rng_keys,
                         int64_t chains, float *agent_init, void *stream);

/*
 * Batched forward of one MLP in the flat layout above: y [rows,out] = net(x [rows,in]) (models/model_utils.py:31-39;
 * behind Critic_DQN / Actor_TD3.net / Critic_Q / reward_net calls of the one-step API).
 */
int lenv_mlp_forward(const lenv_mlp_desc *d /*HOST*/, const float *params, const float *x, int64_t rows, float *y, void *stream);

/*
 * RewardEnv on a vector-state real env, one-step API (envs/reward_env.py:29-133).
 * lenv_rn_num_params: parameters of build_reward_net for a reward type -- MLP on state_dim (types 1,2,5,6) or
 * state_dim+info_dim (3,4,7,8) inputs, Linear(info_dim,1,bias=False) for 101/102, 0 for type 0; LENV_ERR_UNSUPPORTED for an
 * unknown type (the reference raises NotImplementedError).  HOST.
 * lenv_rn_shape_rows: RewardEnv._calc_reward for `rows` transitions: s, s2 [rows,state_dim], info [rows,info_dim] (the
 * real env's info values in dict order, fp32; may be NULL for the types that do not read it -- otherwise a missing info
 * is LENV_ERR_INVALID, the reference's ValueError), r [rows] the real reward rounded to fp32; out [rows] fp32.
 */
int64_t lenv_rn_num_params(int32_t type, int32_t state_dim, int32_t info_dim, int32_t hidden, int32_t layers);
int lenv_rn_shape_rows(int32_t type, const lenv_mlp_desc *rn /*HOST, types 1-8*/, int32_t state_dim, int32_t info_dim, double gamma,
                       const float *theta, const float *s, const float *s2, const float *info, const float *r, int64_t rows,
                       float *out, void *stream);

/* HalfCheetah-v3 STAND-IN reset / step for n instances (state [n,17] float64, action [n,6], obs [n,17]); same contract as
 * lenv_real_env_reset / lenv_real_env_step. */
int lenv_cheetah_standin_reset(const uint64_t *keys, const int64_t *episode, int64_t n, double *state, float *obs,
                               int32_t *elapsed, void *stream);
int lenv_cheetah_standin_step(int32_t max_steps, int64_t n, const float *action, double *state, int32_t *elapsed,
                              float *obs, float *reward, float *done, void *stream);

/* The continuous real envs of the TD3 path by id (LENV_ENV_CHEETAH_STANDIN: state [n,17], action [n,6], obs [n,17];
 * LENV_ENV_PENDULUM = gym 0.17.3 Pendulum-v0: state [n,2] = (theta, theta_dot), action [n,1], obs [n,3]; LENV_ENV_CMC =
 * MountainCarContinuous-v0: state [n,2] = (position, velocity), action [n,1], obs [n,2], done at the flag); replaces
 * gym.make(...).reset / .step + TimeLimit behind EnvWrapper (envs/env_wrapper.py:58-75).  Other ids: LENV_ERR_UNSUPPORTED. */
int lenv_cont_env_reset(int32_t env_id, const uint64_t *keys, const int64_t *episode, int64_t n, double *state, float *obs,
                        int32_t *elapsed, void *stream);
int lenv_cont_env_step(int32_t env_id, int32_t max_steps, int64_t n, const float *action, double *state, int32_t *elapsed,
                       float *obs, float *reward, float *done, void *stream);

/* Counter-RNG key of a chain (same function as the oracle's): kind 0 = theta, 1 = theta+eps, 2 = theta-eps. HOST. */
/* out[c][i] = (2u - 1) * bounds[i], u = unit(rng(rng_keys[c], rng_stream, i)), i < p: freshly initialised parameter vectors
 * (nn.Linear default init with bounds[i] = 1/sqrt(fan_in)) keyed by the chains' counter-RNG keys.  rng_stream 10 reproduces
 * lenv_nes_draw's agent_init; ICM modules (lenv_dueling_se_inner_loop_icm) are drawn from stream 12. */
int lenv_chain_uniform_init(const uint64_t *rng_keys, int64_t chains, uint32_t rng_stream, int64_t p, const float *bounds, float *out,
                            void *stream);
/* one uniform in [0,1) of a chain's counter RNG (HOST function, no device work): unit(rng(key, stream, index)) */
double lenv_rng_unit(uint64_t key, uint32_t stream, uint64_t index);
uint64_t lenv_chain_key(uint64_t seed, uint64_t generation, uint64_t worker, uint64_t kind);

/*
 * GTN_Worker.calc_best_score (agents/GTN_worker.py:234-254) for `pop` workers from the per-chain scores
 * laid out [pop,3] = (orig, add, sub): result[p] = {score_best, score_orig, sign, 0} (double).
 */
int lenv_nes_worker_best(const double *chain_scores, int64_t pop, int32_t mirrored, double *result, void *stream);
/* the same for num_grad_evals = G evaluations per direction (agents/GTN_worker.py:90-104,234-242): chain_scores [pop,1+2G] =
 * (orig, add_1..add_G, sub_1..sub_G); grad_eval_type 0 = 'mean' (statistics.mean: the exactly rounded mean), 1 = 'minmax'
 * (min of both lists, as the reference computes it); anything else LENV_ERR_UNSUPPORTED. */
int lenv_nes_worker_best_multi(const double *chain_scores, int64_t pop, int32_t num_grad_evals, int32_t mirrored,
                               int32_t grad_eval_type, double *result, void *stream);

/*
 * One NES generation's stochastic inputs in a single launch, all from the counter RNG (reproduced by every rank and by the
 * CPU oracle): eps [pop, p_theta] = N(0,1) * noise_std (GTN_Worker.get_random_noise, agents/GTN_worker.py:156-163);
 * agent_init [chains, p_agent] = U(-bound_i, bound_i) (the nn.Linear default init of the fresh agent every calc_score builds,
 * agents/agent_utils.py:15-66); rng_keys [chains] = lenv_chain_key(seed, generation, worker_lo + c / chains_per_worker,
 * c % chains_per_worker).  Any of the three outputs may be NULL.
 */
int lenv_nes_draw(uint64_t seed, uint64_t generation, int64_t pop, int64_t p_theta, float noise_std, float *eps,
                  int64_t chains, int32_t chains_per_worker, int64_t worker_lo, int64_t p_agent, const float *bounds,
                  float *agent_init, uint64_t *rng_keys, void *stream);

/* The same with the generation read from device memory (generation_dev[0]) when the kernel runs: a captured generation
 * (HIP graph) is replayed for generation after generation without touching its kernel arguments; lenv_nes_rank_update_keep
 * advances the counter. */
int lenv_nes_draw_dev(uint64_t seed, const int64_t *generation_dev, int64_t pop, int64_t p_theta, float noise_std, float *eps,
                      int64_t chains, int32_t chains_per_worker, int64_t worker_lo, int64_t p_agent, const float *bounds,
                      float *agent_init, uint64_t *rng_keys, void *stream);

/* result[w][3] = min(status[0..n)) for w < pop: this rank's worst chain status rides in the fitness records through the
 * all-gather (replaces the second host read-back of a generation). */
int lenv_nes_status_fold(const int32_t *status, int64_t n, double *result, int64_t pop, void *stream);

/*
 * GTN_Master.score_transform + update_env (agents/GTN_master.py:197-298) on device.
 * gathered [pop,4] as produced by lenv_nes_worker_best (after the all-gather); rank_table [pop] doubles =
 * weight by rank for the rank-only transforms 1,2,3 (host-computed with the reference's numpy formulas);
 * weights_out [pop] receives score_transform_list; theta [P] is updated in place:
 * theta <- theta*(1-wd); for i in worker order: theta += (ss*w_i) * (sign_i*eps_i).
 */
int lenv_nes_rank_update(int32_t score_transform_type, const double *gathered, const double *rank_table, int64_t pop,
                         float *theta, const float *eps, int64_t p_theta, double step_size, int32_t nes_step_size,
                         double weight_decay, double *weights_out, void *stream);

/* lenv_nes_rank_update for a generation that runs as ONE captured graph (draw -> fused inner loop -> worker_best -> status_fold
 * -> rank update) with the host read-back behind it: theta_prev [P] (may be NULL) receives theta as it was before the update --
 * what GTN_Master.save_good_model saves and what stays when the run quits on "solved" (agents/GTN_master.py:95-101,118-131:
 * the reference decides both BEFORE update_env) -- and generation_dev[0] (may be NULL) is incremented for the next replay. */
int lenv_nes_rank_update_keep(int32_t score_transform_type, const double *gathered, const double *rank_table, int64_t pop,
                              float *theta, const float *eps, int64_t p_theta, double step_size, int32_t nes_step_size,
                              double weight_decay, double *weights_out, float *theta_prev, int64_t *generation_dev, void *stream);

#ifdef __cplusplus
}
#endif
#endif
