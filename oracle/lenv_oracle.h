/*
 * lenv_oracle.h -- CPU ORACLE for the NES inner-loop hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the algorithm the
 * reference (automl/learning_environments, /root/reference) runs on its NES
 * inner loop: GTN_Worker.calc_score -> BaseAgent.train/test -> EnvWrapper.step /
 * DDQN.learn, plus the GTN_Master aggregation.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product (learning_environments_amd)
 * never does.
 *
 * Parity status: PINNED for everything defined by /root/reference + torch
 * (SE/RN forward, Q-nets, DDQN.learn, NES noise/mirroring, score_transform,
 * update_env) against golden vectors produced by importing the reference
 * (oracle/gen_golden.py -> tests/golden/).  UNPINNED for the third-party
 * real-env physics (gym==0.17.3 CartPole-v0 / Acrobot-v1, requirements.txt:47),
 * whose source is not under /root/reference: restated from the published
 * equations.
 *
 * Canonical floating-point order (what "bit-exact vs the oracle" means for the
 * HIP kernels; the reference's own order is MKL/oneDNN-defined):
 *   - every dot product is a sequential fp32 fmaf chain in index order starting
 *     from 0.0f, bias added last with a plain add (this IS torch-CPU's batched
 *     `linear` order for K <= 128, verified bitwise);
 *   - batch-gradient sums are accumulated in micro-chunks of `grad_chunk`
 *     samples (sequential fmaf inside a chunk, chunk partials added in order);
 *   - tanh is the table-driven piecewise cubic of lenv_tanh_table.h, sin / cos are the polynomial routines in
 *     lenv_oracle.c (no libm);
 *   - element-wise update formulas follow torch 2.10's CPU kernels op by op
 *     (lerp = fma, addcmul = fma, addcdiv = plain, tanh' = dh*fma(-h,h,1)).
 */
#ifndef LENV_ORACLE_H
#define LENV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_ACT_IDENTITY = 0, ORC_ACT_RELU = 1, ORC_ACT_LEAKYRELU = 2, ORC_ACT_TANH = 3, ORC_ACT_PRELU = 4 };
enum { ORC_ENV_CARTPOLE = 0, ORC_ENV_ACROBOT = 1, ORC_ENV_MOUNTAINCAR = 3 };   /* 2 = the HalfCheetah stand-in (TD3) */
enum { ORC_RNG_COUNTER = 0, ORC_RNG_TAPE = 1 };

/* models/model_utils.py:4-39 -- Linear(in,H) act [Linear(H,H) act]x(L-1) Linear(H,out) */
typedef struct {
    int32_t in_dim, hidden, layers, out_dim, act;
    float prelu; /* single shared PReLU slope (init 0.25, never perturbed) */
    /* model_utils.py:22-37 `use_layer_norm`: ONE shared nn.LayerNorm(hidden) after every hidden Linear but the first (before the
     * activation); its weight/bias [hidden] sit right after the second Linear in the flat vector (Module.parameters() order).
     * Only orc_mlp_forward / orc_mlp_num_params read it. */
    int32_t use_layer_norm;
} orc_mlp_desc;

/* Inner-loop configuration: agents/DDQN.py:15-38 + agents/base_agent.py:9-26 + env config. */
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
    int32_t grad_chunk;
    int32_t rng_mode;
    int32_t agent_kind;   /* 0 = DDQN (agents/DDQN.py), 1 = DuelingDDQN (agents/DuelingDDQN.py) */
    int32_t feature_dim;  /* DuelingDDQN: width of the feature vector / of the two head hidden layers */
    double solved_reward;
    double gamma, lr, tau;
    double eps_init, eps_min, eps_decay;
    double adam_beta1, adam_beta2, adam_eps;
    int64_t step_budget;   /* see include/lenv_hip.h: env-step budget standing in for time_remaining (base_agent.py:30-47) */
    /* Intrinsic Curiosity Module inside learn() (agents/DDQN.py:40-58,74-76; models/icm_baseline.py): config section `icm` */
    int32_t icm_enabled, icm_feature_dim, icm_hidden;
    /* `use_layer_norm` of the env's config section with se_layers >= 2 (models/model_utils.py:22-37): the shared nn.LayerNorm behind hidden
     * Linear 2..L of each SE net.  NES perturbs and updates nn.Linear parameters only (GTN_worker.py:156-175, GTN_master.py:281-296), so the
     * module keeps its initial affine (weight 1, bias 0) and theta stays the Linear parameters */
    int32_t se_layer_norm;
    double icm_lr, icm_beta, icm_eta;
    /* gtn.synthetic_env_type 1: the agent trains on a RewardEnv over the REAL env (envs/reward_env.py:61-133) instead of the
     * VirtualEnv; se_hidden / se_layers / se_act then describe the reward network (state_dim -> 1), theta holds its parameters;
     * reward_env_type 0, 1, 2, 5, 6 (the real CartPole / Acrobot step returns no info vector) */
    int32_t synthetic_env_type, reward_env_type;
    /* same_action_num (base_agent.py:20,104,194; env_wrapper.py:24-29,56-61): env steps per chosen action; 0 and 1 both mean 1 */
    int32_t same_action_num;
    /* `use_layer_norm` of the agent's config section (models/model_utils.py:22-37): ONE shared nn.LayerNorm(hidden) behind every hidden
     * Linear but the first of the Q-net (Critic_DQN) / of the DuelingDDQN's feature stream (the heads have one hidden layer: none there) */
    int32_t q_layer_norm;
    /* test_mode (what BaseAgent.train is called with; base_agent.py:64,134-148):
     *   0 = train(env, test_env=real): GTN_Worker.calc_score (GTN_worker.py:195-199) -- the meter is fed by the mean of test_episodes real-env
     *       test episodes after every training episode, early-out = real rule on it (avg >= solved_reward);
     *   1 = train(env, test_env=None): every downstream evaluation (experiments/syn_env_evaluate_cartpole_vary_hp_2.py:38-41) -- NO per-episode
     *       tests, the meter is fed by the training env's own episode reward (fp32 sum of the step rewards, base_agent.py:121,138), break_env =
     *       the training env: a VirtualEnv ends on |avg - avg_last| / (|avg_last| + 1e-9) < early_out_virtual_diff once episode >=
     *       init_episodes + early_out_num (base_agent.py:49-56, AverageMeter.get_mean_last utils.py:97-98); a RewardEnv / real env on
     *       avg >= solved_reward.  episode_test_mean[] then carries the training episode rewards (the function's reward_train list). */
    int32_t test_mode;
    double early_out_virtual_diff;
} orc_ddqn_cfg;

/* RNG tapes (parity mode): values the reference drew, in per-stream order. */
typedef struct {
    const double *eps_uniform;  int64_t n_eps_uniform;   /* random.random()            DDQN.py:98 */
    const int32_t *rand_action; int64_t n_rand_action;   /* action_space.sample()      env_wrapper.py:88 */
    const int32_t *replay_idx;  int64_t n_replay_idx;    /* np.random.randint          utils.py:35 */
    const double *train_reset;  int64_t n_train_reset;   /* SE reset_env.reset() rows  virtual_env.py:36 */
    const double *test_reset;   int64_t n_test_reset;    /* real_env.reset() rows */
} orc_tapes;

/* Optional per-step trace of the training loop (parity debugging). */
typedef struct {
    int64_t cap;         /* rows available */
    int64_t n;           /* rows written */
    int32_t *episode;    /* [cap] */
    int32_t *action;     /* [cap] */
    int32_t *explored;   /* [cap] 1 if random action */
    float *state;        /* [cap,S] state before the step */
    float *next_state;   /* [cap,S] */
    float *reward;       /* [cap] */
    float *done;         /* [cap] */
    float *loss;         /* [cap] mse loss of the learn step (NaN when no learn) */
} orc_trace;

typedef struct {
    double score;             /* statistics.mean(final test returns)  GTN_worker.py:209 */
    int32_t episodes_run;     /* train episodes executed */
    int64_t train_steps;      /* SE steps */
    int64_t learn_steps;
    int64_t test_steps;       /* real-env steps (per-episode tests + final test) */
} orc_chain_result;

/* ---- deterministic math ---- */
float orc_tanhf(float x);
int64_t orc_tanhf_scan(uint32_t lo_bits, uint32_t hi_bits, float slack, float *max_out);
double orc_sin(double x);
double orc_cos(double x);

/* ---- RNG (counter mode) ---- */
uint64_t orc_mix64(uint64_t x);
uint64_t orc_chain_key(uint64_t seed, uint64_t generation, uint64_t worker, uint64_t kind);
uint64_t orc_rng_u64(uint64_t key, uint32_t stream, uint64_t n);
uint32_t orc_rng_replay_below(uint64_t key, uint64_t n, uint32_t size);
void orc_nes_draw(uint64_t seed, uint64_t generation, int64_t pop, int64_t P, float noise_std, float *eps, int64_t chains,
                  int64_t cpw, int64_t worker_lo, int64_t p_agent, const float *bounds, float *agent_init, uint64_t *rng_keys);

/* ---- MLP ---- */
int64_t orc_mlp_num_params(const orc_mlp_desc *d);
/* batched forward; x [B,in], y [B,out]; hidden_out (optional) [B,H] = last hidden activations */
int orc_mlp_forward(const orc_mlp_desc *d, const float *params, const float *x, int64_t B, float *y, float *hidden_out);

/* envs/virtual_env.py:43-54 + env_wrapper.py:16-47 for a population of perturbations:
 * W_c = theta + sign[c]*eps[worker[c]]; in = [onehot(action), state]. */
int orc_se_step_population(const orc_mlp_desc *state_net, const orc_mlp_desc *reward_net, const orc_mlp_desc *done_net,
                           const float *theta, const float *eps, const int32_t *worker, const float *sign,
                           int64_t chains, const float *state, const int32_t *action,
                           float *next_state, float *reward, float *done);

/* ---- real envs (gym 0.17.3, UNPINNED) ---- */
void orc_cartpole_step(double st[4], int action, double *reward, int *done);
void orc_acrobot_step(double st[4], int action, double *reward, int *done);
void orc_acrobot_obs(const double st[4], double obs[6]);

/* ---- DDQN pieces (agents/DDQN.py:60-110) ---- */
/* TD forward for a minibatch: q_sa, target y, per-sample loss term; rows = [s,a,s2,r,d] with stride row_stride */
int orc_qnet_td_forward(const orc_mlp_desc *q, const float *online, const float *target,
                        const float *rows, int64_t row_stride, int64_t B, int32_t S, double gamma,
                        float *q_sa, float *y, int32_t *argmax_next);
/* one full learn step on an explicit minibatch; updates online/target/m/v in place; returns loss */
float orc_ddqn_learn(const orc_ddqn_cfg *cfg, float *online, float *target, float *adam_m, float *adam_v,
                     int64_t step /*1-based*/, double *b1pow, double *b2pow,
                     const float *rows, int64_t row_stride);

/* ---- DuelingDDQN (agents/DuelingDDQN.py:59-110, models/actor_critic.py:94-122) ----
 * parameters: feature MLP (S -> H x layers -> F) | value MLP (F -> F -> 1) | advantage MLP (F -> F -> A), flat in
 * state-dict order.  q = V + (Adv - mean(Adv over ALL batch x action elements)) (actor_critic.py:121). */
int64_t orc_dueling_num_params(const orc_ddqn_cfg *cfg);
int orc_dueling_forward(const orc_ddqn_cfg *cfg, const float *params, const float *x, int64_t B, float *q /*[B,A]*/);
float orc_dueling_learn(const orc_ddqn_cfg *cfg, float *online, float *target, float *adam_m, float *adam_v,
                        double *b1pow, double *b2pow, const float *rows, int64_t row_stride);

/* ---- one chain = GTN_Worker.calc_score (GTN_worker.py:187-221) ---- */
int orc_ddqn_se_chain(const orc_ddqn_cfg *cfg, const float *se_params /*perturbed, [P_theta]*/,
                      const float *agent_init /*[P_agent]*/, uint64_t rng_key, const orc_tapes *tapes,
                      double *episode_test_mean /*[train_episodes] or NULL*/, int32_t *episode_len /*[train_episodes] or NULL*/,
                      double *final_test_returns /*[test_episodes] or NULL*/, orc_trace *trace, orc_chain_result *res);
/* the same + the trained online net (flat state-dict order) */
int orc_ddqn_se_chain_params(const orc_ddqn_cfg *cfg, const float *se_params, const float *agent_init, uint64_t rng_key,
                             const orc_tapes *tapes, double *episode_test_mean, int32_t *episode_len,
                             double *final_test_returns, orc_trace *trace, orc_chain_result *res, float *final_online);

/* population driver (multi-threaded over chains; used by the cpu_baseline leg) */
int orc_ddqn_se_population(const orc_ddqn_cfg *cfg, const float *theta, const float *eps, int64_t pop, int64_t p_theta,
                           const float *agent_init /*[3*pop,P_agent]*/, uint64_t seed, uint64_t generation,
                           int64_t worker_offset, int threads, double *chain_scores /*[3*pop]*/,
                           orc_chain_result *results /*[3*pop] or NULL*/);

/* ---- config 4: tabular Q-learning on a potential-shaped RewardEnv over a grid MDP ----
 * agents/QL.py:13-105, envs/reward_env.py:29-149, envs/gridworld.py:24-124 (MDP given as transition tables) */
typedef struct {
    int32_t n_states, n_actions, start_state, max_steps;
    int32_t rn_hidden, rn_layers, rn_act;
    float rn_prelu;
    int32_t reward_env_type;            /* 0,1,2,5,6 (types with an info vector are not part of the grid path) */
    int32_t train_episodes, test_episodes, init_episodes, early_out_num, batch_size;
    int32_t rng_mode;
    int32_t agent_kind;                 /* 0 QL (agents/QL.py), 1 SARSA (agents/SARSA.py) */
    int32_t count_based;                /* ql_cb / sarsa_cb (agent_utils.py:57-64): reward += beta / (sqrt(n(s,a)) + 1e-9) */
    double solved_reward, alpha, gamma, eps_init, eps_min, eps_decay, beta;
    int64_t step_budget;
    int32_t same_action_num;            /* env steps per chosen action; 0 and 1 both mean 1 */
    int32_t rn_layer_norm;              /* the ENV section's use_layer_norm with rn_layers >= 2: the reward net's (never perturbed) LayerNorm */
    int32_t test_mode;                  /* as orc_ddqn_cfg::test_mode */
    double early_out_virtual_diff;      /* never read: a grid RewardEnv is not a VirtualEnv */
} orc_ql_cfg;

typedef struct {
    int64_t cap, n;
    int32_t *action;      /* action | explored<<16 */
    int32_t *state;       /* state before the step */
    int32_t *next_state;
    float *reward;        /* shaped reward returned by the RewardEnv */
    float *done;
} orc_ql_trace;

/* RewardEnv._calc_reward (reward_env.py:67-133) for every (s,a) of the grid MDP -> shaped[n_states*n_actions] */
int orc_rn_shaped_rewards(const orc_ql_cfg *cfg, const float *rn_params, const int32_t *next_state, const double *reward,
                          float *phi /*[n_states] or NULL*/, float *shaped);
/* GTN_Worker.calc_score with QL on a RewardEnv (GTN_worker.py:187-221) */
int orc_ql_rn_chain(const orc_ql_cfg *cfg, const float *rn_params, const float *shaped_override /*[N*A] or NULL*/,
                    const int32_t *next_state, const double *reward, const uint8_t *done, uint64_t rng_key, const orc_tapes *tapes, double *episode_test_mean,
                    int32_t *episode_len, double *final_test_returns, double *q_table_out, orc_ql_trace *trace,
                    orc_chain_result *res);

/* ---- config 5: TD3 on a continuous-state RewardEnv (agents/TD3.py:13-135, envs/reward_env.py:61-133) ----
 * real env: the documented HalfCheetah-v3 stand-in (tools/gen_cheetah_standin.py). */
enum { ORC_ENV_CHEETAH_STANDIN = 2, ORC_ENV_PENDULUM = 4, ORC_ENV_CMC = 5 };
typedef struct {
    int32_t env_id, state_dim, action_dim, max_steps;
    int32_t rn_hidden, rn_layers, rn_act;
    float rn_prelu;
    int32_t reward_env_type;                 /* 0-8, 101, 102 (reward_env.py:29-59) */
    int32_t info_dim;                        /* length of the real env's info vector (4 for the stand-in) */
    int32_t hidden, layers, act;             /* actor / critic MLPs (models/actor_critic.py:11-19,64-71) */
    float prelu;
    int32_t batch_size, rb_size, train_episodes, test_episodes, init_episodes, early_out_num, policy_delay, rng_mode;
    double solved_reward, gamma, lr, tau, action_std, policy_std, policy_std_clip, max_action;
    double adam_beta1, adam_beta2, adam_eps;
    int64_t step_budget;
    /* TD3(icm=True), select_agent "td3_icm" (agents/TD3.py:44-60,68-70): config section `icm` */
    int32_t icm_enabled, icm_feature_dim, icm_hidden;
    /* `use_layer_norm` of the td3 section (models/model_utils.py:22-37): ONE shared nn.LayerNorm(hidden) behind every hidden Linear but the
     * first of the actor and of each critic (three modules, one per net), weight | bias behind the net's second Linear */
    int32_t use_layer_norm;
    double icm_lr, icm_beta, icm_eta;
    /* virtual_env != 0 (gtn.synthetic_env_type 0, default_config_halfcheetah.yaml): the agent trains on a VirtualEnv
     * (envs/virtual_env.py:43-54) instead of the RewardEnv -- rn_params then holds state_net | reward_net | done_net, each
     * (action_dim + state_dim) -> rn_hidden x rn_layers -> {state_dim, 1, 1} with rn_act; the episode ends on done > 0.5 */
    int32_t virtual_env;
    /* same_action_num (base_agent.py:20, env_wrapper.py:24,57): env steps per chosen action; 0 and 1 both mean 1 */
    int32_t same_action_num;
    /* the ENV section's use_layer_norm with rn_layers >= 2: the reward net / the three SE nets normalise behind hidden Linear 2..L; NES perturbs
     * nn.Linear modules only, rn_params stays Linear-only and the module keeps weight 1 / bias 0 */
    int32_t rn_layer_norm;
    int32_t test_mode;                  /* as orc_ddqn_cfg::test_mode */
    double early_out_virtual_diff;
} orc_td3_cfg;

typedef struct {
    const float *rand_action;  int64_t n_rand_action;   /* rows of A: Box.sample() returned by get_random_action  env_wrapper.py:87-90 */
    const float *act_noise;    int64_t n_act_noise;     /* rows of A: torch.randn(action_dim) in select_train_action  TD3.py:123 */
    const float *test_noise;   int64_t n_test_noise;    /* rows of A: torch.randn(action_dim) in select_test_action   TD3.py:128 */
    const float *policy_noise; int64_t n_policy_noise;  /* rows of B*A: torch.randn_like(actions) in learn            TD3.py:75 */
    const int32_t *replay_idx; int64_t n_replay_idx;    /* rows of B */
    const double *train_reset; int64_t n_train_reset;   /* rows of S */
    const double *test_reset;  int64_t n_test_reset;    /* rows of S */
} orc_td3_tapes;

typedef struct {
    int64_t cap, n;
    float *action;      /* [cap,A] action passed to env.step */
    float *state;       /* [cap,S] */
    float *next_state;  /* [cap,S] */
    float *reward;      /* [cap] shaped reward */
} orc_td3_trace;

double orc_log(double x);
double orc_normal(uint64_t key, uint32_t stream, uint64_t n);      /* Box-Muller on two counter draws */
void orc_cheetah_step(double x[17], const float a[6], double *reward);
int64_t orc_td3_actor_params(const orc_td3_cfg *cfg);
int64_t orc_td3_critic_params(const orc_td3_cfg *cfg);
/* actor forward: tanh(net(s)) * max_action; critic forward: net(cat(s,a)) */
int orc_td3_actor_forward(const orc_td3_cfg *cfg, const float *actor, const float *s, int64_t B, float *out);
int orc_td3_critic_forward(const orc_td3_cfg *cfg, const float *critic, const float *s, const float *a, int64_t B, float *out);
/* one TD3.learn step; params = [actor | critic1 | critic2], targets/m/v same layout; pows: {b1c,b2c,b1a,b2a} */
int orc_td3_learn(const orc_td3_cfg *cfg, float *params, float *targets, float *adam_m, float *adam_v, double pows[4],
                  int64_t total_it /*1-based*/, const float *rows, int64_t row_stride, const float *policy_noise /*[B,A] N(0,1)*/,
                  float *losses /*[2] critic, actor (optional)*/);
/* number of reward-net parameters for a RewardEnv type (reward_env.py:29-59): MLP on S (types 1,2,5,6) or S+info_dim
 * (3,4,7,8) inputs, Linear(info_dim,1,bias=False) for 101/102, 0 for type 0 (its dummy net is never evaluated) */
int64_t orc_rn_num_params(int type, int S, int info_dim, int hidden, int layers);
/* RewardEnv._calc_reward (reward_env.py:68-133) for n rows of a vector-state env: s,s2 [n,S], info [n,info_dim], r [n]
 * (the real reward already rounded to fp32); fp32, left to right.  Returns -1 for unknown types. */
int orc_rn_shape_rows(int type, int S, int info_dim, int hidden, int layers, int use_layer_norm, int act, float prelu, double gamma,
                      const float *rn_params, const float *s, const float *s2, const float *info, const float *r, int64_t n,
                      float *out);
int orc_td3_rn_chain(const orc_td3_cfg *cfg, const float *rn_params, const float *agent_init /*[actor|critic1|critic2]*/,
                     uint64_t rng_key, const orc_td3_tapes *tapes, double *episode_test_mean, int32_t *episode_len,
                     double *final_test_returns, orc_td3_trace *trace, orc_chain_result *res);
/* the same + the trained agent: final_params [actor | critic_1 | critic_2] */
int orc_td3_rn_chain_params(const orc_td3_cfg *cfg, const float *rn_params, const float *agent_init, uint64_t rng_key,
                            const orc_td3_tapes *tapes, double *episode_test_mean, int32_t *episode_len, double *final_test_returns,
                            orc_td3_trace *trace, orc_chain_result *res, float *final_params);

/* ---- TD3_discrete_vary on a VirtualEnv over a discrete-action real env (agents/TD3_discrete_vary.py:62-117,159-171,
 * models/actor_critic.py:22-35, agents/base_agent.py:64-227 with discretize_action) ---- */
typedef struct {
    int32_t env_id, state_dim, action_dim, max_steps;       /* CartPole-v0 4 / 2, Acrobot-v1 6 / 3, MountainCar-v0 2 / 3 */
    int32_t se_hidden, se_layers, se_act;                    /* the VirtualEnv's three nets on cat(one_hot(argmax action), state) */
    float se_prelu;
    int32_t hidden, layers, act;                             /* actor state->action_dim and critics (state+action_dim)->1 */
    float prelu;
    int32_t use_layer_norm;                                  /* model_utils.py:22-37: one shared LayerNorm behind hidden Linear 2..L */
    int32_t gumbel_hard;                                     /* actor_critic.py:31,35 */
    int32_t batch_size, rb_size, train_episodes, test_episodes, init_episodes, early_out_num, policy_delay, rng_mode;
    double solved_reward, gamma, lr, tau, action_std, policy_std, policy_std_clip, max_action;
    double gumbel_temp;                                      /* annealed over the first 2000 learn calls to gumbel_temp / 20 (:59-60,64-68) */
    double adam_beta1, adam_beta2, adam_eps;
    int64_t step_budget;
    int32_t se_layer_norm;                                   /* the ENV section's use_layer_norm, as orc_td3_cfg::rn_layer_norm */
    int32_t test_mode;                  /* as orc_ddqn_cfg::test_mode */
    double early_out_virtual_diff;
} orc_td3d_cfg;

typedef struct {
    const int32_t *rand_action; int64_t n_rand_action;     /* Discrete.sample() of the init episodes (env_wrapper.py:87-92) */
    const float *act_noise;     int64_t n_act_noise;       /* rows of A: torch.randn(action_dim) in select_train_action :167 */
    const float *test_noise;    int64_t n_test_noise;      /* rows of A: the same in select_test_action :171 */
    const float *policy_noise;  int64_t n_policy_noise;    /* rows of A, B per learn call: randn_like(actions) :76 */
    const float *gumbel_act;    int64_t n_gumbel_act;      /* rows of A: the Gumbel(0,1) draw of F.gumbel_softmax in select_train_action */
    const float *gumbel_test;   int64_t n_gumbel_test;     /* rows of A: ... in select_test_action */
    const float *gumbel_target; int64_t n_gumbel_target;   /* rows of A, B per learn call: actor_target(next_states) :77 */
    const float *gumbel_actor;  int64_t n_gumbel_actor;    /* rows of A, B per policy update: actor(states) :101 */
    const int32_t *replay_idx;  int64_t n_replay_idx;      /* elements, B per learn call */
    const double *train_reset;  int64_t n_train_reset;     /* rows of 4: the reset env's state */
    const double *test_reset;   int64_t n_test_reset;
} orc_td3d_tapes;

int64_t orc_td3d_actor_params(const orc_td3d_cfg *cfg);
int64_t orc_td3d_critic_params(const orc_td3d_cfg *cfg);
float orc_td3d_temperature(const orc_td3d_cfg *cfg, int64_t learn_calls_so_far);      /* np.linspace(t, t/20, 2000)[min(n, 1999)] as fp32 */
float orc_gumbel(uint64_t key, uint32_t stream, uint64_t n);                           /* counter mode: -log(-log(u)) */
/* Actor_TD3_discrete.forward on B rows: gumbel_softmax(net(s) * max_action, tau, hard) with the given Gumbel draws [B,A] */
int orc_td3d_actor_forward(const orc_td3d_cfg *cfg, const float *actor, const float *s, const float *gumbel, float tau, int64_t B, float *out);
/* one TD3_discrete_vary.learn call; params = [actor | critic_1 | critic_2]; total_it = calls so far INCLUDING this one */
int orc_td3d_learn(const orc_td3d_cfg *cfg, float *params, float *targets, float *adam_m, float *adam_v, double pows[4],
                   int64_t total_it, const float *rows, int64_t row_stride, const float *policy_noise /*[B,A]*/,
                   const float *gumbel_target /*[B,A]*/, const float *gumbel_actor /*[B,A], read on policy updates*/);
int orc_td3d_chain(const orc_td3d_cfg *cfg, const float *se_params, const float *agent_init, uint64_t rng_key,
                   const orc_td3d_tapes *tapes, double *episode_test_mean, int32_t *episode_len, double *final_test_returns,
                   orc_td3_trace *trace, orc_chain_result *res, float *final_params);

/* ---- NES master/worker math ---- */
/* GTN_worker.py:234-254: mirrored sampling pick; out[p] = {score_best, sign} */
void orc_worker_best(const double *score_add, const double *score_sub, int64_t pop, int mirrored, double *score_best, float *sign);
/* the same with num_grad_evals = G evaluations per direction (score lists [pop,G]); grad_eval_type 0 'mean', 1 'minmax' */
int orc_worker_best_multi(const double *score_add, const double *score_sub, int64_t pop, int G, int mirrored, int grad_eval_type,
                          double *score_best, float *sign);
/* GTN_master.py:197-265; ties broken by lower index first (documented stable order) */
int orc_score_transform(int type, const double *scores, const double *scores_orig, int64_t n, double *out);
/* GTN_master.py:267-298: theta <- theta*(1-wd); theta += ss*w_i*sign_i*eps_i sequentially over i */
void orc_update_env(float *theta, const float *eps, const float *sign, const double *weights, int64_t pop, int64_t p_theta,
                    const uint8_t *linear_mask /*[p_theta] 1 = nn.Linear param*/, double step_size, int nes_step_size, double weight_decay);

#ifdef __cplusplus
}
#endif
#endif
