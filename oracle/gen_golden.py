#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the read-only reference (/root/reference).

TEST INFRASTRUCTURE.  Runs only in the build container (the reference never travels to
the GPU box); its outputs -- small .npz files of inputs/expected outputs -- are committed
under tests/golden/ and are what pins oracle/lenv_oracle.c to the reference.

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz

Recipe = SURVEY.md Appendix C: shims for gym/seaborn/ConfigSpace on PYTHONPATH ahead of
the reference, every RNG source pinned, RNG draws recorded as tapes.
"""
import contextlib
import copy
import io
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("LENV_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "shims"))
os.makedirs("/tmp/lenv_golden_cwd", exist_ok=True)
os.chdir("/tmp/lenv_golden_cwd")  # GTN_Base creates ./results/GTN_sync relative to cwd

import torch  # noqa: E402
import yaml  # noqa: E402
import gym  # noqa: E402  (the shim)
from gym.utils import seeding  # noqa: E402

torch.set_num_threads(1)


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def load_cfg(name):
    with open(os.path.join(REF, name)) as f:
        return yaml.safe_load(f)


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)
    random.seed(s)
    seeding.set_counter(1000 + s)


def pack_linear_params(state_dict, prefix):
    """Flat fp32 vector of the nn.Linear weights/biases under `prefix`, in state-dict order
    (W0,b0,...,Wout,bout); PReLU slopes (1-element '<idx>.weight') are skipped."""
    parts, seen = [], set()
    for k, v in state_dict.items():
        if k.startswith(prefix) and not (k.endswith("weight") and _is_prelu_key(state_dict, k)):
            if v.data_ptr() in seen:                 # the shared nn.LayerNorm appears once per POSITION in a state dict: Module.parameters() order keeps the first
                continue
            seen.add(v.data_ptr())
            parts.append(v.detach().cpu().numpy().astype(np.float32).reshape(-1))
    return np.concatenate(parts)


def pack_linear_only(state_dict, prefix):
    """The nn.Linear weights / biases under `prefix` in state-dict order -- THE layout of theta (GTN_worker.py:156-175 perturbs nn.Linear
    modules only): a LayerNorm's 1-D weight and its bias are left out, as PReLU slopes are."""
    parts = []
    for k, v in state_dict.items():
        if k.startswith(prefix) and k.endswith("weight") and v.dim() == 2:
            parts.append(v.detach().cpu().numpy().astype(np.float32).reshape(-1))
            b = state_dict.get(k[:-len("weight")] + "bias")
            if b is not None:
                parts.append(b.detach().cpu().numpy().astype(np.float32).reshape(-1))
    return np.concatenate(parts)


def _is_prelu_key(sd, k):
    # a PReLU module has a weight but no bias sibling
    return (k[:-len("weight")] + "bias") not in sd


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, {k: (np.asarray(v).shape) for k, v in arrays.items()})


# ------------------------------------------------------------------------------------------------
# G1: VirtualEnv.step through EnvWrapper.step (envs/virtual_env.py:43-54, env_wrapper.py:16-47)
# ------------------------------------------------------------------------------------------------
def gen_g1():
    from envs.env_factory import EnvFactory
    out = {}
    cases = []
    for env_yaml, env_name in (("default_config_cartpole_syn_env.yaml", "CartPole-v0"),
                               ("default_config_acrobot_syn_env.yaml", "Acrobot-v1")):
        for act in ("leakyrelu", "relu", "tanh", "prelu", "identity"):
            for layers in (1, 2):
                cases.append((env_yaml, env_name, act, layers))
    for ci, (env_yaml, env_name, act, layers) in enumerate(cases):
        cfg = load_cfg(env_yaml)
        cfg["envs"][env_name]["activation_fn"] = act
        cfg["envs"][env_name]["hidden_layer"] = layers
        if layers == 2:
            cfg["envs"][env_name]["hidden_size"] = 40 if ci % 4 == 1 else 24
        seed_all(100 + ci)
        with quiet():
            fac = EnvFactory(cfg)
            venv = fac.generate_virtual_env()
        sd = venv.state_dict()
        S, A = venv.get_state_dim(), venv.get_action_dim()
        theta = np.concatenate([pack_linear_params(sd, "env.state_net."), pack_linear_params(sd, "env.reward_net."),
                                pack_linear_params(sd, "env.done_net.")])
        n = 12
        states = torch.randn(n, S) * 0.7
        actions = torch.randint(0, A, (n,))
        ns, rw, dn = [], [], []
        with torch.no_grad():
            for i in range(n):
                venv.env.state = states[i].clone()
                a, b, c = venv.step(actions[i:i + 1].float())
                ns.append(a.numpy().copy()); rw.append(b.numpy().copy()); dn.append(c.numpy().copy())
        pre = "c%02d_" % ci
        out[pre + "meta"] = np.array([S, A, cfg["envs"][env_name]["hidden_size"], layers,
                                      ["identity", "relu", "leakyrelu", "tanh", "prelu"].index(act)], np.int64)
        out[pre + "theta"] = theta
        out[pre + "state"] = states.numpy()
        out[pre + "action"] = actions.numpy().astype(np.int32)
        out[pre + "next_state"] = np.stack(ns)
        out[pre + "reward"] = np.concatenate(rw)
        out[pre + "done"] = np.concatenate(dn)
    out["n_cases"] = np.array(len(cases))
    save("g1_virtual_env_step", **out)


# ------------------------------------------------------------------------------------------------
# G3: Critic_DQN forward on a batch (models/actor_critic.py:84-91)
# ------------------------------------------------------------------------------------------------
def gen_g3():
    from models.actor_critic import Critic_DQN
    out = {}
    cases = [(4, 2, 57, 1, "tanh"), (6, 3, 112, 1, "leakyrelu"), (4, 2, 24, 2, "relu"), (6, 3, 40, 2, "tanh")]
    for ci, (S, A, H, L, act) in enumerate(cases):
        cfg = {"agents": {"ddqn": {"hidden_size": H, "hidden_layer": L, "activation_fn": act}}}
        seed_all(300 + ci)
        net = Critic_DQN(S, A, "ddqn", cfg)
        x = torch.randn(37, S)
        with torch.no_grad():
            y = net(x)
            y1 = torch.stack([net(x[i]) for i in range(5)])   # batch-1 (gemv) path
        pre = "c%d_" % ci
        out[pre + "meta"] = np.array([S, A, H, L, ["identity", "relu", "leakyrelu", "tanh", "prelu"].index(act)], np.int64)
        out[pre + "params"] = pack_linear_params(net.state_dict(), "net.")
        out[pre + "x"] = x.numpy()
        out[pre + "y"] = y.numpy()
        out[pre + "y_single"] = y1.numpy()
    out["n_cases"] = np.array(len(cases))
    save("g3_critic_dqn_forward", **out)


# ------------------------------------------------------------------------------------------------
# G4: DDQN.learn steps on explicit minibatches (agents/DDQN.py:60-94)
# ------------------------------------------------------------------------------------------------
def gen_g4():
    from agents.DDQN import DDQN
    from envs.env_factory import EnvFactory
    from utils import ReplayBuffer
    out = {}
    variants = [("default_config_cartpole_syn_env.yaml", {}, 4),
                ("default_config_cartpole_syn_env.yaml", {"activation_fn": "relu", "hidden_layer": 2, "hidden_size": 24, "batch_size": 64}, 3),
                ("default_config_acrobot_syn_env.yaml", {}, 3)]
    for vi, (yml, over, nsteps) in enumerate(variants):
        cfg = load_cfg(yml)
        cfg["agents"]["ddqn"].update(over)
        seed_all(400 + vi)
        with quiet():
            fac = EnvFactory(cfg)
            real_env = fac.generate_real_env()
            agent = DDQN(env=real_env, config=cfg)
        S, A = real_env.get_state_dim(), real_env.get_action_dim()
        B = cfg["agents"]["ddqn"]["batch_size"]
        rb = ReplayBuffer(state_dim=S, action_dim=1, device="cpu", max_size=600)
        for i in range(500):
            rb.add(torch.randn(S) * 0.5, torch.tensor([float(np.random.randint(A))]), torch.randn(S) * 0.5,
                   torch.randn(1) * 0.3 + 0.5, torch.randn(1) * 0.2)
        # make target differ from online so the Polyak step is exercised
        with torch.no_grad():
            for p in agent.model_target.parameters():
                p.add_(torch.randn_like(p) * 0.05)
        pre = "v%d_" % vi
        a = cfg["agents"]["ddqn"]
        out[pre + "meta"] = np.array([S, A, a["hidden_size"], a["hidden_layer"],
                                      ["identity", "relu", "leakyrelu", "tanh", "prelu"].index(a["activation_fn"]), B, nsteps], np.int64)
        out[pre + "hparams"] = np.array([a["gamma"], a["lr"], a["tau"]], np.float64)
        out[pre + "online0"] = pack_linear_params(agent.model.state_dict(), "net.")
        out[pre + "target0"] = pack_linear_params(agent.model_target.state_dict(), "net.")
        idxs, rows_all, onl, tgt, losses, ms, vs = [], [], [], [], [], [], []
        for step in range(nsteps):
            idx = np.random.randint(0, rb.size, size=B)
            rb.sample = lambda batch_size, _idx=idx: rb._sample_idx(_idx)
            rows = np.concatenate([rb.state[idx].numpy(), rb.action[idx].numpy(), rb.next_state[idx].numpy(),
                                   rb.reward[idx].numpy(), rb.done[idx].numpy()], axis=1)
            loss = agent.learn(rb, real_env, episode=5)
            idxs.append(idx); rows_all.append(rows); losses.append(float(loss.item()))
            onl.append(pack_linear_params(agent.model.state_dict(), "net."))
            tgt.append(pack_linear_params(agent.model_target.state_dict(), "net."))
            st = agent.optimizer.state_dict()["state"]
            order = list(range(len(st)))
            ms.append(np.concatenate([st[i]["exp_avg"].numpy().reshape(-1) for i in order]))
            vs.append(np.concatenate([st[i]["exp_avg_sq"].numpy().reshape(-1) for i in order]))
        out[pre + "rows"] = np.stack(rows_all).astype(np.float32)
        out[pre + "loss"] = np.array(losses, np.float64)
        out[pre + "online"] = np.stack(onl)
        out[pre + "target"] = np.stack(tgt)
        out[pre + "adam_m"] = np.stack(ms)
        out[pre + "adam_v"] = np.stack(vs)
    out["n_variants"] = np.array(len(variants))
    save("g4_ddqn_learn", **out)


# ------------------------------------------------------------------------------------------------
# G6: GTN_Worker noise / perturbation / mirrored pick (agents/GTN_worker.py:156-254)
# ------------------------------------------------------------------------------------------------
def se_theta(envw):
    sd = envw.state_dict()
    return np.concatenate([pack_linear_only(sd, "env.state_net."), pack_linear_only(sd, "env.reward_net."),
                           pack_linear_only(sd, "env.done_net.")])


def gen_g6():
    from agents.GTN import GTN_Worker
    cfg = load_cfg("default_config_cartpole_syn_env.yaml")
    with quiet():
        w = GTN_Worker(id=0, bohb_id=0)
        seed_all(600)
        w.config = cfg
        w.late_init(cfg)
        w.synthetic_env.load_state_dict(w.synthetic_env_orig.state_dict())
        theta = se_theta(w.synthetic_env_orig)
        w.get_random_noise()
        eps = se_theta(w.eps)
        w.add_noise_to_synthetic_env()
        plus = se_theta(w.synthetic_env)
        w.subtract_noise_from_synthetic_env()
        minus = se_theta(w.synthetic_env)
        cases = []
        for add, sub in ((10.0, 20.0), (20.0, 10.0), (15.0, 15.0)):
            w.eps.load_state_dict(_sd_from_flat(w.eps, eps))
            w.subtract_noise_from_synthetic_env()
            best = w.calc_best_score(score_sub=[sub], score_add=[add])
            cases.append((add, sub, best, se_theta(w.eps), se_theta(w.synthetic_env)))
    save("g6_worker_noise", theta=theta, eps=eps, theta_plus=plus, theta_minus=minus,
         noise_std=np.array(cfg["agents"]["gtn"]["noise_std"]),
         score_add=np.array([c[0] for c in cases]), score_sub=np.array([c[1] for c in cases]),
         score_best=np.array([c[2] for c in cases]), eps_after=np.stack([c[3] for c in cases]),
         env_after=np.stack([c[4] for c in cases]))



def gen_g6m():
    """calc_best_score with num_grad_evals = 3 score lists, 'mean' (statistics.mean) and 'minmax', mirrored or not."""
    from agents.GTN import GTN_Worker
    cfg = load_cfg("default_config_cartpole_syn_env.yaml")
    rng = np.random.RandomState(61)
    n = 40
    add = rng.uniform(5, 200, (n, 3)) / rng.choice([1.0, 3.0, 7.0, 10.0], (n, 1))
    sub = rng.uniform(5, 200, (n, 3)) / rng.choice([1.0, 3.0, 7.0, 10.0], (n, 1))
    add[0] = sub[0]                                    # a tie: +eps is kept
    add[1] = [0.1, 0.2, 0.3]; sub[1] = [0.3, 0.2, 0.1]   # equal exact means, different float sums
    out = {}
    with quiet():
        w = GTN_Worker(id=0, bohb_id=0)
        seed_all(601)
        w.config = cfg
        w.late_init(cfg)
        w.get_random_noise()
        eps = se_theta(w.eps)
        for gtype in ("mean", "minmax"):
            for mirrored in (True, False):
                w.grad_eval_type = gtype
                w.mirrored_sampling = mirrored
                best, sign = [], []
                for i in range(n):
                    w.eps.load_state_dict(_sd_from_flat(w.eps, eps))
                    b = w.calc_best_score(score_sub=[float(v) for v in sub[i]], score_add=[float(v) for v in add[i]])
                    best.append(b)
                    sign.append(-1.0 if np.array_equal(se_theta(w.eps), -eps) else 1.0)
                out["best_%s_%d" % (gtype, mirrored)] = np.array(best)
                out["sign_%s_%d" % (gtype, mirrored)] = np.array(sign, np.float32)
    save("g6m_worker_best_multi", score_add=add, score_sub=sub, **out)


def _sd_from_flat(envw, flat):
    sd = copy.deepcopy(envw.state_dict())
    off = 0
    for k in sd:
        if not _is_prelu_key(sd, k) or not k.endswith("weight"):
            n = sd[k].numel()
            sd[k] = torch.from_numpy(flat[off:off + n].reshape(tuple(sd[k].shape)).copy())
            off += n
    assert off == flat.size
    return sd


# ------------------------------------------------------------------------------------------------
# G7: GTN_Master.score_transform (all types) and update_env (agents/GTN_master.py:197-298)
# ------------------------------------------------------------------------------------------------
def gen_g7():
    from agents.GTN import GTN_Master
    cfg = load_cfg("default_config_cartpole_syn_env.yaml")
    n = 8
    cfg["agents"]["gtn"]["num_workers"] = n
    out = {}
    rng = np.random.RandomState(7)
    scores = rng.uniform(10, 200, n)
    scores_orig = rng.uniform(10, 200, n)
    tied = np.array([200.0, 13.0, 200.0, 57.0, 13.0, 200.0, 99.0, 57.0])
    with quiet():
        seed_all(700)
        m = GTN_Master(cfg, bohb_id=0)
        for t in range(8):
            m.score_transform_type = t
            m.score_list = list(scores); m.score_orig_list = list(scores_orig)
            m.score_transform()
            out["tf%d" % t] = np.array(m.score_transform_list, np.float64)
            m.score_list = list(tied); m.score_orig_list = list(scores_orig)
            m.score_transform()
            out["tf%d_tied" % t] = np.array(m.score_transform_list, np.float64)
        theta0 = se_theta(m.synthetic_env_orig)
        eps = []
        for e in m.eps_list:
            flat = (rng.randn(theta0.size) * 0.0124).astype(np.float32)
            e.load_state_dict(_sd_from_flat(e, flat))
            eps.append(flat)
        m.score_transform_type = 3
        m.score_list = list(scores); m.score_orig_list = list(scores_orig)
        m.score_transform()
        w = np.array(m.score_transform_list)
        m.update_env()
        theta1 = se_theta(m.synthetic_env_orig)
        m.nes_step_size = True
        m.weight_decay = 0.01
        m.update_env()
        theta2 = se_theta(m.synthetic_env_orig)
    save("g7_master", scores=scores, scores_orig=scores_orig, tied=tied, theta0=theta0, eps=np.stack(eps), weights=w,
         step_size=np.array(cfg["agents"]["gtn"]["step_size"]), theta1=theta1, theta2=theta2, **out)


# ------------------------------------------------------------------------------------------------
# G8: full GTN_Worker.calc_score with recorded RNG tapes + per-step trace (cfg 1 shapes)
# ------------------------------------------------------------------------------------------------
class Recorder(object):
    def __init__(self):
        self.eps_uniform, self.rand_action, self.replay_idx = [], [], []
        self.resets = []          # (id(env), state)
        self.steps = []           # dicts
        self.losses = []
        self.active = False


def gen_g8(name, train_episodes, done_bias_shift, seed, max_steps=None, env_yaml="default_config_cartpole_syn_env.yaml",
           env_name="CartPole-v0", env_cls="CartPoleEnv", agent_key="ddqn", agent_over=None, env_over=None, vary_seed=None,
           icm_over=None, reward_env_type=None, record_q_gap=False, perturb_ulp=False):
    import agents.GTN_worker as gw
    from agents.GTN import GTN_Worker
    import gym.envs as genvs
    import gym.spaces as gspaces
    cfg = load_cfg(env_yaml)
    cfg["agents"][agent_key]["train_episodes"] = train_episodes
    cfg["agents"][agent_key]["print_rate"] = int(1e9)
    cfg["agents"][agent_key].update(agent_over or {})
    cfg["envs"][env_name].update(env_over or {})
    cfg["agents"]["gtn"]["agent_name"] = {"ddqn": "DDQN", "duelingddqn": "DuelingDDQN"}[agent_key]
    if icm_over is not None:
        # the agent with its Intrinsic Curiosity Module (select_agent: "ddqn_icm" -> DDQN(icm=True), agents/DDQN.py:40-58,74-76)
        cfg["agents"]["gtn"]["agent_name"] += "_icm"
        cfg["agents"]["icm"].update(icm_over)
    if vary_seed is not None:
        # the *_vary agent of the same family (agents/DDQN_vary.py, DuelingDDQN_vary.py): its ConfigSpace draw comes from
        # the stand-in under oracle/shims (seeded here); the sampled values are recorded in the fixture
        import ConfigSpace
        ConfigSpace.RANDOM.seed(vary_seed)
        cfg["agents"]["gtn"]["agent_name"] += "_vary"
        cfg["agents"][agent_key + "_vary"] = {"vary_hp": True}
    cfg["agents"]["gtn"]["synthetic_env_type"] = 0
    if reward_env_type is not None:
        # the agent trains on a RewardEnv over the real env (envs/reward_env.py) instead of a VirtualEnv
        cfg["agents"]["gtn"]["synthetic_env_type"] = 1
        cfg["envs"][env_name]["reward_env_type"] = reward_env_type
    if max_steps:
        cfg["envs"][env_name]["max_steps"] = max_steps
    rec = Recorder()

    orig_random, orig_randint = random.random, np.random.randint
    cls = getattr(genvs, env_cls)
    orig_reset, orig_sample = cls.reset, gspaces.Discrete.sample

    def rec_random():
        v = orig_random()
        if rec.active:
            rec.eps_uniform.append(v)
        return v

    def rec_randint(*a, **k):
        v = orig_randint(*a, **k)
        if rec.active:
            rec.replay_idx.append(np.asarray(v).copy())
        return v

    def rec_reset(self):
        obs = orig_reset(self)
        if rec.active:
            rec.resets.append((id(self), np.array(self.state, np.float64).copy()))
        return obs

    def rec_sample(self):
        v = orig_sample(self)
        if rec.active:
            rec.rand_action.append(v)
        return v

    orig_select_agent = gw.select_agent
    holder = {}

    def wrapped_select_agent(config, agent_name):
        agent = orig_select_agent(config=config, agent_name=agent_name)
        if perturb_ulp:
            # the reference against ITSELF: every weight of the fresh agent moved by one unit in the last place.  How fast the two
            # reference runs drift apart is the yardstick for any other implementation's drift (test_long_horizon_*)
            with torch.no_grad():
                for prm in agent.model.parameters():
                    prm.copy_(torch.from_numpy(np.nextafter(prm.numpy(), np.float32(np.inf))))
                agent.model_target.load_state_dict(agent.model.state_dict())      # (DDQN.py:36: the target starts as a copy)
        holder["agent"] = agent
        holder["hp"] = {k: agent.full_config["agents"][agent_key][k] for k in ("lr", "batch_size", "hidden_size", "hidden_layer")} \
            if hasattr(agent, "full_config") else {}
        holder["init"] = pack_linear_params(agent.model.state_dict(), "net.") if hasattr(agent.model, "net") \
            else _pack_dueling(agent.model.state_dict())
        if getattr(agent, "icm", None):
            # ICMModel parameters in state-dict order (features, inverse, forward_pre, residual blocks 1-4, forward_post)
            holder["icm_init"] = np.concatenate([v.detach().cpu().numpy().astype(np.float32).reshape(-1)
                                                 for v in agent.icm.model.state_dict().values()])
        orig_learn = agent.learn

        def learn(replay_buffer, env, episode):
            loss = orig_learn(replay_buffer=replay_buffer, env=env, episode=episode)
            rec.losses.append(float(loss.item()))
            return loss

        agent.learn = learn
        return agent

    with quiet():
        w = GTN_Worker(id=0, bohb_id=0)
        seed_all(seed)
        w.config = cfg
        w.late_init(cfg)
        w.timeout = 1e9
        if done_bias_shift:
            with torch.no_grad():
                w.synthetic_env_orig.env.done_net[-1].bias.add_(done_bias_shift)
        theta = se_theta(w.synthetic_env_orig) if reward_env_type is None \
            else pack_linear_only(w.synthetic_env_orig.state_dict(), "env.reward_net.")
        env = w.synthetic_env_orig
        orig_step = env.step

        def rec_step(action, state=None):
            s_before = env.env.state.detach().numpy().copy() if reward_env_type is None \
                else np.asarray(env.env.state, np.float32).copy()
            ns, r, d = orig_step(action=action, state=state) if reward_env_type is None else orig_step(action=action)
            rec.steps.append(dict(state=s_before, action=int(action.item()), next_state=ns.detach().numpy().copy(),
                                  reward=float(r.item()), done=float(d.item()), n_rand=len(rec.rand_action)))
            if record_q_gap:
                # how decisive was the choice?  Q(s) of the agent as it stood when it chose (learn() runs after the step): best minus second
                # best.  The long-horizon fixtures carry it so that a test can say WHY a greedy action of a replay differs, when one does.
                with torch.no_grad():
                    q = holder["agent"].model(torch.as_tensor(s_before, dtype=torch.float32).reshape(1, -1)).reshape(-1)
                top = torch.sort(q, descending=True).values
                rec.steps[-1]["q_gap"] = float((top[0] - top[1]).item())
            return ns, r, d

        env.step = rec_step
        random.random, np.random.randint = rec_random, rec_randint
        cls.reset, gspaces.Discrete.sample = rec_reset, rec_sample
        gw.select_agent = wrapped_select_agent
        train_reset_id = id(env.env.reset_env.env.unwrapped) if reward_env_type is None else id(env.env.real_env.unwrapped)
        try:
            rec.active = True
            # replicate calc_score but keep the per-episode lists (GTN_worker.py:187-209)
            agent = gw.select_agent(config=w.config, agent_name=w.agent_name)
            real_env = w.env_factory.generate_real_env()
            reward_list_train, episode_length_train, _ = agent.train(env=env, test_env=real_env, time_remaining=1e9)
            reward_list_test, _, _ = agent.test(env=real_env, time_remaining=1e9)
            rec.active = False
        finally:
            random.random, np.random.randint = orig_random, orig_randint
            cls.reset, gspaces.Discrete.sample = orig_reset, orig_sample
            gw.select_agent = orig_select_agent
    import statistics
    score = statistics.mean(reward_list_test)
    def pad4(rows):                       # the reset tapes carry four doubles per episode (MountainCar's state has two)
        rows = np.array(rows, np.float64).reshape(len(rows), -1)
        return np.concatenate([rows, np.zeros((rows.shape[0], 4 - rows.shape[1]))], axis=1) if rows.shape[1] < 4 else rows
    train_reset = pad4([s for (i, s) in rec.resets if i == train_reset_id])
    test_reset = pad4([s for (i, s) in rec.resets if i != train_reset_id])
    a = cfg["agents"][agent_key]
    B = holder["hp"].get("batch_size", a["batch_size"])
    n = len(rec.steps)
    explored = np.zeros(n, np.int32)
    prev = 0
    for k, st in enumerate(rec.steps):
        explored[k] = 1 if st["n_rand"] > prev else 0
        prev = st["n_rand"]
    import json
    extra = {}
    if record_q_gap:
        extra["tr_q_gap"] = np.array([s["q_gap"] for s in rec.steps], np.float32)
    if "icm_init" in holder:
        extra["icm_init"] = holder["icm_init"]
        extra["icm_final"] = np.concatenate([v.detach().cpu().numpy().astype(np.float32).reshape(-1)
                                             for v in holder["agent"].icm.model.state_dict().values()])
    if perturb_ulp:
        # only what the comparison needs: the perturbed run shares config, theta and tapes' PROVENANCE with the unperturbed fixture
        save(name, tr_action=np.array([s["action"] for s in rec.steps], np.int32), tr_explored=explored,
             tr_next_state=np.stack([s["next_state"] for s in rec.steps]).astype(np.float32),
             losses=np.array(rec.losses, np.float64), reward_list_train=np.array(reward_list_train, np.float64),
             reward_list_test=np.array(reward_list_test, np.float64), score=np.array(score))
        return
    save(name, config_json=np.array(json.dumps(cfg)), hp_json=np.array(json.dumps(holder["hp"])), **extra,
         theta=theta, agent_init=holder["init"],
         train_episodes=np.array(train_episodes), max_steps=np.array(cfg["envs"][env_name]["max_steps"]),
         tape_eps_uniform=np.array(rec.eps_uniform, np.float64), tape_rand_action=np.array(rec.rand_action, np.int32),
         tape_replay_idx=(np.stack(rec.replay_idx).astype(np.int32) if rec.replay_idx else np.zeros((0, B), np.int32)),
         tape_train_reset=train_reset, tape_test_reset=test_reset,
         tr_state=np.stack([s["state"] for s in rec.steps]).astype(np.float32),
         tr_action=np.array([s["action"] for s in rec.steps], np.int32), tr_explored=explored,
         tr_next_state=np.stack([s["next_state"] for s in rec.steps]).astype(np.float32),
         tr_reward=np.array([s["reward"] for s in rec.steps], np.float32),
         tr_done=np.array([s["done"] for s in rec.steps], np.float32),
         losses=np.array(rec.losses, np.float64),
         reward_list_train=np.array(reward_list_train, np.float64),
         episode_length_train=np.array(episode_length_train, np.int32),
         reward_list_test=np.array(reward_list_test, np.float64), score=np.array(score))


# ------------------------------------------------------------------------------------------------
# G1LN: build_nn_from_config with use_layer_norm (models/model_utils.py:22-37): forward of the reference's module
# ------------------------------------------------------------------------------------------------
def gen_g1ln():
    from models.model_utils import build_nn_from_config
    out = {}
    cases = [(5, 3, 24, 2, "relu"), (7, 2, 33, 3, "tanh"), (4, 4, 16, 1, "leakyrelu"), (6, 1, 40, 3, "leakyrelu")]
    for ci, (din, dout, H, L, act) in enumerate(cases):
        seed_all(1200 + ci)
        net = build_nn_from_config(din, dout, {"hidden_size": H, "hidden_layer": L, "activation_fn": act, "use_layer_norm": True})
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, torch.nn.LayerNorm):          # the default weight 1 / bias 0 would hide a swapped or missing affine
                    m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.3, 0.3)
        x = torch.randn(9, din) * 1.3
        with torch.no_grad():
            y = net(x)
        pre = "c%d_" % ci
        out[pre + "meta"] = np.array([din, dout, H, L, ["identity", "relu", "leakyrelu", "tanh", "prelu"].index(act)], np.int64)
        out[pre + "params"] = np.concatenate([p.detach().numpy().astype(np.float32).reshape(-1) for p in net.parameters()])
        out[pre + "keys"] = np.array(list(net.state_dict().keys()))
        out[pre + "x"] = x.numpy(); out[pre + "y"] = y.numpy()
    out["n_cases"] = np.array(len(cases))
    save("g1ln_mlp_layer_norm", **out)


# ------------------------------------------------------------------------------------------------
# G10: gridworld transition tables for every layout (envs/gridworld.py, pure reference code)
# ------------------------------------------------------------------------------------------------
def gen_g10():
    import envs.gridworld as gw
    out = {}
    names = ["EmptyRoom22", "EmptyRoom23", "EmptyRoom33", "WallRoom", "HoleRoom", "HoleRoomLarge", "HoleRoomLargeShifted", "Cliff"]
    for name in names:
        env = getattr(gw, name)()
        m, n = len(env.grid), len(env.grid[0])
        N = m * n
        nxt = np.zeros((N, 4), np.int32); rew = np.zeros((N, 4), np.float64); dn = np.zeros((N, 4), np.uint8)
        for s in range(N):
            for a in range(4):
                env.state = env._obs_to_state(s)
                if env.grid[env.state[0]][env.state[1]] == '#':
                    continue          # wall cells are unreachable
                obs, r, d, _ = env.step(a)
                nxt[s, a], rew[s, a], dn[s, a] = obs, r, d
        out[name + "_next"] = nxt; out[name + "_reward"] = rew; out[name + "_done"] = dn
        out[name + "_start"] = np.array(env.reset())
        out[name + "_walls"] = np.array([env.grid[s // n][s % n] == '#' for s in range(N)])
    save("g10_gridworld_tables", names=np.array(names), **out)


# ------------------------------------------------------------------------------------------------
# G2: RewardEnv._calc_reward on Cliff for the info-free types (envs/reward_env.py:67-133)
# ------------------------------------------------------------------------------------------------
def gen_g2():
    from envs.env_factory import EnvFactory
    out = {}
    types = [0, 1, 2, 5, 6]
    for t in types:
        for act, layers in (("prelu", 1), ("tanh", 2)):
            cfg = load_cfg("default_config_gridworld_reward_env.yaml")
            cfg["envs"]["Cliff"]["reward_env_type"] = t
            cfg["envs"]["Cliff"]["activation_fn"] = act
            cfg["envs"]["Cliff"]["hidden_layer"] = layers
            seed_all(200 + t)
            with quiet():
                renv = EnvFactory(cfg).generate_reward_env()
            renv.set_agent_params(same_action_num=1, gamma=0.8)
            theta = pack_linear_params(renv.state_dict(), "env.reward_net.") if t != 0 else np.zeros(1, np.float32)
            if t == 0:
                theta = pack_linear_params(renv.state_dict(), "env.reward_net.")
            vals = np.zeros((48, 4), np.float64)
            with torch.no_grad():
                for s in range(48):
                    for a in range(4):
                        renv.env.real_env.reset()
                        renv.env.real_env.env.state = renv.env.real_env.env._obs_to_state(s)
                        renv.env.state = s
                        _, r, _, _ = renv.env.step(a)
                        vals[s, a] = r
            key = "t%d_%s%d_" % (t, act, layers)
            out[key + "theta"] = theta
            out[key + "shaped"] = vals
    save("g2_reward_env_cliff", types=np.array(types), **out)



# ------------------------------------------------------------------------------------------------
# G2F: RewardEnv.step on the vector-state stand-in, all 11 reward types incl. the info-vector ones
# ------------------------------------------------------------------------------------------------
def gen_g2f():
    import json
    from envs.env_factory import EnvFactory
    out = {}
    types = [0, 1, 2, 3, 4, 5, 6, 7, 8, 101, 102]
    n_steps = 6
    for t in types:
        cfg = _td3_cfg(env_over={"reward_env_type": t, "hidden_size": 24, "max_steps": n_steps})
        seed_all(2600 + t)
        with quiet():
            renv = EnvFactory(cfg).generate_reward_env()
        renv.set_agent_params(same_action_num=1, gamma=0.98)
        sd = renv.state_dict()
        if t in (101, 102):
            theta = sd["env.reward_net.weight"].numpy().reshape(-1).astype(np.float32)
        else:
            theta = pack_linear_params(sd, "env.reward_net.")
        rng = np.random.RandomState(77 + t)
        s0 = renv.reset().numpy().astype(np.float64)          # fp32 view of the env's fp64 state
        s0_exact = np.array(renv.env.real_env.env.state, np.float64)
        acts = rng.uniform(-1, 1, (n_steps, 6)).astype(np.float32)
        ns, rs, ds = [], [], []
        with torch.no_grad():
            for k in range(n_steps):
                n_, r_, d_ = renv.step(torch.from_numpy(acts[k].copy()))
                ns.append(n_.numpy().copy()); rs.append(float(r_)); ds.append(float(d_))
        key = "t%d_" % t
        out[key + "theta"] = theta
        out[key + "reset_state"] = s0_exact
        out[key + "actions"] = acts
        out[key + "next_states"] = np.array(ns, np.float32)
        out[key + "shaped"] = np.array(rs, np.float64)
        out[key + "done"] = np.array(ds, np.float32)
        out[key + "sd_keys"] = np.array(list(sd.keys()))
    save("g2f_reward_env_cheetah_info", types=np.array(types), hidden=24, gamma=0.98, config_json=json.dumps(_td3_cfg(env_over={"hidden_size": 24, "max_steps": n_steps})), **out)


# ------------------------------------------------------------------------------------------------
# CKPT: a reference-format checkpoint {'model': state_dict, 'config': dict} (GTN_master.py:133-139) + step outputs
# ------------------------------------------------------------------------------------------------
def gen_ckpt():
    from envs.env_factory import EnvFactory
    cfg = load_cfg("default_config_cartpole_syn_env.yaml")
    cfg["agents"]["ddqn"].update(train_episodes=3, test_episodes=3, init_episodes=1, print_rate=int(1e9))
    cfg["envs"]["CartPole-v0"]["max_steps"] = 30
    seed_all(4100)
    with quiet():
        venv = EnvFactory(cfg).generate_virtual_env()
    with torch.no_grad():
        venv.env.done_net[-1].bias.fill_(-5.0)
    path = os.path.join(OUT, "ckpt_cartpole_se_reference.pt")
    torch.save({'model': venv.state_dict(), 'config': cfg}, path)           # exactly GTN_Master.save_model's payload
    # what the reference computes with the re-loaded checkpoint (experiments/syn_env_evaluate_cartpole_vary_hp_2.py:12-23)
    sd = torch.load(path)
    with quiet():
        v2 = EnvFactory(sd['config']).generate_virtual_env()
    v2.load_state_dict(sd['model'])
    rng = np.random.RandomState(9)
    states = rng.uniform(-0.5, 0.5, (8, 4)).astype(np.float32)
    actions = rng.randint(0, 2, 8)
    ns, rs, ds = [], [], []
    with torch.no_grad():
        for k in range(8):
            n_, r_, d_ = v2.step(action=torch.tensor([float(actions[k])]), state=torch.from_numpy(states[k]))
            ns.append(n_.numpy().copy()); rs.append(float(r_)); ds.append(float(d_))
    save("ckpt_cartpole_se_reference_steps", states=states, actions=actions.astype(np.int32), next_states=np.array(ns, np.float32),
         rewards=np.array(rs, np.float32), dones=np.array(ds, np.float32), theta=se_theta(v2))
    print("wrote", path, os.path.getsize(path))

# ------------------------------------------------------------------------------------------------
# G9: full GTN_Worker.calc_score on the Cliff RewardEnv with QL (cfg 4) + tapes + per-step trace
# ------------------------------------------------------------------------------------------------
def gen_g9(name, seed, eps_over=None, agent_name="QL", agent_over=None, env_over=None):
    import json
    import statistics
    import agents.GTN_worker as gw
    from agents.GTN import GTN_Worker
    import gym.spaces as gspaces
    cfg = load_cfg("default_config_gridworld_reward_env.yaml")
    sec = "sarsa" if agent_name.lower().startswith("sarsa") else "ql"
    cfg["agents"]["gtn"]["agent_name"] = agent_name
    cfg["agents"][sec]["print_rate"] = int(1e9)
    cfg["agents"][sec].update(agent_over or {})
    cfg["envs"][cfg["env_name"]].update(env_over or {})
    if eps_over is not None:
        cfg["agents"][sec]["eps_init"] = eps_over
        cfg["agents"][sec]["eps_min"] = eps_over
    rec = Recorder()
    orig_random = random.random
    orig_sample = gspaces.Discrete.sample

    def rec_random():
        v = orig_random()
        if rec.active:
            rec.eps_uniform.append(v)
        return v

    def rec_sample(self):
        v = orig_sample(self)
        if rec.active:
            rec.rand_action.append(v)
        return v

    with quiet():
        w = GTN_Worker(id=0, bohb_id=0)
        seed_all(seed)
        w.config = cfg
        w.late_init(cfg)
        w.timeout = 1e9
        env = w.synthetic_env_orig
        theta = pack_linear_only(env.state_dict(), "env.reward_net.")
        orig_step = env.step

        def rec_step(action, state=None):
            s_before = int(env.env.state)
            ns, r, d = orig_step(action=action, state=state)
            rec.steps.append(dict(state=s_before, action=int(action.item()), next_state=int(ns.item()), reward=float(r.item()),
                                  done=float(d.item()), n_rand=len(rec.rand_action)))
            return ns, r, d

        env.step = rec_step
        random.random = rec_random
        gspaces.Discrete.sample = rec_sample
        try:
            rec.active = True
            agent = gw.select_agent(config=w.config, agent_name=w.agent_name)
            real_env = w.env_factory.generate_real_env()
            reward_list_train, episode_length_train, _ = agent.train(env=env, test_env=real_env, time_remaining=1e9)
            reward_list_test, _, _ = agent.test(env=real_env, time_remaining=1e9)
            rec.active = False
        finally:
            random.random = orig_random
            gspaces.Discrete.sample = orig_sample
    explored = np.zeros(len(rec.steps), np.int32)
    prev = 0
    for k, st in enumerate(rec.steps):
        explored[k] = 1 if st["n_rand"] > prev else 0
        prev = st["n_rand"]
    # the reference's own shaped-reward table for this theta (what RewardEnv.step returned for every (s,a))
    shaped = np.zeros((48, 4), np.float32)
    env.step = orig_step
    with torch.no_grad():
        for s_ in range(48):
            for a_ in range(4):
                env.env.real_env.reset()
                env.env.real_env.env.state = env.env.real_env.env._obs_to_state(s_)
                env.env.state = s_
                shaped[s_, a_] = env.env.step(a_)[1]
    save(name, config_json=np.array(json.dumps(cfg)), theta=theta, shaped_ref=shaped,
         tape_eps_uniform=np.array(rec.eps_uniform, np.float64), tape_rand_action=np.array(rec.rand_action, np.int32),
         tr_state=np.array([s["state"] for s in rec.steps], np.int32), tr_action=np.array([s["action"] for s in rec.steps], np.int32),
         tr_explored=explored, tr_next_state=np.array([s["next_state"] for s in rec.steps], np.int32),
         tr_reward=np.array([s["reward"] for s in rec.steps], np.float32), tr_done=np.array([s["done"] for s in rec.steps], np.float32),
         q_table=np.array(agent.q_table, np.float64), reward_list_train=np.array(reward_list_train, np.float64),
         episode_length_train=np.array(episode_length_train, np.int32), reward_list_test=np.array(reward_list_test, np.float64),
         score=np.array(statistics.mean(reward_list_test)))


# ------------------------------------------------------------------------------------------------
# G3d / G4d: Critic_DuelingDQN forward (global advantage mean) and DuelingDDQN.learn steps
# ------------------------------------------------------------------------------------------------
def _pack_dueling(sd, prefix=""):
    return np.concatenate([pack_linear_params(sd, prefix + "feature_stream."), pack_linear_params(sd, prefix + "value_stream."),
                           pack_linear_params(sd, prefix + "advantage_stream.")])


def gen_g4d():
    from agents.DuelingDDQN import DuelingDDQN
    from envs.env_factory import EnvFactory
    from utils import ReplayBuffer
    out = {}
    variants = [("default_config_acrobot.yaml", {"hidden_size": 24, "feature_dim": 16, "batch_size": 32}, 3),
                ("default_config_acrobot.yaml", {"hidden_size": 20, "feature_dim": 12, "batch_size": 16, "hidden_layer": 1,
                                                 "activation_fn": "tanh"}, 3),
                ("default_config_cartpole_syn_env.yaml", {"hidden_size": 16, "feature_dim": 16, "batch_size": 24}, 2)]
    for vi, (yml, over, nsteps) in enumerate(variants):
        cfg = load_cfg(yml)
        cfg["agents"]["duelingddqn"].update(over)
        seed_all(450 + vi)
        with quiet():
            fac = EnvFactory(cfg)
            real_env = fac.generate_real_env()
            agent = DuelingDDQN(env=real_env, config=cfg)
        S, A = real_env.get_state_dim(), real_env.get_action_dim()
        a = cfg["agents"]["duelingddqn"]
        B = a["batch_size"]
        rb = ReplayBuffer(state_dim=S, action_dim=1, device="cpu", max_size=300)
        for i in range(200):
            rb.add(torch.randn(S) * 0.5, torch.tensor([float(np.random.randint(A))]), torch.randn(S) * 0.5,
                   torch.randn(1) * 0.3 - 0.5, torch.randn(1) * 0.2)
        with torch.no_grad():
            for p in agent.model_target.parameters():
                p.add_(torch.randn_like(p) * 0.05)
        pre = "v%d_" % vi
        out[pre + "meta"] = np.array([S, A, a["hidden_size"], a["hidden_layer"], a["feature_dim"],
                                      ["identity", "relu", "leakyrelu", "tanh", "prelu"].index(a["activation_fn"]), B, nsteps], np.int64)
        out[pre + "hparams"] = np.array([a["gamma"], a["lr"], a["tau"]], np.float64)
        out[pre + "online0"] = _pack_dueling(agent.model.state_dict())
        out[pre + "target0"] = _pack_dueling(agent.model_target.state_dict())
        x = torch.randn(9, S)
        with torch.no_grad():
            out[pre + "fwd_x"] = x.numpy()
            out[pre + "fwd_q"] = agent.model(x).numpy()
            out[pre + "fwd_q_single"] = torch.stack([agent.model(x[i]) for i in range(4)]).numpy()
        rows_all, onl, tgt, losses = [], [], [], []
        for step in range(nsteps):
            idx = np.random.randint(0, rb.size, size=B)
            rb.sample = lambda batch_size, _idx=idx: rb._sample_idx(_idx)
            rows = np.concatenate([rb.state[idx].numpy(), rb.action[idx].numpy(), rb.next_state[idx].numpy(),
                                   rb.reward[idx].numpy(), rb.done[idx].numpy()], axis=1)
            loss = agent.learn(rb, real_env, episode=50)
            rows_all.append(rows); losses.append(float(loss.item()))
            onl.append(_pack_dueling(agent.model.state_dict()))
            tgt.append(_pack_dueling(agent.model_target.state_dict()))
        out[pre + "rows"] = np.stack(rows_all).astype(np.float32)
        out[pre + "loss"] = np.array(losses, np.float64)
        out[pre + "online"] = np.stack(onl)
        out[pre + "target"] = np.stack(tgt)
    out["n_variants"] = np.array(len(variants))
    save("g4d_dueling_learn", **out)


# ------------------------------------------------------------------------------------------------
# config 5: TD3 on the HalfCheetah stand-in RewardEnv.  G4t: TD3.learn steps; G8t: full calc_score with tapes
# ------------------------------------------------------------------------------------------------
def _pack_td3(agent, target=False):
    if target:
        nets = (agent.actor_target, agent.critic_target_1, agent.critic_target_2)
    else:
        nets = (agent.actor, agent.critic_1, agent.critic_2)
    return np.concatenate([pack_linear_params(n.state_dict(), "net.") for n in nets])


def _td3_cfg(over=None, env_over=None, cfg_yaml="default_config_halfcheetah_reward_env.yaml", env_name="HalfCheetah-v3"):
    cfg = load_cfg(cfg_yaml)
    cfg["agents"]["td3"].update(over or {})
    cfg["agents"]["td3"]["print_rate"] = int(1e9)
    cfg["envs"][env_name].update(env_over or {})
    return cfg


def gen_g4t():
    from agents.TD3 import TD3
    from envs.env_factory import EnvFactory
    from utils import ReplayBuffer
    out = {}
    variants = [({"hidden_size": 24, "batch_size": 16, "policy_delay": 1}, 3), ({"hidden_size": 20, "batch_size": 12, "policy_delay": 2,
                                                                                 "hidden_layer": 1, "activation_fn": "tanh"}, 4)]
    for vi, (over, nsteps) in enumerate(variants):
        cfg = _td3_cfg(over)
        seed_all(470 + vi)
        with quiet():
            fac = EnvFactory(cfg)
            real_env = fac.generate_real_env()
            agent = TD3(env=real_env, max_action=real_env.get_max_action(), config=cfg)
        a = cfg["agents"]["td3"]
        B, S, A = a["batch_size"], 17, 6
        rb = ReplayBuffer(state_dim=S, action_dim=A, device="cpu", max_size=300)
        for i in range(120):
            rb.add(torch.randn(S) * 0.5, torch.rand(A) * 2 - 1, torch.randn(S) * 0.5, torch.randn(1) * 0.3, torch.zeros(1) + (i % 17 == 0))
        with torch.no_grad():
            for net in (agent.actor_target, agent.critic_target_1, agent.critic_target_2):
                for p in net.parameters():
                    p.add_(torch.randn_like(p) * 0.03)
        pre = "v%d_" % vi
        out[pre + "meta"] = np.array([a["hidden_size"], a["hidden_layer"], ["identity", "relu", "leakyrelu", "tanh", "prelu"].index(a["activation_fn"]),
                                      B, a["policy_delay"], nsteps], np.int64)
        out[pre + "hparams"] = np.array([a["gamma"], a["lr"], a["tau"], a["policy_std"], a["policy_std_clip"]], np.float64)
        out[pre + "params0"] = _pack_td3(agent)
        out[pre + "targets0"] = _pack_td3(agent, True)
        x = torch.randn(5, S); act = torch.rand(5, A) * 2 - 1
        with torch.no_grad():
            out[pre + "fwd_s"] = x.numpy(); out[pre + "fwd_a"] = act.numpy()
            out[pre + "fwd_actor"] = agent.actor(x).numpy()
            out[pre + "fwd_critic1"] = agent.critic_1(x, act).numpy().reshape(-1)
        rows_all, noises, pars, tars = [], [], [], []
        orig_randn_like = torch.randn_like
        for step in range(nsteps):
            idx = np.random.randint(0, rb.size, size=B)
            rb.sample = lambda batch_size, _idx=idx: rb._sample_idx(_idx)
            rows = np.concatenate([rb.state[idx].numpy(), rb.action[idx].numpy(), rb.next_state[idx].numpy(),
                                   rb.reward[idx].numpy(), rb.done[idx].numpy()], axis=1)
            holder = {}

            def rec_randn_like(t, *a_, **k_):
                v = orig_randn_like(t, *a_, **k_)
                holder["n"] = v.numpy().copy()
                return v
            torch.randn_like = rec_randn_like
            try:
                agent.learn(rb, real_env, episode=50)
            finally:
                torch.randn_like = orig_randn_like
            rows_all.append(rows); noises.append(holder["n"])
            pars.append(_pack_td3(agent)); tars.append(_pack_td3(agent, True))
        out[pre + "rows"] = np.stack(rows_all).astype(np.float32)
        out[pre + "policy_noise"] = np.stack(noises).astype(np.float32)
        out[pre + "params"] = np.stack(pars)
        out[pre + "targets"] = np.stack(tars)
    out["n_variants"] = np.array(len(variants))
    save("g4t_td3_learn", **out)


def gen_g8t(name, seed, agent_over=None, env_over=None, vary_seed=None, icm_over=None, virtual=False,
            cfg_yaml="default_config_halfcheetah_reward_env.yaml", env_name="HalfCheetah-v3", env_cls="CheetahStandinEnv", perturb_ulp=False):
    import json
    import statistics
    import agents.GTN_worker as gw
    from agents.GTN import GTN_Worker
    import gym.envs as genvs
    import gym.spaces as gspaces
    cfg = _td3_cfg(agent_over or {"train_episodes": 4, "init_episodes": 2, "batch_size": 16, "hidden_size": 24, "test_episodes": 2},
                   env_over or {"max_steps": 7, "hidden_size": 20}, cfg_yaml=cfg_yaml, env_name=env_name)
    drawn = {}
    if vary_seed is None and cfg["agents"]["gtn"]["agent_name"].lower() == "td3_vary":
        # default_config_pendulum.yaml / default_config_halfcheetah.yaml name td3_vary without shipping its section (TD3_vary.py:16 raises
        # KeyError there); with vary_hp off the agent IS TD3 on the yaml's td3 section (TD3_vary.py:16-21)
        cfg["agents"]["gtn"]["agent_name"] = "td3"
    cfg["device"] = "cpu"                         # (default_config_pendulum.yaml ships cuda:0)
    if virtual:                                   # default_config_halfcheetah.yaml: TD3 on a VirtualEnv (synthetic_env_type 0)
        cfg["agents"]["gtn"]["synthetic_env_type"] = 0
    if icm_over is not None:                      # select_agent "td3_icm": TD3(icm=True), agents/TD3.py:44-60,68-70
        cfg["agents"]["gtn"]["agent_name"] = "td3_icm"
        cfg["agents"].setdefault("icm", {"lr": 1e-4, "beta": 0.2, "eta": 0.5, "feature_dim": 32, "hidden_size": 128}).update(icm_over)
    if vary_seed is not None:
        # TD3_vary (agents/TD3_vary.py:24-58): the draw of the ConfigSpace stand-in is recorded in the fixture
        import ConfigSpace
        ConfigSpace.RANDOM.seed(vary_seed)
        cfg["agents"]["gtn"]["agent_name"] = "TD3_vary"
        cfg["agents"]["td3_vary"] = {"vary_hp": True}
        orig_sample = ConfigSpace.ConfigurationSpace.sample_configuration

        def rec_sample(self):
            d = orig_sample(self)
            drawn.update(d)
            return d
        ConfigSpace.ConfigurationSpace.sample_configuration = rec_sample
    rec = dict(rand=[], act_noise=[], test_noise=[], policy_noise=[], replay=[], resets=[], steps=[], purpose=None, active=False)
    orig_randn, orig_randn_like, orig_randint = torch.randn, torch.randn_like, np.random.randint
    env_class = getattr(genvs, env_cls)
    orig_box_sample, orig_reset = gspaces.Box.sample, env_class.reset

    def rec_randn(*a, **k):
        v = orig_randn(*a, **k)
        if rec["active"] and rec["purpose"] in ("act_noise", "test_noise"):
            rec[rec["purpose"]].append(v.numpy().copy())
        return v

    def rec_randn_like(t, *a, **k):
        v = orig_randn_like(t, *a, **k)
        if rec["active"] and rec["purpose"] == "learn":
            rec["policy_noise"].append(v.numpy().copy())
        return v

    def rec_randint(*a, **k):
        v = orig_randint(*a, **k)
        if rec["active"]:
            rec["replay"].append(np.asarray(v).copy())
        return v

    def rec_box_sample(self):
        v = orig_box_sample(self)
        if rec["active"]:
            rec["rand"].append(np.asarray(v).copy())
        return v

    def rec_reset(self):
        obs = orig_reset(self)
        if rec["active"]:
            rec["resets"].append((id(self), np.array(self.state, np.float64).copy()))
        return obs

    orig_select_agent = gw.select_agent
    holder = {}

    def wrapped_select_agent(config, agent_name):
        agent = orig_select_agent(config=config, agent_name=agent_name)
        if perturb_ulp:                            # (see gen_g8: the reference against itself, one ulp apart)
            with torch.no_grad():
                for net, tgt in ((agent.actor, agent.actor_target), (agent.critic_1, agent.critic_target_1), (agent.critic_2, agent.critic_target_2)):
                    for prm in net.parameters():
                        prm.copy_(torch.from_numpy(np.nextafter(prm.numpy(), np.float32(np.inf))))
                    tgt.load_state_dict(net.state_dict())
        holder["init"] = _pack_td3(agent)
        holder["agent"] = agent
        if getattr(agent, "icm", None):
            holder["icm_init"] = np.concatenate([v.detach().cpu().numpy().astype(np.float32).reshape(-1)
                                                 for v in agent.icm.model.state_dict().values()])

        def wrap(fn, purpose):
            def inner(*a, **k):
                prev = rec["purpose"]
                rec["purpose"] = purpose
                try:
                    return fn(*a, **k)
                finally:
                    rec["purpose"] = prev
            return inner
        agent.select_train_action = wrap(agent.select_train_action, "act_noise")
        agent.select_test_action = wrap(agent.select_test_action, "test_noise")
        agent.learn = wrap(agent.learn, "learn")
        return agent

    with quiet():
        w = GTN_Worker(id=0, bohb_id=0)
        seed_all(seed)
        w.config = cfg
        w.late_init(cfg)
        w.timeout = 1e9
        env = w.synthetic_env_orig
        theta = se_theta(env) if virtual else pack_linear_only(env.state_dict(), "env.reward_net.")
        orig_step = env.step

        def rec_step(action, state=None):
            s_before = env.env.state.detach().numpy().astype(np.float32).copy() if virtual \
                else np.asarray(env.env.state, np.float64).astype(np.float32)
            ns, r, d = orig_step(action=action, state=state)
            rec["steps"].append(dict(state=s_before, action=action.detach().numpy().astype(np.float32).copy(), next_state=ns.detach().numpy().copy(),
                                     reward=float(r.item()), done=float(d.item())))
            return ns, r, d
        env.step = rec_step
        torch.randn, torch.randn_like, np.random.randint = rec_randn, rec_randn_like, rec_randint
        gspaces.Box.sample, env_class.reset = rec_box_sample, rec_reset
        gw.select_agent = wrapped_select_agent
        train_reset_id = id(env.env.reset_env.env.unwrapped) if virtual else id(env.env.real_env.unwrapped)
        try:
            rec["active"] = True
            agent = gw.select_agent(config=w.config, agent_name=w.agent_name)
            real_env = w.env_factory.generate_real_env()
            reward_list_train, episode_length_train, _ = agent.train(env=env, test_env=real_env, time_remaining=1e9)
            reward_list_test, _, _ = agent.test(env=real_env, time_remaining=1e9)
            rec["active"] = False
        finally:
            torch.randn, torch.randn_like, np.random.randint = orig_randn, orig_randn_like, orig_randint
            gspaces.Box.sample, env_class.reset = orig_box_sample, orig_reset
            gw.select_agent = orig_select_agent
    if vary_seed is not None:
        ConfigSpace.ConfigurationSpace.sample_configuration = orig_sample
    extra = {}
    if "icm_init" in holder:
        extra["icm_init"] = holder["icm_init"]
        extra["icm_final"] = np.concatenate([v.detach().cpu().numpy().astype(np.float32).reshape(-1)
                                             for v in holder["agent"].icm.model.state_dict().values()])
    if perturb_ulp:
        save(name, tr_action=np.stack([s["action"] for s in rec["steps"]]), tr_next_state=np.stack([s["next_state"] for s in rec["steps"]]).astype(np.float32),
             tr_reward=np.array([s["reward"] for s in rec["steps"]], np.float32), reward_list_train=np.array(reward_list_train, np.float64),
             reward_list_test=np.array(reward_list_test, np.float64), score=np.array(statistics.mean(reward_list_test)), final_params=_pack_td3(agent))
        return
    save(name, config_json=np.array(json.dumps(cfg)), hp_json=np.array(json.dumps(drawn)), theta=theta, agent_init=holder["init"], **extra,
         tape_rand_action=np.stack(rec["rand"][1::2]).astype(np.float32),          # get_random_action samples twice, returns the 2nd
         tape_act_noise=np.stack(rec["act_noise"]).astype(np.float32), tape_test_noise=np.stack(rec["test_noise"]).astype(np.float32),
         tape_policy_noise=np.stack(rec["policy_noise"]).astype(np.float32).reshape(-1, np.stack(rec["rand"]).shape[-1]),
         tape_replay_idx=np.stack(rec["replay"]).astype(np.int32),
         tape_train_reset=np.array([s for (i, s) in rec["resets"] if i == train_reset_id]),
         tape_test_reset=np.array([s for (i, s) in rec["resets"] if i != train_reset_id]),
         tr_state=np.stack([s["state"] for s in rec["steps"]]), tr_action=np.stack([s["action"] for s in rec["steps"]]),
         tr_next_state=np.stack([s["next_state"] for s in rec["steps"]]).astype(np.float32),
         tr_reward=np.array([s["reward"] for s in rec["steps"]], np.float32),
         reward_list_train=np.array(reward_list_train, np.float64), episode_length_train=np.array(episode_length_train, np.int32),
         reward_list_test=np.array(reward_list_test, np.float64), score=np.array(statistics.mean(reward_list_test)),
         final_params=_pack_td3(agent))


# ------------------------------------------------------------------------------------------------
# G4TD / G8TD: TD3_discrete_vary (agents/TD3_discrete_vary.py) -- learn calls on a fixed buffer, and whole calc_score runs on
# a CartPole / Acrobot VirtualEnv through GTN_Worker.  The Gumbel(0,1) draws of F.gumbel_softmax are recorded where torch
# makes them (Tensor.exponential_, then -log as torch computes it) next to the Gaussian noises, replay indices and resets.
# ------------------------------------------------------------------------------------------------
def _params_flat(net):
    """Module.parameters() order (the shared LayerNorm appears once, behind the second Linear)."""
    return np.concatenate([p.detach().cpu().numpy().astype(np.float32).reshape(-1) for p in net.parameters()])


def _pack_td3d(agent, target=False):
    nets = (agent.actor_target, agent.critic_target_1, agent.critic_target_2) if target else (agent.actor, agent.critic_1, agent.critic_2)
    return np.concatenate([_params_flat(n) for n in nets])


def _td3d_cfg(cfg_yaml, env_name, over=None, env_over=None):
    cfg = load_cfg(cfg_yaml)
    cfg["agents"]["td3_discrete_vary"].update(over or {})
    cfg["agents"]["td3_discrete_vary"]["print_rate"] = int(1e9)
    cfg["envs"][env_name].update(env_over or {})
    cfg["agents"]["gtn"]["agent_name"] = "TD3_discrete_vary"
    cfg["agents"]["gtn"]["synthetic_env_type"] = 0
    return cfg


class _GumbelTap(object):
    """Records every Tensor.exponential_() result as the Gumbel draw torch derives from it (-log(e))."""
    def __init__(self):
        self.orig = torch.Tensor.exponential_
        self.sink = None

    def __enter__(self):
        tap = self

        def rec_exponential(t, *a, **k):
            v = tap.orig(t, *a, **k)
            if tap.sink is not None:
                tap.sink.append((-(v.clone().log())).numpy().copy())
            return v
        torch.Tensor.exponential_ = rec_exponential
        return self

    def __exit__(self, *exc):
        torch.Tensor.exponential_ = self.orig


def gen_g4td():
    from agents.TD3_discrete_vary import TD3_discrete_vary
    from envs.env_factory import EnvFactory
    from utils import ReplayBuffer
    out = {}
    variants = [("default_config_cartpole_syn_env.yaml", "CartPole-v0", {"hidden_size": 24, "batch_size": 16, "policy_delay": 1, "hidden_layer": 2,
                                                                        "use_layer_norm": True, "gumbel_softmax_hard": False}, 4),
                ("default_config_acrobot_syn_env.yaml", "Acrobot-v1", {"hidden_size": 20, "batch_size": 12, "policy_delay": 2, "hidden_layer": 3,
                                                                      "use_layer_norm": True, "activation_fn": "tanh", "gumbel_softmax_hard": True}, 4),
                ("default_config_cartpole_syn_env.yaml", "CartPole-v0", {"hidden_size": 18, "batch_size": 10, "policy_delay": 1, "hidden_layer": 2,
                                                                        "activation_fn": "leakyrelu", "gumbel_softmax_temp": 0.7}, 3)]
    for vi, (yaml_name, env_name, over, nsteps) in enumerate(variants):
        over = dict(over, vary_hp=False)
        cfg = _td3d_cfg(yaml_name, env_name, over)
        seed_all(1470 + vi)
        with quiet():
            fac = EnvFactory(cfg)
            real_env = fac.generate_real_env()
            agent = TD3_discrete_vary(env=real_env, min_action=real_env.get_min_action(), max_action=real_env.get_max_action(), config=cfg)
        a = cfg["agents"]["td3_discrete_vary"]
        B, S, A = a["batch_size"], real_env.get_state_dim(), real_env.get_action_dim()
        rb = ReplayBuffer(state_dim=S, action_dim=A, device="cpu", max_size=300)
        for i in range(120):
            rb.add(torch.randn(S) * 0.5, torch.softmax(torch.randn(A) * 2, 0) + torch.randn(A) * 0.1, torch.randn(S) * 0.5, torch.randn(1) * 0.3,
                   torch.zeros(1) + (i % 17 == 0))
        with torch.no_grad():
            for net in (agent.actor_target, agent.critic_target_1, agent.critic_target_2):
                for p in net.parameters():
                    p.add_(torch.randn_like(p) * 0.03)
            for net in (agent.actor, agent.critic_1, agent.critic_2, agent.actor_target, agent.critic_target_1, agent.critic_target_2):
                for m in net.modules():
                    if isinstance(m, torch.nn.LayerNorm):      # default 1 / 0 would hide a swapped or missing affine
                        m.weight.uniform_(0.6, 1.4); m.bias.uniform_(-0.2, 0.2)
        agent.total_it = 3 * vi                                  # a temperature off the schedule's first entry
        pre = "v%d_" % vi
        out[pre + "meta"] = np.array([S, A, a["hidden_size"], a["hidden_layer"], ["identity", "relu", "leakyrelu", "tanh", "prelu"].index(a["activation_fn"]),
                                      B, a["policy_delay"], nsteps, int(bool(a.get("use_layer_norm", False))), int(bool(a["gumbel_softmax_hard"])),
                                      agent.total_it], np.int64)
        out[pre + "hparams"] = np.array([a["gamma"], a["lr"], a["tau"], a["policy_std"], a["policy_std_clip"], float(real_env.get_max_action()),
                                         a["gumbel_softmax_temp"]], np.float64)
        out[pre + "params0"] = _pack_td3d(agent)
        out[pre + "targets0"] = _pack_td3d(agent, True)
        # forward of the actor with recorded Gumbel draws
        x = torch.randn(5, S)
        sink = []
        with _GumbelTap() as tap, torch.no_grad():
            tap.sink = sink
            out[pre + "fwd_actor"] = agent.actor(x, 0.8).numpy()
        out[pre + "fwd_s"] = x.numpy(); out[pre + "fwd_gumbel"] = sink[0]
        rows_all, noises, gts, gas, pars, tars = [], [], [], [], [], []
        orig_randn_like = torch.randn_like
        for step in range(nsteps):
            idx = np.random.randint(0, rb.size, size=B)
            rb.sample = lambda batch_size, _idx=idx: rb._sample_idx(_idx)
            rows = np.concatenate([rb.state[idx].numpy(), rb.action[idx].numpy(), rb.next_state[idx].numpy(),
                                   rb.reward[idx].numpy(), rb.done[idx].numpy()], axis=1)
            holder = {}

            def rec_randn_like(t, *a_, **k_):
                v = orig_randn_like(t, *a_, **k_)
                holder["n"] = v.numpy().copy()
                return v
            torch.randn_like = rec_randn_like
            sink = []
            try:
                with _GumbelTap() as tap:
                    tap.sink = sink
                    agent.learn(rb, real_env, episode=50)
            finally:
                torch.randn_like = orig_randn_like
            rows_all.append(rows); noises.append(holder["n"]); gts.append(sink[0])
            gas.append(sink[1] if len(sink) > 1 else np.zeros_like(sink[0]))
            pars.append(_pack_td3d(agent)); tars.append(_pack_td3d(agent, True))
        out[pre + "rows"] = np.stack(rows_all).astype(np.float32)
        out[pre + "policy_noise"] = np.stack(noises).astype(np.float32)
        out[pre + "gumbel_target"] = np.stack(gts).astype(np.float32)
        out[pre + "gumbel_actor"] = np.stack(gas).astype(np.float32)
        out[pre + "params"] = np.stack(pars)
        out[pre + "targets"] = np.stack(tars)
    out["n_variants"] = np.array(len(variants))
    save("g4td_td3_discrete_learn", **out)


def gen_g8td(name, seed, cfg_yaml="default_config_cartpole_syn_env.yaml", env_name="CartPole-v0", env_cls="CartPoleEnv", agent_over=None,
             env_over=None, done_bias_shift=0.0, vary_seed=None):
    import json
    import statistics
    import agents.GTN_worker as gw
    from agents.GTN import GTN_Worker
    import gym.envs as genvs
    import gym.spaces as gspaces
    over = dict(train_episodes=4, init_episodes=2, batch_size=16, hidden_size=24, hidden_layer=2, test_episodes=2, vary_hp=False)
    over.update(agent_over or {})
    cfg = _td3d_cfg(cfg_yaml, env_name, over, env_over or {"max_steps": 7, "hidden_size": 20})
    drawn = {}
    if vary_seed is not None:
        import ConfigSpace
        ConfigSpace.RANDOM.seed(vary_seed)
        cfg["agents"]["td3_discrete_vary"]["vary_hp"] = True
        orig_cs_sample = ConfigSpace.ConfigurationSpace.sample_configuration

        def rec_cs_sample(self):
            d = orig_cs_sample(self)
            drawn.update(d)
            return d
        ConfigSpace.ConfigurationSpace.sample_configuration = rec_cs_sample
    rec = dict(rand=[], act_noise=[], test_noise=[], policy_noise=[], replay=[], resets=[], steps=[], gumbel_act=[], gumbel_test=[],
               gumbel_learn=[], learn_calls=[], purpose=None, active=False)
    orig_randn, orig_randn_like, orig_randint = torch.randn, torch.randn_like, np.random.randint
    env_class = getattr(genvs, env_cls)
    orig_disc_sample, orig_reset = gspaces.Discrete.sample, env_class.reset

    def rec_randn(*a, **k):
        v = orig_randn(*a, **k)
        if rec["active"] and rec["purpose"] in ("act_noise", "test_noise"):
            rec[rec["purpose"]].append(v.numpy().copy())
        return v

    def rec_randn_like(t, *a, **k):
        v = orig_randn_like(t, *a, **k)
        if rec["active"] and rec["purpose"] == "learn":
            rec["policy_noise"].append(v.numpy().copy())
        return v

    def rec_randint(*a, **k):
        v = orig_randint(*a, **k)
        if rec["active"]:
            rec["replay"].append(np.asarray(v).copy())
        return v

    def rec_disc_sample(self):
        v = orig_disc_sample(self)
        if rec["active"]:
            rec["rand"].append(int(v))
        return v

    def rec_reset(self):
        obs = orig_reset(self)
        if rec["active"]:
            rec["resets"].append((id(self), np.array(self.state, np.float64).copy()))
        return obs

    orig_select_agent = gw.select_agent
    holder = {}
    tap = _GumbelTap()

    def wrapped_select_agent(config, agent_name):
        agent = orig_select_agent(config=config, agent_name=agent_name)
        holder["init"] = _pack_td3d(agent)
        holder["agent"] = agent

        def wrap(fn, purpose, sink_name):
            def inner(*a, **k):
                prev, prev_sink = rec["purpose"], tap.sink
                rec["purpose"] = purpose
                tap.sink = rec[sink_name]
                n0 = len(rec[sink_name])
                try:
                    return fn(*a, **k)
                finally:
                    rec["purpose"], tap.sink = prev, prev_sink
                    if purpose == "learn":
                        rec["learn_calls"].append(len(rec[sink_name]) - n0)       # 1 = critics only, 2 = policy update too
            return inner
        agent.select_train_action = wrap(agent.select_train_action, "act_noise", "gumbel_act")
        agent.select_test_action = wrap(agent.select_test_action, "test_noise", "gumbel_test")
        agent.learn = wrap(agent.learn, "learn", "gumbel_learn")
        return agent

    with quiet():
        w = GTN_Worker(id=0, bohb_id=0)
        seed_all(seed)
        w.config = cfg
        w.late_init(cfg)
        w.timeout = 1e9
        env = w.synthetic_env_orig
        if done_bias_shift:
            with torch.no_grad():
                env.env.done_net[-1].bias.add_(done_bias_shift)
        theta = se_theta(env)
        orig_step = env.step

        def rec_step(action, state=None):
            s_before = env.env.state.detach().numpy().astype(np.float32).copy()
            ns, r, d = orig_step(action=action, state=state)
            rec["steps"].append(dict(state=s_before, action=int(action.item()), next_state=ns.detach().numpy().copy(),
                                     reward=float(r.item()), done=float(d.item())))
            return ns, r, d
        env.step = rec_step
        torch.randn, torch.randn_like, np.random.randint = rec_randn, rec_randn_like, rec_randint
        gspaces.Discrete.sample, env_class.reset = rec_disc_sample, rec_reset
        gw.select_agent = wrapped_select_agent
        train_reset_id = id(env.env.reset_env.env.unwrapped)
        try:
            with tap:
                rec["active"] = True
                agent = gw.select_agent(config=w.config, agent_name=w.agent_name)
                real_env = w.env_factory.generate_real_env()
                reward_list_train, episode_length_train, rbuf = agent.train(env=env, test_env=real_env, time_remaining=1e9)
                reward_list_test, _, _ = agent.test(env=real_env, time_remaining=1e9)
                rec["active"] = False
        finally:
            torch.randn, torch.randn_like, np.random.randint = orig_randn, orig_randn_like, orig_randint
            gspaces.Discrete.sample, env_class.reset = orig_disc_sample, orig_reset
            gw.select_agent = orig_select_agent
            if vary_seed is not None:
                ConfigSpace.ConfigurationSpace.sample_configuration = orig_cs_sample
    A = real_env.get_action_dim()
    # the learn calls' Gumbel draws: the first of a call belongs to actor_target(next_states), the second (policy updates) to actor(states)
    gt, ga, i = [], [], 0
    for n in rec["learn_calls"]:
        gt.append(rec["gumbel_learn"][i])
        if n > 1:
            ga.append(rec["gumbel_learn"][i + 1])
        i += n

    def pad4(rows):
        rows = np.array(rows, np.float64).reshape(len(rows), -1)
        return np.concatenate([rows, np.zeros((rows.shape[0], 4 - rows.shape[1]))], axis=1) if rows.shape[1] < 4 else rows

    def rows_of(lst):
        return np.concatenate([np.asarray(v, np.float32).reshape(-1, A) for v in lst]) if lst else np.zeros((0, A), np.float32)
    save(name, config_json=np.array(json.dumps(cfg)), hp_json=np.array(json.dumps(drawn)), theta=theta, agent_init=holder["init"],
         tape_rand_action=np.array(rec["rand"], np.int32), tape_act_noise=rows_of(rec["act_noise"]), tape_test_noise=rows_of(rec["test_noise"]),
         tape_policy_noise=rows_of(rec["policy_noise"]), tape_gumbel_act=rows_of(rec["gumbel_act"]), tape_gumbel_test=rows_of(rec["gumbel_test"]),
         tape_gumbel_target=rows_of(gt), tape_gumbel_actor=rows_of(ga),
         tape_replay_idx=(np.concatenate([np.asarray(v, np.int32).reshape(-1) for v in rec["replay"]]) if rec["replay"] else np.zeros(0, np.int32)),
         tape_train_reset=pad4([s for (i_, s) in rec["resets"] if i_ == train_reset_id]),
         tape_test_reset=pad4([s for (i_, s) in rec["resets"] if i_ != train_reset_id]),
         tr_state=np.stack([s["state"] for s in rec["steps"]]), tr_action=np.array([s["action"] for s in rec["steps"]], np.int32),
         tr_next_state=np.stack([s["next_state"] for s in rec["steps"]]).astype(np.float32),
         tr_reward=np.array([s["reward"] for s in rec["steps"]], np.float32), tr_done=np.array([s["done"] for s in rec["steps"]], np.float32),
         rb_action=rbuf.action[:rbuf.size].numpy().astype(np.float32),
         reward_list_train=np.array(reward_list_train, np.float64), episode_length_train=np.array(episode_length_train, np.int32),
         reward_list_test=np.array(reward_list_test, np.float64), score=np.array(statistics.mean(reward_list_test)),
         final_params=_pack_td3d(agent))


# ------------------------------------------------------------------------------------------------
# G11: the sync-file transport written by the REFERENCE, both directions (agents/GTN_master.py:147-195,267-298,
# agents/GTN_worker.py:76-154): the payload files themselves are the fixture (tensors + plain dicts: data), next to the
# values the reference's master computes from the workers' results.
# ------------------------------------------------------------------------------------------------
def gen_g11():
    import shutil
    from agents.GTN import GTN_Master, GTN_Worker
    cfg = load_cfg("default_config_cartpole_syn_env.yaml")
    n = 2
    cfg["agents"]["gtn"].update(num_workers=n, max_iterations=1, mode="single", time_sleep_master=0.02, time_sleep_worker=0.02,
                                quit_when_solved=False)
    cfg["agents"]["ddqn"].update(train_episodes=2, test_episodes=2, init_episodes=1, print_rate=int(1e9), batch_size=16)
    cfg["envs"]["CartPole-v0"]["max_steps"] = 12
    # a deterministic clock (0.25 s per reading) and no sleeping: `time_elapsed` / `timeout` are part of the payloads, and the
    # fixtures must regenerate byte for byte; everything here runs sequentially, so nothing ever has to wait for a file
    import time as _time
    real_time, real_sleep, ticks = _time.time, _time.sleep, [0]

    def fake_time():
        ticks[0] += 1
        return 1.7e9 + 0.25 * ticks[0]
    _time.time, _time.sleep = fake_time, (lambda s: None)
    try:
        _gen_g11_body(cfg, n, GTN_Master, GTN_Worker, shutil)
    finally:
        _time.time, _time.sleep = real_time, real_sleep


def _gen_g11_body(cfg, n, GTN_Master, GTN_Worker, shutil):
    with quiet():
        seed_all(1100)
        m = GTN_Master(cfg, bohb_id=-1)                      # bohb_id < 0: quit_flag on the last iteration (:163-166)
        m.clean_working_dir()
        workers = [GTN_Worker(id=i, bohb_id=-1) for i in range(n)]        # the constructor deletes the worker's stale sync files
        theta0 = se_theta(m.synthetic_env_orig)
        m.write_worker_inputs(0)
        for i in range(n):
            shutil.copyfile(m.get_input_file_name(i), os.path.join(OUT, "g11_ref_master_input_w%d.pt" % i))
        for i, w in enumerate(workers):
            seed_all(1101 + i)
            w.run()                                          # read input -> 3 calc_score -> calc_best_score -> write result -> quit
            shutil.copyfile(w.get_result_file_name(i), os.path.join(OUT, "g11_ref_worker_result_w%d.pt" % i))
        m.read_worker_results()
        eps = np.stack([se_theta(e) for e in m.eps_list])
        m.score_transform()
        weights = np.array(m.score_transform_list, np.float64)
        m.update_env()
        theta1 = se_theta(m.synthetic_env_orig)
    save("g11_file_transport", theta0=theta0, theta1=theta1, eps=eps, weights=weights, score=np.array(m.score_list, np.float64),
         score_orig=np.array(m.score_orig_list, np.float64), time_elapsed=np.array(m.time_elapsed_list, np.float64),
         timeout=np.array(float(m.time_max)), step_size=np.array(cfg["agents"]["gtn"]["step_size"]),
         score_transform_type=np.array(cfg["agents"]["gtn"]["score_transform_type"]))


# ------------------------------------------------------------------------------------------------
# G12: the reference's OWN evaluation harness -- experiments/syn_env_evaluate_cartpole_vary_hp_2.py:25-48 train_test_agents
# (DDQN_vary agents trained with agent.train(env=train_env) -- NO test env: the meter is fed by the training env's episode reward,
# early-out on the virtual env by early_out_virtual_diff (base_agent.py:49-56,134-148) -- then ONE agent.test on the real env),
# called the way experiments/syn_env_run_vary_hp.py:32-117 calls it: mode 0 = train on the REAL env, modes 1/2 = on the loaded SE.
# ------------------------------------------------------------------------------------------------
def gen_ckpt_b():
    """A second reference-written checkpoint for G12: the same SE with the reward net's output bias moved to ~1 per step and a done
    net that ends episodes (a stand-in for a TRAINED CartPole SE, whose episode reward is its length): the virtual early-out then
    fires after a few evaluations, not at the first one and not never."""
    from envs.env_factory import EnvFactory
    cfg = load_cfg("default_config_cartpole.yaml")          # what experiments/GTNC_evaluate_cartpole_vary_hp.py:46 trains its checkpoints from
    # 40-step episodes; a solved threshold their 10-episode mean can cross (mode 0 trains on the real env: real early-out rule)
    cfg["envs"]["CartPole-v0"].update(max_steps=40, solved_reward=16.0)
    seed_all(4200)
    with quiet():
        venv = EnvFactory(cfg).generate_virtual_env()
    with torch.no_grad():
        venv.env.reward_net[-1].bias.add_(1.0)
        venv.env.done_net[-1].bias.add_(0.3)
    path = os.path.join(OUT, "ckpt_cartpole_se_reference_b.pt")
    torch.save({'model': venv.state_dict(), 'config': cfg}, path)           # exactly GTN_Master.save_model's payload
    print("wrote", path, os.path.getsize(path))


def gen_g12(name, mode, seed, agents_num=2, vary=True, vary_seed=77, ckpt="ckpt_cartpole_se_reference_b.pt",
            module="experiments.syn_env_evaluate_cartpole_vary_hp_2", agent_key="ddqn", env_cls="CartPoleEnv"):
    """module / agent_key: the sibling script and the config section its agent reads (experiments/syn_env_evaluate_cartpole_vary_hp_2_DuelingDDQN.py:
    DuelingDDQN_vary, section `duelingddqn`)."""
    import importlib
    import json
    import shutil
    import tempfile
    import ConfigSpace
    ev = importlib.import_module(module)
    import gym.envs as genvs
    import gym.spaces as gspaces
    tmp = tempfile.mkdtemp(prefix="lenv_g12_")
    shutil.copy(os.path.join(OUT, ckpt), os.path.join(tmp, "model.pt"))
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "experiments"))      # (the sibling scripts read "../default_config_cartpole.yaml" relative to their own directory; nothing is written there)
    try:
        with quiet():
            venv, real_env, config = ev.load_envs_and_config(file_name="model.pt", model_dir=tmp, device="cpu")
    finally:
        os.chdir(cwd)
    theta = se_theta(venv)
    # syn_env_run_vary_hp.py:47-54 (mode 0): train_env = test_env = the real env; :82-91: train_env = the loaded virtual env
    train_env = real_env if mode == 0 else venv
    recs, holders = [], []
    state = {"rec": None, "phase": None}
    orig_random, orig_randint = random.random, np.random.randint
    cls = getattr(genvs, env_cls)
    orig_reset, orig_sample = cls.reset, gspaces.Discrete.sample

    def rec_random():
        v = orig_random()
        if state["phase"] == "train":
            state["rec"].eps_uniform.append(v)
        return v

    def rec_randint(*a, **k):
        v = orig_randint(*a, **k)
        if state["phase"] == "train":
            state["rec"].replay_idx.append(np.asarray(v).copy())
        return v

    def rec_reset(self):
        obs = orig_reset(self)
        if state["phase"] is not None:
            state["rec"].resets.append((state["phase"], np.array(self.state, np.float64).copy()))
        return obs

    def rec_sample(self):
        v = orig_sample(self)
        if state["phase"] == "train":
            state["rec"].rand_action.append(v)
        return v

    orig_step = train_env.step

    def rec_step(action, state_=None):
        ns, r, d = orig_step(action=action, state=state_) if mode != 0 else orig_step(action=action)
        if state["phase"] == "train":
            rec = state["rec"]
            rec.steps.append(dict(action=int(np.asarray(action.detach().cpu().numpy()).reshape(-1)[0]), next_state=ns.detach().numpy().reshape(-1).copy(),
                                  reward=float(r.item()), done=float(d.item()), n_rand=len(rec.rand_action)))
        return ns, r, d

    orig_select_agent = ev.select_agent

    def wrapped_select_agent(config, agent_name):
        if not vary:
            # mode 1 of the experiment family with the base hyper-parameters: DDQN_vary with vary_hp off IS DDQN (DDQN_vary.py:16-21).
            # train_test_agents has just forced vary_hp = True (:30); flip it back before the agent is built
            config['agents'][agent_key + '_vary']['vary_hp'] = False
        agent = orig_select_agent(config=config, agent_name=agent_name)
        rec = Recorder()
        state["rec"] = rec
        recs.append(rec)
        h = {"agent": agent,
             "hp": {k: agent.full_config["agents"][agent_key][k] for k in ("lr", "batch_size", "hidden_size", "hidden_layer")},
             "init": pack_linear_params(agent.model.state_dict(), "net.") if hasattr(agent.model, "net") else _pack_dueling(agent.model.state_dict())}
        holders.append(h)
        orig_learn, orig_train, orig_test = agent.learn, agent.train, agent.test

        def learn(replay_buffer, env, episode):
            loss = orig_learn(replay_buffer=replay_buffer, env=env, episode=episode)
            rec.losses.append(float(loss.item()))
            return loss

        def train(env, test_env=None, time_remaining=1e9):
            assert test_env is None                       # the harness calls agent.train(env=train_env) (:40)
            state["phase"] = "train"
            out = orig_train(env=env, test_env=test_env, time_remaining=time_remaining)
            state["phase"] = None
            h["reward_train"], h["episode_length"] = list(out[0]), list(out[1])
            return out

        def test(env, time_remaining=1e9):
            state["phase"] = "test"
            out = orig_test(env=env, time_remaining=time_remaining)
            state["phase"] = None
            return out

        agent.learn, agent.train, agent.test = learn, train, test
        return agent

    seed_all(seed)
    ConfigSpace.RANDOM.seed(vary_seed)
    train_env.step = rec_step
    random.random, np.random.randint = rec_random, rec_randint
    cls.reset, gspaces.Discrete.sample = rec_reset, rec_sample
    ev.select_agent = wrapped_select_agent
    try:
        with quiet():
            reward_list, train_steps_needed, episodes_needed = ev.train_test_agents(train_env=train_env, test_env=real_env, config=config,
                                                                                   agents_num=agents_num)
    finally:
        random.random, np.random.randint = orig_random, orig_randint
        cls.reset, gspaces.Discrete.sample = orig_reset, orig_sample
        ev.select_agent = orig_select_agent
        shutil.rmtree(tmp, ignore_errors=True)
    out = dict(config_json=np.array(json.dumps(config)), mode=np.array(mode), agents_num=np.array(agents_num), theta=theta,
               reward_list=np.array(reward_list, np.float64), train_steps_needed=np.array(train_steps_needed, np.int64),
               episodes_needed=np.array(episodes_needed, np.int64))
    for i, (rec, h) in enumerate(zip(recs, holders)):
        pre = "a%d_" % i
        B = h["hp"]["batch_size"]
        n = len(rec.steps)
        explored = np.zeros(n, np.int32)
        prev = 0
        for k, st in enumerate(rec.steps):
            explored[k] = 1 if st["n_rand"] > prev else 0
            prev = st["n_rand"]
        out.update({pre + "hp_json": np.array(json.dumps(h["hp"])), pre + "agent_init": h["init"],
                    pre + "tape_eps_uniform": np.array(rec.eps_uniform, np.float64),
                    pre + "tape_rand_action": np.array(rec.rand_action, np.int32),
                    pre + "tape_replay_idx": (np.stack(rec.replay_idx).astype(np.int32) if rec.replay_idx else np.zeros((0, B), np.int32)),
                    pre + "tape_train_reset": np.array([s_ for (ph, s_) in rec.resets if ph == "train"], np.float64).reshape(-1, 4),
                    pre + "tape_test_reset": np.array([s_ for (ph, s_) in rec.resets if ph == "test"], np.float64).reshape(-1, 4),
                    pre + "tr_action": np.array([s_["action"] for s_ in rec.steps], np.int32), pre + "tr_explored": explored,
                    pre + "tr_next_state": np.stack([s_["next_state"] for s_ in rec.steps]).astype(np.float32),
                    pre + "tr_reward": np.array([s_["reward"] for s_ in rec.steps], np.float32),
                    pre + "tr_done": np.array([s_["done"] for s_ in rec.steps], np.float32),
                    pre + "losses": np.array(rec.losses, np.float64),
                    pre + "reward_train": np.array(h["reward_train"], np.float64),
                    pre + "episode_length": np.array(h["episode_length"], np.int32)})
    save(name, **out)
    print(name, "episodes", [int(e[0]) for e in episodes_needed], "steps", [int(t[0]) for t in train_steps_needed],
          "hp", [h["hp"] for h in holders], "reward_test", [np.mean(r) for r in reward_list])


def gen_ckpt_c():
    """A reference-written Acrobot-v1 SE checkpoint for G12A (the Acrobot harness script): default_config_acrobot.yaml's shapes, 30-step
    episodes, reward net biased to about -1 per step (the real env's reward) and a done net that rarely fires, so that the virtual rule says
    "no" a few times before "yes"."""
    from envs.env_factory import EnvFactory
    cfg = load_cfg("default_config_acrobot.yaml")
    cfg["envs"]["Acrobot-v1"].update(max_steps=30)
    seed_all(4300)
    with quiet():
        venv = EnvFactory(cfg).generate_virtual_env()
    with torch.no_grad():
        venv.env.reward_net[-1].bias.add_(-1.0)
        venv.env.done_net[-1].bias.add_(-0.3)
    path = os.path.join(OUT, "ckpt_acrobot_se_reference_c.pt")
    torch.save({'model': venv.state_dict(), 'config': cfg}, path)           # exactly GTN_Master.save_model's payload
    print("wrote", path, os.path.getsize(path))


def gen_g13():
    """G13: the result file the REFERENCE's run_vary_hp writes (experiments/syn_env_run_vary_hp.py:32-139 -> utils.save_lists, utils.py:144-160)
    for mode 2 on two copies of the G12 checkpoint, two agents each -- the payload itself is the fixture (plain lists, a config dict and a
    pandas DataFrame: data), like the G11 sync files."""
    import importlib
    import shutil
    import tempfile
    import ConfigSpace
    ev = importlib.import_module("experiments.syn_env_evaluate_cartpole_vary_hp_2")
    rv = importlib.import_module("experiments.syn_env_run_vary_hp")
    tmp = tempfile.mkdtemp(prefix="lenv_g13_")
    for nm in ("CartPole-v0_4_QQQQQQ.pt", "CartPole-v0_7_CCCCCC.pt"):
        shutil.copy(os.path.join(OUT, "ckpt_cartpole_se_reference_b.pt"), os.path.join(tmp, nm))
    cwd = os.getcwd()
    out_dir = tempfile.mkdtemp(prefix="lenv_g13_out_")
    try:
        # the script reads "../default_config_cartpole.yaml" relative to experiments/; save_lists writes into the CURRENT directory: the
        # loader runs with cwd = experiments/ (nothing is written there), run_vary_hp's save with cwd = out_dir
        def load(file_name, model_dir, device):
            os.chdir(os.path.join(REF, "experiments"))
            try:
                return ev.load_envs_and_config(file_name=file_name, model_dir=model_dir, device=device)
            finally:
                os.chdir(out_dir)
        os.chdir(out_dir)
        seed_all(1300)
        ConfigSpace.RANDOM.seed(77)
        with quiet():
            rv.run_vary_hp(mode=2, experiment_name="g13", model_num=2, agents_num=2, model_dir=tmp, custom_load_envs_and_config=load,
                           custom_train_test_agents=ev.train_test_agents, env_name="CartPole", pool=None, device="cpu")
    finally:
        os.chdir(cwd)
    shutil.copyfile(os.path.join(out_dir, "2_g13.pt"), os.path.join(OUT, "g13_ref_run_vary_hp_mode2.pt"))
    d = torch.load(os.path.join(OUT, "g13_ref_run_vary_hp_mode2.pt"), weights_only=False)
    print("g13:", {k: type(v).__name__ for k, v in d.items()}, d["env_reward_overview"].shape, list(d["env_reward_overview"].index), d["episode_length_needed"])
    shutil.rmtree(tmp, ignore_errors=True)
    shutil.rmtree(out_dir, ignore_errors=True)


def gen_g12t(name, seed, vary_seed, agents_num=2, ckpt="ckpt_cartpole_se_reference_b.pt",
             module="experiments.syn_env_evaluate_cartpole_vary_hp_2_TD3_discrete"):
    """G12T: the TD3_discrete sibling script's train_test_agents (experiments/syn_env_evaluate_cartpole_vary_hp_2_TD3_discrete.py:44-67:
    `td3_discrete_vary` agents configured from default_config_cartpole.yaml's `td3_discrete_vary_layer_norm_2` section, comparability
    settings init_episodes 10 / early_out_num 10 / early_out_virtual_diff 0.01, agent.train(env=train_env) without a test env, one
    agent.test on the real env) on the loaded SE (mode 2 of syn_env_run_vary_hp.py).  Taps as in gen_g8td, one recorder per agent."""
    import importlib
    import json
    import shutil
    import tempfile
    import ConfigSpace
    ev = importlib.import_module(module)
    import gym.envs as genvs
    import gym.spaces as gspaces
    tmp = tempfile.mkdtemp(prefix="lenv_g12t_")
    shutil.copy(os.path.join(OUT, ckpt), os.path.join(tmp, "model.pt"))
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "experiments"))      # (the script reads "../default_config_cartpole.yaml"; nothing is written there)
    try:
        with quiet():
            venv, real_env, config = ev.load_envs_and_config(file_name="model.pt", model_dir=tmp, device="cpu")
    finally:
        os.chdir(cwd)
    theta = se_theta(venv)
    train_env = venv
    A = real_env.get_action_dim()
    recs, holders = [], []
    state = {"rec": None, "phase": None}
    tap = _GumbelTap()
    orig_randn, orig_randn_like, orig_randint = torch.randn, torch.randn_like, np.random.randint
    cls = genvs.CartPoleEnv
    orig_reset, orig_sample = cls.reset, gspaces.Discrete.sample
    orig_cs_sample = ConfigSpace.ConfigurationSpace.sample_configuration
    drawn = []

    def rec_cs_sample(self):
        d = orig_cs_sample(self)
        drawn.append(dict(d))
        return d

    def rec_randn(*a, **k):
        v = orig_randn(*a, **k)
        rec = state["rec"]
        if rec is not None and rec["purpose"] in ("act_noise", "test_noise"):
            rec[rec["purpose"]].append(v.numpy().copy())
        return v

    def rec_randn_like(t, *a, **k):
        v = orig_randn_like(t, *a, **k)
        rec = state["rec"]
        if rec is not None and rec["purpose"] == "learn":
            rec["policy_noise"].append(v.numpy().copy())
        return v

    def rec_randint(*a, **k):
        v = orig_randint(*a, **k)
        if state["phase"] == "train":
            state["rec"]["replay"].append(np.asarray(v).copy())
        return v

    def rec_sample(self):
        v = orig_sample(self)
        if state["phase"] == "train":
            state["rec"]["rand"].append(int(v))
        return v

    def rec_reset(self):
        obs = orig_reset(self)
        if state["phase"] is not None:
            state["rec"]["resets"].append((state["phase"], np.array(self.state, np.float64).copy()))
        return obs

    orig_step = train_env.step

    def rec_step(action, state_=None):
        ns, r, d = orig_step(action=action, state=state_)
        if state["phase"] == "train":
            state["rec"]["steps"].append(dict(action=action.detach().numpy().reshape(-1).astype(np.float32).copy(),
                                              next_state=ns.detach().numpy().reshape(-1).copy(), reward=float(r.item()), done=float(d.item())))
        return ns, r, d

    orig_select_agent = ev.select_agent

    def wrapped_select_agent(config, agent_name):
        agent = orig_select_agent(config=config, agent_name=agent_name)
        rec = dict(rand=[], act_noise=[], test_noise=[], policy_noise=[], replay=[], resets=[], steps=[], gumbel_act=[], gumbel_test=[],
                   gumbel_learn=[], learn_calls=[], purpose=None)
        state["rec"] = rec
        recs.append(rec)
        h = {"agent": agent, "hp": dict(drawn[-1]), "init": _pack_td3d(agent)}
        holders.append(h)

        def wrap(fn, purpose, sink_name):
            def inner(*a, **k):
                prev, prev_sink = rec["purpose"], tap.sink
                rec["purpose"] = purpose
                tap.sink = rec[sink_name]
                n0 = len(rec[sink_name])
                try:
                    return fn(*a, **k)
                finally:
                    rec["purpose"], tap.sink = prev, prev_sink
                    if purpose == "learn":
                        rec["learn_calls"].append(len(rec[sink_name]) - n0)       # 1 = critics only, 2 = policy update too
            return inner
        agent.select_train_action = wrap(agent.select_train_action, "act_noise", "gumbel_act")
        agent.select_test_action = wrap(agent.select_test_action, "test_noise", "gumbel_test")
        agent.learn = wrap(agent.learn, "learn", "gumbel_learn")
        orig_train, orig_test = agent.train, agent.test

        def train(env, test_env=None, time_remaining=1e9):
            assert test_env is None                       # the harness calls agent.train(env=train_env) (:56)
            state["phase"] = "train"
            out = orig_train(env=env, test_env=test_env, time_remaining=time_remaining)
            state["phase"] = None
            h["reward_train"], h["episode_length"] = list(out[0]), list(out[1])
            h["rb_action"] = out[2].action[:out[2].size].numpy().astype(np.float32).copy()
            return out

        def test(env, time_remaining=1e9):
            state["phase"] = "test"
            out = orig_test(env=env, time_remaining=time_remaining)
            state["phase"] = None
            h["final_params"] = _pack_td3d(agent)
            return out

        agent.train, agent.test = train, test
        return agent

    seed_all(seed)
    ConfigSpace.RANDOM.seed(vary_seed)
    train_env.step = rec_step
    torch.randn, torch.randn_like, np.random.randint = rec_randn, rec_randn_like, rec_randint
    cls.reset, gspaces.Discrete.sample = rec_reset, rec_sample
    ConfigSpace.ConfigurationSpace.sample_configuration = rec_cs_sample
    ev.select_agent = wrapped_select_agent
    try:
        with quiet(), tap:
            reward_list, train_steps_needed, episodes_needed = ev.train_test_agents(train_env=train_env, test_env=real_env, config=config,
                                                                                   agents_num=agents_num)
    finally:
        torch.randn, torch.randn_like, np.random.randint = orig_randn, orig_randn_like, orig_randint
        cls.reset, gspaces.Discrete.sample = orig_reset, orig_sample
        ConfigSpace.ConfigurationSpace.sample_configuration = orig_cs_sample
        ev.select_agent = orig_select_agent
        shutil.rmtree(tmp, ignore_errors=True)
    config["agents"]["gtn"]["agent_name"] = "TD3_discrete_vary"
    config["agents"]["gtn"]["synthetic_env_type"] = 0
    out = dict(config_json=np.array(json.dumps(config)), mode=np.array(2), agents_num=np.array(agents_num), theta=theta,
               reward_list=np.array(reward_list, np.float64), train_steps_needed=np.array(train_steps_needed, np.int64),
               episodes_needed=np.array(episodes_needed, np.int64))

    def rows_of(lst):
        return np.concatenate([np.asarray(v, np.float32).reshape(-1, A) for v in lst]) if lst else np.zeros((0, A), np.float32)
    for i, (rec, h) in enumerate(zip(recs, holders)):
        pre = "a%d_" % i
        gt, ga, k = [], [], 0
        for n in rec["learn_calls"]:                        # first Gumbel draw of a learn call: actor_target(next_states); second: actor(states)
            gt.append(rec["gumbel_learn"][k])
            if n > 1:
                ga.append(rec["gumbel_learn"][k + 1])
            k += n
        out.update({pre + "hp_json": np.array(json.dumps(h["hp"])), pre + "agent_init": h["init"], pre + "final_params": h["final_params"],
                    pre + "tape_rand_action": np.array(rec["rand"], np.int32), pre + "tape_act_noise": rows_of(rec["act_noise"]),
                    pre + "tape_test_noise": rows_of(rec["test_noise"]), pre + "tape_policy_noise": rows_of(rec["policy_noise"]),
                    pre + "tape_gumbel_act": rows_of(rec["gumbel_act"]), pre + "tape_gumbel_test": rows_of(rec["gumbel_test"]),
                    pre + "tape_gumbel_target": rows_of(gt), pre + "tape_gumbel_actor": rows_of(ga),
                    pre + "tape_replay_idx": (np.concatenate([np.asarray(v, np.int32).reshape(-1) for v in rec["replay"]]) if rec["replay"]
                                              else np.zeros(0, np.int32)),
                    pre + "tape_train_reset": np.array([s_ for (ph, s_) in rec["resets"] if ph == "train"], np.float64).reshape(-1, 4),
                    pre + "tape_test_reset": np.array([s_ for (ph, s_) in rec["resets"] if ph == "test"], np.float64).reshape(-1, 4),
                    pre + "tr_action": np.stack([s_["action"] for s_ in rec["steps"]]).astype(np.float32),
                    pre + "tr_next_state": np.stack([s_["next_state"] for s_ in rec["steps"]]).astype(np.float32),
                    pre + "tr_reward": np.array([s_["reward"] for s_ in rec["steps"]], np.float32),
                    pre + "tr_done": np.array([s_["done"] for s_ in rec["steps"]], np.float32),
                    pre + "rb_action": h["rb_action"],
                    pre + "reward_train": np.array(h["reward_train"], np.float64),
                    pre + "episode_length": np.array(h["episode_length"], np.int32)})
    save(name, **out)
    print(name, "episodes", [int(e[0]) for e in episodes_needed], "steps", [int(t[0]) for t in train_steps_needed],
          "hp", [h["hp"] for h in holders], "reward_test", [np.mean(r) for r in reward_list])


def main():
    # no arguments (or "all"): every fixture under tests/golden is regenerated (the full-shape runs g8df / g8tf take minutes each)
    ALL = ["g1", "g1ln", "g2", "g3", "g4", "g4d", "g4t", "g6", "g7", "g8", "g8d", "g8t", "g9", "g10", "g2f", "ckpt", "g8w", "g6m", "g9x",
           "g8tv", "g8ts", "g8p", "g8c", "g8cf", "g8co", "g8pf", "g8ti", "g8tf", "g8tln", "g8df", "g8l2", "g8ln", "g8seln", "g8tseln", "g8tdseln", "g9ln", "g8m", "g8r", "g8rl", "g8i", "g8v", "g11", "g4td", "g8td", "g8k", "g9k", "g8long"]
    which = sys.argv[1:] or ALL
    if "all" in which:
        which = ALL
    os.makedirs(OUT, exist_ok=True)
    if "g12" in which:
        # (after "ckpt": g12 reads the committed checkpoints)
        gen_ckpt_b()
        gen_g12("g12_train_test_agents_cartpole_mode2_vary", mode=2, seed=1201, vary=True)
        gen_g12("g12_train_test_agents_cartpole_mode1_plain", mode=1, seed=1202, vary=False)
        gen_g12("g12_train_test_agents_cartpole_mode0_real_env", mode=0, seed=1203, vary=True)
        # the DuelingDDQN sibling script (settings in the `duelingddqn` section, DuelingDDQN_vary agents); vary_seed 1 draws small nets
        # (batch 105 / width 42 / 3 layers and 58 / 63 / 1) so that the CPU oracle replays the run in seconds
        gen_g12("g12d_train_test_agents_cartpole_mode2_dueling_vary", mode=2, seed=1204, vary=True, vary_seed=1,
                module="experiments.syn_env_evaluate_cartpole_vary_hp_2_DuelingDDQN", agent_key="duelingddqn")
    if "g13" in which:
        gen_g13()
    if "g12a" in which:
        # the Acrobot harness script (experiments/syn_env_evaluate_acrobot_vary_hp_2.py: the same function on an Acrobot-v1 SE; DDQN_vary over
        # default_config_acrobot.yaml's 128 x 2 DDQN, vary_seed 1 draws 42 x 3 / batch 105 and 63 x 1 / batch 58)
        gen_ckpt_c()
        gen_g12("g12a_train_test_agents_acrobot_mode2_vary", mode=2, seed=1206, vary=True, vary_seed=1, ckpt="ckpt_acrobot_se_reference_c.pt",
                module="experiments.syn_env_evaluate_acrobot_vary_hp_2", env_cls="AcrobotEnv")
    if "g12t" in which:
        # the TD3_discrete sibling script (td3_discrete_vary + LayerNorm section of default_config_cartpole.yaml); a vary_seed whose draws are
        # small nets so that the CPU oracle replays the run in seconds
        gen_g12t("g12t_train_test_agents_cartpole_mode2_td3_discrete_vary", seed=1205, vary_seed=int(os.environ.get("LENV_G12T_VARY_SEED", "1")))
    if "g1" in which:
        gen_g1()
    if "g1ln" in which:
        gen_g1ln()
    if "g3" in which:
        gen_g3()
    if "g4" in which:
        gen_g4()
    if "g6" in which:
        gen_g6()
    if "g6m" in which:
        gen_g6m()
    if "g11" in which:
        gen_g11()
    if "g7" in which:
        gen_g7()
    if "g4t" in which:
        gen_g4t()
    if "g8t" in which:
        gen_g8t("g8t_calc_score_cheetah_td3", seed=830)
    if "g8tln" in which:
        # `use_layer_norm: True` in the td3 section (models/model_utils.py:22-37): actor and critics with two / three hidden layers, one
        # shared nn.LayerNorm per net (at one / two positions)
        gen_g8t("g8tln_calc_score_cheetah_td3_layernorm", seed=831,
                agent_over={"train_episodes": 4, "init_episodes": 2, "batch_size": 16, "hidden_size": 24, "test_episodes": 2, "use_layer_norm": True})
        gen_g8t("g8tln3_calc_score_cheetah_td3_layernorm_3layer", seed=832,
                agent_over={"train_episodes": 4, "init_episodes": 2, "batch_size": 12, "hidden_size": 20, "hidden_layer": 3, "test_episodes": 2,
                            "policy_delay": 2, "use_layer_norm": True})
    if "g8tseln" in which:
        # `use_layer_norm: True` in the ENV's section: the two-hidden-layer reward net of a Pendulum RewardEnv (type 2) and the three SE nets
        # of a HalfCheetah VirtualEnv normalise behind their second Linear; the module is never perturbed, theta = the nn.Linear parameters
        gen_g8t("g8trnln_calc_score_pendulum_td3_reward_net_layernorm", seed=872, cfg_yaml="default_config_pendulum_reward_env.yaml",
                env_name="Pendulum-v0", env_cls="PendulumEnv",
                agent_over={"train_episodes": 4, "init_episodes": 2, "batch_size": 16, "hidden_size": 24, "test_episodes": 2},
                env_over={"max_steps": 8, "hidden_size": 20, "hidden_layer": 2, "activation_fn": "leakyrelu", "reward_env_type": 2,
                          "use_layer_norm": True})
        gen_g8t("g8tseln_calc_score_cheetah_td3_virtual_env_layernorm", seed=835, virtual=True,
                agent_over={"train_episodes": 3, "init_episodes": 1, "batch_size": 16, "hidden_size": 24, "test_episodes": 1},
                env_over={"max_steps": 8, "hidden_size": 20, "hidden_layer": 2, "activation_fn": "tanh", "reward_env_type": 0,
                          "use_layer_norm": True})
    if "g8tdseln" in which:
        gen_g8td("g8tdseln_calc_score_cartpole_td3_discrete_se_layernorm", seed=1834, agent_over={"train_episodes": 3, "init_episodes": 1, "test_episodes": 3},
                 env_over={"max_steps": 30, "hidden_size": 20, "hidden_layer": 2, "use_layer_norm": True})
    if "g9ln" in which:
        gen_g9("g9ln_calc_score_cliff_ql_reward_net_layernorm", seed=908, eps_over=0.2,
               env_over={"hidden_layer": 2, "hidden_size": 24, "use_layer_norm": True})
    if "g4td" in which:
        gen_g4td()
    if "g8td" in which:
        # TD3_discrete_vary (agents/TD3_discrete_vary.py) through GTN_Worker on a VirtualEnv: CartPole (A = 2) plain, Acrobot (A = 3)
        # with the shared LayerNorm and the hard (straight-through) Gumbel softmax, CartPole with three hidden layers + LayerNorm
        gen_g8td("g8td_calc_score_cartpole_td3_discrete", seed=1830, agent_over={"train_episodes": 3, "init_episodes": 1, "test_episodes": 3},
                 env_over={"max_steps": 30, "hidden_size": 20})
        gen_g8td("g8tdl_calc_score_acrobot_td3_discrete_layer_norm", seed=1831, cfg_yaml="default_config_acrobot_syn_env.yaml", env_name="Acrobot-v1",
                 env_cls="AcrobotEnv", agent_over={"use_layer_norm": True, "gumbel_softmax_hard": True, "policy_delay": 2, "activation_fn": "tanh"},
                 env_over={"max_steps": 9, "hidden_size": 16}, done_bias_shift=-0.3)
        gen_g8td("g8td3_calc_score_cartpole_td3_discrete_3_layers", seed=1832, agent_over={"use_layer_norm": True, "hidden_layer": 3, "hidden_size": 20,
                                                                                         "policy_delay": 1, "batch_size": 12}, done_bias_shift=-0.3)
        gen_g8td("g8tdv_calc_score_cartpole_td3_discrete_vary", seed=1833, vary_seed=11, agent_over={"use_layer_norm": True}, done_bias_shift=-0.3)
    if "g8tv" in which:
        gen_g8t("g8tv_calc_score_cheetah_td3_vary", seed=832, vary_seed=8,
                agent_over={"train_episodes": 3, "init_episodes": 1, "batch_size": 64, "hidden_size": 48, "hidden_layer": 2, "test_episodes": 1},
                env_over={"max_steps": 10, "hidden_size": 24})          # draws batch 145, width 108, 3 hidden layers
    if "g8ts" in which:
        # default_config_halfcheetah.yaml's combination: TD3 on a VirtualEnv (three SE nets on cat(action, state), two hidden
        # layers here), tested on the real (stand-in) env
        gen_g8t("g8ts_calc_score_cheetah_td3_virtual_env", seed=834, virtual=True,
                agent_over={"train_episodes": 3, "init_episodes": 1, "batch_size": 16, "hidden_size": 24, "test_episodes": 1},
                env_over={"max_steps": 8, "hidden_size": 20, "hidden_layer": 2, "activation_fn": "leakyrelu", "reward_env_type": 0})
    if "g8p" in which:
        # default_config_pendulum.yaml's combination: TD3 on a VirtualEnv of Pendulum-v0 (3 obs, 1 action, max_action 2), tested on
        # the real Pendulum; and default_config_pendulum_reward_env.yaml's: TD3 on a RewardEnv (type 2) over the real Pendulum
        gen_g8t("g8p_calc_score_pendulum_td3_virtual_env", seed=870, virtual=True, cfg_yaml="default_config_pendulum_reward_env.yaml",
                env_name="Pendulum-v0", env_cls="PendulumEnv",
                agent_over={"train_episodes": 3, "init_episodes": 1, "batch_size": 16, "hidden_size": 24, "test_episodes": 2},
                env_over={"max_steps": 9, "hidden_size": 20, "hidden_layer": 2, "activation_fn": "leakyrelu", "reward_env_type": 0})
        gen_g8t("g8pr_calc_score_pendulum_td3_reward_env", seed=871, cfg_yaml="default_config_pendulum_reward_env.yaml",
                env_name="Pendulum-v0", env_cls="PendulumEnv",
                agent_over={"train_episodes": 4, "init_episodes": 2, "batch_size": 16, "hidden_size": 24, "test_episodes": 2},
                env_over={"max_steps": 8, "hidden_size": 20, "hidden_layer": 1, "activation_fn": "prelu", "reward_env_type": 2})
    if "g8c" in which:
        # default_config_cmc.yaml's combination: TD3 on a VirtualEnv of MountainCarContinuous-v0 (2 obs, 1 action), tested on the real
        # env; and default_config_cmc_reward_env.yaml's: TD3 on a RewardEnv (type 2, tanh) over the real env, whose episodes end at
        # the flag
        gen_g8t("g8c_calc_score_cmc_td3_virtual_env", seed=880, virtual=True, cfg_yaml="default_config_cmc_reward_env.yaml",
                env_name="MountainCarContinuous-v0", env_cls="Continuous_MountainCarEnv",
                agent_over={"train_episodes": 3, "init_episodes": 1, "batch_size": 16, "hidden_size": 24, "test_episodes": 2},
                env_over={"max_steps": 9, "hidden_size": 20, "hidden_layer": 2, "activation_fn": "leakyrelu", "reward_env_type": 0})
        gen_g8t("g8cr_calc_score_cmc_td3_reward_env", seed=881, cfg_yaml="default_config_cmc_reward_env.yaml",
                env_name="MountainCarContinuous-v0", env_cls="Continuous_MountainCarEnv",
                agent_over={"train_episodes": 4, "init_episodes": 2, "batch_size": 16, "hidden_size": 24, "test_episodes": 2},
                env_over={"max_steps": 8, "hidden_size": 20, "hidden_layer": 1, "activation_fn": "tanh", "reward_env_type": 2})
    if "g8cf" in which:
        # default_config_cmc.yaml at its REAL shapes (actor 2-128-128-1, twin critics 3-128-128-1 relu, batch 256, policy_delay 2,
        # same_action_num 2, SE nets 3-96-96-{2,1,1} leakyrelu): two learning episodes of 15 agent steps = 30 learn steps, 15 policy steps
        gen_g8t("g8cf_calc_score_cmc_td3_virtual_env_fullshape", seed=883, virtual=True, cfg_yaml="default_config_cmc.yaml",
                env_name="MountainCarContinuous-v0", env_cls="Continuous_MountainCarEnv",
                agent_over={"train_episodes": 3, "init_episodes": 1, "test_episodes": 1}, env_over={"max_steps": 30})
    if "g8co" in which:
        # default_config_cmc_syn_env_opt.yaml at its REAL shapes (actor 2-64-1, twin critics 3-64-1 leakyrelu -- ONE hidden layer --, batch 256,
        # policy_delay 2, same_action_num 2, SE nets 3-128-128-128-{2,1,1} relu): two learning episodes of 20 agent steps = 40 learn steps,
        # 20 policy steps -- the shape the DIRECT instantiations of the TD3 GEMM-queue kernel run
        gen_g8t("g8co_calc_score_cmc_td3_syn_env_opt_fullshape", seed=886, virtual=True, cfg_yaml="default_config_cmc_syn_env_opt.yaml",
                env_name="MountainCarContinuous-v0", env_cls="Continuous_MountainCarEnv",
                agent_over={"train_episodes": 3, "init_episodes": 1, "test_episodes": 1}, env_over={"max_steps": 40})
    if "g8pf" in which:
        # the td3 sections of default_config_pendulum.yaml (SE nets 4-32-32-x) and default_config_halfcheetah.yaml (SE nets
        # 23-128-128-128-x) at their REAL shapes: batch 256, policy_delay 2, ten test episodes per test phase
        gen_g8t("g8pf_calc_score_pendulum_td3_virtual_env_fullshape", seed=884, virtual=True, cfg_yaml="default_config_pendulum.yaml",
                env_name="Pendulum-v0", env_cls="PendulumEnv", agent_over={"train_episodes": 3, "init_episodes": 1}, env_over={"max_steps": 14})
        gen_g8t("g8hf_calc_score_cheetah_td3_virtual_env_fullshape", seed=885, virtual=True, cfg_yaml="default_config_halfcheetah.yaml",
                agent_over={"train_episodes": 3, "init_episodes": 1}, env_over={"max_steps": 10})
    if "g8ti" in which:
        gen_g8t("g8ti_calc_score_cheetah_td3_icm", seed=833,
                agent_over={"train_episodes": 3, "init_episodes": 1, "batch_size": 16, "hidden_size": 24, "test_episodes": 1},
                env_over={"max_steps": 8, "hidden_size": 20}, icm_over={"feature_dim": 12, "hidden_size": 20})
    if "g8tf" in which:
        # BASELINE configs[4] at its REAL network shapes (actor 17-128-128-6, critics 23-128-128-1, B 192, RN 17-128-1)
        gen_g8t("g8tf_calc_score_cheetah_td3_fullshape", seed=831,
                agent_over={"train_episodes": 3, "init_episodes": 1, "batch_size": 192, "hidden_size": 128, "hidden_layer": 2, "test_episodes": 1},
                env_over={"max_steps": 12, "hidden_size": 128})
    if "g4d" in which:
        gen_g4d()
    if "g8df" in which:
        # BASELINE configs[2] at its REAL network shapes (feature 6-128-128-128, heads 128-128-{1,3}, B 128, SE hidden 128)
        gen_g8("g8df_calc_score_acrobot_dueling_fullshape", train_episodes=3, done_bias_shift=0.0, seed=811, max_steps=20,
               env_yaml="default_config_acrobot.yaml", env_name="Acrobot-v1", env_cls="AcrobotEnv", agent_key="duelingddqn",
               agent_over={"hidden_size": 128, "hidden_layer": 2, "feature_dim": 128, "batch_size": 128, "init_episodes": 1, "test_episodes": 2},
               env_over={"hidden_size": 128, "solved_reward": 0.5})
    if "g8l2" in which:
        # the DDQN of default_config_acrobot.yaml (Critic_DQN 6-128-128-3, relu, B 128) on a synthetic Acrobot: the
        # multi-layer Q-net that only the GEMM-tiled kernel takes
        gen_g8("g8l2_calc_score_acrobot_ddqn_2layer", train_episodes=3, done_bias_shift=0.0, seed=812, max_steps=20,
               env_yaml="default_config_acrobot.yaml", env_name="Acrobot-v1", env_cls="AcrobotEnv", agent_key="ddqn",
               agent_over={"init_episodes": 1, "test_episodes": 2}, env_over={"hidden_size": 128, "solved_reward": 0.5})
    if "g8ln" in which:
        # `use_layer_norm: True` (models/model_utils.py:22-37) in the agent's section: a DDQN whose Critic_DQN has two hidden layers (ONE
        # shared nn.LayerNorm behind the second Linear) and a DuelingDDQN whose feature stream has three (the same module at two positions)
        gen_g8("g8ln_calc_score_acrobot_ddqn_layernorm", train_episodes=3, done_bias_shift=0.0, seed=813, max_steps=20,
               env_yaml="default_config_acrobot.yaml", env_name="Acrobot-v1", env_cls="AcrobotEnv", agent_key="ddqn",
               agent_over={"hidden_size": 40, "hidden_layer": 2, "batch_size": 32, "init_episodes": 1, "test_episodes": 2, "use_layer_norm": True},
               env_over={"hidden_size": 32, "solved_reward": 0.5})
        gen_g8("g8dln_calc_score_acrobot_dueling_layernorm", train_episodes=3, done_bias_shift=0.0, seed=814, max_steps=20,
               env_yaml="default_config_acrobot.yaml", env_name="Acrobot-v1", env_cls="AcrobotEnv", agent_key="duelingddqn",
               agent_over={"hidden_size": 24, "hidden_layer": 3, "feature_dim": 16, "batch_size": 32, "init_episodes": 1, "test_episodes": 2,
                           "use_layer_norm": True},
               env_over={"hidden_size": 32, "solved_reward": 0.5})
    if "g8seln" in which:
        # `use_layer_norm: True` in the ENV's section with a two-hidden-layer SE: the three SE nets normalise behind their second Linear; NES
        # never touches the module, theta stays the nn.Linear parameters
        gen_g8("g8seln_calc_score_acrobot_ddqn_se_layernorm", train_episodes=3, done_bias_shift=0.0, seed=815, max_steps=20,
               env_yaml="default_config_acrobot.yaml", env_name="Acrobot-v1", env_cls="AcrobotEnv", agent_key="ddqn",
               agent_over={"hidden_size": 40, "hidden_layer": 2, "batch_size": 32, "init_episodes": 1, "test_episodes": 2},
               env_over={"hidden_size": 32, "hidden_layer": 2, "solved_reward": 0.5, "use_layer_norm": True})
    if "g8m" in which:
        # default_config_mountaincar.yaml's pair: MountainCar-v0 SE + DDQN with two hidden layers (GEMM-tiled kernel, plain-DQN mode)
        gen_g8("g8m_calc_score_mountaincar_ddqn", train_episodes=3, done_bias_shift=0.0, seed=860, max_steps=25,
               env_yaml="default_config_mountaincar.yaml", env_name="MountainCar-v0", env_cls="MountainCarEnv",
               agent_over={"hidden_size": 48, "batch_size": 32, "init_episodes": 1, "test_episodes": 2},
               env_over={"hidden_size": 32, "solved_reward": 0.5})
        gen_g8("g8mr_calc_score_mountaincar_ddqn_reward_env", train_episodes=3, done_bias_shift=0.0, seed=861, max_steps=25,
               env_yaml="default_config_mountaincar.yaml", env_name="MountainCar-v0", env_cls="MountainCarEnv", reward_env_type=2,
               agent_over={"hidden_size": 40, "batch_size": 24, "init_episodes": 1, "test_episodes": 2},
               env_over={"hidden_size": 24, "solved_reward": 0.5, "info_dim": 0})
    if "g8rl" in which:
        # the same RewardEnv mode with a two-hidden-layer reward net (4-24-24-1), plain and with `use_layer_norm` in the env's section
        gen_g8("g8rl_calc_score_cartpole_ddqn_reward_env_2layer", train_episodes=3, done_bias_shift=0.0, seed=852, max_steps=30,
               env_yaml="default_config_cartpole_reward_env.yaml", reward_env_type=2,
               agent_over={"init_episodes": 1, "test_episodes": 2, "batch_size": 24}, env_over={"hidden_size": 24, "hidden_layer": 2})
        gen_g8("g8rln_calc_score_cartpole_ddqn_reward_env_layernorm", train_episodes=3, done_bias_shift=0.0, seed=853, max_steps=30,
               env_yaml="default_config_cartpole_reward_env.yaml", reward_env_type=1,
               agent_over={"init_episodes": 1, "test_episodes": 2, "batch_size": 24},
               env_over={"hidden_size": 24, "hidden_layer": 2, "activation_fn": "leakyrelu", "use_layer_norm": True})
    if "g8r" in which:
        # default_config_cartpole_reward_env.yaml's experiment: DDQN on a RewardEnv over the real CartPole (potential-shaped,
        # type 2, PReLU reward net 4-64-1) -- the env transition is the real one, the reward goes through the network
        gen_g8("g8r_calc_score_cartpole_ddqn_reward_env", train_episodes=4, done_bias_shift=0.0, seed=850, max_steps=40,
               env_yaml="default_config_cartpole_reward_env.yaml", reward_env_type=2,
               agent_over={"init_episodes": 1, "test_episodes": 2, "batch_size": 32})
        gen_g8("g8r6_calc_score_cartpole_ddqn_reward_env_t6", train_episodes=3, done_bias_shift=0.0, seed=851, max_steps=30,
               env_yaml="default_config_cartpole_reward_env.yaml", reward_env_type=6,
               agent_over={"init_episodes": 1, "test_episodes": 2, "batch_size": 24}, env_over={"activation_fn": "tanh", "hidden_size": 24})
    if "g8k" in which:
        # same_action_num > 1 outside the TD3 family (agents/base_agent.py:104,194; envs/env_wrapper.py:24-29,56-61; no shipped DDQN
        # config sets it): DDQN on a CartPole VirtualEnv with 2 steps per action (odd max_steps: the last action of an episode gets
        # one env step in the real test env), DuelingDDQN on an Acrobot VirtualEnv with 3, DDQN on the CartPole RewardEnv with 2
        gen_g8("g8k_calc_score_cartpole_ddqn_same_action_2", train_episodes=4, done_bias_shift=-0.3, seed=860, max_steps=25,
               agent_over={"init_episodes": 1, "test_episodes": 3, "batch_size": 24, "same_action_num": 2})
        gen_g8("g8kd_calc_score_acrobot_duelingddqn_same_action_3", train_episodes=3, done_bias_shift=-0.3, seed=861, max_steps=20,
               env_yaml="default_config_acrobot.yaml", env_name="Acrobot-v1", env_cls="AcrobotEnv", agent_key="duelingddqn",
               agent_over={"init_episodes": 1, "test_episodes": 2, "batch_size": 16, "hidden_size": 24, "feature_dim": 20, "same_action_num": 3},
               env_over={"hidden_size": 20, "solved_reward": 0.5})
        gen_g8("g8kr_calc_score_cartpole_ddqn_reward_env_same_action_2", train_episodes=4, done_bias_shift=0.0, seed=862, max_steps=31,
               env_yaml="default_config_cartpole_reward_env.yaml", reward_env_type=2,
               agent_over={"init_episodes": 1, "test_episodes": 2, "batch_size": 32, "same_action_num": 2})
    if "g8i" in which:
        # DDQN / DuelingDDQN with the ICM baseline inside learn() (models/icm_baseline.py): CartPole = BCE inverse loss on one
        # action logit, Acrobot = cross-entropy over three
        gen_g8("g8i_calc_score_cartpole_ddqn_icm", train_episodes=3, done_bias_shift=0.0, seed=840, max_steps=12,
               agent_over={"init_episodes": 1, "test_episodes": 2, "batch_size": 24, "hidden_size": 32},
               icm_over={"feature_dim": 16, "hidden_size": 24})
        gen_g8("g8ia_calc_score_acrobot_dueling_icm", train_episodes=3, done_bias_shift=0.0, seed=841, max_steps=10,
               env_yaml="default_config_acrobot.yaml", env_name="Acrobot-v1", env_cls="AcrobotEnv", agent_key="duelingddqn",
               agent_over={"hidden_size": 24, "feature_dim": 16, "batch_size": 20, "init_episodes": 1, "test_episodes": 2},
               env_over={"hidden_size": 32, "solved_reward": 0.5}, icm_over={"feature_dim": 12, "hidden_size": 20})
    if "g8v" in which:
        # DDQN_vary / DuelingDDQN_vary (agents/DDQN_vary.py:26-59): the agent draws lr / batch_size / hidden_size / hidden_layer
        # and trains with them; the fixture records the draw and the run
        gen_g8("g8v_calc_score_cartpole_ddqn_vary", train_episodes=3, done_bias_shift=0.0, seed=820, max_steps=25, vary_seed=14,
               agent_over={"init_episodes": 1, "test_episodes": 2})            # draws batch 204, width 129, 2 hidden layers
        gen_g8("g8v2_calc_score_cartpole_ddqn_vary_wide", train_episodes=3, done_bias_shift=0.0, seed=822, max_steps=15, vary_seed=4,
               agent_over={"init_episodes": 1, "test_episodes": 2})            # draws batch 555, width 161, 1 hidden layer
        gen_g8("g8vd_calc_score_acrobot_dueling_vary", train_episodes=3, done_bias_shift=0.0, seed=821, max_steps=12, vary_seed=8,      # batch 145, width 108, 3 layers
               env_yaml="default_config_acrobot.yaml", env_name="Acrobot-v1", env_cls="AcrobotEnv", agent_key="duelingddqn",
               agent_over={"hidden_size": 48, "feature_dim": 32, "batch_size": 64, "init_episodes": 1, "test_episodes": 2},
               env_over={"hidden_size": 32, "solved_reward": 0.5})
    if "g8d" in which:
        gen_g8("g8d_calc_score_acrobot_dueling", train_episodes=4, done_bias_shift=0.0, seed=810, max_steps=25,
               env_yaml="default_config_acrobot.yaml", env_name="Acrobot-v1", env_cls="AcrobotEnv", agent_key="duelingddqn",
               agent_over={"hidden_size": 24, "feature_dim": 16, "batch_size": 32, "init_episodes": 1, "test_episodes": 3},
               env_over={"hidden_size": 32, "solved_reward": 0.5})
    if "g2" in which:
        gen_g2()
    if "g2f" in which:
        gen_g2f()
    if "ckpt" in which:
        gen_ckpt()
    if "g9" in which:
        gen_g9("g9_calc_score_cliff_a", seed=900)
        gen_g9("g9_calc_score_cliff_b", seed=901, eps_over=0.2)
    if "g9x" in which:
        # the other tabular agents of select_agent (agent_utils.py:57-64) and the init_episodes gate of BaseAgent.train
        gen_g9("g9s_calc_score_cliff_sarsa", seed=902, agent_name="SARSA")
        gen_g9("g9c_calc_score_cliff_ql_cb", seed=903, agent_name="QL_cb", agent_over={"beta": 0.3}, eps_over=0.05)
        gen_g9("g9sc_calc_score_cliff_sarsa_cb", seed=904, agent_name="SARSA_cb", agent_over={"beta": 0.3})
        gen_g9("g9i_calc_score_cliff_ql_init2", seed=905, eps_over=0.2, agent_over={"init_episodes": 2, "train_episodes": 12})
    if "g9k" in which:
        # same_action_num > 1 for the tabular agents (no shipped gridworld config sets it): every chosen action is applied twice /
        # three times, the repeats stop at the cliff / goal / TimeLimit, the shaped rewards are summed
        gen_g9("g9k_calc_score_cliff_ql_same_action_2", seed=906, eps_over=0.15, agent_over={"same_action_num": 2})
        gen_g9("g9ks_calc_score_cliff_sarsa_same_action_3", seed=907, agent_name="SARSA", eps_over=0.1, agent_over={"same_action_num": 3})
    if "g10" in which:
        gen_g10()
    if "g8" in which:
        gen_g8("g8_calc_score_cartpole_a", train_episodes=3, done_bias_shift=0.0, seed=800)
        gen_g8("g8_calc_score_cartpole_b", train_episodes=4, done_bias_shift=0.45, seed=801, max_steps=60)
    if "g8long" in which:
        # LONG-HORIZON runs of the reference, one per NN configuration of BASELINE.json: how far does "returns within
        # 1e-4 of the reference on a fixed seed" hold when thousands of learn steps feed back into the greedy actions?
        # configs[1] at EXACTLY the workload bench.py times: 20 train episodes x 200 steps on an SE that never terminates (done bias -10),
        # 3 800 learn steps at B = 199, ten real-env test episodes after every train episode, early-out off (solved_reward 1e9)
        gen_g8("g8long_cartpole_ddqn_bench_workload", train_episodes=20, done_bias_shift=-10.0, seed=880, env_over={"solved_reward": 1e9},
               record_q_gap=True)
        # configs[2] at its real shapes (DuelingDDQN 6-128-128-128 / heads, B 128, SE hidden 128): 3 x 400 steps, 800 learn steps... and
        # one more episode to pass 1 000
        gen_g8("g8long_acrobot_dueling_fullshape", train_episodes=4, done_bias_shift=-10.0, seed=881, max_steps=350,
               env_yaml="default_config_acrobot.yaml", env_name="Acrobot-v1", env_cls="AcrobotEnv", agent_key="duelingddqn",
               agent_over={"hidden_size": 128, "hidden_layer": 2, "feature_dim": 128, "batch_size": 128, "init_episodes": 1, "test_episodes": 2},
               env_over={"hidden_size": 128, "solved_reward": 1e9}, record_q_gap=True)
        # configs[4] at its real shapes (actor 17-128-128-6, critics 23-128-128-1, B 192, RN 17-128-1): 5 x 260 steps, 1 040 learn steps
        gen_g8t("g8long_cheetah_td3_fullshape", seed=882,
                agent_over={"train_episodes": 5, "init_episodes": 1, "batch_size": 192, "hidden_size": 128, "hidden_layer": 2, "test_episodes": 1},
                env_over={"max_steps": 260, "hidden_size": 128, "solved_reward": 1e9})
        # ... and each of the three once more with EVERY weight of the fresh agent one ulp larger (same seeds, so the same random draws):
        # the reference's own sensitivity to a rounding-level difference
        gen_g8("g8long_cartpole_ddqn_bench_workload_ulp", train_episodes=20, done_bias_shift=-10.0, seed=880, env_over={"solved_reward": 1e9},
               perturb_ulp=True)
        gen_g8("g8long_acrobot_dueling_fullshape_ulp", train_episodes=4, done_bias_shift=-10.0, seed=881, max_steps=350,
               env_yaml="default_config_acrobot.yaml", env_name="Acrobot-v1", env_cls="AcrobotEnv", agent_key="duelingddqn",
               agent_over={"hidden_size": 128, "hidden_layer": 2, "feature_dim": 128, "batch_size": 128, "init_episodes": 1, "test_episodes": 2},
               env_over={"hidden_size": 128, "solved_reward": 1e9}, perturb_ulp=True)
        gen_g8t("g8long_cheetah_td3_fullshape_ulp", seed=882,
                agent_over={"train_episodes": 5, "init_episodes": 1, "batch_size": 192, "hidden_size": 128, "hidden_layer": 2, "test_episodes": 1},
                env_over={"max_steps": 260, "hidden_size": 128, "solved_reward": 1e9}, perturb_ulp=True)
    if "g8w" in which:
        # replay ring wraps (ReplayBuffer.add, utils.py:24-32): capacity 37 rows, ~90 env steps
        gen_g8("g8w_calc_score_cartpole_ringwrap", train_episodes=3, done_bias_shift=0.0, seed=802, max_steps=30,
               agent_over={"rb_size": 37, "batch_size": 24})


if __name__ == "__main__":
    main()
