"""Tensor-level host wrappers over the C-ABI (torch is plumbing: device memory + streams).

Every function takes/returns torch CUDA(HIP) tensors and enqueues on torch's current stream.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import DdqnCfg, InnerOut, MlpDesc, QlCfg, QlOut, Tapes, Td3Cfg, Td3Out, Td3Tapes


def require_device():
    if not torch.cuda.is_available():
        raise _lib.LenvError("learning_environments_amd needs a HIP device (MI355X); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, dtype, name):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise ValueError("%s must be a contiguous CUDA tensor of dtype %s" % (name, dtype))
    return t


def mlp_desc(in_dim, hidden, layers, out_dim, act, prelu=0.25, use_layer_norm=False):
    return MlpDesc(int(in_dim), int(hidden), int(layers), int(out_dim), _lib.ACT[act] if isinstance(act, str) else int(act),
                   float(prelu), 1 if use_layer_norm else 0)


def mlp_num_params(d):
    return int(_lib.lib().lenv_mlp_num_params(C.byref(d)))


def mlp_forward(d, params, x):
    """y [rows,out] = MLP(x [rows,in]) with the flat parameter vector `params` (device tensors)."""
    dev = require_device()
    _chk(params, torch.float32, "params"); _chk(x, torch.float32, "x")
    rows = x.shape[0]
    y = torch.empty((rows, d.out_dim), dtype=torch.float32, device=dev)
    rc = _lib.lib().lenv_mlp_forward(C.byref(d), _ptr(params), _ptr(x), rows, _ptr(y), _stream())
    _lib.check(rc, "lenv_mlp_forward")
    return y


def rn_num_params(rtype, state_dim, info_dim, hidden, layers):
    """Parameter count of RewardEnv.build_reward_net for a reward type (reference envs/reward_env.py:29-59)."""
    n = _lib.lib().lenv_rn_num_params(int(rtype), int(state_dim), int(info_dim), int(hidden), int(layers))
    _lib.check(int(n) if n < 0 else 0, "lenv_rn_num_params")
    return int(n)


def rn_shape_rows(rtype, rn_desc, state_dim, info_dim, gamma, theta, s, s2, info, r):
    """RewardEnv._calc_reward (reference envs/reward_env.py:68-133) for rows of a vector-state real env; device tensors.
    `info` may be None for the types that do not read it; for the others a missing info raises ValueError like the
    reference."""
    dev = require_device()
    for t, n in ((s, "s"), (s2, "s2"), (r, "r")):
        _chk(t, torch.float32, n)
    rows = s.shape[0]
    out = torch.empty(rows, dtype=torch.float32, device=dev)
    rc = _lib.lib().lenv_rn_shape_rows(int(rtype), C.byref(rn_desc) if rn_desc is not None else None, int(state_dim), int(info_dim),
                                       float(gamma), _ptr(theta) if theta is not None else None, _ptr(s), _ptr(s2),
                                       _ptr(info) if info is not None else None, _ptr(r), rows, _ptr(out), _stream())
    _lib.check(rc, "lenv_rn_shape_rows")
    return out


def se_descs(S, A, hidden, layers, act, prelu=0.25):
    return (mlp_desc(S + A, hidden, layers, S, act, prelu), mlp_desc(S + A, hidden, layers, 1, act, prelu),
            mlp_desc(S + A, hidden, layers, 1, act, prelu))


def se_step_population(descs, theta, eps, worker, sign, state, action):
    """state [chains,S] or [chains,n,S]; action int32 [chains] or [chains,n].  Returns (next_state, reward, done)."""
    dev = require_device()
    sn, rn, dn = descs
    squeeze = state.dim() == 2
    if squeeze:
        state, action = state.unsqueeze(1), action.unsqueeze(1)
    chains, n, S = state.shape
    _chk(theta, torch.float32, "theta"); _chk(state, torch.float32, "state"); _chk(action, torch.int32, "action")
    _chk(eps, torch.float32, "eps"); _chk(worker, torch.int32, "worker"); _chk(sign, torch.float32, "sign")
    ns = torch.empty((chains, n, S), dtype=torch.float32, device=dev)
    r = torch.empty((chains, n), dtype=torch.float32, device=dev)
    d = torch.empty((chains, n), dtype=torch.float32, device=dev)
    rc = _lib.lib().lenv_se_step_population(C.byref(sn), C.byref(rn), C.byref(dn), _ptr(theta), _ptr(eps), _ptr(worker),
                                            _ptr(sign), chains, n, _ptr(state), _ptr(action), _ptr(ns), _ptr(r), _ptr(d),
                                            _stream())
    _lib.check(rc, "lenv_se_step_population")
    if squeeze:
        return ns[:, 0], r[:, 0], d[:, 0]
    return ns, r, d


def qnet_td_forward(qd, online, target, replay, idx, gamma):
    """online/target [chains,P]; replay [chains,cap,row_stride]; idx int32 [chains,B] -> (q_sa, y) [chains,B]."""
    dev = require_device()
    for t, n in ((online, "online"), (target, "target"), (replay, "replay")):
        _chk(t, torch.float32, n)
    _chk(idx, torch.int32, "idx")
    chains, cap, stride = replay.shape
    B = idx.shape[1]
    q_sa = torch.empty((chains, B), dtype=torch.float32, device=dev)
    y = torch.empty((chains, B), dtype=torch.float32, device=dev)
    rc = _lib.lib().lenv_qnet_td_forward(C.byref(qd), _ptr(online), _ptr(target), _ptr(replay), cap, stride, _ptr(idx),
                                         chains, B, float(gamma), _ptr(q_sa), _ptr(y), _stream())
    _lib.check(rc, "lenv_qnet_td_forward")
    return q_sa, y


def chain_key(seed, generation, worker, kind):
    return int(_lib.lib().lenv_chain_key(seed, generation, worker, kind))


def _alloc_chain_hp(obj):
    """Device arrays of include/lenv_hip.h's lenv_chain_hp for obj.chains chains + the struct that points at them."""
    dev, n = obj.dev, obj.chains
    obj.hp = dict(lr=torch.zeros(n, dtype=torch.float64, device=dev), batch_size=torch.zeros(n, dtype=torch.int32, device=dev),
                  q_hidden=torch.zeros(n, dtype=torch.int32, device=dev), q_layers=torch.zeros(n, dtype=torch.int32, device=dev))
    obj.hp_struct = _lib.ChainHp(_ptr(obj.hp["lr"]), _ptr(obj.hp["batch_size"]), _ptr(obj.hp["q_hidden"]), _ptr(obj.hp["q_layers"]))
    obj.agent_init = torch.zeros((n, obj.p_agent), dtype=torch.float32, device=dev)


def _set_chain_hp(obj, lr, batch_size, hidden_size, hidden_layer, max_batch, max_hidden, max_layers):
    if not obj.vary:
        raise ValueError("the inner loop was built without vary=True")
    n = obj.chains
    if not (len(lr) == len(batch_size) == len(hidden_size) == len(hidden_layer) == n):
        raise ValueError("set_hp: need %d values per hyper-parameter" % n)
    layers = [max(1, int(v)) for v in hidden_layer]
    if max(batch_size) > max_batch or max(hidden_size) > max_hidden or max(layers) > max_layers or min(batch_size) < 1 \
            or min(hidden_size) < 1:
        raise ValueError("set_hp: a chain's hyper-parameters exceed the maxima the inner loop was sized for")
    obj.hp["lr"].copy_(torch.tensor([float(v) for v in lr], dtype=torch.float64))
    obj.hp["batch_size"].copy_(torch.tensor([int(v) for v in batch_size], dtype=torch.int32))
    obj.hp["q_hidden"].copy_(torch.tensor([int(v) for v in hidden_size], dtype=torch.int32))
    obj.hp["q_layers"].copy_(torch.tensor(layers, dtype=torch.int32))


def _draw_icm_init(obj, rng_keys, bounds):
    """Fresh ICMModel parameters per chain into obj.icm_init (nn.Linear default init with `bounds` [p_icm], counter-RNG stream
    12 of every chain key)."""
    _chk(rng_keys, torch.int64, "rng_keys"); _chk(bounds, torch.float32, "bounds")
    rc = _lib.lib().lenv_chain_uniform_init(_ptr(rng_keys), obj.chains, 12, obj.p_icm, _ptr(bounds), _ptr(obj.icm_init), _stream())
    _lib.check(rc, "lenv_chain_uniform_init")
    return obj.icm_init


class InnerLoop(object):
    """Owns the workspace/outputs of lenv_ddqn_se_inner_loop for a fixed (cfg, chains)."""

    def __init__(self, cfg, chains, want_episode_stats=True, want_final_online=False, trace_cap=0, vary=False):
        """vary=True: the *_vary agents -- cfg carries the MAXIMAL batch_size / q_hidden / q_layers, every chain runs with its
        own lr / batch_size / hidden_size / hidden_layer (set_hp) in the GEMM-tiled kernel (lenv_dueling_se_inner_loop_hp)."""
        self.dev = require_device()
        self.cfg, self.chains = cfg, int(chains)
        L = _lib.lib()
        E, T, S = cfg.train_episodes, cfg.test_episodes, cfg.state_dim
        self.vary = bool(vary)
        self.hp = self.hp_struct = self.agent_init = None
        # DuelingDDQN, and DDQN whose Critic_DQN the register-resident kernel refuses (hidden_layer >= 2 / wide layers),
        # run in the GEMM-tiled kernel; `dueling` keeps its name from the first of the two
        self.icm = bool(cfg.icm_enabled)               # ICM agents (ddqn_icm / duelingddqn_icm): GEMM-tiled kernel only
        self.icm_init = self.icm_final = self.icm_io = None
        # (a RewardEnv / real-env cfg with an explicit micro-chunk takes the register-resident kernel's RENV instantiations; with grad_chunk 0 --
        # one sequential batch gradient -- the probe refuses it and the GEMM-tiled kernel runs it)
        self.dueling = self.vary or self.icm or cfg.agent_kind == 1 or (cfg.agent_kind == 0 and L.lenv_ddqn_se_lds_bytes(C.byref(cfg)) <= 0
                                                                        and L.lenv_dueling_num_params(C.byref(cfg)) > 0)
        if self.dueling:
            self.p_agent = int(L.lenv_dueling_num_params(C.byref(cfg)))
            _lib.check(min(self.p_agent, 0), "lenv_dueling_num_params")
            self.ws_bytes = int(L.lenv_dueling_se_workspace_bytes(C.byref(cfg), self.chains))
            self._fn = L.lenv_dueling_se_inner_loop
            if self.vary:
                _alloc_chain_hp(self)
            if self.icm:
                self.p_icm = int(L.lenv_icm_num_params(C.byref(cfg)))
                _lib.check(min(self.p_icm, 0), "lenv_icm_num_params")
                self.icm_init = torch.zeros((self.chains, self.p_icm), dtype=torch.float32, device=self.dev)
                self.icm_final = torch.zeros((self.chains, self.p_icm), dtype=torch.float32, device=self.dev)
                self.icm_io = _lib.IcmIo(_ptr(self.icm_init), _ptr(self.icm_final))
        else:
            self.ws_bytes = int(L.lenv_ddqn_se_workspace_bytes(C.byref(cfg), self.chains))
            qd = mlp_desc(S, cfg.q_hidden, cfg.q_layers, cfg.num_actions, cfg.q_act)
            self.p_agent = mlp_num_params(qd)
            self._fn = L.lenv_ddqn_se_inner_loop
        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.dev)
        self.score = torch.zeros(self.chains, dtype=torch.float64, device=self.dev)
        self.stats = torch.zeros((self.chains, 4), dtype=torch.int64, device=self.dev)
        self.status = torch.zeros(self.chains, dtype=torch.int32, device=self.dev)
        self.episode_test_mean = self.episode_len = self.final_returns = self.final_online = None
        if want_episode_stats:
            self.episode_test_mean = torch.zeros((self.chains, max(E, 1)), dtype=torch.float64, device=self.dev)
            self.episode_len = torch.zeros((self.chains, max(E, 1)), dtype=torch.int32, device=self.dev)
            self.final_returns = torch.zeros((self.chains, T), dtype=torch.float64, device=self.dev)
        if want_final_online:
            self.final_online = torch.zeros((self.chains, self.p_agent), dtype=torch.float32, device=self.dev)
        self.trace_cap = int(trace_cap)
        self.trace = None
        if trace_cap:
            self.trace = dict(action=torch.zeros((self.chains, trace_cap), dtype=torch.int32, device=self.dev),
                              state=torch.zeros((self.chains, trace_cap, S), dtype=torch.float32, device=self.dev),
                              next_state=torch.zeros((self.chains, trace_cap, S), dtype=torch.float32, device=self.dev),
                              reward_done=torch.zeros((self.chains, trace_cap, 2), dtype=torch.float32, device=self.dev))
        tr = self.trace or {}
        self.out = InnerOut(_ptr(self.score), _ptr(self.stats), _ptr(self.status), _ptr(self.episode_test_mean),
                            _ptr(self.episode_len), _ptr(self.final_returns), _ptr(self.final_online), self.trace_cap,
                            _ptr(tr.get("action")), _ptr(tr.get("state")), _ptr(tr.get("next_state")),
                            _ptr(tr.get("reward_done")))

    def set_hp(self, lr, batch_size, hidden_size, hidden_layer):
        """The chains' own hyper-parameters (host sequences of length `chains`; hidden_layer as the config writes it: the
        network has max(1, hidden_layer) hidden layers, models/model_utils.py:33-37)."""
        _set_chain_hp(self, lr, batch_size, hidden_size, hidden_layer, self.cfg.batch_size, self.cfg.q_hidden, self.cfg.q_layers)

    def chain_num_params(self, hidden_size, hidden_layer):
        """Parameter count of one chain's agent at its own shapes (the used prefix of its agent_init / final_online row)."""
        probe = _lib.DdqnCfg.from_buffer_copy(self.cfg)
        probe.q_hidden, probe.q_layers = int(hidden_size), max(1, int(hidden_layer))
        n = int(_lib.lib().lenv_dueling_num_params(C.byref(probe)))
        _lib.check(min(n, 0), "lenv_dueling_num_params")
        return n

    def draw_icm_init(self, rng_keys, bounds):
        return _draw_icm_init(self, rng_keys, bounds)

    def draw_agent_init(self, rng_keys):
        """Fresh agents at every chain's own shapes into self.agent_init (nn.Linear default init, keyed by the chain keys)."""
        _chk(rng_keys, torch.int64, "rng_keys")
        rc = _lib.lib().lenv_dueling_agent_init_hp(C.byref(self.cfg), C.byref(self.hp_struct), _ptr(rng_keys), self.chains,
                                                   _ptr(self.agent_init), _stream())
        _lib.check(rc, "lenv_dueling_agent_init_hp")
        return self.agent_init

    def run(self, theta, eps, worker, sign, agent_init, rng_keys=None, tapes=None):
        """Enqueue one fused inner loop for all chains on the current stream (asynchronous)."""
        if agent_init is None and self.vary:
            agent_init = self.agent_init
        _chk(theta, torch.float32, "theta"); _chk(eps, torch.float32, "eps"); _chk(worker, torch.int32, "worker")
        _chk(sign, torch.float32, "sign"); _chk(agent_init, torch.float32, "agent_init")
        if agent_init.shape != (self.chains, self.p_agent):
            raise ValueError("agent_init must be [chains, %d]" % self.p_agent)
        t = None
        if tapes is not None:
            t = Tapes(_ptr(tapes["eps_uniform"]), tapes["eps_uniform"].shape[1],
                      _ptr(tapes["rand_action"]), tapes["rand_action"].shape[1],
                      _ptr(tapes["replay_idx"]), tapes["replay_idx"].shape[1],
                      _ptr(tapes["train_reset"]), tapes["train_reset"].shape[1],
                      _ptr(tapes["test_reset"]), tapes["test_reset"].shape[1])
        if rng_keys is not None:
            _chk(rng_keys, torch.int64, "rng_keys")
        args = (_ptr(theta), _ptr(eps), _ptr(worker), _ptr(sign), _ptr(agent_init), _ptr(rng_keys),
                C.byref(t) if t is not None else None, self.chains, _ptr(self.workspace), self.ws_bytes, C.byref(self.out), _stream())
        if self.icm:
            rc = _lib.lib().lenv_dueling_se_inner_loop_icm(C.byref(self.cfg), C.byref(self.hp_struct) if self.vary else None,
                                                           C.byref(self.icm_io), *args)
        elif self.vary:
            rc = _lib.lib().lenv_dueling_se_inner_loop_hp(C.byref(self.cfg), C.byref(self.hp_struct), *args)
        else:
            rc = self._fn(C.byref(self.cfg), *args)
        _lib.check(rc, "lenv_dueling_se_inner_loop" if self.dueling else "lenv_ddqn_se_inner_loop")
        return self.score


class QlInnerLoop(object):
    """Owns the outputs of lenv_ql_rn_inner_loop for a fixed (cfg, chains, grid MDP)."""

    def __init__(self, cfg, chains, tables, want_episode_stats=True, trace_cap=0):
        self.dev = require_device()
        self.cfg, self.chains = cfg, int(chains)
        N, A, E, T = cfg.n_states, cfg.n_actions, cfg.train_episodes, cfg.test_episodes
        self.next_state = torch.from_numpy(tables["next_state"].astype("int32")).contiguous().to(self.dev)
        self.reward = torch.from_numpy(tables["reward"].astype("float64")).contiguous().to(self.dev)
        self.done = torch.from_numpy(tables["done"].astype("uint8")).contiguous().to(self.dev)
        H = cfg.rn_hidden            # Linear(N, H) | (layers - 1) x Linear(H, H) | Linear(H, 1)
        self.p_theta = N * H + H + (max(1, cfg.rn_layers) - 1) * (H * H + H) + H + 1
        self.score = torch.zeros(self.chains, dtype=torch.float64, device=self.dev)
        self.stats = torch.zeros((self.chains, 4), dtype=torch.int64, device=self.dev)
        self.status = torch.zeros(self.chains, dtype=torch.int32, device=self.dev)
        self.episode_test_mean = self.episode_len = self.final_returns = self.q_table = self.shaped = None
        if want_episode_stats:
            self.episode_test_mean = torch.zeros((self.chains, max(E, 1)), dtype=torch.float64, device=self.dev)
            self.episode_len = torch.zeros((self.chains, max(E, 1)), dtype=torch.int32, device=self.dev)
            self.final_returns = torch.zeros((self.chains, T), dtype=torch.float64, device=self.dev)
            self.q_table = torch.zeros((self.chains, N * A), dtype=torch.float64, device=self.dev)
            self.shaped = torch.zeros((self.chains, N * A), dtype=torch.float32, device=self.dev)
        self.trace_cap = int(trace_cap)
        self.trace = None
        if trace_cap:
            self.trace = dict(action=torch.zeros((self.chains, trace_cap), dtype=torch.int32, device=self.dev),
                              state=torch.zeros((self.chains, trace_cap, 2), dtype=torch.int32, device=self.dev),
                              reward_done=torch.zeros((self.chains, trace_cap, 2), dtype=torch.float32, device=self.dev))
        tr = self.trace or {}
        self.out = QlOut(_ptr(self.score), _ptr(self.stats), _ptr(self.status), _ptr(self.episode_test_mean),
                         _ptr(self.episode_len), _ptr(self.final_returns), _ptr(self.q_table), _ptr(self.shaped),
                         self.trace_cap, _ptr(tr.get("action")), _ptr(tr.get("state")), _ptr(tr.get("reward_done")))

    def run(self, theta, eps, worker, sign, rng_keys=None, tapes=None, shaped_override=None):
        _chk(theta, torch.float32, "theta"); _chk(eps, torch.float32, "eps"); _chk(worker, torch.int32, "worker")
        _chk(sign, torch.float32, "sign"); _chk(shaped_override, torch.float32, "shaped_override")
        if theta is not None and theta.numel() != self.p_theta and self.cfg.reward_env_type != 0:
            raise ValueError("theta must hold %d reward-net parameters" % self.p_theta)
        t = None
        if tapes is not None:
            t = Tapes(_ptr(tapes["eps_uniform"]), tapes["eps_uniform"].shape[1], _ptr(tapes["rand_action"]),
                      tapes["rand_action"].shape[1], None, 0, None, 0, None, 0)
        if rng_keys is not None:
            _chk(rng_keys, torch.int64, "rng_keys")
        rc = _lib.lib().lenv_ql_rn_inner_loop(C.byref(self.cfg), _ptr(theta), _ptr(eps), _ptr(worker), _ptr(sign),
                                              _ptr(shaped_override), _ptr(self.next_state), _ptr(self.reward), _ptr(self.done),
                                              _ptr(rng_keys), C.byref(t) if t is not None else None, self.chains,
                                              C.byref(self.out), _stream())
        _lib.check(rc, "lenv_ql_rn_inner_loop")
        return self.score


class Td3InnerLoop(object):
    """Owns the workspace/outputs of lenv_td3_rn_inner_loop for a fixed (cfg, chains)."""

    def __init__(self, cfg, chains, want_episode_stats=True, want_final_params=False, trace_cap=0, vary=False):
        """vary=True: TD3_vary -- cfg carries the maximal batch_size / hidden / layers, every chain runs with its own
        hyper-parameters (set_hp) through lenv_td3_rn_inner_loop_hp."""
        self.dev = require_device()
        self.cfg, self.chains = cfg, int(chains)
        self.vary = bool(vary)
        self.hp = self.hp_struct = self.agent_init = None
        L = _lib.lib()
        pa, pc = C.c_int64(), C.c_int64()
        self.p_agent = int(L.lenv_td3_num_params(C.byref(cfg), C.byref(pa), C.byref(pc)))
        _lib.check(min(self.p_agent, 0), "lenv_td3_num_params")
        if self.vary:
            _alloc_chain_hp(self)
        self.icm = bool(cfg.icm_enabled)               # TD3(icm=True): select_agent "td3_icm" / "td3_icm_vary"
        self.icm_init = self.icm_final = self.icm_io = None
        if self.icm:
            self.p_icm = int(L.lenv_td3_icm_num_params(C.byref(cfg)))
            _lib.check(min(self.p_icm, 0), "lenv_td3_icm_num_params")
            self.icm_init = torch.zeros((self.chains, self.p_icm), dtype=torch.float32, device=self.dev)
            self.icm_final = torch.zeros((self.chains, self.p_icm), dtype=torch.float32, device=self.dev)
            self.icm_io = _lib.IcmIo(_ptr(self.icm_init), _ptr(self.icm_final))
        self.p_actor, self.p_critic = pa.value, pc.value
        self.p_theta = cfg.state_dim * cfg.rn_hidden + 2 * cfg.rn_hidden + 1
        self.ws_bytes = int(L.lenv_td3_rn_workspace_bytes(C.byref(cfg), self.chains))
        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.dev)
        E, T, S, A = cfg.train_episodes, cfg.test_episodes, cfg.state_dim, cfg.action_dim
        self.score = torch.zeros(self.chains, dtype=torch.float64, device=self.dev)
        self.stats = torch.zeros((self.chains, 4), dtype=torch.int64, device=self.dev)
        self.status = torch.zeros(self.chains, dtype=torch.int32, device=self.dev)
        self.episode_test_mean = self.episode_len = self.final_returns = self.final_params = None
        if want_episode_stats:
            self.episode_test_mean = torch.zeros((self.chains, max(E, 1)), dtype=torch.float64, device=self.dev)
            self.episode_len = torch.zeros((self.chains, max(E, 1)), dtype=torch.int32, device=self.dev)
            self.final_returns = torch.zeros((self.chains, T), dtype=torch.float64, device=self.dev)
        if want_final_params:
            self.final_params = torch.zeros((self.chains, self.p_agent), dtype=torch.float32, device=self.dev)
        self.trace_cap = int(trace_cap)
        self.trace = None
        if trace_cap:
            self.trace = dict(action=torch.zeros((self.chains, trace_cap, A), dtype=torch.float32, device=self.dev),
                              state=torch.zeros((self.chains, trace_cap, S), dtype=torch.float32, device=self.dev),
                              next_state=torch.zeros((self.chains, trace_cap, S), dtype=torch.float32, device=self.dev),
                              reward=torch.zeros((self.chains, trace_cap), dtype=torch.float32, device=self.dev))
        tr = self.trace or {}
        self.out = Td3Out(_ptr(self.score), _ptr(self.stats), _ptr(self.status), _ptr(self.episode_test_mean), _ptr(self.episode_len),
                          _ptr(self.final_returns), _ptr(self.final_params), self.trace_cap, _ptr(tr.get("action")),
                          _ptr(tr.get("state")), _ptr(tr.get("next_state")), _ptr(tr.get("reward")))

    def set_hp(self, lr, batch_size, hidden_size, hidden_layer):
        _set_chain_hp(self, lr, batch_size, hidden_size, hidden_layer, self.cfg.batch_size, self.cfg.hidden, self.cfg.layers)

    def chain_num_params(self, hidden_size, hidden_layer):
        probe = _lib.Td3Cfg.from_buffer_copy(self.cfg)
        probe.hidden, probe.layers = int(hidden_size), max(1, int(hidden_layer))
        n = int(_lib.lib().lenv_td3_num_params(C.byref(probe), None, None))
        _lib.check(min(n, 0), "lenv_td3_num_params")
        return n

    def draw_icm_init(self, rng_keys, bounds):
        return _draw_icm_init(self, rng_keys, bounds)

    def draw_agent_init(self, rng_keys):
        """Fresh actor | critic_1 | critic_2 at every chain's own shapes into self.agent_init."""
        _chk(rng_keys, torch.int64, "rng_keys")
        rc = _lib.lib().lenv_td3_agent_init_hp(C.byref(self.cfg), C.byref(self.hp_struct), _ptr(rng_keys), self.chains,
                                               _ptr(self.agent_init), _stream())
        _lib.check(rc, "lenv_td3_agent_init_hp")
        return self.agent_init

    def run(self, theta, eps, worker, sign, agent_init, rng_keys=None, tapes=None):
        if agent_init is None and self.vary:
            agent_init = self.agent_init
        _chk(theta, torch.float32, "theta"); _chk(eps, torch.float32, "eps"); _chk(worker, torch.int32, "worker")
        _chk(sign, torch.float32, "sign"); _chk(agent_init, torch.float32, "agent_init")
        if agent_init.shape != (self.chains, self.p_agent):
            raise ValueError("agent_init must be [chains, %d]" % self.p_agent)
        t = None
        if tapes is not None:
            t = Td3Tapes(_ptr(tapes["rand_action"]), tapes["rand_action"].shape[1], _ptr(tapes["act_noise"]), tapes["act_noise"].shape[1],
                         _ptr(tapes["test_noise"]), tapes["test_noise"].shape[1], _ptr(tapes["policy_noise"]), tapes["policy_noise"].shape[1],
                         _ptr(tapes["replay_idx"]), tapes["replay_idx"].shape[1], _ptr(tapes["train_reset"]), tapes["train_reset"].shape[1],
                         _ptr(tapes["test_reset"]), tapes["test_reset"].shape[1])
        if rng_keys is not None:
            _chk(rng_keys, torch.int64, "rng_keys")
        args = (_ptr(theta), _ptr(eps), _ptr(worker), _ptr(sign), _ptr(agent_init), _ptr(rng_keys),
                C.byref(t) if t is not None else None, self.chains, _ptr(self.workspace), self.ws_bytes, C.byref(self.out), _stream())
        if self.icm:
            rc = _lib.lib().lenv_td3_rn_inner_loop_icm(C.byref(self.cfg), C.byref(self.hp_struct) if self.vary else None,
                                                       C.byref(self.icm_io), *args)
        elif self.vary:
            rc = _lib.lib().lenv_td3_rn_inner_loop_hp(C.byref(self.cfg), C.byref(self.hp_struct), *args)
        else:
            rc = _lib.lib().lenv_td3_rn_inner_loop(C.byref(self.cfg), *args)
        _lib.check(rc, "lenv_td3_rn_inner_loop")
        return self.score


class Td3DiscreteInnerLoop(object):
    """Owns the workspace/outputs of lenv_td3d_inner_loop (TD3_discrete_vary on a VirtualEnv) for a fixed (cfg, chains).
    vary=True: cfg carries the maximal batch_size / hidden / layers, every chain runs with its own draw (set_hp)."""

    def __init__(self, cfg, chains, want_episode_stats=True, want_final_params=False, trace_cap=0, vary=False):
        self.dev = require_device()
        self.cfg, self.chains = cfg, int(chains)
        self.vary = bool(vary)
        self.hp = self.hp_struct = self.agent_init = None
        L = _lib.lib()
        pa, pc = C.c_int64(), C.c_int64()
        self.p_agent = int(L.lenv_td3d_num_params(C.byref(cfg), C.byref(pa), C.byref(pc)))
        _lib.check(min(self.p_agent, 0), "lenv_td3d_num_params")
        self.p_actor, self.p_critic = pa.value, pc.value
        self.p_theta = int(L.lenv_td3d_se_num_params(C.byref(cfg)))
        if self.vary:
            _alloc_chain_hp(self)
        else:
            self.agent_init = torch.zeros((self.chains, self.p_agent), dtype=torch.float32, device=self.dev)
        self.ws_bytes = int(L.lenv_td3d_workspace_bytes(C.byref(cfg), self.chains))
        self.workspace = torch.empty(self.ws_bytes, dtype=torch.uint8, device=self.dev)
        E, T, S, A = cfg.train_episodes, cfg.test_episodes, cfg.state_dim, cfg.action_dim
        self.score = torch.zeros(self.chains, dtype=torch.float64, device=self.dev)
        self.stats = torch.zeros((self.chains, 4), dtype=torch.int64, device=self.dev)
        self.status = torch.zeros(self.chains, dtype=torch.int32, device=self.dev)
        self.episode_test_mean = self.episode_len = self.final_returns = self.final_params = None
        if want_episode_stats:
            self.episode_test_mean = torch.zeros((self.chains, max(E, 1)), dtype=torch.float64, device=self.dev)
            self.episode_len = torch.zeros((self.chains, max(E, 1)), dtype=torch.int32, device=self.dev)
            self.final_returns = torch.zeros((self.chains, T), dtype=torch.float64, device=self.dev)
        if want_final_params:
            self.final_params = torch.zeros((self.chains, self.p_agent), dtype=torch.float32, device=self.dev)
        self.trace_cap = int(trace_cap)
        self.trace = None
        if trace_cap:
            self.trace = dict(action=torch.zeros((self.chains, trace_cap, A), dtype=torch.float32, device=self.dev),
                              state=torch.zeros((self.chains, trace_cap, S), dtype=torch.float32, device=self.dev),
                              next_state=torch.zeros((self.chains, trace_cap, S), dtype=torch.float32, device=self.dev),
                              reward=torch.zeros((self.chains, trace_cap), dtype=torch.float32, device=self.dev))
        tr = self.trace or {}
        self.out = Td3Out(_ptr(self.score), _ptr(self.stats), _ptr(self.status), _ptr(self.episode_test_mean), _ptr(self.episode_len),
                          _ptr(self.final_returns), _ptr(self.final_params), self.trace_cap, _ptr(tr.get("action")),
                          _ptr(tr.get("state")), _ptr(tr.get("next_state")), _ptr(tr.get("reward")))

    def set_hp(self, lr, batch_size, hidden_size, hidden_layer):
        _set_chain_hp(self, lr, batch_size, hidden_size, hidden_layer, self.cfg.batch_size, self.cfg.hidden, self.cfg.layers)

    def chain_num_params(self, hidden_size, hidden_layer):
        probe = _lib.Td3dCfg.from_buffer_copy(self.cfg)
        probe.hidden, probe.layers = int(hidden_size), max(1, int(hidden_layer))
        n = int(_lib.lib().lenv_td3d_num_params(C.byref(probe), None, None))
        _lib.check(min(n, 0), "lenv_td3d_num_params")
        return n

    def draw_agent_init(self, rng_keys):
        """Fresh actor | critic_1 | critic_2 (nn.Linear default init, LayerNorm 1 / 0) at every chain's own shapes."""
        _chk(rng_keys, torch.int64, "rng_keys")
        rc = _lib.lib().lenv_td3d_agent_init(C.byref(self.cfg), C.byref(self.hp_struct) if self.vary else None, _ptr(rng_keys), self.chains,
                                             _ptr(self.agent_init), _stream())
        _lib.check(rc, "lenv_td3d_agent_init")
        return self.agent_init

    def run(self, theta, eps, worker, sign, agent_init=None, rng_keys=None, tapes=None):
        if agent_init is None:
            agent_init = self.agent_init
        _chk(theta, torch.float32, "theta"); _chk(eps, torch.float32, "eps"); _chk(worker, torch.int32, "worker")
        _chk(sign, torch.float32, "sign"); _chk(agent_init, torch.float32, "agent_init")
        if agent_init.shape != (self.chains, self.p_agent):
            raise ValueError("agent_init must be [chains, %d]" % self.p_agent)
        if theta.numel() != self.p_theta:
            raise ValueError("theta must hold %d SE parameters" % self.p_theta)
        t = None
        if tapes is not None:
            vals = []
            for k in _lib.TD3D_TAPE_KEYS:
                vals += [_ptr(tapes[k]), tapes[k].shape[1]]
            t = _lib.Td3dTapes(*vals)
        if rng_keys is not None:
            _chk(rng_keys, torch.int64, "rng_keys")
        rc = _lib.lib().lenv_td3d_inner_loop(C.byref(self.cfg), C.byref(self.hp_struct) if self.vary else None, _ptr(theta), _ptr(eps),
                                             _ptr(worker), _ptr(sign), _ptr(agent_init), _ptr(rng_keys), C.byref(t) if t is not None else None,
                                             self.chains, _ptr(self.workspace), self.ws_bytes, C.byref(self.out), _stream())
        _lib.check(rc, "lenv_td3d_inner_loop")
        return self.score


def rn_shape_population(cfg, theta, eps, worker, sign, next_state, reward, chains=1):
    """(phi [chains,N], shaped [chains,N,A]) of a population of perturbed reward networks on a grid MDP."""
    dev = require_device()
    N, A = cfg.n_states, cfg.n_actions
    phi = torch.empty((chains, N), dtype=torch.float32, device=dev)
    shaped = torch.empty((chains, N, A), dtype=torch.float32, device=dev)
    rc = _lib.lib().lenv_rn_shape_population(C.byref(cfg), _ptr(theta), _ptr(eps), _ptr(worker), _ptr(sign), chains,
                                             _ptr(next_state), _ptr(reward), _ptr(phi), _ptr(shaped), _stream())
    _lib.check(rc, "lenv_rn_shape_population")
    return phi, shaped


GRAD_EVAL_TYPES = {"mean": 0, "minmax": 1}


def nes_worker_best(chain_scores, pop, mirrored=True, num_grad_evals=1, grad_eval_type="mean", out=None):
    """GTN_Worker.calc_best_score for `pop` workers; chain_scores [pop, 1+2G] = (orig, add_1..G, sub_1..G)."""
    dev = require_device()
    _chk(chain_scores, torch.float64, "chain_scores")
    if grad_eval_type not in GRAD_EVAL_TYPES:
        raise NotImplementedError('Unknown parameter for grad_eval_type: ' + str(grad_eval_type))
    result = out if out is not None else torch.empty((pop, 4), dtype=torch.float64, device=dev)
    _chk(result, torch.float64, "out")
    rc = _lib.lib().lenv_nes_worker_best_multi(_ptr(chain_scores), pop, int(num_grad_evals), 1 if mirrored else 0,
                                               GRAD_EVAL_TYPES[grad_eval_type], _ptr(result), _stream())
    _lib.check(rc, "lenv_nes_worker_best_multi")
    return result


def nes_draw(seed, generation, pop, p_theta, noise_std, chains, chains_per_worker, worker_lo, bounds, want_keys=True):
    """(eps [pop,p_theta], agent_init [chains,p_agent] or None, rng_keys int64 [chains]) of one generation, one launch.
    `generation` may be a device int64 tensor [1] (captured generations: read when the kernel runs)."""
    dev = require_device()
    eps = torch.empty((pop, p_theta), dtype=torch.float32, device=dev)
    init = torch.empty((chains, bounds.numel()), dtype=torch.float32, device=dev) if bounds is not None and chains > 0 else None
    keys = torch.empty(chains, dtype=torch.int64, device=dev) if want_keys and chains > 0 else None
    tail = (pop, p_theta, float(noise_std), _ptr(eps), chains, int(chains_per_worker), int(worker_lo),
            bounds.numel() if bounds is not None else 0, _ptr(bounds), _ptr(init), _ptr(keys), _stream())
    if torch.is_tensor(generation):
        rc = _lib.lib().lenv_nes_draw_dev(int(seed) & (2 ** 64 - 1), _ptr(_chk(generation, torch.int64, "generation")), *tail)
    else:
        rc = _lib.lib().lenv_nes_draw(int(seed) & (2 ** 64 - 1), int(generation), *tail)
    _lib.check(rc, "lenv_nes_draw")
    return eps, init, keys


def nes_status_fold(status, result):
    rc = _lib.lib().lenv_nes_status_fold(_ptr(status), status.numel(), _ptr(result), result.shape[0], _stream())
    _lib.check(rc, "lenv_nes_status_fold")


def nes_rank_update(score_transform_type, gathered, rank_table, theta, eps, step_size, nes_step_size=False, weight_decay=0.0,
                    theta_prev=None, generation=None):
    """In-place theta update; returns the score_transform weights [pop] (float64).  theta_prev / generation (device tensors):
    the captured-generation form, lenv_nes_rank_update_keep (theta_prev <- theta before the update, generation[0] += 1)."""
    dev = require_device()
    _chk(gathered, torch.float64, "gathered"); _chk(rank_table, torch.float64, "rank_table")
    _chk(theta, torch.float32, "theta"); _chk(eps, torch.float32, "eps")
    pop = gathered.shape[0]
    weights = torch.empty(pop, dtype=torch.float64, device=dev)
    head = (int(score_transform_type), _ptr(gathered), _ptr(rank_table), pop, _ptr(theta), _ptr(eps),
            theta.numel() if theta is not None else 0, float(step_size), 1 if nes_step_size else 0, float(weight_decay), _ptr(weights))
    if theta_prev is not None or generation is not None:
        rc = _lib.lib().lenv_nes_rank_update_keep(*head, _ptr(_chk(theta_prev, torch.float32, "theta_prev")),
                                                  _ptr(_chk(generation, torch.int64, "generation")), _stream())
    else:
        rc = _lib.lib().lenv_nes_rank_update(*head, _stream())
    _lib.check(rc, "lenv_nes_rank_update")
    return weights


class HipNesEngine(object):
    """The compute engine GTN_Master/GTN_Worker drive: every method is a thin call into liblenv_hip.so.
    (tests substitute an oracle-backed object with the same methods to exercise the host/distributed logic on CPU)"""
    name = "hip"

    def __init__(self):
        self.device = require_device()

    def make_inner(self, cfg, chains, **kw):
        return InnerLoop(cfg, chains, **kw)

    def make_inner_td3(self, cfg, chains, **kw):
        return Td3InnerLoop(cfg, chains, **kw)

    def make_inner_td3d(self, cfg, chains, **kw):
        return Td3DiscreteInnerLoop(cfg, chains, **kw)

    def inner_scores_td3(self, inner, theta, eps, worker, sign, agent_init, rng_keys):
        return inner.run(theta, eps, worker, sign, agent_init, rng_keys=rng_keys)

    def make_inner_ql(self, cfg, chains, tables, **kw):
        return QlInnerLoop(cfg, chains, tables, **kw)

    def inner_scores_ql(self, inner, theta, eps, worker, sign, rng_keys):
        return inner.run(theta, eps, worker, sign, rng_keys=rng_keys)

    def inner_scores(self, inner, theta, eps, worker, sign, agent_init, rng_keys):
        return inner.run(theta, eps, worker, sign, agent_init, rng_keys=rng_keys)

    def check_status(self, inner):
        st = inner.status.cpu()
        if int(st.min()) != 0:
            raise _lib.LenvError("inner loop reported status %s" % st.tolist())

    def run_checked(self, inner, *args, **kw):
        """inner.run(...) + host check of the chain statuses (synchronises).  A launch whose teams of workgroups could not assemble
        (status -10: a foreign kernel held CUs, include/lenv_hip.h lenv_ddqn_cfg::team_size) is repeated once with one workgroup per
        chain -- the chains are deterministic functions of their inputs -- and the inner loop keeps that setting."""
        out = inner.run(*args, **kw)
        st = inner.status.cpu()
        if int(st.min()) == _lib.STATUS_TEAM_GAVE_UP and hasattr(inner.cfg, "team_size") and inner.cfg.team_size != 1:
            inner.cfg.team_size = 1
            out = inner.run(*args, **kw)
            st = inner.status.cpu()
        if int(st.min()) != 0:
            raise _lib.LenvError("inner loop reported status %s" % st.tolist())
        return out

    def worker_best(self, chain_scores, pop, mirrored, num_grad_evals=1, grad_eval_type="mean", out=None):
        return nes_worker_best(chain_scores, pop, mirrored, num_grad_evals, grad_eval_type, out=out)

    def draw(self, seed, generation, pop, p_theta, noise_std, chains, chains_per_worker, worker_lo, bounds):
        return nes_draw(seed, generation, pop, p_theta, noise_std, chains, chains_per_worker, worker_lo, bounds)

    def status_fold(self, inner, result):
        nes_status_fold(inner.status, result)

    def rank_update(self, score_transform_type, gathered, rank_table, theta, eps, step_size, nes_step_size, weight_decay,
                    theta_prev=None, generation=None):
        return nes_rank_update(score_transform_type, gathered, rank_table, theta, eps, step_size, nes_step_size, weight_decay,
                               theta_prev=theta_prev, generation=generation)

    graph_capable = True            # every call above only enqueues on the current stream: a generation can be captured
