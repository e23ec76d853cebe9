"""ctypes binding of the CPU oracle (oracle/lenv_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the learning_environments_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liblenv_oracle.so")

ACT = {"identity": 0, "relu": 1, "leakyrelu": 2, "tanh": 3, "prelu": 4}
ENV = {"CartPole-v0": 0, "Acrobot-v1": 1, "MountainCar-v0": 3}


class MlpDesc(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32), ("out_dim", C.c_int32),
                ("act", C.c_int32), ("prelu", C.c_float), ("use_layer_norm", C.c_int32)]


class DdqnCfg(C.Structure):
    _fields_ = [("env_id", C.c_int32), ("state_dim", C.c_int32), ("num_actions", C.c_int32), ("max_steps", C.c_int32),
                ("se_hidden", C.c_int32), ("se_layers", C.c_int32), ("se_act", C.c_int32), ("se_prelu", C.c_float),
                ("q_hidden", C.c_int32), ("q_layers", C.c_int32), ("q_act", C.c_int32), ("q_prelu", C.c_float),
                ("batch_size", C.c_int32), ("rb_size", C.c_int32),
                ("train_episodes", C.c_int32), ("test_episodes", C.c_int32), ("init_episodes", C.c_int32),
                ("early_out_num", C.c_int32), ("grad_chunk", C.c_int32), ("rng_mode", C.c_int32),
                ("agent_kind", C.c_int32), ("feature_dim", C.c_int32),
                ("solved_reward", C.c_double), ("gamma", C.c_double), ("lr", C.c_double), ("tau", C.c_double),
                ("eps_init", C.c_double), ("eps_min", C.c_double), ("eps_decay", C.c_double),
                ("adam_beta1", C.c_double), ("adam_beta2", C.c_double), ("adam_eps", C.c_double),
                ("step_budget", C.c_int64),
                ("icm_enabled", C.c_int32), ("icm_feature_dim", C.c_int32), ("icm_hidden", C.c_int32), ("se_layer_norm", C.c_int32),
                ("icm_lr", C.c_double), ("icm_beta", C.c_double), ("icm_eta", C.c_double),
                ("synthetic_env_type", C.c_int32), ("reward_env_type", C.c_int32),
                ("same_action_num", C.c_int32), ("q_layer_norm", C.c_int32), ("test_mode", C.c_int32), ("early_out_virtual_diff", C.c_double)]


class Tapes(C.Structure):
    _fields_ = [("eps_uniform", C.POINTER(C.c_double)), ("n_eps_uniform", C.c_int64),
                ("rand_action", C.POINTER(C.c_int32)), ("n_rand_action", C.c_int64),
                ("replay_idx", C.POINTER(C.c_int32)), ("n_replay_idx", C.c_int64),
                ("train_reset", C.POINTER(C.c_double)), ("n_train_reset", C.c_int64),
                ("test_reset", C.POINTER(C.c_double)), ("n_test_reset", C.c_int64)]


class Trace(C.Structure):
    _fields_ = [("cap", C.c_int64), ("n", C.c_int64), ("episode", C.POINTER(C.c_int32)), ("action", C.POINTER(C.c_int32)),
                ("explored", C.POINTER(C.c_int32)), ("state", C.POINTER(C.c_float)), ("next_state", C.POINTER(C.c_float)),
                ("reward", C.POINTER(C.c_float)), ("done", C.POINTER(C.c_float)), ("loss", C.POINTER(C.c_float))]


class QlCfg(C.Structure):
    _fields_ = [("n_states", C.c_int32), ("n_actions", C.c_int32), ("start_state", C.c_int32), ("max_steps", C.c_int32),
                ("rn_hidden", C.c_int32), ("rn_layers", C.c_int32), ("rn_act", C.c_int32), ("rn_prelu", C.c_float),
                ("reward_env_type", C.c_int32), ("train_episodes", C.c_int32), ("test_episodes", C.c_int32),
                ("init_episodes", C.c_int32), ("early_out_num", C.c_int32), ("batch_size", C.c_int32), ("rng_mode", C.c_int32),
                ("agent_kind", C.c_int32), ("count_based", C.c_int32),
                ("solved_reward", C.c_double), ("alpha", C.c_double), ("gamma", C.c_double), ("eps_init", C.c_double),
                ("eps_min", C.c_double), ("eps_decay", C.c_double), ("beta", C.c_double), ("step_budget", C.c_int64), ("same_action_num", C.c_int32), ("rn_layer_norm", C.c_int32), ("test_mode", C.c_int32), ("early_out_virtual_diff", C.c_double)]


class QlTrace(C.Structure):
    _fields_ = [("cap", C.c_int64), ("n", C.c_int64), ("action", C.POINTER(C.c_int32)), ("state", C.POINTER(C.c_int32)),
                ("next_state", C.POINTER(C.c_int32)), ("reward", C.POINTER(C.c_float)), ("done", C.POINTER(C.c_float))]


class Td3Cfg(C.Structure):
    _fields_ = [("env_id", C.c_int32), ("state_dim", C.c_int32), ("action_dim", C.c_int32), ("max_steps", C.c_int32),
                ("rn_hidden", C.c_int32), ("rn_layers", C.c_int32), ("rn_act", C.c_int32), ("rn_prelu", C.c_float),
                ("reward_env_type", C.c_int32), ("info_dim", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32), ("act", C.c_int32),
                ("prelu", C.c_float), ("batch_size", C.c_int32), ("rb_size", C.c_int32), ("train_episodes", C.c_int32),
                ("test_episodes", C.c_int32), ("init_episodes", C.c_int32), ("early_out_num", C.c_int32),
                ("policy_delay", C.c_int32), ("rng_mode", C.c_int32),
                ("solved_reward", C.c_double), ("gamma", C.c_double), ("lr", C.c_double), ("tau", C.c_double),
                ("action_std", C.c_double), ("policy_std", C.c_double), ("policy_std_clip", C.c_double),
                ("max_action", C.c_double), ("adam_beta1", C.c_double), ("adam_beta2", C.c_double), ("adam_eps", C.c_double),
                ("step_budget", C.c_int64),
                ("icm_enabled", C.c_int32), ("icm_feature_dim", C.c_int32), ("icm_hidden", C.c_int32), ("use_layer_norm", C.c_int32),
                ("icm_lr", C.c_double), ("icm_beta", C.c_double), ("icm_eta", C.c_double),
                ("virtual_env", C.c_int32), ("same_action_num", C.c_int32), ("rn_layer_norm", C.c_int32), ("test_mode", C.c_int32), ("early_out_virtual_diff", C.c_double)]


class Td3Tapes(C.Structure):
    _fields_ = [("rand_action", C.POINTER(C.c_float)), ("n_rand_action", C.c_int64),
                ("act_noise", C.POINTER(C.c_float)), ("n_act_noise", C.c_int64),
                ("test_noise", C.POINTER(C.c_float)), ("n_test_noise", C.c_int64),
                ("policy_noise", C.POINTER(C.c_float)), ("n_policy_noise", C.c_int64),
                ("replay_idx", C.POINTER(C.c_int32)), ("n_replay_idx", C.c_int64),
                ("train_reset", C.POINTER(C.c_double)), ("n_train_reset", C.c_int64),
                ("test_reset", C.POINTER(C.c_double)), ("n_test_reset", C.c_int64)]


class Td3dCfg(C.Structure):
    """orc_td3d_cfg: TD3_discrete_vary on a VirtualEnv over a discrete-action real env."""
    _fields_ = [("env_id", C.c_int32), ("state_dim", C.c_int32), ("action_dim", C.c_int32), ("max_steps", C.c_int32),
                ("se_hidden", C.c_int32), ("se_layers", C.c_int32), ("se_act", C.c_int32), ("se_prelu", C.c_float),
                ("hidden", C.c_int32), ("layers", C.c_int32), ("act", C.c_int32), ("prelu", C.c_float),
                ("use_layer_norm", C.c_int32), ("gumbel_hard", C.c_int32),
                ("batch_size", C.c_int32), ("rb_size", C.c_int32), ("train_episodes", C.c_int32), ("test_episodes", C.c_int32),
                ("init_episodes", C.c_int32), ("early_out_num", C.c_int32), ("policy_delay", C.c_int32), ("rng_mode", C.c_int32),
                ("solved_reward", C.c_double), ("gamma", C.c_double), ("lr", C.c_double), ("tau", C.c_double),
                ("action_std", C.c_double), ("policy_std", C.c_double), ("policy_std_clip", C.c_double), ("max_action", C.c_double),
                ("gumbel_temp", C.c_double), ("adam_beta1", C.c_double), ("adam_beta2", C.c_double), ("adam_eps", C.c_double),
                ("step_budget", C.c_int64), ("se_layer_norm", C.c_int32), ("test_mode", C.c_int32), ("early_out_virtual_diff", C.c_double)]


class Td3dTapes(C.Structure):
    _fields_ = [("rand_action", C.POINTER(C.c_int32)), ("n_rand_action", C.c_int64),
                ("act_noise", C.POINTER(C.c_float)), ("n_act_noise", C.c_int64),
                ("test_noise", C.POINTER(C.c_float)), ("n_test_noise", C.c_int64),
                ("policy_noise", C.POINTER(C.c_float)), ("n_policy_noise", C.c_int64),
                ("gumbel_act", C.POINTER(C.c_float)), ("n_gumbel_act", C.c_int64),
                ("gumbel_test", C.POINTER(C.c_float)), ("n_gumbel_test", C.c_int64),
                ("gumbel_target", C.POINTER(C.c_float)), ("n_gumbel_target", C.c_int64),
                ("gumbel_actor", C.POINTER(C.c_float)), ("n_gumbel_actor", C.c_int64),
                ("replay_idx", C.POINTER(C.c_int32)), ("n_replay_idx", C.c_int64),
                ("train_reset", C.POINTER(C.c_double)), ("n_train_reset", C.c_int64),
                ("test_reset", C.POINTER(C.c_double)), ("n_test_reset", C.c_int64)]


class Td3Trace(C.Structure):
    _fields_ = [("cap", C.c_int64), ("n", C.c_int64), ("action", C.POINTER(C.c_float)), ("state", C.POINTER(C.c_float)),
                ("next_state", C.POINTER(C.c_float)), ("reward", C.POINTER(C.c_float))]


class ChainResult(C.Structure):
    _fields_ = [("score", C.c_double), ("episodes_run", C.c_int32), ("train_steps", C.c_int64),
                ("learn_steps", C.c_int64), ("test_steps", C.c_int64)]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "lenv_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_tanhf.restype = C.c_float
        _lib.orc_tanhf.argtypes = [C.c_float]
        _lib.orc_tanhf_scan.restype = C.c_int64
        _lib.orc_tanhf_scan.argtypes = [C.c_uint32, C.c_uint32, C.c_float, C.POINTER(C.c_float)]
        _lib.orc_sin.restype = C.c_double
        _lib.orc_sin.argtypes = [C.c_double]
        _lib.orc_cos.restype = C.c_double
        _lib.orc_cos.argtypes = [C.c_double]
        _lib.orc_mix64.restype = C.c_uint64
        _lib.orc_mix64.argtypes = [C.c_uint64]
        _lib.orc_chain_key.restype = C.c_uint64
        _lib.orc_chain_key.argtypes = [C.c_uint64] * 4
        _lib.orc_rng_u64.restype = C.c_uint64
        _lib.orc_rng_u64.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
        _lib.orc_mlp_num_params.restype = C.c_int64
        _lib.orc_ddqn_learn.restype = C.c_float
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def mlp_desc(in_dim, hidden, layers, out_dim, act, prelu=0.25, use_layer_norm=False):
    return MlpDesc(in_dim, hidden, layers, out_dim, ACT[act] if isinstance(act, str) else act, prelu, 1 if use_layer_norm else 0)


def mlp_num_params(d):
    return int(lib().orc_mlp_num_params(C.byref(d)))


def tanhf(x):
    x = _f32(x)
    out = np.empty_like(x)
    L = lib()
    flat_in, flat_out = x.reshape(-1), out.reshape(-1)
    for i in range(flat_in.size):
        flat_out[i] = L.orc_tanhf(float(flat_in[i]))
    return out


def tanhf_scan(lo, hi, slack=0.0):
    """(number of violations, max result) of orc_tanhf over every float in [lo, hi] (lo, hi >= 0)."""
    lo_b = int(np.array([lo], np.float32).view(np.uint32)[0])
    hi_b = int(np.array([hi], np.float32).view(np.uint32)[0])
    mx = C.c_float(0.0)
    bad = lib().orc_tanhf_scan(lo_b, hi_b, C.c_float(slack), C.byref(mx))
    return int(bad), float(mx.value)


def mlp_forward(d, params, x, want_hidden=False):
    params, x = _f32(params), _f32(x)
    assert params.size == mlp_num_params(d), (params.size, mlp_num_params(d))
    x2 = x.reshape(-1, d.in_dim)
    B = x2.shape[0]
    y = np.empty((B, d.out_dim), np.float32)
    h = np.empty((B, d.hidden), np.float32) if want_hidden else None
    rc = lib().orc_mlp_forward(C.byref(d), _p(params, C.c_float), _p(x2, C.c_float), C.c_int64(B), _p(y, C.c_float),
                               _p(h, C.c_float) if want_hidden else None)
    assert rc == 0
    return (y, h) if want_hidden else y


def se_descs(S, A, hidden, layers, act, prelu=0.25):
    return (mlp_desc(S + A, hidden, layers, S, act, prelu), mlp_desc(S + A, hidden, layers, 1, act, prelu),
            mlp_desc(S + A, hidden, layers, 1, act, prelu))


def se_step_population(descs, theta, eps, worker, sign, state, action):
    sn, rn, dn = descs
    theta = _f32(theta)
    state = _f32(state)
    chains, S = state.shape
    action = np.ascontiguousarray(action, dtype=np.int32)
    ns = np.empty((chains, S), np.float32)
    r = np.empty(chains, np.float32)
    d = np.empty(chains, np.float32)
    if eps is not None:
        eps = _f32(eps)
        worker = np.ascontiguousarray(worker, dtype=np.int32)
        sign = _f32(sign)
    rc = lib().orc_se_step_population(C.byref(sn), C.byref(rn), C.byref(dn), _p(theta, C.c_float),
                                      _p(eps, C.c_float) if eps is not None else None,
                                      _p(worker, C.c_int32) if eps is not None else None,
                                      _p(sign, C.c_float) if eps is not None else None,
                                      C.c_int64(chains), _p(state, C.c_float), _p(action, C.c_int32),
                                      _p(ns, C.c_float), _p(r, C.c_float), _p(d, C.c_float))
    assert rc == 0
    return ns, r, d


def qnet_td_forward(qd, online, target, rows, S, gamma):
    online, target, rows = _f32(online), _f32(target), _f32(rows)
    B, stride = rows.shape
    q_sa = np.empty(B, np.float32)
    y = np.empty(B, np.float32)
    am = np.empty(B, np.int32)
    rc = lib().orc_qnet_td_forward(C.byref(qd), _p(online, C.c_float), _p(target, C.c_float), _p(rows, C.c_float),
                                   C.c_int64(stride), C.c_int64(B), C.c_int32(S), C.c_double(gamma),
                                   _p(q_sa, C.c_float), _p(y, C.c_float), _p(am, C.c_int32))
    assert rc == 0
    return q_sa, y, am


def ddqn_learn(cfg, online, target, m, v, step, b1pow, b2pow, rows):
    """In-place on copies; returns (loss, online, target, m, v, b1pow, b2pow)."""
    online, target, m, v = [_f32(t).copy() for t in (online, target, m, v)]
    rows = _f32(rows)
    b1, b2 = C.c_double(b1pow), C.c_double(b2pow)
    loss = lib().orc_ddqn_learn(C.byref(cfg), _p(online, C.c_float), _p(target, C.c_float), _p(m, C.c_float),
                                _p(v, C.c_float), C.c_int64(step), C.byref(b1), C.byref(b2), _p(rows, C.c_float),
                                C.c_int64(rows.shape[1]))
    return float(loss), online, target, m, v, b1.value, b2.value


def make_tapes(eps_uniform, rand_action, replay_idx, train_reset, test_reset):
    keep = [np.ascontiguousarray(eps_uniform, np.float64), np.ascontiguousarray(rand_action, np.int32),
            np.ascontiguousarray(replay_idx, np.int32).reshape(-1),
            np.ascontiguousarray(train_reset, np.float64).reshape(-1, 4),
            np.ascontiguousarray(test_reset, np.float64).reshape(-1, 4)]
    t = Tapes(_p(keep[0], C.c_double), keep[0].size, _p(keep[1], C.c_int32), keep[1].size,
              _p(keep[2], C.c_int32), keep[2].size, _p(keep[3], C.c_double), keep[3].shape[0],
              _p(keep[4], C.c_double), keep[4].shape[0])
    t._keep = keep
    return t


def icm_num_params(cfg):
    L = lib()
    L.orc_icm_num_params.restype = C.c_int64
    return int(L.orc_icm_num_params(cfg.state_dim, cfg.num_actions, cfg.icm_feature_dim, cfg.icm_hidden))


def ddqn_se_chain(cfg, se_params, agent_init, rng_key=0, tapes=None, trace_cap=0, icm_init=None, want_final_online=False):
    """icm_init: fresh ICMModel parameters (state-dict order) for an agent with cfg.icm_enabled; the result then carries
    "icm_final" (the ICM parameters after the last learn step).  want_final_online: "final_online" = the trained online net."""
    se_params, agent_init = _f32(se_params), _f32(agent_init)
    E, T, S = cfg.train_episodes, cfg.test_episodes, cfg.state_dim
    ep_mean = np.full(max(E, 1), np.nan)
    ep_len = np.zeros(max(E, 1), np.int32)
    final = np.zeros(max(T, 1))
    res = ChainResult()
    tr = None
    arrs = None
    if trace_cap:
        arrs = dict(episode=np.zeros(trace_cap, np.int32), action=np.zeros(trace_cap, np.int32),
                    explored=np.zeros(trace_cap, np.int32), state=np.zeros((trace_cap, S), np.float32),
                    next_state=np.zeros((trace_cap, S), np.float32), reward=np.zeros(trace_cap, np.float32),
                    done=np.zeros(trace_cap, np.float32), loss=np.zeros(trace_cap, np.float32))
        tr = Trace(trace_cap, 0, _p(arrs["episode"], C.c_int32), _p(arrs["action"], C.c_int32),
                   _p(arrs["explored"], C.c_int32), _p(arrs["state"], C.c_float), _p(arrs["next_state"], C.c_float),
                   _p(arrs["reward"], C.c_float), _p(arrs["done"], C.c_float), _p(arrs["loss"], C.c_float))
    icm_final = None
    if icm_init is not None:
        icm_init = _f32(icm_init)
        assert icm_init.size == icm_num_params(cfg), (icm_init.size, icm_num_params(cfg))
        icm_final = np.zeros_like(icm_init)
        rc = lib().orc_ddqn_se_chain_icm(C.byref(cfg), _p(se_params, C.c_float), _p(agent_init, C.c_float), _p(icm_init, C.c_float),
                                         C.c_uint64(rng_key), C.byref(tapes) if tapes is not None else None, _p(ep_mean, C.c_double),
                                         _p(ep_len, C.c_int32), _p(final, C.c_double), C.byref(tr) if tr is not None else None,
                                         C.byref(res), _p(icm_final, C.c_float))
    elif want_final_online:
        final_online = np.zeros_like(agent_init)
        rc = lib().orc_ddqn_se_chain_params(C.byref(cfg), _p(se_params, C.c_float), _p(agent_init, C.c_float), C.c_uint64(rng_key),
                                            C.byref(tapes) if tapes is not None else None, _p(ep_mean, C.c_double),
                                            _p(ep_len, C.c_int32), _p(final, C.c_double), C.byref(tr) if tr is not None else None,
                                            C.byref(res), _p(final_online, C.c_float))
    else:
        rc = lib().orc_ddqn_se_chain(C.byref(cfg), _p(se_params, C.c_float), _p(agent_init, C.c_float), C.c_uint64(rng_key),
                                     C.byref(tapes) if tapes is not None else None, _p(ep_mean, C.c_double),
                                     _p(ep_len, C.c_int32), _p(final, C.c_double), C.byref(tr) if tr is not None else None,
                                     C.byref(res))
    out = dict(rc=rc, score=res.score, episodes_run=res.episodes_run, train_steps=res.train_steps,
               learn_steps=res.learn_steps, test_steps=res.test_steps, episode_test_mean=ep_mean[:E],
               episode_len=ep_len[:E], final_test_returns=final[:T])
    if want_final_online and icm_init is None:
        out["final_online"] = final_online
    if tr is not None:
        n = tr.n
        out["trace"] = {k: v[:n] for k, v in arrs.items()}
    if icm_final is not None:
        out["icm_final"] = icm_final
    return out


def ddqn_se_population(cfg, theta, eps, agent_init, seed, generation, worker_offset=0, threads=1, want_results=False):
    theta, eps, agent_init = _f32(theta), _f32(eps), _f32(agent_init)
    pop, p_theta = eps.shape
    assert agent_init.shape[0] == 3 * pop
    scores = np.zeros(3 * pop)
    results = (ChainResult * (3 * pop))()
    rc = lib().orc_ddqn_se_population(C.byref(cfg), _p(theta, C.c_float), _p(eps, C.c_float), C.c_int64(pop),
                                      C.c_int64(p_theta), _p(agent_init, C.c_float), C.c_uint64(seed),
                                      C.c_uint64(generation), C.c_int64(worker_offset), C.c_int(threads),
                                      _p(scores, C.c_double), results)
    assert rc == 0, rc
    if want_results:
        return scores, [dict(score=r.score, episodes_run=r.episodes_run, train_steps=r.train_steps,
                             learn_steps=r.learn_steps, test_steps=r.test_steps) for r in results]
    return scores


def chain_key(seed, generation, worker, kind):
    return int(lib().orc_chain_key(seed, generation, worker, kind))


def nes_draw(seed, generation, pop, p_theta, noise_std, chains, cpw, worker_lo, bounds):
    """(eps, agent_init or None, rng_keys uint64) of one generation: the CPU twin of lenv_nes_draw."""
    eps = np.empty((pop, p_theta), np.float32)
    init = np.empty((chains, bounds.size), np.float32) if bounds is not None and chains > 0 else None
    keys = np.empty(chains, np.uint64)
    b = _f32(bounds) if bounds is not None else None
    L = lib()
    L.orc_nes_draw.restype = None
    L.orc_nes_draw(C.c_uint64(int(seed) & (2 ** 64 - 1)), C.c_uint64(int(generation)), C.c_int64(pop), C.c_int64(p_theta),
                   C.c_float(noise_std), _p(eps, C.c_float), C.c_int64(chains), C.c_int64(cpw), C.c_int64(worker_lo),
                   C.c_int64(bounds.size if bounds is not None else 0), _p(b, C.c_float) if b is not None else None,
                   _p(init, C.c_float) if init is not None else None, keys.ctypes.data_as(C.POINTER(C.c_uint64)))
    return eps, init, keys


def worker_best(score_add, score_sub, mirrored=True):
    a = np.ascontiguousarray(score_add, np.float64)
    s = np.ascontiguousarray(score_sub, np.float64)
    best = np.empty_like(a)
    sign = np.empty(a.size, np.float32)
    lib().orc_worker_best(_p(a, C.c_double), _p(s, C.c_double), C.c_int64(a.size), C.c_int(1 if mirrored else 0),
                          _p(best, C.c_double), _p(sign, C.c_float))
    return best, sign


def score_transform(type_, scores, scores_orig):
    s = np.ascontiguousarray(scores, np.float64)
    so = np.ascontiguousarray(scores_orig, np.float64)
    out = np.empty_like(s)
    rc = lib().orc_score_transform(C.c_int(type_), _p(s, C.c_double), _p(so, C.c_double), C.c_int64(s.size), _p(out, C.c_double))
    if rc:
        raise ValueError("Unknown rank transform type: " + str(type_))
    return out


def update_env(theta, eps, sign, weights, step_size, nes_step_size=False, weight_decay=0.0):
    theta = _f32(theta).copy()
    eps = _f32(eps)
    sign = _f32(sign)
    w = np.ascontiguousarray(weights, np.float64)
    pop, p_theta = eps.shape
    lib().orc_update_env(_p(theta, C.c_float), _p(eps, C.c_float), _p(sign, C.c_float), _p(w, C.c_double),
                         C.c_int64(pop), C.c_int64(p_theta), None, C.c_double(step_size),
                         C.c_int(1 if nes_step_size else 0), C.c_double(weight_decay))
    return theta


def ddqn_cfg_from_config(config, grad_chunk=13, rng_mode=0, **overrides):
    """Reference YAML dict -> oracle config (DDQN agent + virtual env).  Mirrors the fields read at
    agents/DDQN.py:15-38, agents/base_agent.py:9-26, envs/env_factory.py:45-59."""
    env_name = config["env_name"]
    e = config["envs"][env_name]
    agent_key = config["agents"]["gtn"]["agent_name"].lower() if "gtn" in config["agents"] else "ddqn"
    if agent_key.endswith("_vary"):                  # the *_vary agents read their base agent's section (DDQN_vary.py:14)
        agent_key = agent_key[:-5]
    icm = agent_key.endswith("_icm")                 # select_agent: ddqn_icm / duelingddqn_icm = the agent with icm=True
    if icm:
        agent_key = agent_key[:-4]
        ic = config["agents"]["icm"]                 # DDQN.py:43-49
        overrides.setdefault("icm_enabled", 1)
        overrides.setdefault("icm_feature_dim", int(ic["feature_dim"]))
        overrides.setdefault("icm_hidden", int(ic["hidden_size"]))
        overrides.setdefault("icm_lr", float(ic["lr"]))
        overrides.setdefault("icm_beta", float(ic["beta"]))
        overrides.setdefault("icm_eta", float(ic["eta"]))
    a = config["agents"][agent_key]
    S, A = {"CartPole-v0": (4, 2), "Acrobot-v1": (6, 3), "MountainCar-v0": (2, 3)}[env_name]
    overrides.setdefault("same_action_num", int(a["same_action_num"]))
    overrides.setdefault("agent_kind", 1 if agent_key == "duelingddqn" else 0)
    overrides.setdefault("feature_dim", int(a.get("feature_dim", 0)))
    overrides.setdefault("q_layer_norm", 1 if a.get("use_layer_norm", False) else 0)      # model_utils.py:22-29
    overrides.setdefault("se_layer_norm", 1 if e.get("use_layer_norm", False) else 0)     # the SE nets' (never perturbed) LayerNorm
    cfg = DdqnCfg(env_id=ENV[env_name], state_dim=S, num_actions=A, max_steps=int(e["max_steps"]),
                  se_hidden=int(e["hidden_size"]), se_layers=int(e["hidden_layer"]), se_act=ACT[e["activation_fn"]],
                  se_prelu=0.25, q_hidden=int(a["hidden_size"]), q_layers=max(1, int(a["hidden_layer"])),   # hidden_layer 0 == 1 (model_utils.py:33)
                  q_act=ACT[a["activation_fn"]], q_prelu=0.25, batch_size=int(a["batch_size"]),
                  rb_size=int(a["rb_size"]), train_episodes=int(a["train_episodes"]),
                  test_episodes=int(a["test_episodes"]), init_episodes=int(a["init_episodes"]),
                  early_out_num=int(a["early_out_num"]), grad_chunk=grad_chunk, rng_mode=rng_mode,
                  solved_reward=float(e["solved_reward"]), gamma=float(a["gamma"]), lr=float(a["lr"]),
                  tau=float(a["tau"]), eps_init=float(a["eps_init"]), eps_min=float(a["eps_min"]),
                  eps_decay=float(a["eps_decay"]), adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8,
                  step_budget=int(a.get("step_budget", 0)), early_out_virtual_diff=float(a.get("early_out_virtual_diff", 0.0)))
    if "gtn" in config["agents"] and int(config["agents"]["gtn"].get("synthetic_env_type", 0)) == 1:
        # the agent trains on a RewardEnv over the real env; the `envs` section describes the reward network (env_factory.py:45-59)
        cfg.synthetic_env_type, cfg.reward_env_type = 1, int(e["reward_env_type"])
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


def ql_cfg_from_config(config, tables, rng_mode=0, **overrides):
    """Reference YAML dict (QL agent + reward env on a gridworld) -> oracle config.  Fields read at agents/QL.py:13-27,
    agents/base_agent.py:9-26, envs/reward_env.py:8-27."""
    env_name = config["env_name"]
    e = config["envs"][env_name]
    name = config["agents"].get("gtn", {}).get("agent_name", "ql").lower()
    if name not in ("ql", "ql_cb", "sarsa", "sarsa_cb"):
        name = "ql"
    a = config["agents"]["sarsa" if name.startswith("sarsa") else "ql"]
    cfg = QlCfg(agent_kind=1 if name.startswith("sarsa") else 0, count_based=1 if name.endswith("_cb") else 0,
                beta=float(a.get("beta", 0.0)), n_states=tables["n_states"], n_actions=tables["n_actions"], start_state=tables["start_state"],
                max_steps=int(e["max_steps"]), rn_hidden=int(e["hidden_size"]), rn_layers=int(e["hidden_layer"]),
                rn_act=ACT[e["activation_fn"]], rn_prelu=0.25, reward_env_type=int(e["reward_env_type"]),
                train_episodes=int(a["train_episodes"]), test_episodes=int(a["test_episodes"]),
                init_episodes=int(a["init_episodes"]), early_out_num=int(a["early_out_num"]), batch_size=int(a["batch_size"]),
                rng_mode=rng_mode, solved_reward=float(e["solved_reward"]), alpha=float(a["alpha"]), gamma=float(a["gamma"]),
                eps_init=float(a["eps_init"]), eps_min=float(a["eps_min"]), eps_decay=float(a["eps_decay"]),
                step_budget=int(a.get("step_budget", 0)))
    cfg.same_action_num = int(a["same_action_num"])
    cfg.rn_layer_norm = 1 if e.get("use_layer_norm", False) else 0     # the reward net's own (never perturbed) LayerNorm
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


def rn_shaped_rewards(cfg, rn_params, tables):
    rn_params = _f32(rn_params)
    nxt = np.ascontiguousarray(tables["next_state"], np.int32)
    rew = np.ascontiguousarray(tables["reward"], np.float64)
    N, A = nxt.shape
    phi = np.empty(N, np.float32)
    shaped = np.empty((N, A), np.float32)
    rc = lib().orc_rn_shaped_rewards(C.byref(cfg), _p(rn_params, C.c_float), _p(nxt, C.c_int32), _p(rew, C.c_double),
                                     _p(phi, C.c_float), _p(shaped, C.c_float))
    assert rc == 0
    return phi, shaped


def ql_rn_chain(cfg, rn_params, tables, rng_key=0, tapes=None, trace_cap=0, shaped_override=None):
    rn_params = _f32(rn_params)
    nxt = np.ascontiguousarray(tables["next_state"], np.int32)
    rew = np.ascontiguousarray(tables["reward"], np.float64)
    dn = np.ascontiguousarray(tables["done"], np.uint8)
    N, A = nxt.shape
    E, T = cfg.train_episodes, cfg.test_episodes
    ep_mean = np.full(max(E, 1), np.nan)
    ep_len = np.zeros(max(E, 1), np.int32)
    final = np.zeros(max(T, 1))
    q = np.zeros((N, A))
    res = ChainResult()
    tr, arrs = None, None
    if trace_cap:
        arrs = dict(action=np.zeros(trace_cap, np.int32), state=np.zeros(trace_cap, np.int32),
                    next_state=np.zeros(trace_cap, np.int32), reward=np.zeros(trace_cap, np.float32),
                    done=np.zeros(trace_cap, np.float32))
        tr = QlTrace(trace_cap, 0, _p(arrs["action"], C.c_int32), _p(arrs["state"], C.c_int32),
                     _p(arrs["next_state"], C.c_int32), _p(arrs["reward"], C.c_float), _p(arrs["done"], C.c_float))
    so = _f32(shaped_override) if shaped_override is not None else None
    rc = lib().orc_ql_rn_chain(C.byref(cfg), _p(rn_params, C.c_float), _p(so, C.c_float) if so is not None else None,
                               _p(nxt, C.c_int32), _p(rew, C.c_double),
                               _p(dn, C.c_uint8), C.c_uint64(rng_key), C.byref(tapes) if tapes is not None else None,
                               _p(ep_mean, C.c_double), _p(ep_len, C.c_int32), _p(final, C.c_double), _p(q, C.c_double),
                               C.byref(tr) if tr is not None else None, C.byref(res))
    out = dict(rc=rc, score=res.score, episodes_run=res.episodes_run, train_steps=res.train_steps,
               learn_steps=res.learn_steps, test_steps=res.test_steps, episode_test_mean=ep_mean[:E],
               episode_len=ep_len[:E], final_test_returns=final[:T], q_table=q)
    if tr is not None:
        out["trace"] = {k: v[:tr.n] for k, v in arrs.items()}
    return out


def dueling_num_params(cfg):
    lib().orc_dueling_num_params.restype = C.c_int64
    return int(lib().orc_dueling_num_params(C.byref(cfg)))


def dueling_forward(cfg, params, x):
    params, x = _f32(params), _f32(x).reshape(-1, cfg.state_dim)
    q = np.empty((x.shape[0], cfg.num_actions), np.float32)
    rc = lib().orc_dueling_forward(C.byref(cfg), _p(params, C.c_float), _p(x, C.c_float), C.c_int64(x.shape[0]), _p(q, C.c_float))
    assert rc == 0
    return q


def dueling_learn(cfg, online, target, m, v, b1pow, b2pow, rows):
    online, target, m, v = [_f32(t).copy() for t in (online, target, m, v)]
    rows = _f32(rows)
    b1, b2 = C.c_double(b1pow), C.c_double(b2pow)
    lib().orc_dueling_learn.restype = C.c_float
    loss = lib().orc_dueling_learn(C.byref(cfg), _p(online, C.c_float), _p(target, C.c_float), _p(m, C.c_float),
                                   _p(v, C.c_float), C.byref(b1), C.byref(b2), _p(rows, C.c_float), C.c_int64(rows.shape[1]))
    return float(loss), online, target, m, v, b1.value, b2.value


def td3_cfg_from_config(config, rng_mode=0, **overrides):
    """Reference YAML dict (TD3 agent + reward env on the HalfCheetah stand-in) -> oracle config; fields read at
    agents/TD3.py:13-29, agents/base_agent.py:9-26, envs/reward_env.py:8-27."""
    env_name = config["env_name"]
    env_id, S, A, max_action = TD3_ENVS[env_name]
    e = config["envs"][env_name]
    a = config["agents"]["td3"]
    cfg = Td3Cfg(env_id=env_id, state_dim=S, action_dim=A, max_steps=int(e["max_steps"]), rn_hidden=int(e["hidden_size"]),
                 rn_layers=int(e["hidden_layer"]), rn_act=ACT[e["activation_fn"]], rn_prelu=0.25,
                 reward_env_type=int(e["reward_env_type"]), info_dim=int(e.get("info_dim", 0)), hidden=int(a["hidden_size"]),
                 layers=int(a["hidden_layer"]), act=ACT[a["activation_fn"]], prelu=0.25, batch_size=int(a["batch_size"]), rb_size=int(a["rb_size"]),
                 train_episodes=int(a["train_episodes"]), test_episodes=int(a["test_episodes"]),
                 init_episodes=int(a["init_episodes"]), early_out_num=int(a["early_out_num"]),
                 policy_delay=int(a["policy_delay"]), rng_mode=rng_mode, solved_reward=float(e["solved_reward"]),
                 gamma=float(a["gamma"]), lr=float(a["lr"]), tau=float(a["tau"]), action_std=float(a["action_std"]),
                 policy_std=float(a["policy_std"]), policy_std_clip=float(a["policy_std_clip"]), max_action=max_action,
                 adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8, step_budget=int(a.get("step_budget", 0)),
                 early_out_virtual_diff=float(a.get("early_out_virtual_diff", 0.0)))
    if "gtn" in config["agents"] and int(config["agents"]["gtn"].get("synthetic_env_type", 1)) == 0:
        cfg.virtual_env = 1                           # the agent trains on a VirtualEnv: the `envs` section describes the three SE nets
    cfg.same_action_num = int(a["same_action_num"])
    cfg.use_layer_norm = 1 if a.get("use_layer_norm", False) else 0      # model_utils.py:22-29
    cfg.rn_layer_norm = 1 if e.get("use_layer_norm", False) else 0       # the env nets' own (never perturbed) LayerNorm
    name = config["agents"]["gtn"]["agent_name"].lower() if "gtn" in config["agents"] else "td3"
    if name.replace("_vary", "").endswith("_icm"):   # select_agent "td3_icm": TD3(icm=True), agents/TD3.py:44-60
        ic = config["agents"]["icm"]
        cfg.icm_enabled, cfg.icm_feature_dim, cfg.icm_hidden = 1, int(ic["feature_dim"]), int(ic["hidden_size"])
        cfg.icm_lr, cfg.icm_beta, cfg.icm_eta = float(ic["lr"]), float(ic["beta"]), float(ic["eta"])
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


# continuous real envs of the TD3 path: env id, observation dim, action dim, EnvWrapper.get_max_action (env_wrapper.py:106-110)
TD3_ENVS = {"HalfCheetah-v3": (2, 17, 6, 1.0), "Pendulum-v0": (4, 3, 1, 2.0), "MountainCarContinuous-v0": (5, 2, 1, 1.0)}
TD3_STATE_WORDS = {2: 17, 4: 2, 5: 2}          # fp64 words of the env's own state (= width of the reset tapes)


def td3_param_counts(cfg):
    L = lib()
    L.orc_td3_actor_params.restype = C.c_int64
    L.orc_td3_critic_params.restype = C.c_int64
    return int(L.orc_td3_actor_params(C.byref(cfg))), int(L.orc_td3_critic_params(C.byref(cfg)))


def td3_actor_forward(cfg, actor, s):
    actor, s = _f32(actor), _f32(s).reshape(-1, cfg.state_dim)
    out = np.empty((s.shape[0], cfg.action_dim), np.float32)
    lib().orc_td3_actor_forward(C.byref(cfg), _p(actor, C.c_float), _p(s, C.c_float), C.c_int64(s.shape[0]), _p(out, C.c_float))
    return out


def td3_critic_forward(cfg, critic, s, a):
    critic, s, a = _f32(critic), _f32(s).reshape(-1, cfg.state_dim), _f32(a).reshape(-1, cfg.action_dim)
    out = np.empty(s.shape[0], np.float32)
    lib().orc_td3_critic_forward(C.byref(cfg), _p(critic, C.c_float), _p(s, C.c_float), _p(a, C.c_float), C.c_int64(s.shape[0]),
                                 _p(out, C.c_float))
    return out


def td3_learn(cfg, params, targets, m, v, pows, total_it, rows, policy_noise):
    params, targets, m, v = [_f32(t).copy() for t in (params, targets, m, v)]
    rows, policy_noise = _f32(rows), _f32(policy_noise)
    pw = (C.c_double * 4)(*pows)
    losses = (C.c_float * 2)(float("nan"), float("nan"))
    rc = lib().orc_td3_learn(C.byref(cfg), _p(params, C.c_float), _p(targets, C.c_float), _p(m, C.c_float), _p(v, C.c_float),
                             pw, C.c_int64(total_it), _p(rows, C.c_float), C.c_int64(rows.shape[1]), _p(policy_noise, C.c_float), losses)
    assert rc == 0
    return params, targets, m, v, list(pw), (losses[0], losses[1])


def make_td3_tapes(rand_action, act_noise, test_noise, policy_noise, replay_idx, train_reset, test_reset, A=6, S=17):
    keep = [np.ascontiguousarray(rand_action, np.float32).reshape(-1, A), np.ascontiguousarray(act_noise, np.float32).reshape(-1, A),
            np.ascontiguousarray(test_noise, np.float32).reshape(-1, A), np.ascontiguousarray(policy_noise, np.float32).reshape(-1, A),
            np.ascontiguousarray(replay_idx, np.int32).reshape(-1), np.ascontiguousarray(train_reset, np.float64).reshape(-1, S),
            np.ascontiguousarray(test_reset, np.float64).reshape(-1, S)]
    t = Td3Tapes(_p(keep[0], C.c_float), keep[0].shape[0], _p(keep[1], C.c_float), keep[1].shape[0],
                 _p(keep[2], C.c_float), keep[2].shape[0], _p(keep[3], C.c_float), keep[3].shape[0],
                 _p(keep[4], C.c_int32), keep[4].size, _p(keep[5], C.c_double), keep[5].shape[0], _p(keep[6], C.c_double), keep[6].shape[0])
    t._keep = keep
    return t


def td3_icm_num_params(cfg):
    L = lib()
    L.orc_icm_num_params_continuous.restype = C.c_int64
    return int(L.orc_icm_num_params_continuous(cfg.state_dim, cfg.action_dim, cfg.icm_feature_dim, cfg.icm_hidden))


def td3_rn_chain(cfg, rn_params, agent_init, rng_key=0, tapes=None, trace_cap=0, icm_init=None, want_final_params=False):
    rn_params, agent_init = _f32(rn_params), _f32(agent_init)
    E, T, S, A = cfg.train_episodes, cfg.test_episodes, cfg.state_dim, cfg.action_dim
    ep_mean = np.full(max(E, 1), np.nan)
    ep_len = np.zeros(max(E, 1), np.int32)
    final = np.zeros(max(T, 1))
    res = ChainResult()
    tr, arrs = None, None
    if trace_cap:
        arrs = dict(action=np.zeros((trace_cap, A), np.float32), state=np.zeros((trace_cap, S), np.float32),
                    next_state=np.zeros((trace_cap, S), np.float32), reward=np.zeros(trace_cap, np.float32))
        tr = Td3Trace(trace_cap, 0, _p(arrs["action"], C.c_float), _p(arrs["state"], C.c_float), _p(arrs["next_state"], C.c_float),
                      _p(arrs["reward"], C.c_float))
    icm_final = None
    if icm_init is not None:
        icm_init = _f32(icm_init)
        assert icm_init.size == td3_icm_num_params(cfg), (icm_init.size, td3_icm_num_params(cfg))
        icm_final = np.zeros_like(icm_init)
        rc = lib().orc_td3_rn_chain_icm(C.byref(cfg), _p(rn_params, C.c_float), _p(agent_init, C.c_float), _p(icm_init, C.c_float),
                                        C.c_uint64(rng_key), C.byref(tapes) if tapes is not None else None, _p(ep_mean, C.c_double),
                                        _p(ep_len, C.c_int32), _p(final, C.c_double), C.byref(tr) if tr is not None else None,
                                        C.byref(res), _p(icm_final, C.c_float))
    elif want_final_params:
        final_params = np.zeros_like(agent_init)
        rc = lib().orc_td3_rn_chain_params(C.byref(cfg), _p(rn_params, C.c_float), _p(agent_init, C.c_float), C.c_uint64(rng_key),
                                           C.byref(tapes) if tapes is not None else None, _p(ep_mean, C.c_double), _p(ep_len, C.c_int32),
                                           _p(final, C.c_double), C.byref(tr) if tr is not None else None, C.byref(res), _p(final_params, C.c_float))
    else:
        rc = lib().orc_td3_rn_chain(C.byref(cfg), _p(rn_params, C.c_float), _p(agent_init, C.c_float), C.c_uint64(rng_key),
                                    C.byref(tapes) if tapes is not None else None, _p(ep_mean, C.c_double), _p(ep_len, C.c_int32),
                                    _p(final, C.c_double), C.byref(tr) if tr is not None else None, C.byref(res))
    out = dict(rc=rc, score=res.score, episodes_run=res.episodes_run, train_steps=res.train_steps, learn_steps=res.learn_steps,
               test_steps=res.test_steps, episode_test_mean=ep_mean[:E], episode_len=ep_len[:E], final_test_returns=final[:T])
    if want_final_params and icm_init is None:
        out["final_params"] = final_params
    if tr is not None:
        out["trace"] = {k: v[:tr.n] for k, v in arrs.items()}
    if icm_final is not None:
        out["icm_final"] = icm_final
    return out


# ---- TD3_discrete_vary (oracle/lenv_oracle_td3d.inc) ----
TD3D_TAPE_KEYS = ("rand_action", "act_noise", "test_noise", "policy_noise", "gumbel_act", "gumbel_test", "gumbel_target", "gumbel_actor",
                  "replay_idx", "train_reset", "test_reset")


TD3D_ENVS = {"CartPole-v0": (0, 4, 2), "Acrobot-v1": (1, 6, 3), "MountainCar-v0": (3, 2, 3)}      # env id, observation dim, actions


def td3d_cfg_from_config(config, rng_mode=0, hp=None, **overrides):
    """Reference YAML dict (TD3_discrete_vary on a VirtualEnv) -> oracle config; fields read at agents/TD3_discrete_vary.py:30-42,
    models/actor_critic.py:27-31, models/model_utils.py:5-29, agents/base_agent.py:9-26, envs/env_wrapper.py:106-110 (max_action 1).
    hp: the recorded draw of vary_hyperparameters (lr / batch_size / hidden_size / hidden_layer), if any."""
    env_name = config["env_name"]
    env_id, S, A = TD3D_ENVS[env_name]
    e = config["envs"][env_name]
    a = dict(config["agents"]["td3_discrete_vary"])
    a.update(hp or {})

    def val(v):
        return float(v[1]) if isinstance(v, list) else v
    cfg = Td3dCfg(env_id=env_id, state_dim=S, action_dim=A, max_steps=int(val(e["max_steps"])), se_hidden=int(val(e["hidden_size"])),
                  se_layers=int(val(e["hidden_layer"])), se_act=ACT[e["activation_fn"]], se_prelu=0.25, hidden=int(a["hidden_size"]),
                  layers=max(1, int(a["hidden_layer"])), act=ACT[a["activation_fn"]], prelu=0.25,
                  use_layer_norm=int(bool(a.get("use_layer_norm", False))), gumbel_hard=int(bool(a["gumbel_softmax_hard"])),
                  se_layer_norm=int(bool(e.get("use_layer_norm", False))),
                  batch_size=int(a["batch_size"]), rb_size=int(a["rb_size"]), train_episodes=int(a["train_episodes"]),
                  test_episodes=int(a["test_episodes"]), init_episodes=int(a["init_episodes"]), early_out_num=int(a["early_out_num"]),
                  policy_delay=int(a["policy_delay"]), rng_mode=rng_mode, solved_reward=float(val(e["solved_reward"])),
                  gamma=float(a["gamma"]), lr=float(a["lr"]), tau=float(a["tau"]), action_std=float(a["action_std"]),
                  policy_std=float(a["policy_std"]), policy_std_clip=float(a["policy_std_clip"]), max_action=1.0,
                  gumbel_temp=float(a["gumbel_softmax_temp"]), adam_beta1=0.9, adam_beta2=0.999, adam_eps=1e-8,
                  step_budget=int(a.get("step_budget", 0)), early_out_virtual_diff=float(a.get("early_out_virtual_diff", 0.0)))
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


def td3d_num_params(cfg):
    L = lib()
    L.orc_td3d_actor_params.restype = C.c_int64
    L.orc_td3d_critic_params.restype = C.c_int64
    pa, pc = int(L.orc_td3d_actor_params(C.byref(cfg))), int(L.orc_td3d_critic_params(C.byref(cfg)))
    return pa + 2 * pc, pa, pc


def td3d_temperature(cfg, n):
    lib().orc_td3d_temperature.restype = C.c_float
    return float(lib().orc_td3d_temperature(C.byref(cfg), C.c_int64(n)))


def gumbel(key, stream, n):
    lib().orc_gumbel.restype = C.c_float
    return float(lib().orc_gumbel(C.c_uint64(key), C.c_uint32(stream), C.c_uint64(n)))


def td3d_actor_forward(cfg, actor, s, gumbels, tau):
    actor, s, gumbels = _f32(actor), _f32(s).reshape(-1, cfg.state_dim), _f32(gumbels).reshape(-1, cfg.action_dim)
    out = np.zeros((s.shape[0], cfg.action_dim), np.float32)
    rc = lib().orc_td3d_actor_forward(C.byref(cfg), _p(actor, C.c_float), _p(s, C.c_float), _p(gumbels, C.c_float), C.c_float(tau),
                                      C.c_int64(s.shape[0]), _p(out, C.c_float))
    assert rc == 0
    return out


def td3d_learn(cfg, params, targets, m, v, pows, total_it, rows, policy_noise, gumbel_target, gumbel_actor):
    params, targets, m, v = [_f32(t).copy() for t in (params, targets, m, v)]
    rows, policy_noise, gumbel_target, gumbel_actor = _f32(rows), _f32(policy_noise), _f32(gumbel_target), _f32(gumbel_actor)
    pw = (C.c_double * 4)(*pows)
    rc = lib().orc_td3d_learn(C.byref(cfg), _p(params, C.c_float), _p(targets, C.c_float), _p(m, C.c_float), _p(v, C.c_float), pw,
                              C.c_int64(total_it), _p(rows, C.c_float), C.c_int64(rows.shape[1]), _p(policy_noise, C.c_float),
                              _p(gumbel_target, C.c_float), _p(gumbel_actor, C.c_float))
    assert rc == 0
    return params, targets, m, v, list(pw)


def make_td3d_tapes(A, **t):
    """Keyword arrays named as in TD3D_TAPE_KEYS (missing ones: empty)."""
    def arr(name, dtype, cols):
        a = np.ascontiguousarray(t.get(name, np.zeros((0, cols) if cols else 0)), dtype)
        return a.reshape(-1, cols) if cols else a.reshape(-1)
    keep = dict(rand_action=arr("rand_action", np.int32, 0), act_noise=arr("act_noise", np.float32, A), test_noise=arr("test_noise", np.float32, A),
                policy_noise=arr("policy_noise", np.float32, A), gumbel_act=arr("gumbel_act", np.float32, A),
                gumbel_test=arr("gumbel_test", np.float32, A), gumbel_target=arr("gumbel_target", np.float32, A),
                gumbel_actor=arr("gumbel_actor", np.float32, A), replay_idx=arr("replay_idx", np.int32, 0),
                train_reset=arr("train_reset", np.float64, 4), test_reset=arr("test_reset", np.float64, 4))
    ct = dict(rand_action=C.c_int32, replay_idx=C.c_int32, train_reset=C.c_double, test_reset=C.c_double)
    args = []
    for k in TD3D_TAPE_KEYS:
        a = keep[k]
        args += [_p(a, ct.get(k, C.c_float)), a.shape[0]]
    tp = Td3dTapes(*args)
    tp._keep = keep
    return tp


def td3d_chain(cfg, se_params, agent_init, rng_key=0, tapes=None, trace_cap=0):
    se_params, agent_init = _f32(se_params), _f32(agent_init)
    E, T, S, A = cfg.train_episodes, cfg.test_episodes, cfg.state_dim, cfg.action_dim
    P = td3d_num_params(cfg)[0]
    assert agent_init.size == P, (agent_init.size, P)
    ep_mean = np.full(max(E, 1), np.nan)
    ep_len = np.zeros(max(E, 1), np.int32)
    final = np.zeros(max(T, 1))
    final_params = np.zeros(P, np.float32)
    res = ChainResult()
    tr, arrs = None, None
    if trace_cap:
        arrs = dict(action=np.zeros((trace_cap, A), np.float32), state=np.zeros((trace_cap, S), np.float32),
                    next_state=np.zeros((trace_cap, S), np.float32), reward=np.zeros(trace_cap, np.float32))
        tr = Td3Trace(trace_cap, 0, _p(arrs["action"], C.c_float), _p(arrs["state"], C.c_float), _p(arrs["next_state"], C.c_float),
                      _p(arrs["reward"], C.c_float))
    rc = lib().orc_td3d_chain(C.byref(cfg), _p(se_params, C.c_float), _p(agent_init, C.c_float), C.c_uint64(rng_key),
                              C.byref(tapes) if tapes is not None else None, _p(ep_mean, C.c_double), _p(ep_len, C.c_int32),
                              _p(final, C.c_double), C.byref(tr) if tr is not None else None, C.byref(res), _p(final_params, C.c_float))
    out = dict(rc=rc, score=res.score, episodes_run=res.episodes_run, train_steps=res.train_steps, learn_steps=res.learn_steps,
               test_steps=res.test_steps, episode_test_mean=ep_mean[:E], episode_len=ep_len[:E], final_test_returns=final[:T],
               final_params=final_params)
    if tr is not None:
        out["trace"] = {k: v[:tr.n] for k, v in arrs.items()}
    return out


def rn_num_params(rtype, S, info_dim, hidden, layers):
    lib().orc_rn_num_params.restype = C.c_int64
    return int(lib().orc_rn_num_params(int(rtype), int(S), int(info_dim), int(hidden), int(layers)))


def rn_shape_rows(rtype, S, info_dim, hidden, layers, act, prelu, gamma, rn_params, s, s2, info, r, use_layer_norm=False):
    """RewardEnv._calc_reward (reward_env.py:68-133) for rows of a vector-state env; all 11 reward types."""
    s, s2, r = _f32(s).reshape(-1, S), _f32(s2).reshape(-1, S), _f32(r).reshape(-1)
    n = s.shape[0]
    info = _f32(info).reshape(n, info_dim) if info is not None and info_dim > 0 else None
    rn_params = _f32(rn_params if rn_params is not None and len(rn_params) else np.zeros(1, np.float32))
    out = np.zeros(n, np.float32)
    rc = lib().orc_rn_shape_rows(int(rtype), int(S), int(info_dim), int(hidden), int(layers), 1 if use_layer_norm else 0,
                                 int(ACT[act]) if isinstance(act, str) else int(act),
                                 C.c_float(prelu), C.c_double(gamma), _p(rn_params, C.c_float), _p(s, C.c_float), _p(s2, C.c_float),
                                 _p(info, C.c_float) if info is not None else None, _p(r, C.c_float), C.c_int64(n), _p(out, C.c_float))
    if rc != 0:
        raise ValueError("orc_rn_shape_rows rc=%d" % rc)
    return out


def worker_best_multi(score_add, score_sub, mirrored=True, grad_eval_type="mean"):
    """GTN_Worker.calc_best_score for num_grad_evals = G: score_add / score_sub [pop,G] -> (score_best [pop], sign [pop])."""
    a, b = np.ascontiguousarray(score_add, np.float64), np.ascontiguousarray(score_sub, np.float64)
    pop, G = a.shape
    best, sign = np.zeros(pop), np.zeros(pop, np.float32)
    rc = lib().orc_worker_best_multi(_p(a, C.c_double), _p(b, C.c_double), C.c_int64(pop), C.c_int(G), C.c_int(1 if mirrored else 0),
                                     C.c_int({"mean": 0, "minmax": 1}[grad_eval_type]), _p(best, C.c_double), _p(sign, C.c_float))
    if rc != 0:
        raise ValueError("orc_worker_best_multi rc=%d" % rc)
    return best, sign


# ------------------------------------------------------------------------------------------------
# *_vary agents (agents/DDQN_vary.py:26-59, DuelingDDQN_vary.py:24-69, TD3_vary.py:24-58): hyper-parameter draws
# ------------------------------------------------------------------------------------------------
STREAM_AGENT_INIT, STREAM_VARY_HP = 10, 11


def rng_unit(key, stream, index):
    L = lib()
    L.orc_rng_u64.restype = C.c_uint64
    L.orc_rng_u64.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
    return float(L.orc_rng_u64(int(key), int(stream), int(index)) >> 11) * (1.0 / 9007199254740992.0)


def vary_hyperparameters(section, units):
    """ConfigSpace 0.4.13's draw of the four varied hyper-parameters from four uniforms given in name order
    (batch_size, hidden_layer, hidden_size, lr), restated with numpy like the library computes it."""
    def flt(u, lo, hi, log):
        a, b = (np.log(lo), np.log(hi)) if log else (lo, hi)
        v = u * (b - a) + a
        v = np.exp(v) if log else v
        return float(np.clip(v, lo, hi))

    def integer(u, lo, hi, log):
        return int(np.clip(np.rint(flt(u, lo - 0.49999, hi + 0.49999, log)), lo, hi))

    lr, b, h, l = section["lr"], section["batch_size"], section["hidden_size"], section["hidden_layer"]
    ub, ul, uh, ulr = units
    return {"lr": flt(ulr, lr / 3, lr * 3, True), "batch_size": integer(ub, int(b / 3), int(b * 3), True),
            "hidden_size": integer(uh, int(h / 3), int(h * 3), True), "hidden_layer": integer(ul, l - 1, l + 1, False)}


def hp_overrides(hp):
    """ddqn_cfg_from_config overrides for one drawn set of hyper-parameters."""
    return dict(lr=float(hp["lr"]), batch_size=int(hp["batch_size"]), q_hidden=int(hp["hidden_size"]),
                q_layers=max(1, int(hp["hidden_layer"])))


def vary_chain_hp(section, key):
    return vary_hyperparameters(section, [rng_unit(key, STREAM_VARY_HP, i) for i in range(4)])


def rng_u64_array(key, stream, n):
    """orc_rng_u64(key, stream, i) for i < n, vectorised (uint64 wrap-around arithmetic)."""
    def mix(x):
        x = x ^ (x >> np.uint64(30)); x = x * np.uint64(0xbf58476d1ce4e5b9)
        x = x ^ (x >> np.uint64(27)); x = x * np.uint64(0x94d049bb133111eb)
        return x ^ (x >> np.uint64(31))
    with np.errstate(over="ignore"):
        k = np.uint64(key)
        idx = np.arange(n, dtype=np.uint64)
        x = k + np.uint64(0x9e3779b97f4a7c15) * ((np.uint64(stream) << np.uint64(56)) ^ idx)
        return mix(mix(x) ^ k)


STREAM_ICM_INIT = 12


def icm_layer_dims(cfg):
    """ICMModel's nn.Linear layers in state-dict order (models/icm_baseline.py:42-78); a TD3 cfg = continuous actions."""
    S, F, H = cfg.state_dim, cfg.icm_feature_dim, cfg.icm_hidden
    A = cfg.num_actions if hasattr(cfg, "num_actions") else cfg.action_dim
    Ai = 1 if (A == 2 and hasattr(cfg, "num_actions")) else A
    C = F + Ai
    return ([(S, H), (H, H), (H, F)] + [(2 * F, H), (H, H), (H, Ai)] + [(C, H), (H, H), (H, F)] + [(C, F), (C, F)] * 4
            + [(F, H), (H, F)])


def agent_init_from_key(key, layer_dims, stream=STREAM_AGENT_INIT):
    """Fresh agent of an MLP stack given as [(fan_in, fan_out), ...] in state-dict order: nn.Linear's default init
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)), u = unit(rng(key, stream, i)) as in orc_nes_draw (stream 12: ICM modules)."""
    bounds = np.concatenate([np.full(fi * fo + fo, 1.0 / np.sqrt(float(fi)), np.float32) for fi, fo in layer_dims])
    u = ((rng_u64_array(key, stream, bounds.size) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)).astype(np.float32)
    return (u * np.float32(2.0) - np.float32(1.0)) * bounds
