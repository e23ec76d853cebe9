"""RewardEnv: a real env whose reward is shaped/replaced by a learned reward network (reference
envs/reward_env.py:7-149), MI355X edition.

Same constructor kwargs, attributes and state-dict keys (`reward_net.0.weight`, `reward_net.1.weight` (PReLU slope),
`reward_net.2.weight`, ...).  For discrete-state real envs (gridworlds) the network only ever sees one-hot states, so
`step` reads the shaped reward of (state, action) from a table that `lenv_rn_shape_population` evaluates on the device
whenever the parameters change; the info-vector reward types (3,4,7,8,101,102) raise ValueError there exactly like the
reference, because gridworlds return an empty info dict (reward_env.py:95-96).  For vector-state real envs (the
HalfCheetah stand-in) all 11 types go through `lenv_rn_shape_rows`."""
import numpy as np
import torch
import torch.nn as nn

from .. import _lib, engine
from ..models.model_utils import FlatParams, build_nn_from_config, linear_params
from .grid_env import GridEnv


class RewardEnv(nn.Module):
    def __init__(self, real_env, kwargs):
        super().__init__()
        self.env_name = str(kwargs["env_name"])
        self.device = str(kwargs["device"])
        self.state_dim = int(kwargs["state_dim"])
        self.action_dim = int(kwargs["action_dim"])
        self.info_dim = int(kwargs["info_dim"])
        self.solved_reward = float(kwargs["solved_reward"])
        self.reward_env_type = int(kwargs["reward_env_type"])
        self._max_episode_steps = int(kwargs["max_steps"])
        self.action_space = kwargs["action_space"]
        self.observation_space = kwargs["observation_space"]
        self.activation_fn = kwargs["activation_fn"]
        self.hidden_size = int(kwargs["hidden_size"])
        self.hidden_layer = int(kwargs["hidden_layer"])
        self.real_env = real_env
        self.reward_net = self.build_reward_net(kwargs)
        self.gamma = None
        self._flat = None
        self._table = None
        self._table_key = None
        self.state = self.reset()

    def build_reward_net(self, kwargs):
        t = self.reward_env_type
        if t < 100:
            if t == 0:
                input_dim = 1
            elif t in (1, 2, 5, 6):
                input_dim = self.state_dim
            elif t in (3, 4, 7, 8):
                input_dim = self.state_dim + self.info_dim
            else:
                raise NotImplementedError('Unknown reward_env_type: ' + str(t))
            return build_nn_from_config(input_dim=input_dim, output_dim=1, nn_config=kwargs)
        if t in (101, 102):
            return nn.Linear(self.info_dim, 1, bias=False)
        raise NotImplementedError('Unknown reward_env_type: ' + str(t))

    # ---- flat parameter buffer theta (nn.Linear params of reward_net) on the HIP device ----
    def flat_params(self):
        if self._flat is None:
            dev = engine.require_device()
            self.to(dev)
            self._flat = FlatParams(self, dev)
        return self._flat.flat

    def _apply(self, fn, *a, **k):
        self._flat = None
        self._table_key = None
        return super()._apply(fn, *a, **k)

    def params_changed(self):
        """Called by whoever writes the flat parameter buffer behind torch's back (GTN_Master after lenv_nes_rank_update,
        in-place ops on flat_params()): such writes bump no Parameter._version, so the shaped-reward table is dropped."""
        self._table_key = None

    def ql_cfg(self):
        if not isinstance(self.real_env, GridEnv):
            raise NotImplementedError("RewardEnv over a continuous-state real env: next row of the scope table")
        t = self.real_env.tables
        return _lib.QlCfg(n_states=t["n_states"], n_actions=t["n_actions"], start_state=t["start_state"],
                          max_steps=self._max_episode_steps, rn_hidden=self.hidden_size, rn_layers=self.hidden_layer,
                          rn_act=_lib.ACT[self.activation_fn], rn_prelu=0.25, reward_env_type=self.reward_env_type,
                          train_episodes=0, test_episodes=1, init_episodes=0, early_out_num=1, batch_size=1, rng_mode=0,
                          solved_reward=self.solved_reward, alpha=1.0, gamma=float(self.gamma if self.gamma is not None else 0.0),
                          eps_init=0.0, eps_min=0.0, eps_decay=0.0)

    def shaped_table(self):
        """[n_states, n_actions] shaped reward of every transition under the current parameters (device-evaluated)."""
        key = (tuple(p._version for p in linear_params(self)), self.gamma)
        if self._table is None or key != self._table_key:
            t = self.real_env.tables
            dev = engine.require_device()
            nxt = torch.from_numpy(t["next_state"]).contiguous().to(dev)
            rew = torch.from_numpy(t["reward"]).contiguous().to(dev)
            _, shaped = engine.rn_shape_population(self.ql_cfg(), self.flat_params(), None, None, None, nxt, rew, 1)
            self._table = shaped[0].cpu()
            self._table_key = key
        return self._table

    def step(self, action):
        state = self.state
        next_state, reward, done, info = self.real_env.step(action)
        reward_res = self._calc_reward(state=state, next_state=next_state, reward=reward, info=info, action=action)
        self.state = next_state
        return next_state, reward_res, done, {}

    def _calc_reward(self, state, next_state, reward, info, action=None):
        """reference envs/reward_env.py:68-133."""
        if 'TimeLimit.truncated' in info:
            info.pop('TimeLimit.truncated')
        t = self.reward_env_type
        if t in (3, 4, 7, 8, 101, 102) and not info:
            raise ValueError('No info dict provided by environment')
        if isinstance(self.real_env, GridEnv):
            return self.shaped_table()[int(state), int(action)].item()
        from ..models.model_utils import mlp_desc
        dev = engine.require_device()
        r32 = torch.tensor([reward], dtype=torch.float32, device=dev)
        if t == 0:
            return r32.item()
        s = torch.from_numpy(np.asarray(state, np.float32).reshape(1, -1)).to(dev)
        s2 = torch.from_numpy(np.asarray(next_state, np.float32).reshape(1, -1)).to(dev)
        inf = torch.tensor([list(info.values())], dtype=torch.float32, device=dev) if info else None
        desc = mlp_desc(self.reward_net, self.activation_fn) if t < 100 else None
        theta = self.flat_params()
        if desc is not None and desc.use_layer_norm:          # the LayerNorm block is not part of theta: lenv_mlp_desc layout for the one-step entry
            from ..models.model_utils import mlp_params
            theta = torch.cat([p.detach().reshape(-1).to(theta.device, torch.float32) for p in mlp_params(self.reward_net)])
        out = engine.rn_shape_rows(t, desc, self.state_dim, self.info_dim if inf is not None else 0, self.gamma, theta, s, s2, inf, r32)
        return out.item()

    def seed(self, seed):
        return self.real_env.seed(seed)

    def render(self):
        return self.real_env.render()

    def reset(self):
        self.state = self.real_env.reset()
        return self.state

    def close(self):
        return self.real_env.close()

    def set_agent_params(self, gamma):
        self.gamma = gamma
