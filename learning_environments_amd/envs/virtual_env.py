"""VirtualEnv: the learned Synthetic Environment (reference envs/virtual_env.py:8-54), MI355X edition.

Same constructor kwargs, attributes and state-dict keys (`state_net.{0,2}.{weight,bias}`, `reward_net.*`,
`done_net.*`) as the reference, so checkpoints and callers are interchangeable.  `step` does not run the
nn.Sequential modules: the three MLPs are evaluated by the hand-written kernel behind
lenv_se_step_population on the flat parameter buffer the modules alias.
"""
import torch
import torch.nn as nn

from .. import engine
from ..models.model_utils import FlatParams, build_nn_from_config, linear_params, mlp_desc, mlp_params
from ..utils import from_one_hot_encoding, to_one_hot_encoding


class VirtualEnv(nn.Module):
    def __init__(self, kwargs):
        super().__init__()
        self.env_name = str(kwargs["env_name"])
        self.device = str(kwargs["device"])
        self.state_dim = int(kwargs["state_dim"])
        self.action_dim = int(kwargs["action_dim"])
        self.solved_reward = float(kwargs["solved_reward"])
        self._max_episode_steps = int(kwargs["max_steps"])
        self.action_space = kwargs["action_space"]
        self.observation_space = kwargs["observation_space"]
        self.reset_env = kwargs["reset_env"]
        self.activation_fn = kwargs["activation_fn"]

        self.state_net = build_nn_from_config(self.state_dim + self.action_dim, self.state_dim, kwargs)
        self.reward_net = build_nn_from_config(self.state_dim + self.action_dim, 1, kwargs)
        self.done_net = build_nn_from_config(self.state_dim + self.action_dim, 1, kwargs)
        self._flat = None
        self.state = None

    # ---- flat parameter buffer theta = state_net | reward_net | done_net on the HIP device ----
    def flat_params(self):
        if self._flat is None:
            dev = engine.require_device()
            self.to(dev)
            self._flat = FlatParams(self, dev)
        return self._flat.flat

    def step_params(self):
        """What lenv_se_step_population reads: theta itself, or -- `use_layer_norm` nets -- a copy with each net's LayerNorm weight | bias
        behind its second Linear (lenv_mlp_desc layout; the module is not part of theta: NES never touches it)."""
        flat = self.flat_params()
        nets = (self.state_net, self.reward_net, self.done_net)
        if not any(isinstance(m, torch.nn.LayerNorm) for net in nets for m in net):
            return flat
        return torch.cat([p.detach().reshape(-1).to(flat.device, torch.float32) for net in nets for p in mlp_params(net)])

    def step_eps(self, eps):
        """eps rows [pop, P_theta] in the layout of step_params(): zero columns where the (never perturbed) LayerNorm blocks sit."""
        nets = (self.state_net, self.reward_net, self.done_net)
        if eps is None or not any(isinstance(m, torch.nn.LayerNorm) for net in nets for m in net):
            return eps
        cols, off = [], 0
        lin = set(id(p) for p in linear_params(self))
        for net in nets:
            for p in mlp_params(net):
                if id(p) in lin:
                    cols.append(torch.arange(off, off + p.numel(), device=eps.device))
                off += p.numel()
        wide = torch.zeros((eps.shape[0], off), dtype=eps.dtype, device=eps.device)
        wide[:, torch.cat(cols)] = eps
        return wide

    def descs(self):
        return (mlp_desc(self.state_net, self.activation_fn), mlp_desc(self.reward_net, self.activation_fn),
                mlp_desc(self.done_net, self.activation_fn))

    def _apply(self, fn, *a, **k):
        self._flat = None          # .to()/.cuda() re-home parameters; re-alias lazily
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict=True, **k):
        if self._flat is None:
            return super().load_state_dict(state_dict, strict=strict, **k)
        own = self.state_dict()
        missing = [key for key in own if key not in state_dict]
        unexpected = [key for key in state_dict if key not in own]
        if strict and (missing or unexpected):
            raise RuntimeError("load_state_dict: missing %s unexpected %s" % (missing, unexpected))
        with torch.no_grad():
            for key, v in state_dict.items():
                if key in own:
                    own[key].copy_(v)      # in place: keeps the flat-buffer aliasing
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def reset(self):
        # reference :35-41 -- real-env reset state as fp32
        self.state = torch.as_tensor(self.reset_env.reset()).clone().float()
        if len(self.state) > self.state_dim:
            self.state = from_one_hot_encoding(self.state)
        elif len(self.state) < self.state_dim:
            self.state = to_one_hot_encoding(self.state, self.state_dim)
        return self.state

    def step(self, action, state=None):
        """action: one-hot [A] (or [n,A]); state: [S] (or [n,S]).  Returns device tensors (next_state, reward, done)
        shaped like the reference's: [S],[1],[1] (or [n,S],[n,1],[n,1])."""
        if self.state is None and state is None:
            self.reset()
        s = self.state if state is None else state
        dev = engine.require_device()
        batched = action.dim() > 1
        s2 = s.reshape(-1, self.state_dim).to(dev, torch.float32).contiguous()
        n = s2.shape[0]
        def forward_path(act_rows):
            # three lenv_mlp_forward launches on cat(action, state) (virtual_env.py:43-54); weights read from HBM: any width / depth
            x = torch.cat([act_rows.reshape(-1, self.action_dim).to(dev, torch.float32), s2], dim=1).contiguous()
            self.flat_params()
            return [engine.mlp_forward(dsc, torch.cat([p.detach().reshape(-1) for p in mlp_params(net)]), x)
                    for net, dsc in zip((self.state_net, self.reward_net, self.done_net), self.descs())]

        if not hasattr(self.action_space, "n"):
            # a continuous action space (HalfCheetah / Pendulum / MountainCarContinuous SEs): the action vector goes in as it comes
            ns, r, d = forward_path(action)
        else:
            a_idx = torch.argmax(action.reshape(-1, self.action_dim), dim=1).to(torch.int32).to(dev)
            try:
                ns, r, d = engine.se_step_population(self.descs(), self.step_params(), None, None, None,
                                                     s2.unsqueeze(0), a_idx.reshape(1, n).contiguous())
                ns, r, d = ns[0], r[0].unsqueeze(-1), d[0].unsqueeze(-1)
            except NotImplementedError:                    # an SE whose parameters exceed the one-launch kernel's LDS
                ns, r, d = forward_path(action)
        if not batched:
            ns, r, d = ns[0], r[0], d[0]
        self.state = ns
        return ns, r, d
