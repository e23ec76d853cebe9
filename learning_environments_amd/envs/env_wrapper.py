"""EnvWrapper: the uniform tensor API over real / virtual envs (reference envs/env_wrapper.py:9-167).

Same methods, argument meaning and return conventions: `step(action, state=None)` returns the 3-tuple
`(next_state, reward, done)` of CPU fp32 tensors.  New (batched fast path, not in the reference):
`step_population` evaluates many NES perturbations of a virtual env in one kernel launch.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import engine
from ..utils import from_one_hot_encoding, to_one_hot_encoding
from .spaces import Discrete
from .virtual_env import VirtualEnv


class EnvWrapper(nn.Module):
    def __init__(self, env):
        super().__init__()
        self.env = env
        self.same_action_num = 1

    def step(self, action, state=None):
        if self.is_virtual_env():
            reward_sum = None
            if self.has_discrete_action_space():
                action = to_one_hot_encoding(action, self.get_action_dim())
            for _ in range(self.same_action_num):
                state, reward, done = self.env.step(action=action, state=state)
                reward_sum = reward if reward_sum is None else reward_sum + reward
            reward = reward_sum.to("cpu")
            next_state = state.to("cpu")
            done = done.to("cpu")
            if self.has_discrete_state_space():
                next_state = from_one_hot_encoding(next_state)
            return next_state, reward, done
        return self._step_real(action)

    def _step_real(self, action):
        """Real / reward env (reference :48-70): numpy action (an int for discrete spaces), the action repeated `same_action_num`
        times with summed rewards until the env reports done; 0-dim reward / done tensors, scalar states become 1-element
        vectors."""
        act = action.detach().cpu().numpy()
        if self.has_discrete_action_space():
            act = int(act.astype(int)[0])
        total, obs, done = 0, None, False
        for _ in range(self.same_action_num):
            obs, reward, done, _ = self.env.step(act)
            total += reward
            if done:
                break
        as_f32 = lambda v: torch.tensor(v, device="cpu", dtype=torch.float32)
        obs_t = as_f32(obs)
        return (obs_t.unsqueeze(0) if obs_t.dim() == 0 else obs_t), as_f32(total), as_f32(done)

    def step_population(self, actions, states, eps=None, worker=None, sign=None):
        """Batched fast path: chain c steps the SE with weights theta + sign[c]*eps[worker[c]].
        actions int32 [chains], states fp32 [chains,S] (device).  Returns device tensors."""
        if not self.is_virtual_env():
            raise NotImplementedError("step_population is defined for virtual envs")
        if not self.has_discrete_action_space():
            raise NotImplementedError("step_population takes action indices: virtual envs over a discrete action space")
        return engine.se_step_population(self.env.descs(), self.env.step_params(), self.env.step_eps(eps), worker, sign, states, actions)

    def reset(self):
        """CPU fp32 state of a fresh episode; a virtual env over a discrete observation space hands back the index."""
        first = self.env.reset()
        if torch.is_tensor(first):
            out = first.cpu()
        elif isinstance(first, np.ndarray):
            out = torch.from_numpy(first).float().cpu()
        else:                                          # a gridworld's integer state
            out = torch.tensor([first], device="cpu", dtype=torch.float32)
        discrete_virtual = self.is_virtual_env() and self.has_discrete_state_space()
        return from_one_hot_encoding(out) if discrete_virtual else out

    def get_random_action(self):
        space = self.env.action_space
        if isinstance(space, Discrete):
            return torch.tensor([int(np.random.randint(space.n))], device="cpu", dtype=torch.float32)
        return torch.from_numpy(np.random.uniform(space.low, space.high).astype(np.float32))

    def get_state_dim(self):
        if self.env.observation_space.shape:
            return self.env.observation_space.shape[0]
        return self.env.observation_space.n

    def get_action_dim(self):
        if self.env.action_space.shape:
            return self.env.action_space.shape[0]
        return self.env.action_space.n

    def get_max_action(self):
        return 2 if self.env.env_name == 'Pendulum-v0' else 1

    def get_min_action(self):
        return 0 if self.env.env_name == 'CartPole-v0' else -self.get_max_action()

    def has_discrete_action_space(self):
        return isinstance(self.env.action_space, Discrete)

    def has_discrete_state_space(self):
        return isinstance(self.env.observation_space, Discrete)

    def render(self):
        return None

    def close(self):
        if not self.is_virtual_env():
            return self.env.close()

    def get_solved_reward(self):
        return self.env.solved_reward

    def max_episode_steps(self):
        return self.env._max_episode_steps

    def can_be_solved(self):
        return self.env.solved_reward < 1e9

    def seed(self, seed):
        """Real envs forward the seed; a virtual env has no RNG of its own to seed (its resets come from `reset_env`)."""
        if self.is_virtual_env():
            print("EnvWrapper.seed: a virtual env is not seeded (its reset states come from the real env)")
            return 0
        return self.env.seed(seed)

    def is_virtual_env(self):
        return isinstance(self.env, VirtualEnv)

    def set_agent_params(self, same_action_num, gamma):
        self.same_action_num = same_action_num
        if hasattr(self.env, "set_agent_params"):
            self.env.set_agent_params(gamma=gamma)
