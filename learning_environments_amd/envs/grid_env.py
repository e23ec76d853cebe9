"""GridEnv: gym.Env look-alike over a compiled grid MDP (reference envs/gridworld.py:24-124 + gym TimeLimit).
Host-side table lookup only (integer bookkeeping, no arithmetic): the hot path steps these MDPs inside
`lenv_ql_rn_inner_loop`; this class serves the one-env-one-step EnvWrapper API."""
from .gridworld import transition_tables
from .spaces import Discrete


class GridEnv(object):
    def __init__(self, env_name):
        self.env_name = env_name
        self.tables = transition_tables(env_name)
        self.action_space = Discrete(self.tables["n_actions"])
        self.observation_space = Discrete(self.tables["n_states"])
        self._max_episode_steps = None
        self._elapsed_steps = 0
        self.state = None

    def seed(self, seed=None):
        return [seed]

    def reset(self):
        self._elapsed_steps = 0
        self.state = int(self.tables["start_state"])
        return self.state

    def step(self, action):
        action = int(action)
        assert self.action_space.contains(action)
        s = self.state
        self.state = int(self.tables["next_state"][s, action])
        reward = self.tables["reward"][s, action].item()
        done = bool(self.tables["done"][s, action])
        info = {}
        self._elapsed_steps += 1
        if self._max_episode_steps is not None and self._elapsed_steps >= self._max_episode_steps:
            info['TimeLimit.truncated'] = not done
            done = True
        self.last_transition = (s, action)
        return self.state, reward, done, info

    def render(self, mode='human'):
        return None

    def close(self):
        return None
