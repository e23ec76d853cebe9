"""gym.make + classic-control envs restated from gym 0.17.3.
Float64 python/numpy arithmetic in the published operation order."""
import math
import numpy as np
from numpy import sin, cos, pi

from gym.core import Env
from gym import spaces
from gym.utils import seeding


class EnvSpec(object):
    def __init__(self, id, max_episode_steps=None, reward_threshold=None):
        self.id = id
        self.max_episode_steps = max_episode_steps
        self.reward_threshold = reward_threshold


class CartPoleEnv(Env):
    """gym/envs/classic_control/cartpole.py @0.17.3"""

    def __init__(self):
        self.gravity = 9.8
        self.masscart = 1.0
        self.masspole = 0.1
        self.total_mass = (self.masspole + self.masscart)
        self.length = 0.5  # actually half the pole's length
        self.polemass_length = (self.masspole * self.length)
        self.force_mag = 10.0
        self.tau = 0.02  # seconds between state updates
        self.kinematics_integrator = 'euler'
        self.theta_threshold_radians = 12 * 2 * math.pi / 360
        self.x_threshold = 2.4
        high = np.array([self.x_threshold * 2, np.finfo(np.float32).max,
                         self.theta_threshold_radians * 2, np.finfo(np.float32).max], dtype=np.float32)
        self.action_space = spaces.Discrete(2)
        self.observation_space = spaces.Box(-high, high, dtype=np.float32)
        self.seed()
        self.viewer = None
        self.state = None
        self.steps_beyond_done = None

    def seed(self, seed=None):
        self.np_random, seed = seeding.np_random(seed)
        return [seed]

    def step(self, action):
        x, x_dot, theta, theta_dot = self.state
        force = self.force_mag if action == 1 else -self.force_mag
        costheta = math.cos(theta)
        sintheta = math.sin(theta)
        temp = (force + self.polemass_length * theta_dot ** 2 * sintheta) / self.total_mass
        thetaacc = (self.gravity * sintheta - costheta * temp) / \
                   (self.length * (4.0 / 3.0 - self.masspole * costheta ** 2 / self.total_mass))
        xacc = temp - self.polemass_length * thetaacc * costheta / self.total_mass
        x = x + self.tau * x_dot
        x_dot = x_dot + self.tau * xacc
        theta = theta + self.tau * theta_dot
        theta_dot = theta_dot + self.tau * thetaacc
        self.state = (x, x_dot, theta, theta_dot)
        done = bool(x < -self.x_threshold or x > self.x_threshold
                    or theta < -self.theta_threshold_radians or theta > self.theta_threshold_radians)
        if not done:
            reward = 1.0
        elif self.steps_beyond_done is None:
            self.steps_beyond_done = 0
            reward = 1.0
        else:
            self.steps_beyond_done += 1
            reward = 0.0
        return np.array(self.state), reward, done, {}

    def reset(self):
        self.state = self.np_random.uniform(low=-0.05, high=0.05, size=(4,))
        self.steps_beyond_done = None
        return np.array(self.state)

    def render(self, mode='human'):
        return None


def _wrap(x, m, M):
    diff = M - m
    while x > M:
        x = x - diff
    while x < m:
        x = x + diff
    return x


def _bound(x, m, M):
    return min(max(x, m), M)


def _rk4(derivs, y0, t):
    Ny = len(y0)
    yout = np.zeros((len(t), Ny), np.float64)
    yout[0] = y0
    for i in np.arange(len(t) - 1):
        thist = t[i]
        dt = t[i + 1] - thist
        dt2 = dt / 2.0
        y0 = yout[i]
        k1 = np.asarray(derivs(y0, thist))
        k2 = np.asarray(derivs(y0 + dt2 * k1, thist + dt2))
        k3 = np.asarray(derivs(y0 + dt2 * k2, thist + dt2))
        k4 = np.asarray(derivs(y0 + dt * k3, thist + dt))
        yout[i + 1] = y0 + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
    return yout


class AcrobotEnv(Env):
    """gym/envs/classic_control/acrobot.py @0.17.3 ("book" dynamics)"""
    dt = .2
    LINK_LENGTH_1 = 1.
    LINK_LENGTH_2 = 1.
    LINK_MASS_1 = 1.
    LINK_MASS_2 = 1.
    LINK_COM_POS_1 = 0.5
    LINK_COM_POS_2 = 0.5
    LINK_MOI = 1.
    MAX_VEL_1 = 4 * pi
    MAX_VEL_2 = 9 * pi
    AVAIL_TORQUE = [-1., 0., +1]
    torque_noise_max = 0.
    book_or_nips = "book"

    def __init__(self):
        self.viewer = None
        high = np.array([1.0, 1.0, 1.0, 1.0, self.MAX_VEL_1, self.MAX_VEL_2], dtype=np.float32)
        self.observation_space = spaces.Box(low=-high, high=high, dtype=np.float32)
        self.action_space = spaces.Discrete(3)
        self.state = None
        self.seed()

    def seed(self, seed=None):
        self.np_random, seed = seeding.np_random(seed)
        return [seed]

    def reset(self):
        self.state = self.np_random.uniform(low=-0.1, high=0.1, size=(4,))
        return self._get_ob()

    def step(self, a):
        s = self.state
        torque = self.AVAIL_TORQUE[a]
        s_augmented = np.append(s, torque)
        ns = _rk4(self._dsdt, s_augmented, [0, self.dt])
        ns = ns[-1]
        ns = ns[:4]
        ns[0] = _wrap(ns[0], -pi, pi)
        ns[1] = _wrap(ns[1], -pi, pi)
        ns[2] = _bound(ns[2], -self.MAX_VEL_1, self.MAX_VEL_1)
        ns[3] = _bound(ns[3], -self.MAX_VEL_2, self.MAX_VEL_2)
        self.state = ns
        terminal = self._terminal()
        reward = -1. if not terminal else 0.
        return (self._get_ob(), reward, terminal, {})

    def _get_ob(self):
        s = self.state
        return np.array([cos(s[0]), sin(s[0]), cos(s[1]), sin(s[1]), s[2], s[3]])

    def _terminal(self):
        s = self.state
        return bool(-cos(s[0]) - cos(s[1] + s[0]) > 1.)

    def _dsdt(self, s_augmented, t):
        m1 = self.LINK_MASS_1
        m2 = self.LINK_MASS_2
        l1 = self.LINK_LENGTH_1
        lc1 = self.LINK_COM_POS_1
        lc2 = self.LINK_COM_POS_2
        I1 = self.LINK_MOI
        I2 = self.LINK_MOI
        g = 9.8
        a = s_augmented[-1]
        s = s_augmented[:-1]
        theta1 = s[0]
        theta2 = s[1]
        dtheta1 = s[2]
        dtheta2 = s[3]
        d1 = m1 * lc1 ** 2 + m2 * (l1 ** 2 + lc2 ** 2 + 2 * l1 * lc2 * cos(theta2)) + I1 + I2
        d2 = m2 * (lc2 ** 2 + l1 * lc2 * cos(theta2)) + I2
        phi2 = m2 * lc2 * g * cos(theta1 + theta2 - pi / 2.)
        phi1 = - m2 * l1 * lc2 * dtheta2 ** 2 * sin(theta2) \
               - 2 * m2 * l1 * lc2 * dtheta2 * dtheta1 * sin(theta2) \
               + (m1 * lc1 + m2 * l1) * g * cos(theta1 - pi / 2) + phi2
        ddtheta2 = (a + d2 / d1 * phi1 - m2 * l1 * lc2 * dtheta1 ** 2 * sin(theta2) - phi2) \
                   / (m2 * lc2 ** 2 + I2 - d2 ** 2 / d1)
        ddtheta1 = -(d2 * ddtheta2 + phi1) / d1
        return (dtheta1, dtheta2, ddtheta1, ddtheta2, 0.)

    def render(self, mode='human'):
        return None


class MountainCarEnv(Env):
    """gym/envs/classic_control/mountain_car.py @0.17.3 (restated from memory like the other classic-control envs)"""

    def __init__(self, goal_velocity=0):
        self.min_position = -1.2
        self.max_position = 0.6
        self.max_speed = 0.07
        self.goal_position = 0.5
        self.goal_velocity = goal_velocity
        self.force = 0.001
        self.gravity = 0.0025
        self.low = np.array([self.min_position, -self.max_speed], dtype=np.float32)
        self.high = np.array([self.max_position, self.max_speed], dtype=np.float32)
        self.viewer = None
        self.action_space = spaces.Discrete(3)
        self.observation_space = spaces.Box(self.low, self.high, dtype=np.float32)
        self.seed()

    def seed(self, seed=None):
        self.np_random, seed = seeding.np_random(seed)
        return [seed]

    def step(self, action):
        position, velocity = self.state
        velocity += (action - 1) * self.force + math.cos(3 * position) * (-self.gravity)
        velocity = np.clip(velocity, -self.max_speed, self.max_speed)
        position += velocity
        position = np.clip(position, self.min_position, self.max_position)
        if (position == self.min_position and velocity < 0):
            velocity = 0
        done = bool(position >= self.goal_position and velocity >= self.goal_velocity)
        reward = -1.0
        self.state = (position, velocity)
        return np.array(self.state), reward, done, {}

    def reset(self):
        self.state = np.array([self.np_random.uniform(low=-0.6, high=-0.4), 0])
        return np.array(self.state)

    def render(self, mode='human'):
        return None


class CheetahStandinEnv(Env):
    """HalfCheetah-v3 STAND-IN (tools/gen_cheetah_standin.py): MuJoCo is unavailable, so config 5 runs -- on the reference side
    too -- on this fixed 17-obs / 6-action saturated linear system.  Plain python float arithmetic, left to right."""

    def __init__(self):
        import json
        import os
        d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cheetah_standin.json")))
        self.A = [float.fromhex(v) for v in d["A"]]
        self.B = [float.fromhex(v) for v in d["B"]]
        self.c = [float.fromhex(v) for v in d["c"]]
        high = np.full(17, np.inf, dtype=np.float32)
        self.observation_space = spaces.Box(-high, high, dtype=np.float32)
        self.action_space = spaces.Box(low=-1.0, high=1.0, shape=(6,), dtype=np.float32)
        self.state = None
        self.seed()

    def seed(self, seed=None):
        self.np_random, seed = seeding.np_random(seed)
        return [seed]

    def reset(self):
        self.state = self.np_random.uniform(low=-0.1, high=0.1, size=(17,))
        return np.array(self.state)

    def step(self, action):
        x = [float(v) for v in self.state]
        a = [float(v) for v in np.asarray(action).reshape(-1)]
        nx = []
        for i in range(17):
            acc = self.c[i]
            for j in range(17):
                acc = acc + self.A[i * 17 + j] * x[j]
            for k in range(6):
                acc = acc + self.B[i * 6 + k] * a[k]
            nx.append(min(max(acc, -10.0), 10.0))
        ctrl = 0.0
        for k in range(6):
            ctrl = ctrl + a[k] * a[k]
        reward = nx[8] - 0.1 * ctrl
        self.state = np.array(nx)
        info = {"x_position": nx[0], "x_velocity": nx[8], "reward_run": nx[8], "reward_ctrl": -0.1 * ctrl}
        return np.array(nx), reward, False, info

    def render(self, mode='human'):
        return None


class PendulumEnv(Env):
    """gym==0.17.3 classic_control/pendulum.py (Pendulum-v0), restated; third party, not under /root/reference.  The torque
    arrives as a float32 array (EnvWrapper.step: action.cpu().detach().numpy()); with the numpy of the reference's era
    (value-based casting) `u ** 2` is a float32 square and every other product promotes u to float64 -- written out
    explicitly here so the arithmetic does not depend on the installed numpy."""

    def __init__(self, g=10.0):
        self.max_speed = 8
        self.max_torque = 2.
        self.dt = .05
        self.g = g
        self.m = 1.
        self.l = 1.
        high = np.array([1., 1., self.max_speed], dtype=np.float32)
        self.action_space = spaces.Box(low=-self.max_torque, high=self.max_torque, shape=(1,), dtype=np.float32)
        self.observation_space = spaces.Box(low=-high, high=high, dtype=np.float32)
        self.state = None
        self.seed()

    def seed(self, seed=None):
        self.np_random, seed = seeding.np_random(seed)
        return [seed]

    def step(self, u):
        th, thdot = float(self.state[0]), float(self.state[1])
        g, m, l, dt = self.g, self.m, self.l, self.dt
        u32 = np.float32(np.clip(np.asarray(u, np.float32), np.float32(-self.max_torque), np.float32(self.max_torque)).reshape(-1)[0])
        ud = float(u32)
        self.last_u = ud
        usq = float(np.float32(u32 * u32))
        costs = _angle_normalize(th) ** 2 + .1 * thdot ** 2 + .001 * usq
        newthdot = thdot + (-3 * g / (2 * l) * math.sin(th + math.pi) + 3. / (m * l ** 2) * ud) * dt
        newth = th + newthdot * dt
        newthdot = min(max(newthdot, -float(self.max_speed)), float(self.max_speed))
        self.state = np.array([newth, newthdot])
        return self._get_obs(), -costs, False, {}

    def reset(self):
        high = np.array([np.pi, 1])
        self.state = self.np_random.uniform(low=-high, high=high)
        self.last_u = None
        return self._get_obs()

    def _get_obs(self):
        theta, thetadot = float(self.state[0]), float(self.state[1])
        return np.array([math.cos(theta), math.sin(theta), thetadot])

    def render(self, mode='human'):
        return None


class Continuous_MountainCarEnv(Env):
    """gym==0.17.3 classic_control/continuous_mountain_car.py (MountainCarContinuous-v0), restated; third party, not under
    /root/reference.  The action arrives as a float32 array; with the numpy of the reference's era `force * power` promotes the
    float32 force to float64 -- written out explicitly here so the arithmetic does not depend on the installed numpy."""

    def __init__(self, goal_velocity=0):
        self.min_action = -1.0
        self.max_action = 1.0
        self.min_position = -1.2
        self.max_position = 0.6
        self.max_speed = 0.07
        self.goal_position = 0.45
        self.goal_velocity = goal_velocity
        self.power = 0.0015
        self.low_state = np.array([self.min_position, -self.max_speed], dtype=np.float32)
        self.high_state = np.array([self.max_position, self.max_speed], dtype=np.float32)
        self.action_space = spaces.Box(low=self.min_action, high=self.max_action, shape=(1,), dtype=np.float32)
        self.observation_space = spaces.Box(low=self.low_state, high=self.high_state, dtype=np.float32)
        self.state = None
        self.seed()

    def seed(self, seed=None):
        self.np_random, seed = seeding.np_random(seed)
        return [seed]

    def step(self, action):
        position = float(self.state[0])
        velocity = float(self.state[1])
        a0 = float(np.asarray(action, np.float32).reshape(-1)[0])
        force = min(max(a0, self.min_action), self.max_action)
        velocity += force * self.power - 0.0025 * math.cos(3 * position)
        if (velocity > self.max_speed): velocity = self.max_speed
        if (velocity < -self.max_speed): velocity = -self.max_speed
        position += velocity
        if (position > self.max_position): position = self.max_position
        if (position < self.min_position): position = self.min_position
        if (position == self.min_position and velocity < 0): velocity = 0
        done = bool(position >= self.goal_position and velocity >= self.goal_velocity)
        reward = 0
        if done:
            reward = 100.0
        reward -= math.pow(a0, 2) * 0.1
        self.state = np.array([position, velocity])
        return self.state, reward, done, {}

    def reset(self):
        self.state = np.array([self.np_random.uniform(low=-0.6, high=-0.4), 0])
        return np.array(self.state)

    def render(self, mode='human'):
        return None


def _angle_normalize(x):
    return (((x + math.pi) % (2 * math.pi)) - math.pi)


_REGISTRY = {
    'CartPole-v0': (CartPoleEnv, EnvSpec('CartPole-v0', 200, 195.0)),
    'CartPole-v1': (CartPoleEnv, EnvSpec('CartPole-v1', 500, 475.0)),
    'Acrobot-v1': (AcrobotEnv, EnvSpec('Acrobot-v1', 500, -100.0)),
    'MountainCar-v0': (MountainCarEnv, EnvSpec('MountainCar-v0', 200, -110.0)),
    'HalfCheetah-v3': (CheetahStandinEnv, EnvSpec('HalfCheetah-v3', 1000, 4800.0)),
    'Pendulum-v0': (PendulumEnv, EnvSpec('Pendulum-v0', 200, None)),
    'MountainCarContinuous-v0': (Continuous_MountainCarEnv, EnvSpec('MountainCarContinuous-v0', 999, 90.0)),
}


def register(id, entry_point, max_episode_steps=None, reward_threshold=None):
    _REGISTRY[id] = (entry_point, EnvSpec(id, max_episode_steps, reward_threshold))


def make(id, **kwargs):
    from gym.wrappers import TimeLimit
    if id not in _REGISTRY:
        raise KeyError("gym shim: unknown env id " + str(id))
    cls, spec = _REGISTRY[id]
    env = cls(**kwargs)
    env.spec = spec
    if spec.max_episode_steps is not None:
        env = TimeLimit(env, max_episode_steps=spec.max_episode_steps)
    return env
