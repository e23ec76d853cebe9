"""Independent cross-check of the restated gym 0.17.3 classic-control physics (the third-party source
is not under /root/reference, so these equations are *unpinned*).  The oracle's closed-form accelerations are compared with
a SECOND derivation that shares no formula with them: the Euler-Lagrange equations written as a mass-matrix system
M(q) qdd = f(q, qd, u), assembled here from the Lagrangian and solved numerically, plus the power balance dE/dt = u . v.
A transcription error in either restatement (a sign, a missing Coriolis term, the 4/3 rod factor) breaks these tests.
The HIP kernels are held bit-exact to the same oracle functions by tests/test_gpu_parity.py."""
import ctypes as C

import numpy as np

from oracle import oracle as orc


def _cartpole_step(st, action):
    L = orc.lib()
    s = (C.c_double * 4)(*st)
    r, d = C.c_double(), C.c_int()
    L.orc_cartpole_step(s, int(action), C.byref(r), C.byref(d))
    return np.array(list(s)), r.value, d.value


def _acrobot_step(st, action):
    L = orc.lib()
    s = (C.c_double * 4)(*st)
    r, d = C.c_double(), C.c_int()
    L.orc_acrobot_step(s, int(action), C.byref(r), C.byref(d))
    return np.array(list(s)), r.value, d.value


def test_cartpole_accelerations_match_lagrangian_mass_matrix():
    """Cart (mass 1) + uniform rod pole (mass 0.1, half-length 0.5) hinged on it, theta from upright:
    L = 1/2 M xd^2 + m l xd thd cos(th) + 2/3 m l^2 thd^2 - m g l cos(th)   =>
    [ M          m l cos th ] [xdd ]   [ F + m l thd^2 sin th ]
    [ m l cos th 4/3 m l^2  ] [thdd] = [ m g l sin th         ]"""
    g, mc, mp, l, F, tau = 9.8, 1.0, 0.1, 0.5, 10.0, 0.02
    M = mc + mp
    rng = np.random.RandomState(0)
    for _ in range(200):
        st = rng.uniform([-2.0, -3.0, -0.2, -3.0], [2.0, 3.0, 0.2, 3.0])
        action = int(rng.randint(2))
        nxt, reward, done = _cartpole_step(st, action)
        x, xd, th, thd = st
        # gym's 'euler' kinematics: positions advance with the OLD velocities, velocities with the accelerations
        assert np.allclose(nxt[0], x + tau * xd, rtol=0, atol=1e-15) and np.allclose(nxt[2], th + tau * thd, rtol=0, atol=1e-15)
        acc = np.array([(nxt[1] - xd) / tau, (nxt[3] - thd) / tau])
        f = F if action == 1 else -F
        A = np.array([[M, mp * l * np.cos(th)], [mp * l * np.cos(th), 4.0 / 3.0 * mp * l * l]])
        b = np.array([f + mp * l * thd * thd * np.sin(th), mp * g * l * np.sin(th)])
        np.testing.assert_allclose(acc, np.linalg.solve(A, b), rtol=0, atol=2e-10)
        # power balance: dE/dt = F * xd with E = T + V of the same Lagrangian
        xdd, thdd = acc
        dE = (M * xd * xdd + mp * l * (xdd * thd * np.cos(th) + xd * thdd * np.cos(th) - xd * thd * thd * np.sin(th))
              + 4.0 / 3.0 * mp * l * l * thd * thdd - mp * g * l * thd * np.sin(th))
        assert abs(dE - f * xd) < 1e-9
        assert reward == 1.0
        assert done == int(abs(nxt[0]) > 2.4 or abs(nxt[2]) > 12 * 2 * np.pi / 360)


def _acrobot_acc_lagrangian(q, qd, torque):
    """Two-link pendulum, angles: th1 from the downward vertical, th2 relative to link 1; m1 = m2 = 1, l1 = 1, lc1 = lc2 = 0.5,
    I1 = I2 = 1, g = 9.8; torque on joint 2.  Standard manipulator form M(q) qdd + C(q, qd) + G(q) = [0, tau]."""
    m1 = m2 = 1.0; l1 = 1.0; lc1 = lc2 = 0.5; I1 = I2 = 1.0; g = 9.8
    th1, th2 = q
    d11 = m1 * lc1 ** 2 + m2 * (l1 ** 2 + lc2 ** 2 + 2 * l1 * lc2 * np.cos(th2)) + I1 + I2
    d12 = m2 * (lc2 ** 2 + l1 * lc2 * np.cos(th2)) + I2
    d22 = m2 * lc2 ** 2 + I2
    h = m2 * l1 * lc2 * np.sin(th2)
    c1 = -h * qd[1] ** 2 - 2 * h * qd[0] * qd[1]
    c2 = h * qd[0] ** 2
    g1 = (m1 * lc1 + m2 * l1) * g * np.sin(th1) + m2 * lc2 * g * np.sin(th1 + th2)
    g2 = m2 * lc2 * g * np.sin(th1 + th2)
    return np.linalg.solve(np.array([[d11, d12], [d12, d22]]), np.array([-c1 - g1, torque - c2 - g2]))


def _wrap(x, lo, hi):
    d = hi - lo
    while x > hi:
        x -= d
    while x < lo:
        x += d
    return x


def test_acrobot_step_matches_independent_rk4_of_the_lagrangian_form():
    """One env step = one classical RK4 step (dt 0.2) of the 'book' dynamics, then angle wrap and velocity clip.  Here the
    same RK4 runs on the mass-matrix form above (no phi1/phi2/d1/d2 closed forms) -- the results must agree to rounding."""
    rng = np.random.RandomState(1)
    dt = 0.2
    for _ in range(200):
        st = rng.uniform([-np.pi, -np.pi, -4 * np.pi, -9 * np.pi], [np.pi, np.pi, 4 * np.pi, 9 * np.pi]) * np.array([1, 1, 0.5, 0.5])
        action = int(rng.randint(3))
        torque = float(action - 1)
        nxt, reward, done = _acrobot_step(st, action)

        def f(y):
            acc = _acrobot_acc_lagrangian(y[:2], y[2:], torque)
            return np.array([y[2], y[3], acc[0], acc[1]])

        k1 = f(st); k2 = f(st + dt / 2 * k1); k3 = f(st + dt / 2 * k2); k4 = f(st + dt * k3)
        ns = st + dt / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)
        ns[0] = _wrap(ns[0], -np.pi, np.pi); ns[1] = _wrap(ns[1], -np.pi, np.pi)
        ns[2] = np.clip(ns[2], -4 * np.pi, 4 * np.pi); ns[3] = np.clip(ns[3], -9 * np.pi, 9 * np.pi)
        np.testing.assert_allclose(nxt, ns, rtol=0, atol=5e-9)
        terminal = bool(-np.cos(ns[0]) - np.cos(ns[1] + ns[0]) > 1.0)
        assert done == int(terminal) and reward == (0.0 if terminal else -1.0)


def test_acrobot_energy_is_conserved_without_torque():
    """With zero torque the Lagrangian system conserves T + V; a single RK4 step of 0.2 s keeps it to the method's O(dt^5)."""
    m1 = m2 = 1.0; l1 = 1.0; lc1 = lc2 = 0.5; I1 = I2 = 1.0; g = 9.8

    def energy(s):
        th1, th2, w1, w2 = s
        d11 = m1 * lc1 ** 2 + m2 * (l1 ** 2 + lc2 ** 2 + 2 * l1 * lc2 * np.cos(th2)) + I1 + I2
        d12 = m2 * (lc2 ** 2 + l1 * lc2 * np.cos(th2)) + I2
        d22 = m2 * lc2 ** 2 + I2
        T = 0.5 * d11 * w1 * w1 + d12 * w1 * w2 + 0.5 * d22 * w2 * w2
        V = -(m1 * lc1 + m2 * l1) * g * np.cos(th1) - m2 * lc2 * g * np.cos(th1 + th2)
        return T + V

    rng = np.random.RandomState(2)
    for _ in range(100):
        st = rng.uniform(-1.0, 1.0, 4) * np.array([1.0, 1.0, 1.5, 1.5])
        nxt, _, _ = _acrobot_step(st, 1)            # action 1 = zero torque
        assert abs(energy(nxt) - energy(st)) < 2e-3 * max(1.0, abs(energy(st)))


# ------------------------------------------------------------------------------------------------------------------------------
# MountainCar-v0, MountainCarContinuous-v0, Pendulum-v0 (these three were only ever checked oracle-vs-kernel, i.e.
# against the same restatement).  Each gets a second derivation that shares no formula with oracle/lenv_oracle*.c(.inc):
#   * the mountain cars from the published HEIGHT PROFILE h(x) = 0.45 sin(3x) + 0.55 (gym's _height, what the renderer draws): the
#     slope term of the update must be -k h'(x) with k = 0.0025 / 1.35, h' taken numerically -- the restatement writes cos(3x);
#   * the pendulum from its Lagrangian (uniform rod, I = m l^2 / 3 about the pivot, V = m g (l/2) cos th with th = 0 upright):
#     th_dd = (-dV/dth + u) / I with dV/dth taken numerically -- the restatement writes -3g/(2l) sin(th + pi) + 3u/(m l^2);
# plus, for each, a conservation property of the symplectic-Euler update and the published bounds / termination / reward rules.
# ------------------------------------------------------------------------------------------------------------------------------
def _mountaincar_step(st, action):
    L = orc.lib()
    s = (C.c_double * 4)(st[0], st[1], 0.0, 0.0)
    r, d = C.c_double(), C.c_int()
    L.orc_mountaincar_step(s, int(action), C.byref(r), C.byref(d))
    return np.array([s[0], s[1]]), r.value, d.value


def _cmc_step(st, a):
    L = orc.lib()
    s = (C.c_double * 2)(*st)
    act = (C.c_float * 1)(a)
    r, d = C.c_double(), C.c_int()
    L.orc_cmc_step(s, act, C.byref(r), C.byref(d))
    return np.array(list(s)), r.value, d.value


def _pendulum_step(st, u):
    L = orc.lib()
    s = (C.c_double * 2)(*st)
    act = (C.c_float * 1)(u)
    r = C.c_double()
    L.orc_pendulum_step(s, act, C.byref(r))
    return np.array(list(s)), r.value


def _height(x):
    return 0.45 * np.sin(3.0 * x) + 0.55


def _dheight(x, h=1e-6):
    return (_height(x + h) - _height(x - h)) / (2.0 * h)


K_HILL = 0.0025 / 1.35          # gravity constant of both mountain cars over the height profile's amplitude x frequency (0.45 * 3)


def test_mountaincar_update_follows_the_height_profile():
    rng = np.random.RandomState(3)
    for _ in range(400):
        x, v = rng.uniform(-1.15, 0.45), rng.uniform(-0.06, 0.06)
        a = int(rng.randint(3))
        nxt, reward, done = _mountaincar_step([x, v], a)
        v_new = v + (a - 1) * 0.001 - K_HILL * _dheight(x)
        if abs(v_new) < 0.0699 and -1.19 < x + v_new < 0.59:           # away from the clips: the plain symplectic-Euler update
            assert abs(nxt[1] - v_new) < 1e-9 and abs(nxt[0] - (x + v_new)) < 1e-9
        assert -1.2 <= nxt[0] <= 0.6 and abs(nxt[1]) <= 0.07 and reward == -1.0
        assert done == int(nxt[0] >= 0.5 and nxt[1] >= 0)
    # bounds: speed clip, the inelastic left wall, the flag
    nxt, _, _ = _mountaincar_step([-0.9, 0.0699], 2)             # uphill slope behind the car + full throttle: past the speed limit
    assert nxt[1] == 0.07
    nxt, _, done = _mountaincar_step([-1.199, -0.05], 0)
    assert nxt[0] == -1.2 and nxt[1] == 0.0 and done == 0
    nxt, _, done = _mountaincar_step([0.49, 0.05], 2)
    assert done == 1 and nxt[0] >= 0.5


def test_mountaincar_coasting_conserves_energy_over_the_height_profile():
    """No engine force (action 1): E = v^2/2 + k h(x) of the published profile; the update is symplectic Euler, so E oscillates within
    O(step) of its start and never drifts (600 steps = several swings of the valley)."""
    st = np.array([-0.9, 0.0])
    e0 = 0.5 * st[1] ** 2 + K_HILL * _height(st[0])
    es = []
    for _ in range(600):
        st, _, done = _mountaincar_step(st, 1)
        assert not done and -1.2 < st[0] < 0.5
        es.append(0.5 * st[1] ** 2 + K_HILL * _height(st[0]))
    assert max(abs(e - e0) for e in es) < 0.03 * K_HILL * 0.9          # 3 % of the profile's potential range
    assert abs(np.mean(es[-200:]) - np.mean(es[:200])) < 0.003 * K_HILL * 0.9


def test_mountaincar_continuous_update_reward_and_flag():
    rng = np.random.RandomState(4)
    for _ in range(400):
        x, v = rng.uniform(-1.15, 0.4), rng.uniform(-0.06, 0.06)
        a = float(np.float32(rng.uniform(-1.5, 1.5)))                   # beyond [-1, 1]: the force is clipped, the cost is not
        nxt, reward, done = _cmc_step([x, v], a)
        force = min(max(a, -1.0), 1.0)
        v_new = v + force * 0.0015 - K_HILL * _dheight(x)
        if abs(v_new) < 0.0699 and -1.19 < x + v_new < 0.59:
            assert abs(nxt[1] - v_new) < 1e-9 and abs(nxt[0] - (x + v_new)) < 1e-9
        assert -1.2 <= nxt[0] <= 0.6 and abs(nxt[1]) <= 0.07
        assert done == int(nxt[0] >= 0.45 and nxt[1] >= 0)
        assert abs(reward - ((100.0 if done else 0.0) - 0.1 * a * a)) < 1e-12
    nxt, reward, done = _cmc_step([0.44, 0.05], 1.0)
    assert done == 1 and abs(reward - 99.9) < 1e-12
    # coasting: the same conservation property as the discrete car (zero action)
    st = np.array([-0.9, 0.0])
    e0 = K_HILL * _height(st[0])
    for _ in range(400):
        st, _, done = _cmc_step(st, 0.0)
        assert not done and abs(0.5 * st[1] ** 2 + K_HILL * _height(st[0]) - e0) < 0.03 * K_HILL * 0.9


def test_pendulum_acceleration_matches_the_lagrangian_and_the_cost_its_published_form():
    g, m, l, dt = 10.0, 1.0, 1.0, 0.05
    inertia = m * l * l / 3.0
    V = lambda th: m * g * (l / 2.0) * np.cos(th)
    rng = np.random.RandomState(5)
    for _ in range(400):
        th, thd = rng.uniform(-10.0, 10.0), rng.uniform(-7.0, 7.0)
        u_raw = float(np.float32(rng.uniform(-3.0, 3.0)))
        nxt, reward = _pendulum_step([th, thd], u_raw)
        u = min(max(u_raw, -2.0), 2.0)
        dV = (V(th + 1e-6) - V(th - 1e-6)) / 2e-6
        thdd = (-dV + u) / inertia
        v_new = thd + thdd * dt
        assert abs(nxt[0] - (th + v_new * dt)) < 1e-8                   # the angle advances with the UNCLIPPED new speed
        assert abs(nxt[1] - min(max(v_new, -8.0), 8.0)) < 1e-8
        # cost: angle distance to upright via atan2 (the restatement uses fmod), speed and torque terms
        ang = np.arctan2(np.sin(th), np.cos(th))
        if abs(abs(ang) - np.pi) > 1e-6:
            assert abs(-reward - (ang * ang + 0.1 * thd * thd + 0.001 * u * u)) < 1e-6
        # power balance of the same Lagrangian at the sampled state: dE/dt = u * th_d
        assert abs(inertia * thd * thdd + dV * thd - u * thd) < 1e-6 * max(1.0, abs(thd))


def test_pendulum_free_swing_conserves_energy():
    """Zero torque, below the speed clip: E = I th_d^2 / 2 + m g (l/2) cos th; symplectic Euler keeps it within O(dt) without drift."""
    g, m, l = 10.0, 1.0, 1.0
    inertia = m * l * l / 3.0
    E = lambda s: 0.5 * inertia * s[1] ** 2 + m * g * (l / 2.0) * np.cos(s[0])
    st = np.array([2.0, 0.0])
    e0, es = E(st), []
    for _ in range(800):
        st, _ = _pendulum_step(st, 0.0)
        assert abs(st[1]) < 8.0
        es.append(E(st))
    assert max(abs(e - e0) for e in es) < 0.05 * m * g * l            # 5 % of the potential range
    assert abs(np.mean(es[-300:]) - np.mean(es[:300])) < 0.005 * m * g * l
