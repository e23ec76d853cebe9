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
