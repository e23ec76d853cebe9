"""Tabular grid MDPs of the reference (envs/gridworld.py:24-238) as transition tables for the device kernels.

The reference steps a python object cell by cell; here every layout is compiled once into
next_state[s,a] / reward[s,a] / done[s,a] (actions: 0 right, 1 left, 2 down, 3 up -- gridworld.py:13-16), which is
what `lenv_ql_rn_inner_loop` and `lenv_grid_env_step` consume.  Semantics restated from gridworld.py:63-110:
moves are clipped at the border, walls '#' block, holes 'O' are absorbing, reward/done depend on the cell entered
(goal 'G': g_reward, hole: o_reward, else step_cost; done on goal or hole)."""
import numpy as np

G_RIGHT, G_LEFT, G_DOWN, G_UP = 0, 1, 2, 3

# (grid rows, step_cost, g_reward, o_reward)  -- layouts of gridworld.py:125-238
LAYOUTS = {
    "EmptyRoom22": (["S ", " G"], -0.01, 1, -1),
    "EmptyRoom23": (["S  ", "  G"], -0.01, 1, -1),
    "EmptyRoom33": (["S  ", "   ", "  G"], -0.01, 1, -1),
    "EmptyRoom": (["S    ", "     ", "     ", "     ", "    G"], -0.01, 1, -1),
    "WallRoom": (["S #  ", "  #  ", "     ", "  #  ", "  # G"], -0.01, 1, -1),
    "HoleRoom": (["S O  ", "  O  ", "     ", "  O  ", "  O G"], -0.01, 1, -1),
    "HoleRoomLarge": (["S O O  ", "  O O  ", "       ", "  O O  ", "  O O G"], -0.01, 1, -1),
    "HoleRoomLargeShifted": (["S O    ", "  O O  ", "  O O  ", "  O O  ", "    O G"], -0.01, 1, -1),
    "Cliff": (["            ", "            ", "            ", "SOOOOOOOOOOG"], -1, 0, -100),
}


def transition_tables(name):
    """-> dict(n_states, n_actions, start_state, next_state int32[N,4], reward float64[N,4], done uint8[N,4])"""
    if name not in LAYOUTS:
        raise NotImplementedError("unknown gridworld '%s'" % name)
    grid, step_cost, g_reward, o_reward = LAYOUTS[name]
    m, n = len(grid), len(grid[0])
    N = m * n
    nxt = np.zeros((N, 4), np.int32)
    rew = np.zeros((N, 4), np.float64)
    done = np.zeros((N, 4), np.uint8)
    start = None
    for x in range(m):
        for y in range(n):
            if grid[x][y] == 'S':
                start = x * n + y
            for a in range(4):
                if grid[x][y] == 'O':
                    xn, yn = x, y                       # hole -> stuck
                else:
                    dx, dy = {G_LEFT: (0, -1), G_UP: (-1, 0), G_RIGHT: (0, 1), G_DOWN: (1, 0)}[a]
                    xn, yn = min(max(x + dx, 0), m - 1), min(max(y + dy, 0), n - 1)
                    if grid[xn][yn] == '#':
                        xn, yn = x, y
                cell = grid[xn][yn]
                nxt[x * n + y, a] = xn * n + yn
                rew[x * n + y, a] = g_reward if cell == 'G' else (o_reward if cell == 'O' else step_cost)
                done[x * n + y, a] = 1 if cell in ('O', 'G') else 0
    if start is None:
        raise ValueError("No start state found")
    return dict(n_states=N, n_actions=4, start_state=start, next_state=nxt, reward=rew, done=done, shape=(m, n))
