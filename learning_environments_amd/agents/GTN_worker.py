"""GTN_Worker: one NES population member (reference agents/GTN_worker.py:15-254), MI355X edition.

Kept for API/transport compatibility: same constructor, `late_init`, noise methods, `calc_score`,
`calc_best_score`, and the file protocol of `run()` (read `<bohb_id>_<id>_input.pt`, write `..._result.pt`), so it can
serve a reference GTN_Master (or this package's GTN_Master in transport="file" mode).  Every (inner agent, synthetic env)
combination of agents/tasks.py is supported: DDQN / DuelingDDQN on a VirtualEnv, QL / SARSA (+count-based) on a gridworld
RewardEnv, TD3 on the HalfCheetah stand-in RewardEnv.  The three inner loops of an evaluation (theta, theta+eps, theta-eps) run as ONE launch
of the fused kernel (3 chains) instead of three sequential python training runs.  The in-process GTN_Master of this
package does not use worker objects at all.
"""
import os
import time

import numpy as np
import torch

from .. import _lib
from ..envs.env_factory import EnvFactory
from ..models.model_utils import linear_params
from .GTN_base import GTN_Base
from .nes_common import chain_keys, fresh_agent_init, host_worker_best
from .tasks import select_task


class GTN_Worker(GTN_Base):
    def __init__(self, id, bohb_id=-1, engine=None, seed=None):
        super().__init__(bohb_id)
        self.id = id
        self.seed = int(seed) if seed is not None else int(id + bohb_id * id * 1000 + int(time.time()))
        torch.manual_seed(self.seed)
        self.test_counter = 0
        self.quit_flag = False
        self.time_sleep_worker = 3
        self.timeout = None
        self.engine = engine
        self.generation = 0
        self.team_fallbacks = 0                # launches repeated with one workgroup per chain (status -10, _run_chains)
        self._no_teams = False
        self.task = None
        self._task_config = None
        self._inner = {}
        for file in self.worker_files(self.id):
            if os.path.isfile(file):
                os.remove(file)
        # (reference :47 -- also what a launcher waits for before the master writes its first input: the line is flushed after the
        # worker's stale sync files are gone)
        print("Starting GTN Worker with bohb_id {} and id {}".format(bohb_id, id), flush=True)

    # the gtn section's keys a worker keeps as attributes of the same name (reference agents/GTN_worker.py:49-58)
    _GTN_KEYS = ("noise_std", "num_grad_evals", "grad_eval_type", "mirrored_sampling", "time_sleep_worker", "agent_name",
                 "synthetic_env_type", "unsolved_weight")
    _ENV_BUILDERS = {0: "generate_virtual_env", 1: "generate_reward_env"}      # gtn.synthetic_env_type -> EnvFactory method

    def late_init(self, config):
        """Called with the config the master sent (reference :49-74): settings, then three instances of the synthetic env
        (theta, the perturbed copy, and one whose parameters hold the noise), then the fused-kernel task for the inner agent."""
        gtn = config["agents"]["gtn"]
        for key in self._GTN_KEYS:
            setattr(self, key, gtn[key])
        if gtn["mode"] == "single":                      # master and workers on one machine poll ten times as often
            self.time_sleep_worker = self.time_sleep_worker / 10
        if self.engine is None:
            from ..engine import HipNesEngine
            self.engine = HipNesEngine()
        self.config = config
        self.env_factory = EnvFactory(config)
        builder = self._ENV_BUILDERS.get(self.synthetic_env_type)
        if builder is None:
            raise NotImplementedError("Unknown synthetic_env_type value: " + str(self.synthetic_env_type))
        make_env = getattr(self.env_factory, builder)
        tag = "GTN_Worker%s: " % self.id
        self.synthetic_env_orig = make_env(print_str="GTN_Base: ")
        self.synthetic_env, self.eps = make_env(print_str=tag), make_env(print_str=tag)
        # inner agent x synthetic-env type -> fused kernel (agents/agent_utils.py:15-66 select_agent + EnvFactory).  read_worker_input calls
        # late_init every generation (reference :117): the task and its inner loops (workspaces, the team fall-back of _run_chains) are
        # rebuilt only when the master sent a different config
        if self.task is None or config != self._task_config:
            import copy
            self.task = select_task(config, self.engine, self.synthetic_env_orig)
            self._task_config = copy.deepcopy(config)
            self._inner = {}
        self.cfg = self.task.cfg
        self._bounds = self.task.agent_bounds

    # ---- noise handling (reference :156-185) over THE flat NES layout: linear_params() lists the nn.Linear weights and biases of an env
    # in the order theta / eps are flattened in, so the three envs' lists line up element for element ----
    def get_random_noise(self):
        for p_eps in linear_params(self.eps):           # one N(0, 1) * noise_std draw per parameter tensor, weights before biases, layer by layer
            p_eps.data.copy_(torch.normal(mean=torch.zeros_like(p_eps), std=torch.ones_like(p_eps)) * self.noise_std)

    def add_noise_to_synthetic_env(self, add=True):
        triples = zip(linear_params(self.synthetic_env_orig), linear_params(self.synthetic_env), linear_params(self.eps))
        for p_orig, p_virt, p_eps in triples:
            p_virt.data.copy_(p_orig + p_eps if add else p_orig - p_eps)

    def subtract_noise_from_synthetic_env(self):
        self.add_noise_to_synthetic_env(add=False)

    def invert_eps(self):
        for p_eps in linear_params(self.eps):
            p_eps.data.neg_()

    # ---- scoring ----
    def _flat(self, envw):
        return torch.cat([p.detach().reshape(-1) for p in linear_params(envw)]).to(self.engine.device, torch.float32).contiguous()

    def _run_chains(self, thetas):
        """Scores of len(thetas) chains (each its own SE parameter vector) in one launch."""
        n = len(thetas)
        dev = self.engine.device
        if n not in self._inner:
            self._inner[n] = self.task.make_inner(n, want_episode_stats=False)
            if self._no_teams and hasattr(self._inner[n].cfg, "team_size"):
                self._inner[n].cfg.team_size = 1           # a team launch of this worker already gave up once: see below
        inner = self._inner[n]
        # express chain c as theta0 + 1*(theta_c - theta0)?  No: exactness matters -> run with eps rows = theta_c, theta = 0
        zero = torch.zeros_like(thetas[0])
        eps = torch.stack(thetas).contiguous()
        worker = torch.arange(n, dtype=torch.int32, device=dev)
        sign = torch.ones(n, dtype=torch.float32, device=dev)
        g = torch.Generator(device=dev)
        g.manual_seed((self.seed * 1000003 + self.generation * 7919 + self.test_counter) % (2 ** 63 - 1))
        agent_init = fresh_agent_init(self._bounds, n, g, dev) if self.task.needs_agent_init() else None
        keys = chain_keys(self.seed, self.generation * 1000 + self.test_counter, np.full(n, self.id), np.arange(n))
        self.test_counter += 1
        keys_t = torch.from_numpy(keys.view(np.int64)).to(dev)
        scores = self.task.scores(inner, zero, eps, worker, sign, keys_t, agent_init)
        # Several worker processes usually share one GPU: a launch whose teams of workgroups could not assemble next to another
        # worker's kernel reports status -10 (include/lenv_hip.h, lenv_ddqn_cfg::team_size).  The chains are deterministic functions
        # of these inputs, so the launch is repeated once with one workgroup per chain; this worker's inner loops keep that setting
        # (self._no_teams: also the ones it builds later).
        status = getattr(inner, "status", None)
        if status is not None and status.numel() and int(status.min()) == _lib.STATUS_TEAM_GAVE_UP and getattr(inner.cfg, "team_size", 1) != 1:
            inner.cfg.team_size = 1
            self._no_teams = True
            self.team_fallbacks += 1
            scores = self.task.scores(inner, zero, eps, worker, sign, keys_t, agent_init)
        out = scores.cpu().tolist()
        if hasattr(self.engine, "check_status"):
            self.engine.check_status(inner)
        return out

    def calc_score(self, env, time_remaining=1e9):
        """reference :187-221: fresh agent, train on `env` (VirtualEnv :199-209 or RewardEnv :211-221) with per-episode
        real-env tests, final test on the real env, mean test return."""
        if env.is_virtual_env() != (self.synthetic_env_type == 0):
            raise ValueError("calc_score: env kind does not match synthetic_env_type %s" % self.synthetic_env_type)
        return self._run_chains([self._flat(env)])[0]

    def calc_best_score(self, score_sub, score_add):
        """reference :234-254 -- the mirrored pick, on the host (two scalars: statistics.mean's exactly rounded mean or the minimum of
        each side, then the comparison; nes_common.host_worker_best, the one-worker form of the engine's lenv_nes_worker_best_multi).
        Afterwards, as in the reference, eps points in the direction that scored better and synthetic_env holds theta + eps."""
        score_best, sign = host_worker_best([float(v) for v in score_add], [float(v) for v in score_sub], bool(self.mirrored_sampling),
                                            self.grad_eval_type)
        if sign < 0:
            self.invert_eps()                   # -eps won: eps := -eps; synthetic_env already holds theta - (old eps)
        else:
            self.add_noise_to_synthetic_env()
        return score_best

    def evaluate(self):
        """One worker-evaluation (reference run() body :84-104) with the three inner loops batched in one launch."""
        self.get_random_noise()
        th = self._flat(self.synthetic_env_orig)
        e = self._flat(self.eps)
        G = int(self.num_grad_evals)                       # reference :90-104: G evaluations of +eps, then G of -eps
        scores = self._run_chains([th] + [th + e] * G + [th - e] * G)
        score_orig, score_add, score_sub = scores[0], scores[1:1 + G], scores[1 + G:]
        self.subtract_noise_from_synthetic_env()          # state the reference is in before calc_best_score
        score_best = self.calc_best_score(score_add=score_add, score_sub=score_sub)
        return score_best, score_orig

    # ---- file transport, reference :76-154 ----
    def run(self):
        while not self.quit_flag:
            self.read_worker_input()
            time_start = time.time()
            score_best, score_orig = self.evaluate()
            self.write_worker_result(score=score_best, score_orig=score_orig, time_elapsed=time.time() - time_start)
            self.generation += 1
            if self.quit_flag:
                break

    def read_worker_input(self):
        file_name = self.get_input_file_name(id=self.id)
        check_file_name = self.get_input_check_file_name(id=self.id)
        while not os.path.isfile(check_file_name):
            time.sleep(self.time_sleep_worker)
        time.sleep(self.time_sleep_worker)
        data = torch.load(file_name)
        self.timeout = data['timeout']
        self.quit_flag = data['quit_flag']
        self.config = data['config']
        self.late_init(self.config)
        self.synthetic_env_orig.load_state_dict(data['synthetic_env_orig'])
        self.synthetic_env.load_state_dict(data['synthetic_env_orig'])
        os.remove(check_file_name)
        os.remove(file_name)

    def write_worker_result(self, score, score_orig, time_elapsed):
        file_name = self.get_result_file_name(id=self.id)
        check_file_name = self.get_result_check_file_name(id=self.id)
        while os.path.isfile(file_name):
            time.sleep(self.time_sleep_worker)
        data = {"eps": {k: v.cpu() for k, v in self.eps.state_dict().items()},
                "synthetic_env": {k: v.cpu() for k, v in self.synthetic_env.state_dict().items()},
                "time_elapsed": time_elapsed, "score": score, "score_orig": score_orig}
        torch.save(data, file_name)
        torch.save({}, check_file_name)
