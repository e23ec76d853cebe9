"""GTN_Master: the NES master (reference agents/GTN_master.py:15-308), MI355X edition.

Same constructor, attributes (`score_list`, `score_orig_list`, `score_transform_list`, `synthetic_env_orig`,
`model_dir`, `model_name`) and `run()` return tuple.  What changed underneath:

  * the `num_workers` GTN_Worker evaluations of a generation (3 inner loops each, agents/GTN_worker.py:84-102) run
    as ONE fused kernel launch on this rank's share of the population -- no worker processes, no sync files;
  * with torch.distributed initialised (one process per GPU, backend nccl == RCCL over xGMI) the population is
    split in contiguous blocks; the only exchange is ONE all-gather of [score_best, score_orig, sign] per worker
    (replaces write_worker_result/read_worker_results, :178-195); every rank then runs score_transform + update_env
    redundantly on bit-identical inputs, so theta never needs a broadcast (replaces write_worker_inputs, :147-176);
  * noise eps is one [num_workers, P] tensor drawn by ONE kernel (lenv_nes_draw: noise + fresh agents + chain keys) from
    a (seed, generation)-keyed counter RNG that every rank -- and the CPU oracle -- reproduces bit for bit
    (agents/GTN_worker.py:156-163 drew it per worker from a time-seeded global RNG);
  * wall-clock time-outs (calc_worker_timeout, :141-145) have no counterpart on the fused path: chains run to their step
    budgets.

`transport="file"` (or config["agents"]["gtn"]["transport"] == "file") is the compatibility mode: the master then drives
file-based GTN_Worker processes -- this package's or the reference's -- through the sync directory exactly like
reference :147-195 (`write_worker_inputs` / `read_worker_results`), and only the rank transform + theta update run here.
"""
import os
import random
import string
import time

import numpy as np
import torch
import torch.distributed as dist

from ..config import TABULAR_AGENTS
from ..envs.env_factory import EnvFactory
from ..models.model_utils import linear_params
from ..utils import calc_abs_param_sum
from .GTN_base import GTN_Base
from .nes_common import rank_table, shard_bounds
from .tasks import select_task

__all__ = ["GTN_Master", "rank_table"]

TEAM_GAVE_UP = -10          # _lib.STATUS_TEAM_GAVE_UP: chains whose team of workgroups could not assemble


class GTN_Master(GTN_Base):
    def __init__(self, config, bohb_id=-1, bohb_working_dir=None, engine=None, seed=0, verbose=False, transport=None, graph=None):
        super().__init__(bohb_id)
        self.config = config
        self.device = config["device"]
        self.env_name = config['env_name']

        gtn_config = config["agents"]["gtn"]
        self.max_iterations = gtn_config["max_iterations"]
        self.agent_name = gtn_config["agent_name"]
        self.num_workers = gtn_config["num_workers"]
        self.step_size = gtn_config["step_size"]
        self.nes_step_size = gtn_config["nes_step_size"]
        self.weight_decay = gtn_config["weight_decay"]
        self.score_transform_type = gtn_config["score_transform_type"]
        self.noise_std = gtn_config["noise_std"]
        self.mirrored_sampling = gtn_config["mirrored_sampling"]
        self.num_grad_evals = gtn_config["num_grad_evals"]
        self.grad_eval_type = gtn_config["grad_eval_type"]
        self.quit_when_solved = gtn_config["quit_when_solved"]
        self.synthetic_env_type = gtn_config["synthetic_env_type"]
        self.unsolved_weight = gtn_config["unsolved_weight"]
        self.time_mult = gtn_config.get("time_mult", 3)
        self.time_max = gtn_config.get("time_max", 600)
        self.time_sleep_master = gtn_config.get("time_sleep_master", 0.2)
        if gtn_config.get("mode") == 'single':       # reference :37-39
            self.time_sleep_master /= 10
        self.transport = transport or gtn_config.get("transport", "fused")
        if self.transport not in ("fused", "file"):
            raise ValueError("Unknown transport: " + str(self.transport))
        self.seed = int(seed)
        self.verbose = verbose

        if self.score_transform_type not in range(8):
            raise ValueError("Unknown rank transform type: " + str(self.score_transform_type))
        if self.grad_eval_type not in ('mean', 'minmax'):
            raise NotImplementedError('Unknown parameter for grad_eval_type: ' + str(self.grad_eval_type))
        if int(self.num_grad_evals) < 1:
            raise ValueError("num_grad_evals must be >= 1")
        self.cpw = 1 + 2 * int(self.num_grad_evals)       # chains per worker: orig, G x (+eps), G x (-eps)  (GTN_worker.py:84-104)
        if self.agent_name.lower() not in ("ddqn", "duelingddqn", "td3", "ddqn_vary", "duelingddqn_vary", "td3_vary", "ddqn_icm", "duelingddqn_icm", "ddqn_icm_vary", "duelingddqn_icm_vary", "td3_icm", "td3_icm_vary", "td3_discrete_vary") + TABULAR_AGENTS:
            raise NotImplementedError("inner agent '%s' has no fused kernel yet" % self.agent_name)

        self.time_elapsed_list = [None] * self.num_workers
        self.score_list = [None] * self.num_workers
        self.score_orig_list = [None] * self.num_workers
        self.score_transform_list = [None] * self.num_workers

        if engine is None:
            from ..engine import HipNesEngine
            engine = HipNesEngine()     # raises without a HIP device / built library: no CPU fallback
        self.engine = engine
        dev = engine.device

        self.env_factory = EnvFactory(config)
        if self.synthetic_env_type == 0:
            generate_synthetic_env_fn = self.env_factory.generate_virtual_env
        elif self.synthetic_env_type == 1:
            generate_synthetic_env_fn = self.env_factory.generate_reward_env
        else:
            raise NotImplementedError("Unknown synthetic_env_type value: " + str(self.synthetic_env_type))
        self.synthetic_env_orig = generate_synthetic_env_fn(print_str='GTN_Base: ')
        self.real_env = self.env_factory.generate_real_env()

        # theta: one flat device buffer aliased by synthetic_env_orig's nn.Linear parameters
        self.theta = self._flat_theta(dev)
        self.p_theta = self.theta.numel()

        # distributed layout: worker p lives on rank p // ceil(pop/world)
        self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # a process group -- of ANY size, a one-rank communicator included -- means the generation's fitness records go through the
        # collective (and the captured generation is the two graphs around it): what a one-GPU box runs is then exactly what N ranks run
        self.has_group = bool(dist.is_available() and dist.is_initialized())
        self.collectives_run = 0               # all-gathers issued so far (bench.py reports whether one really ran)
        self.w_lo, self.w_hi, self.w_per = shard_bounds(self.num_workers, self.rank, self.world)
        self.n_local = self.w_hi - self.w_lo

        if self.transport == "file":
            if self.world > 1:
                raise ValueError('transport="file" is single-process: the workers are the parallelism')
            self.w_lo, self.w_hi, self.w_per, self.n_local = 0, 0, self.num_workers, 0      # nothing is evaluated in-process
            self.task = self.cfg = self.agent_bounds = None
            self._eps_scratch = generate_synthetic_env_fn(print_str='GTN_Master: ')      # decodes the workers' eps state dicts
        else:
            self.task = select_task(config, engine, self.synthetic_env_orig)
            self.cfg = self.task.cfg
            self.agent_bounds = self.task.agent_bounds
        G = int(self.num_grad_evals)
        self.inner = self.task.make_inner(self.cpw * self.n_local) if self.n_local > 0 else None
        lw = np.arange(self.w_lo, self.w_hi)
        self.chain_worker = torch.from_numpy(np.repeat(lw, self.cpw).astype(np.int32)).to(dev)
        self.chain_sign = torch.tensor(([0.0] + [1.0] * G + [-1.0] * G) * self.n_local, dtype=torch.float32, device=dev)
        self.rank_table = torch.from_numpy(rank_table(self.score_transform_type, self.num_workers)).to(dev)
        self.eps = None
        self._gathered = None
        self._local = None
        # One HIP graph per generation (draw -> fused inner loop -> worker_best -> status_fold -> score_transform + update_env):
        # single-process runs on the HIP engine whose task needs no host work between the kernels (the *_vary tasks draw their
        # hyper-parameters on the host).  graph=False keeps the eager launches.
        # With a process group (any world size) the generation is TWO graphs around the one collective: (draw -> fused inner loop -> worker_best -> status_fold),
        # the eager all-gather of the fitness records, (score_transform + update_env) -- the collective itself is not captured, so the
        # path does not depend on the backend (RCCL, or gloo in the tests).
        capable = (self.transport == "fused" and self.n_local > 0 and getattr(engine, "graph_capable", False)
                   and not hasattr(self.task, "draw_hp"))
        if graph and not capable:
            raise ValueError("graph=True needs the fused transport on the HIP engine, local workers, and a task without host-side draws")
        self.use_graph = capable if graph is None else bool(graph)
        self._graph = self._graph2 = self._graph_gathered = self._gen_t = self._theta_prev = self._gather_buf = None
        self._gen_next = None
        self.graph_replays = 0                 # graph launches so far (tests: the captured path really ran)

        if bohb_working_dir:
            self.model_dir = str(os.path.join(bohb_working_dir, 'GTN_models_' + self.env_name))
        else:
            self.model_dir = str(os.path.join(os.getcwd(), "results", 'GTN_models_' + self.env_name))
        self.model_name = self.get_model_file_name(
            self.env_name + '_' + ''.join(random.choices(string.ascii_uppercase + string.digits, k=6)) + '.pt')
        self.best_score = -float('Inf')
        os.makedirs(self.model_dir, exist_ok=True)
        self._sync_replicas()

    def _sync_replicas(self):
        """theta is replicated, never exchanged per generation -- so it must START identical: rank 0's initial theta and
        model name are broadcast ONCE here (every later update is computed redundantly from bit-identical inputs)."""
        if self.world <= 1:
            return
        dist.broadcast(self.theta, src=0)
        name = [self.model_name]
        dist.broadcast_object_list(name, src=0)
        self.model_name = name[0]

    def _flat_theta(self, dev):
        env = self.synthetic_env_orig.env
        if hasattr(env, "flat_params") and self.engine.name == "hip":
            return env.flat_params()
        from ..models.model_utils import FlatParams
        self._flat_holder = FlatParams(env, dev)
        return self._flat_holder.flat

    def get_model_file_name(self, file_name):
        return os.path.join(self.model_dir, file_name)

    # ---------------------------------------------------------------------------------------------
    def evaluate_population(self, it):
        """One generation's worker evaluations on this rank + the all-gather.  Returns gathered [num_workers,4]
        = (score_best, score_orig, sign, 0) in worker order, identical on every rank."""
        local = self._evaluate_local(it)
        return self._gather(local)

    def _gather(self, local, out=None):
        pop = self.num_workers
        if self.has_group:
            gathered = out if out is not None else torch.empty((self.world * self.w_per, 4), dtype=torch.float64, device=self.engine.device)
            dist.all_gather_into_tensor(gathered, local)      # the ONE collective of a generation (RCCL over xGMI)
            self.collectives_run += 1
            return gathered[:pop]                             # (a leading slice of a contiguous tensor: contiguous)
        return local[:pop]

    def _evaluate_local(self, it):
        """The device work of a generation on this rank (no collective): returns this rank's [w_per, 4] fitness records."""
        dev = self.engine.device
        pop = self.num_workers
        cpw = self.cpw
        # ONE launch draws the generation's stochastic inputs from the (seed, generation)-keyed counter RNG: the noise of the
        # WHOLE population (every rank needs every eps row for the redundant theta update), and the fresh agents + chain
        # keys of this rank's chains.  GTN_Worker.get_random_noise (agents/GTN_worker.py:156-163): N(0,1) * noise_std.
        bounds = self.agent_bounds if self.task.needs_agent_init() else None
        self.eps, local_init, keys_t = self.engine.draw(self.seed, it, pop, self.p_theta, self.noise_std, cpw * self.n_local, cpw,
                                                        self.w_lo, bounds)
        # rows [n_local, w_per) are padding of an uneven split: zero from the allocation on, never written
        if self._local is None:
            self._local = torch.zeros((self.w_per, 4), dtype=torch.float64, device=dev)
        local = self._local
        if self.n_local > 0:
            chain_scores = self.task.scores(self.inner, self.theta, self.eps, self.chain_worker, self.chain_sign, keys_t,
                                            local_init)
            self.engine.worker_best(chain_scores, self.n_local, self.mirrored_sampling, int(self.num_grad_evals),
                                    self.grad_eval_type, out=local[:self.n_local])
            # column 3 carries this rank's worst chain status through the all-gather: every rank sees every rank's
            # failure in the one host read-back of the generation and raises together (no second sync, no hang)
            self.engine.status_fold(self.inner, local[:self.n_local])
        return local

    def step(self, it):
        """One NES generation (the body of the reference's run() loop, :84-106).  Returns (mean_score_orig, solved)."""
        if self.transport == "file":
            return self._step_file(it)
        if self.use_graph:
            return self._step_graph(it)
        t1 = time.time()
        gathered = self.evaluate_population(it)
        self._gathered = gathered
        host = gathered.cpu().numpy()               # the generation's only host sync
        if host[:, 3].min() == TEAM_GAVE_UP and self._disable_teams():
            # a team of workgroups could not assemble (a foreign kernel held CUs): the chains are deterministic functions of
            # (theta, seed, generation), so the generation is simply evaluated again with one workgroup per chain.  Every rank
            # sees the same gathered statuses and takes this branch together.
            gathered = self._gathered = self.evaluate_population(it)
            host = gathered.cpu().numpy()
        if host[:, 3].min() != 0:
            raise RuntimeError("inner loop reported status %d on worker(s) %s (tape underrun / invalid replay index)"
                               % (int(host[:, 3].min()), np.nonzero(host[:, 3])[0].tolist()))
        self.score_list = host[:, 0].tolist()
        self.score_orig_list = host[:, 1].tolist()
        self.time_elapsed_list = [time.time() - t1] * self.num_workers
        mean_score = np.mean(self.score_orig_list)
        solved_flag = self.save_good_model(mean_score)
        if solved_flag and self.quit_when_solved:
            return mean_score, True
        self._transform_and_update(gathered)
        if self.verbose and self.rank == 0:
            self.print_statistics(it=it, time_elapsed=time.time() - t1)
        return mean_score, False

    def _disable_teams(self):
        """React to status -10 (a team member gave up waiting for the others, include/lenv_hip.h lenv_ddqn_cfg::team_size): from now
        on every launch of this master uses one workgroup per chain.  False when the launches were not teamed to begin with."""
        cfg = self.cfg
        if cfg is None or not hasattr(cfg, "team_size") or cfg.team_size == 1:
            return False
        cfg.team_size = 1
        self.team_fallbacks = getattr(self, "team_fallbacks", 0) + 1
        return True

    def _capture_generation(self):
        """Capture one whole generation on a side stream.  Every tensor the kernels touch is allocated inside the capture (the
        graph's private pool) or owned by self, so replays need no argument updates: the generation number lives on the device
        (lenv_nes_draw_dev reads it, lenv_nes_rank_update_keep advances it)."""
        dev = self.engine.device
        self._gen_t = torch.zeros(1, dtype=torch.int64, device=dev)
        self._theta_prev = torch.empty_like(self.theta)
        self._local = torch.zeros((self.w_per, 4), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        # with a process group alive its watchdog thread polls events: only THIS thread's calls belong to the capture
        mode = {"capture_error_mode": "thread_local"} if self.has_group else {}
        if not self.has_group:
            with torch.cuda.graph(g, stream=side):
                gathered = self.evaluate_population(self._gen_t)
                self._weights = self.engine.rank_update(self.score_transform_type, gathered, self.rank_table, self.theta, self.eps,
                                                        self.step_size, self.nes_step_size, self.weight_decay,
                                                        theta_prev=self._theta_prev, generation=self._gen_t)
            self._graph, self._graph2, self._graph_gathered, self._gen_next = g, None, gathered, 0
            return
        # with a process group: graph 1 = this rank's device work up to its fitness records, graph 2 = the redundant rank update on the gathered
        # records; the all-gather runs eagerly between the two replays (same memory pool: graph 2 reads the eps graph 1 drew)
        self._gather_buf = torch.empty((self.world * self.w_per, 4), dtype=torch.float64, device=dev)
        with torch.cuda.graph(g, stream=side, **mode):
            self._evaluate_local(self._gen_t)
        g2 = torch.cuda.CUDAGraph()
        gathered = self._gather_buf[:self.num_workers]
        with torch.cuda.graph(g2, stream=side, pool=g.pool(), **mode):
            self._weights = self.engine.rank_update(self.score_transform_type, gathered, self.rank_table, self.theta, self.eps,
                                                    self.step_size, self.nes_step_size, self.weight_decay,
                                                    theta_prev=self._theta_prev, generation=self._gen_t)
        self._graph, self._graph2, self._graph_gathered, self._gen_next = g, g2, gathered, 0

    def _step_graph(self, it):
        """step() as one graph replay + the generation's single host read-back.  The reference decides save_good_model and
        quit_when_solved BEFORE update_env (agents/GTN_master.py:95-101); here the update has already run when the scores reach the
        host, so the pre-update theta the graph kept (theta_prev) is swapped back in for the save / the early return."""
        t1 = time.time()
        if self._graph is None:
            try:
                self._capture_generation()
            except RuntimeError as e:
                # the capture is an optimisation: if this stack refuses it (a collective library that polls the device from another
                # thread, an allocator in an odd state), the generation runs as eager launches -- same kernels, same results.  Every
                # rank issues exactly one all-gather per generation on either path, so ranks need not agree on it.
                self.use_graph, self._graph, self._graph2 = False, None, None
                self.graph_capture_error = str(e)
                torch.cuda.synchronize()
                return self.step(it)
        if self._gen_next != it:
            self._gen_t.fill_(int(it))
        self._graph.replay()
        self.graph_replays += 1
        if self._graph2 is not None:
            self._gather(self._local, out=self._gather_buf)      # the generation's one collective, between the two graphs
            self._graph2.replay()
            self.graph_replays += 1
        self._gen_next = it + 1
        gathered = self._gathered = self._graph_gathered
        host = gathered.cpu().numpy()               # the generation's only host sync
        if host[:, 3].min() != 0:
            # the captured update_env has already run on scores of failed chains: put theta and the device generation counter
            # back to where the generation started before anything else happens (the eager path never applies such an update)
            self.theta.copy_(self._theta_prev)
            self._theta_changed()
            self._gen_t.fill_(int(it))
            self._gen_next = it
            if host[:, 3].min() == TEAM_GAVE_UP and self._disable_teams():
                self._graph = None                  # re-captured with one workgroup per chain on the next line
                return self._step_graph(it)
            raise RuntimeError("inner loop reported status %d on worker(s) %s (tape underrun / invalid replay index)"
                               % (int(host[:, 3].min()), np.nonzero(host[:, 3])[0].tolist()))
        self.score_list = host[:, 0].tolist()
        self.score_orig_list = host[:, 1].tolist()
        self.time_elapsed_list = [time.time() - t1] * self.num_workers
        mean_score = np.mean(self.score_orig_list)
        save, solved = self._save_decision(mean_score)
        if save:
            updated = self.theta.clone()
            self.theta.copy_(self._theta_prev)
            self._theta_changed()
            self.save_good_model(mean_score)
            if solved and self.quit_when_solved:
                return mean_score, True             # theta stays as it was before this generation's update, like the reference
            self.theta.copy_(updated)
        self.score_transform_list = None
        self._theta_changed()
        if self.verbose and self.rank == 0:
            self.print_statistics(it=it, time_elapsed=time.time() - t1)
        return mean_score, False

    def run(self):
        mean_score_orig_list = []
        for it in range(self.max_iterations):
            mean_score, solved = self.step(it)
            mean_score_orig_list.append(mean_score)
            if solved:
                break
        if len(mean_score_orig_list) > 0:
            return np.mean(self.score_orig_list), mean_score_orig_list, self.model_name
        return 1e9, mean_score_orig_list, self.model_name

    def _save_decision(self, mean_score):
        """(save the model?, counts as solved?) -- the two conditions of reference :118-131."""
        better = mean_score > self.best_score
        if self.synthetic_env_orig.is_virtual_env():
            hit = bool(better and mean_score > self.real_env.get_solved_reward())
            return hit, hit
        return bool(better), False

    def save_good_model(self, mean_score):
        save, solved = self._save_decision(mean_score)
        if save:
            self.save_model()
            self.best_score = mean_score
        return solved

    def save_model(self):
        # reference :133-139 -- {'model': state_dict, 'config': config}
        if self.rank != 0:
            return
        save_dict = {'model': {k: v.detach().cpu() for k, v in self.synthetic_env_orig.state_dict().items()},
                     'config': self.config}
        torch.save(save_dict, os.path.join(self.model_dir, self.model_name))

    def score_transform(self, gathered=None):
        """reference :197-265.  Weights are computed on device together with update_env; this method keeps the
        reference's name and fills `score_transform_list`.  Without an argument it ranks the last evaluated generation
        (INCLUDING the mirrored-sampling sign of every worker)."""
        if gathered is None:
            gathered = self._last_gathered()
        self._weights = self.engine.rank_update(self.score_transform_type, gathered, self.rank_table, None, None, 0.0,
                                                False, 0.0)
        self.score_transform_list = self._weights.cpu().tolist()

    def update_env(self, gathered=None):
        """reference :267-298: theta <- theta*(1-wd); theta += (ss*w_i) * eps_i in worker order (eps_i sign-flipped when
        mirrored sampling picked -eps)."""
        if gathered is None:
            gathered = self._last_gathered()
        self.engine.rank_update(self.score_transform_type, gathered, self.rank_table, self.theta, self.eps,
                                self.step_size, self.nes_step_size, self.weight_decay)
        self._theta_changed()

    def _transform_and_update(self, gathered):
        """score_transform + update_env in one device call (one ranking pass)."""
        self._weights = self.engine.rank_update(self.score_transform_type, gathered, self.rank_table, self.theta, self.eps,
                                                self.step_size, self.nes_step_size, self.weight_decay)
        self.score_transform_list = None     # materialised lazily (needs a host copy)
        self._theta_changed()

    def _theta_changed(self):
        """The kernels write theta through a raw pointer, which bumps no torch version counter: tell the env wrapper so
        caches derived from the parameters (RewardEnv's shaped-reward table) are rebuilt."""
        env = self.synthetic_env_orig.env
        if hasattr(env, "params_changed"):
            env.params_changed()

    def get_score_transform_list(self):
        if self.score_transform_list is None:
            self.score_transform_list = self._weights.cpu().tolist()
        return self.score_transform_list

    def _last_gathered(self):
        """(score_best, score_orig, sign, status) of the generation whose eps is in self.eps.  If the caller edited
        `score_list` / `score_orig_list` (the reference's public attributes) the edited scores are used, the signs stay."""
        if self._gathered is None:
            raise RuntimeError("no generation evaluated yet: pass `gathered` or run step()/read_worker_results() first")
        g = self._gathered.clone()
        g[:, 0] = torch.tensor(self.score_list, dtype=torch.float64, device=g.device)
        g[:, 1] = torch.tensor(self.score_orig_list, dtype=torch.float64, device=g.device)
        return g

    def gathered_from_lists(self, sign_list):
        """Explicit form for callers that bring their own lists: sign_list[i] = -1 where worker i's -eps won."""
        g = np.zeros((self.num_workers, 4))
        g[:, 0], g[:, 1], g[:, 2] = self.score_list, self.score_orig_list, sign_list
        return torch.from_numpy(g).to(self.engine.device)

    # ---- file transport (compatibility mode), reference :141-195 ----
    def calc_worker_timeout(self):
        if self.time_elapsed_list[0] is None:
            return self.time_max
        return float(np.mean(self.time_elapsed_list)) * self.time_mult

    def write_worker_inputs(self, it):
        timeout = self.calc_worker_timeout()
        state = {k: v.detach().cpu() for k, v in self.synthetic_env_orig.state_dict().items()}
        for id in range(self.num_workers):
            file_name = self.get_input_file_name(id=id)
            while os.path.isfile(file_name):          # the worker deletes it to acknowledge the previous input
                time.sleep(self.time_sleep_master)
            time.sleep(self.time_sleep_master)
            quit_flag = (it == self.max_iterations - 1) if self.bohb_id < 0 else False
            torch.save({'timeout': timeout, 'quit_flag': quit_flag, 'config': self.config, 'synthetic_env_orig': state}, file_name)
            torch.save({}, self.get_input_check_file_name(id=id))

    def read_worker_results(self):
        dev = self.engine.device
        self.eps = torch.empty((self.num_workers, self.p_theta), dtype=torch.float32, device=dev)
        for id in range(self.num_workers):
            file_name = self.get_result_file_name(id)
            check_file_name = self.get_result_check_file_name(id)
            while not os.path.isfile(check_file_name):
                time.sleep(self.time_sleep_master)
            data = torch.load(file_name)
            self.time_elapsed_list[id] = data['time_elapsed']
            self.score_list[id] = data['score']
            self.score_orig_list[id] = data['score_orig']
            # the worker's eps already carries the mirrored-sampling sign (GTN_worker.py:180-185 invert_eps)
            self._eps_scratch.load_state_dict(data['eps'])
            self.eps[id] = torch.cat([p.detach().reshape(-1) for p in linear_params(self._eps_scratch)]).to(dev)
            os.remove(check_file_name)
            os.remove(file_name)
        g = np.zeros((self.num_workers, 4))
        g[:, 0], g[:, 1], g[:, 2] = self.score_list, self.score_orig_list, 1.0
        self._gathered = torch.from_numpy(g).to(dev)

    def _step_file(self, it):
        t1 = time.time()
        self.write_worker_inputs(it)
        self.read_worker_results()
        mean_score = np.mean(self.score_orig_list)
        solved_flag = self.save_good_model(mean_score)
        if solved_flag and self.quit_when_solved:
            return mean_score, True
        self._transform_and_update(self._gathered)
        if self.verbose:
            self.print_statistics(it=it, time_elapsed=time.time() - t1)
        return mean_score, False

    def print_statistics(self, it, time_elapsed):
        print('--------------')
        print('GTN iteration:    ' + str(it))
        print('GTN mstr t_elaps: ' + str(time_elapsed))
        print('GTN avg eval score:   ' + str(float(np.mean(self.score_orig_list))))
        print('GTN |theta|_1:    ' + str(float(calc_abs_param_sum(self.synthetic_env_orig))))
        print('--------------')
