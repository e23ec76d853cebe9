"""CPU GTN worker speaking the reference's sync-file protocol, scored by the C oracle.  TEST / BASELINE INFRASTRUCTURE.

This is the "(ii) file-IO worker mode" CPU baseline of SURVEY.md §8(d): one single-threaded process per worker doing what
reference agents/GTN_worker.py:76-154 does -- wait for `<bohb>_<id>_input.pt`, draw eps, run the three inner loops
(theta, theta+eps, theta-eps) sequentially, pick the mirrored-sampling winner, write `<bohb>_<id>_result.pt` -- with the
reference's sleeps (`time_sleep_worker`, /10 in mode 'single').  The arithmetic is oracle/lenv_oracle.c (the pinned CPU
restatement), so only bench.py's cpu_baseline leg and tests/ may run it; nothing in learning_environments_amd imports it.

    python -m oracle.file_worker <id> [--bohb-id B] [--seed S]      (cwd = the directory that holds results/GTN_sync)
"""
import argparse
import math
import os
import sys
import time

import numpy as np


def _sync_file(bohb_id, id, kind):      # reference agents/GTN_base.py:13-29
    return os.path.join(os.getcwd(), "results", "GTN_sync", "%s_%s_%s.pt" % (bohb_id, id, kind))


def _is_linear_key(sd, k):
    """nn.Linear parameters of a state dict (the only ones noise / updates touch, GTN_worker.py:158,167): 2-D weights and
    their biases; the 1-element PReLU slope (`….1.weight`) is not one."""
    if k.endswith("weight"):
        return sd[k].dim() == 2
    return k.endswith("bias")


def flat_linear(sd):
    return np.concatenate([sd[k].detach().cpu().numpy().astype(np.float32).reshape(-1) for k in sd if _is_linear_key(sd, k)])


def unflat_like(sd, flat):
    import torch
    out, off = {}, 0
    for k, v in sd.items():
        if _is_linear_key(sd, k):
            n = v.numel()
            out[k] = torch.from_numpy(flat[off:off + n].reshape(tuple(v.shape)).copy())
            off += n
        else:
            out[k] = v.detach().cpu().clone()
    return out


def agent_bounds(cfg):
    """nn.Linear default-init bound 1/sqrt(fan_in) per parameter of Critic_DQN(S -> H x L -> A), flat state-dict order."""
    dims = [(cfg.state_dim, cfg.q_hidden)] + [(cfg.q_hidden, cfg.q_hidden)] * (cfg.q_layers - 1) + [(cfg.q_hidden, cfg.num_actions)]
    return np.concatenate([np.full(i * o + o, 1.0 / math.sqrt(i), np.float32) for i, o in dims])


def evaluate(orc, config, theta, rng, seed, generation, wid, grad_chunk):
    """One worker-evaluation (GTN_worker.py:84-104 with num_grad_evals = G): returns (score_best, score_orig, eps*sign)."""
    g = config["agents"]["gtn"]
    G = int(g["num_grad_evals"])
    cfg = orc.ddqn_cfg_from_config(config, grad_chunk=grad_chunk, rng_mode=0)
    eps = (rng.standard_normal(theta.size) * g["noise_std"]).astype(np.float32)
    bounds = agent_bounds(cfg)
    scores = []
    for kind, sg in enumerate([0.0] + [1.0] * G + [-1.0] * G):
        w = (np.float32(sg) * eps + theta).astype(np.float32)
        init = ((rng.random(bounds.size).astype(np.float32) * 2 - 1) * bounds).astype(np.float32)
        scores.append(orc.ddqn_se_chain(cfg, w, init, rng_key=orc.chain_key(seed, generation, wid, kind))["score"])
    best, sign = orc.worker_best_multi(np.array([scores[1:1 + G]]), np.array([scores[1 + G:]]), bool(g["mirrored_sampling"]),
                                       g["grad_eval_type"])
    return float(best[0]), float(scores[0]), eps * np.float32(sign[0])


def run(id, bohb_id=-1, seed=None, grad_chunk=17, max_generations=None, initial_sleep=0.2):
    import torch
    torch.set_num_threads(1)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle as orc
    seed = int(seed if seed is not None else id + 1)
    rng = np.random.default_rng(seed)
    # the reference worker polls with 3 s until its first input tells it the configured value (GTN_worker.py:31,62-63); a
    # steady-state sample should not be dominated by that start-up constant, so the configured 'single'-mode value
    # (time_sleep_worker/10 = 0.2 s) is used from the start unless told otherwise
    sleep = float(initial_sleep)
    generation = 0
    quit_flag = False
    while not quit_flag:
        # read_worker_input (GTN_worker.py:116-137)
        f_in, f_chk = _sync_file(bohb_id, id, "input"), _sync_file(bohb_id, id, "input_check")
        while not os.path.isfile(f_chk):
            time.sleep(sleep)
        time.sleep(sleep)
        data = torch.load(f_in)
        config, quit_flag = data["config"], data["quit_flag"]
        sleep = config["agents"]["gtn"]["time_sleep_worker"] / (10 if config["agents"]["gtn"]["mode"] == "single" else 1)
        sd = data["synthetic_env_orig"]
        os.remove(f_chk)
        os.remove(f_in)
        t0 = time.time()
        theta = flat_linear(sd)
        best, orig, eps = evaluate(orc, config, theta, rng, seed, generation, id, grad_chunk)
        # write_worker_result (GTN_worker.py:139-154)
        f_res, f_rchk = _sync_file(bohb_id, id, "result"), _sync_file(bohb_id, id, "result_check")
        while os.path.isfile(f_res):
            time.sleep(sleep)
        torch.save({"eps": unflat_like(sd, eps), "synthetic_env": unflat_like(sd, theta + eps),
                    "time_elapsed": time.time() - t0, "score": best, "score_orig": orig}, f_res)
        torch.save({}, f_rchk)
        generation += 1
        if max_generations is not None and generation >= max_generations:
            break


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("id", type=int)
    ap.add_argument("--bohb-id", type=int, default=-1)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--grad-chunk", type=int, default=17)
    ap.add_argument("--max-generations", type=int, default=None)
    ap.add_argument("--initial-sleep", type=float, default=0.2)
    a = ap.parse_args()
    run(a.id, a.bohb_id, a.seed, a.grad_chunk, a.max_generations, a.initial_sleep)
