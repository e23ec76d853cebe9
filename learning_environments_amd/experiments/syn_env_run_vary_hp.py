"""The caller of the evaluation harness (one row up): reference experiments/syn_env_run_vary_hp.py.

    get_all_files(with_vary_hp, model_num, model_dir, custom_load_envs_and_config, env_name, device, filter_models_list=None)   (:8-29)
    run_vary_hp(mode, experiment_name, model_num, agents_num, model_dir, custom_load_envs_and_config, custom_train_test_agents,
                env_name, pool=None, device="cuda", filter_models_list=None, correlation_exp=False)                             (:32-139)

Same names, arguments, list shapes and output file (`<mode>_<experiment_name>.pt` = utils.save_lists, reference utils.py:144-160).  mode 0:
`model_num` times `agents_num` agents trained on the REAL env; mode 1 / 2: `agents_num` agents on each of the first `model_num` checkpoints
whose config has vary_hp off / on.  Where the reference loops over the models or spreads them over a multiprocessing pool (:66-72,102-110), a
`custom_train_test_agents` that carries a `.fused` attribute (this package's train_test_agents does) gets ALL models in one fused launch:
model_num * agents_num chains, which is what fills an MI355X (40 models x 10 agents = 400 chains); any other callable is called model by
model like the reference's pool-less branch.  `pool` is accepted and ignored (one process drives the GPU).

Several GPUs: inside a torch.distributed process group (one process per GPU, `torchrun ... -m learning_environments_amd.experiments.syn_env_run_vary_hp`)
the models are dealt to the ranks round-robin (model m -> rank m mod N; the chains are keyed by (seed, model index, agent index), so the lists do not
depend on N), every rank runs its share as one launch, ONE all_gather_object of the per-model lists puts the full result on every rank, and rank 0
writes the file.  There is no other communication: the unit (a model's agents) is independent (tests/test_run_vary_hp.py, two gloo ranks)."""
import os

import numpy as np

from ..utils import save_lists


def get_all_files(with_vary_hp, model_num, model_dir, custom_load_envs_and_config, env_name, device, filter_models_list=None):
    file_list = []
    for file_name in os.listdir(model_dir):
        if env_name not in file_name:
            continue
        _, _, config = custom_load_envs_and_config(file_name=file_name, model_dir=model_dir, device=device)
        if config['agents']['ddqn_vary']['vary_hp'] == with_vary_hp:
            file_list.append(file_name)
    # sort file list by random characters/digits -> make randomness deterministic   (reference :18-19)
    file_list = sorted(file_list, key=lambda elem: elem[-9:])
    if len(file_list) < model_num and filter_models_list is None:
        raise ValueError("Not enough saved models")
    if filter_models_list is not None:
        return [f for f in file_list if f in filter_models_list]
    return file_list[:model_num]


def _ranks():
    """(rank, world) of the torch.distributed process group this process belongs to, (0, 1) without one."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return 0, 1


def run_vary_hp(mode, experiment_name, model_num, agents_num, model_dir, custom_load_envs_and_config, custom_train_test_agents, env_name,
                pool=None, device="cuda", filter_models_list=None, correlation_exp=False, out_dir=None):
    if mode not in (0, 1, 2):
        raise ValueError("mode 0: real env, mode 1: syn. env. (no vary), mode 2: syn. env. (vary)")
    train_on_venv = mode != 0
    with_vary_hp = mode == 2
    fused = getattr(custom_train_test_agents, "fused", None)
    rank, world = _ranks()
    env_reward_overview = {}
    reward_list, train_steps_needed, episode_length_needed = [], [], []
    if not train_on_venv:
        file_name = sorted(os.listdir(model_dir))[0] if world > 1 else os.listdir(model_dir)[0]      # (every rank must pick the same file)
        _, real_env, config = custom_load_envs_and_config(file_name=file_name, model_dir=model_dir, device=device)
        names = [real_env.env.env_name + "_" + str(i) for i in range(model_num)]
        mine = list(range(rank, model_num, world))
        if fused is not None:
            per_mine = fused([real_env] * len(mine), real_env, config, agents_num, model_indices=mine) if mine else []
        else:
            per_mine = [custom_train_test_agents(train_env=real_env, test_env=real_env, config=config, agents_num=agents_num) for _ in mine]
    else:
        names = get_all_files(with_vary_hp=with_vary_hp, model_num=model_num, model_dir=model_dir,
                              custom_load_envs_and_config=custom_load_envs_and_config, env_name=env_name, device=device,
                              filter_models_list=filter_models_list)
        mine = list(range(rank, len(names), world))
        loaded = [custom_load_envs_and_config(file_name=names[m], model_dir=model_dir, device=device) for m in mine]
        # the config that is saved: the first file's (the pool branch of the reference keeps it too, :100; rank 0 always holds model 0, and the
        # harness function writes its settings for comparability into this very dict)
        config = loaded[0][2] if loaded else (custom_load_envs_and_config(file_name=names[0], model_dir=model_dir, device=device)[2] if names else None)
        if fused is not None and loaded:
            per_mine = fused([l[0] for l in loaded], loaded[0][1], config, agents_num, model_indices=mine)
        else:
            per_mine = []
            for virtual_env, real_env, cfg_m in loaded:
                per_mine.append(custom_train_test_agents(train_env=virtual_env, test_env=real_env, config=cfg_m, agents_num=agents_num))
                if world == 1:
                    config = cfg_m
    if world > 1:
        # the one exchange step: every rank's (model index, lists) pairs -> the full result on every rank, in model order
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, list(zip(mine, per_mine)))
        per_model = [None] * len(names)
        for part in gathered:
            for m, lists in part:
                per_model[m] = lists
    else:
        per_model = per_mine
    for name, (reward_list_i, train_steps_needed_i, episode_length_needed_i) in zip(names, per_model):
        if correlation_exp and train_on_venv:
            # (:112-117: one entry per model)
            reward_list.append(reward_list_i)
            train_steps_needed.append(train_steps_needed_i)
            episode_length_needed.append(episode_length_needed_i)
        else:
            reward_list += reward_list_i
            train_steps_needed += train_steps_needed_i
            episode_length_needed += episode_length_needed_i
        env_reward_overview[name] = {} if correlation_exp else np.hstack(reward_list_i)
    if rank == 0:
        save_lists(mode=mode, config=config, reward_list=reward_list, train_steps_needed=train_steps_needed,
                   episode_length_needed=episode_length_needed, env_reward_overview=env_reward_overview, experiment_name=experiment_name,
                   out_dir=out_dir)
    return reward_list, train_steps_needed, episode_length_needed


def main(argv=None):
    """The `__main__` block the harness scripts share (syn_env_evaluate_cartpole_vary_hp_2.py:51-101 and siblings): same flags, plus --model_dir /
    --env_name / --agent (the scripts hard-code a results directory, "CartPole" or "Acrobot", and one agent each) and --out_dir.
        python -m learning_environments_amd.experiments.syn_env_run_vary_hp --model_dir DIR [--mode 0|1|2] [--agents_num 10] [--model_num 40]
    Without --mode the three modes run one after the other, like the scripts do."""
    import argparse
    from functools import partial
    from .syn_env_evaluate import HARNESS_AGENTS, load_envs_and_config, train_test_agents, train_test_agents_models
    parser = argparse.ArgumentParser(description=main.__doc__)
    parser.add_argument('--mode', type=int, help='mode 0: real env, mode 1: syn. env. (no vary), mode 2: syn. env. (vary)')
    parser.add_argument('--pool', type=int, help='size of the multiprocessing pool (accepted, unused: one process drives the GPU)')
    parser.add_argument('--agents_num', type=int, help='number of agents evaluated', default=10)
    parser.add_argument('--model_num', type=int, help='number of models evaluated', default=40)
    parser.add_argument('--device', type=str, help='device to be used', default='cuda')
    parser.add_argument('--model_dir', type=str, required=True, help='directory of the saved GTN models (reference format)')
    parser.add_argument('--env_name', type=str, default='CartPole', help='substring of the checkpoints\' file names ("CartPole", "Acrobot")')
    parser.add_argument('--agent', type=str, default='DDQN_vary', help='the sibling script: ' + ', '.join(sorted(HARNESS_AGENTS)))
    parser.add_argument('--train_episodes', type=int, default=1000, help='1000; the Acrobot TD3_discrete script uses 500')
    parser.add_argument('--out_dir', type=str, default=None)
    parser.add_argument('--generalization_gap', action='store_true',
                        help='the *_eval_generalization_gap script: vary_hp off, the fixed optimised DDQN hyper-parameters (4-57-2 tanh, batch 199)')
    parser.add_argument('--correlation', action='store_true',
                        help='the *_correlation scripts: per drawn configuration 100 DDQN agents on the SE and 100 on the real env (mode 2, correlation_exp)')
    args = parser.parse_args(argv)
    if args.agent.lower() not in HARNESS_AGENTS:
        parser.error("unknown --agent %r (one of %s)" % (args.agent, sorted(HARNESS_AGENTS)))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # under torchrun: one process per GPU; the only exchange is one all_gather_object of python lists per mode (gloo carries it)
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        if not dist.is_initialized():
            dist.init_process_group("gloo")
    print("model_num:", args.model_num, "agents_num:", args.agents_num, "pool size:", args.pool, "device:", args.device)
    experiment_name = "%s_transfer_reward_overview_%d_agents_num_%d_model_num" % (args.agent.lower(), args.agents_num, args.model_num)
    harness = partial(train_test_agents, agent_name=args.agent, train_episodes=args.train_episodes)
    harness.fused = partial(train_test_agents_models, agent_name=args.agent, train_episodes=args.train_episodes)
    out = {}
    if args.correlation:
        from .syn_env_evaluate import train_test_agents_correlation
        experiment_name = "ddqn_vary_correlation_syn_real_early_out_num_1000_%d_agents_num_%d_model_num" % (args.agents_num, args.model_num)
        harness = partial(train_test_agents_correlation, train_episodes=args.train_episodes)
        out[2] = run_vary_hp(mode=2, experiment_name=experiment_name, model_num=args.model_num, agents_num=args.agents_num, model_dir=args.model_dir,
                             custom_load_envs_and_config=load_envs_and_config, custom_train_test_agents=harness, env_name=args.env_name, pool=None,
                             device=args.device, correlation_exp=True, out_dir=args.out_dir)
        return out
    if args.generalization_gap:
        from .syn_env_evaluate import train_test_agents_generalization_gap
        experiment_name = "ddqn_generalization_gap_%d_agents_num_%d_model_num" % (args.agents_num, args.model_num)
        harness = train_test_agents_generalization_gap
    for mode in ([args.mode] if args.mode is not None else range(3)):
        out[mode] = run_vary_hp(mode=mode, experiment_name=experiment_name, model_num=args.model_num, agents_num=args.agents_num,
                                model_dir=args.model_dir, custom_load_envs_and_config=load_envs_and_config, custom_train_test_agents=harness,
                                env_name=args.env_name, pool=None, device=args.device, out_dir=args.out_dir)
        print("mode %d: %d agents, mean test return %.2f" % (mode, len(out[mode][0]), float(np.mean([np.mean(r) for r in out[mode][0]]))))
    return out


if __name__ == "__main__":
    main()
