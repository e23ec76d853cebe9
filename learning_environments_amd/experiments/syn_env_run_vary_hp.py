"""The caller of the evaluation harness (one row up): reference experiments/syn_env_run_vary_hp.py.

    get_all_files(with_vary_hp, model_num, model_dir, custom_load_envs_and_config, env_name, device, filter_models_list=None)   (:8-29)
    run_vary_hp(mode, experiment_name, model_num, agents_num, model_dir, custom_load_envs_and_config, custom_train_test_agents,
                env_name, pool=None, device="cuda", filter_models_list=None, correlation_exp=False)                             (:32-139)

Same names, arguments, list shapes and output file (`<mode>_<experiment_name>.pt` = utils.save_lists, reference utils.py:144-160).  mode 0:
`model_num` times `agents_num` agents trained on the REAL env; mode 1 / 2: `agents_num` agents on each of the first `model_num` checkpoints
whose config has vary_hp off / on.  Where the reference loops over the models or spreads them over a multiprocessing pool (:66-72,102-110), a
`custom_train_test_agents` that carries a `.fused` attribute (this package's train_test_agents does) gets ALL models in one fused launch:
model_num * agents_num chains, which is what fills an MI355X (40 models x 10 agents = 400 chains); any other callable is called model by
model like the reference's pool-less branch.  `pool` is accepted and ignored (one process drives the GPU)."""
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


def run_vary_hp(mode, experiment_name, model_num, agents_num, model_dir, custom_load_envs_and_config, custom_train_test_agents, env_name,
                pool=None, device="cuda", filter_models_list=None, correlation_exp=False, out_dir=None):
    if mode not in (0, 1, 2):
        raise ValueError("mode 0: real env, mode 1: syn. env. (no vary), mode 2: syn. env. (vary)")
    train_on_venv = mode != 0
    with_vary_hp = mode == 2
    fused = getattr(custom_train_test_agents, "fused", None)
    env_reward_overview = {}
    reward_list, train_steps_needed, episode_length_needed = [], [], []
    if not train_on_venv:
        file_name = os.listdir(model_dir)[0]
        _, real_env, config = custom_load_envs_and_config(file_name=file_name, model_dir=model_dir, device=device)
        names = [real_env.env.env_name + "_" + str(i) for i in range(model_num)]
        if fused is not None:
            per_model = fused([real_env] * model_num, real_env, config, agents_num)
        else:
            per_model = [custom_train_test_agents(train_env=real_env, test_env=real_env, config=config, agents_num=agents_num) for _ in range(model_num)]
    else:
        names = get_all_files(with_vary_hp=with_vary_hp, model_num=model_num, model_dir=model_dir,
                              custom_load_envs_and_config=custom_load_envs_and_config, env_name=env_name, device=device,
                              filter_models_list=filter_models_list)
        loaded = [custom_load_envs_and_config(file_name=f, model_dir=model_dir, device=device) for f in names]
        if fused is not None and loaded:
            config = loaded[0][2]                          # (the pool branch of the reference keeps the first file's config too, :100)
            per_model = fused([l[0] for l in loaded], loaded[0][1], config, agents_num)
        else:
            per_model = []
            for virtual_env, real_env, config in loaded:
                per_model.append(custom_train_test_agents(train_env=virtual_env, test_env=real_env, config=config, agents_num=agents_num))
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
    save_lists(mode=mode, config=config, reward_list=reward_list, train_steps_needed=train_steps_needed,
               episode_length_needed=episode_length_needed, env_reward_overview=env_reward_overview, experiment_name=experiment_name,
               out_dir=out_dir)
    return reward_list, train_steps_needed, episode_length_needed
