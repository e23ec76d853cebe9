"""ctypes loader for liblenv_hip.so (C-ABI: include/lenv_hip.h).  Fails loudly when the library is absent."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblenv_hip.so")
CSRC = os.path.join(_HERE, "csrc")

ACT = {"identity": 0, "relu": 1, "leakyrelu": 2, "tanh": 3, "prelu": 4}
ENV = {"CartPole-v0": 0, "Acrobot-v1": 1, "HalfCheetah-v3": 2, "MountainCar-v0": 3, "Pendulum-v0": 4, "MountainCarContinuous-v0": 5}
RNG_COUNTER, RNG_TAPE = 0, 1
VARIANT_NO_WAVECHAIN, VARIANT_GENERIC, VARIANT_TEAM_NARROW, VARIANT_NO_DIRECT = 1, 2, 4, 8     # lenv_ddqn_cfg / lenv_td3_cfg kernel_variant bits (A/B timing, kernel-vs-kernel parity tests)
STATUS_TEAM_GAVE_UP = -10                        # a team member waited too long for the others: repeat the launch with team_size 1

ERRORS = {-1: ValueError, -2: NotImplementedError, -3: ValueError, -4: RuntimeError, -5: RuntimeError}


class LenvError(RuntimeError):
    pass


class MlpDesc(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32), ("out_dim", C.c_int32),
                ("act", C.c_int32), ("prelu", C.c_float), ("use_layer_norm", C.c_int32)]


class DdqnCfg(C.Structure):
    _fields_ = [("env_id", C.c_int32), ("state_dim", C.c_int32), ("num_actions", C.c_int32), ("max_steps", C.c_int32),
                ("se_hidden", C.c_int32), ("se_layers", C.c_int32), ("se_act", C.c_int32), ("se_prelu", C.c_float),
                ("q_hidden", C.c_int32), ("q_layers", C.c_int32), ("q_act", C.c_int32), ("q_prelu", C.c_float),
                ("batch_size", C.c_int32), ("rb_size", C.c_int32),
                ("train_episodes", C.c_int32), ("test_episodes", C.c_int32), ("init_episodes", C.c_int32),
                ("early_out_num", C.c_int32), ("grad_chunk", C.c_int32), ("rng_mode", C.c_int32),
                ("agent_kind", C.c_int32), ("feature_dim", C.c_int32),
                ("solved_reward", C.c_double), ("gamma", C.c_double), ("lr", C.c_double), ("tau", C.c_double),
                ("eps_init", C.c_double), ("eps_min", C.c_double), ("eps_decay", C.c_double),
                ("adam_beta1", C.c_double), ("adam_beta2", C.c_double), ("adam_eps", C.c_double),
                ("step_budget", C.c_int64),
                ("icm_enabled", C.c_int32), ("icm_feature_dim", C.c_int32), ("icm_hidden", C.c_int32), ("se_layer_norm", C.c_int32),
                ("icm_lr", C.c_double), ("icm_beta", C.c_double), ("icm_eta", C.c_double),
                ("synthetic_env_type", C.c_int32), ("reward_env_type", C.c_int32),
                ("same_action_num", C.c_int32),
                ("team_size", C.c_int32),        # workgroups per chain: 0 = automatic, 1 = never a team, G = at most G
                ("kernel_variant", C.c_int32),   # VARIANT_* bits, 0 = fastest
                ("q_layer_norm", C.c_int32), ("test_mode", C.c_int32), ("early_out_virtual_diff", C.c_double)]


class Tapes(C.Structure):
    _fields_ = [("eps_uniform", C.c_void_p), ("eps_uniform_stride", C.c_int64),
                ("rand_action", C.c_void_p), ("rand_action_stride", C.c_int64),
                ("replay_idx", C.c_void_p), ("replay_idx_stride", C.c_int64),
                ("train_reset", C.c_void_p), ("train_reset_stride", C.c_int64),
                ("test_reset", C.c_void_p), ("test_reset_stride", C.c_int64)]


class InnerOut(C.Structure):
    _fields_ = [("score", C.c_void_p), ("stats", C.c_void_p), ("status", C.c_void_p),
                ("episode_test_mean", C.c_void_p), ("episode_len", C.c_void_p), ("final_returns", C.c_void_p),
                ("final_online", C.c_void_p), ("trace_cap", C.c_int64), ("trace_action", C.c_void_p),
                ("trace_state", C.c_void_p), ("trace_next_state", C.c_void_p), ("trace_reward_done", C.c_void_p)]


class ChainHp(C.Structure):                  # include/lenv_hip.h: lenv_chain_hp (device arrays [chains])
    _fields_ = [("lr", C.c_void_p), ("batch_size", C.c_void_p), ("q_hidden", C.c_void_p), ("q_layers", C.c_void_p)]


class IcmIo(C.Structure):                    # include/lenv_hip.h: lenv_icm_io
    _fields_ = [("icm_init", C.c_void_p), ("icm_final", C.c_void_p)]


class QlCfg(C.Structure):
    _fields_ = [("n_states", C.c_int32), ("n_actions", C.c_int32), ("start_state", C.c_int32), ("max_steps", C.c_int32),
                ("rn_hidden", C.c_int32), ("rn_layers", C.c_int32), ("rn_act", C.c_int32), ("rn_prelu", C.c_float),
                ("reward_env_type", C.c_int32), ("train_episodes", C.c_int32), ("test_episodes", C.c_int32),
                ("init_episodes", C.c_int32), ("early_out_num", C.c_int32), ("batch_size", C.c_int32), ("rng_mode", C.c_int32),
                ("agent_kind", C.c_int32), ("count_based", C.c_int32),
                ("solved_reward", C.c_double), ("alpha", C.c_double), ("gamma", C.c_double), ("eps_init", C.c_double),
                ("eps_min", C.c_double), ("eps_decay", C.c_double), ("beta", C.c_double), ("step_budget", C.c_int64), ("same_action_num", C.c_int32), ("rn_layer_norm", C.c_int32), ("test_mode", C.c_int32), ("early_out_virtual_diff", C.c_double)]


class QlOut(C.Structure):
    _fields_ = [("score", C.c_void_p), ("stats", C.c_void_p), ("status", C.c_void_p), ("episode_test_mean", C.c_void_p),
                ("episode_len", C.c_void_p), ("final_returns", C.c_void_p), ("q_table", C.c_void_p), ("shaped", C.c_void_p),
                ("trace_cap", C.c_int64), ("trace_action", C.c_void_p), ("trace_state", C.c_void_p),
                ("trace_reward_done", C.c_void_p)]


class Td3Cfg(C.Structure):
    _fields_ = [("env_id", C.c_int32), ("state_dim", C.c_int32), ("action_dim", C.c_int32), ("max_steps", C.c_int32),
                ("rn_hidden", C.c_int32), ("rn_layers", C.c_int32), ("rn_act", C.c_int32), ("rn_prelu", C.c_float),
                ("reward_env_type", C.c_int32), ("info_dim", C.c_int32), ("hidden", C.c_int32), ("layers", C.c_int32), ("act", C.c_int32),
                ("prelu", C.c_float), ("batch_size", C.c_int32), ("rb_size", C.c_int32), ("train_episodes", C.c_int32),
                ("test_episodes", C.c_int32), ("init_episodes", C.c_int32), ("early_out_num", C.c_int32),
                ("policy_delay", C.c_int32), ("rng_mode", C.c_int32),
                ("solved_reward", C.c_double), ("gamma", C.c_double), ("lr", C.c_double), ("tau", C.c_double),
                ("action_std", C.c_double), ("policy_std", C.c_double), ("policy_std_clip", C.c_double),
                ("max_action", C.c_double), ("adam_beta1", C.c_double), ("adam_beta2", C.c_double), ("adam_eps", C.c_double),
                ("step_budget", C.c_int64),
                ("icm_enabled", C.c_int32), ("icm_feature_dim", C.c_int32), ("icm_hidden", C.c_int32), ("use_layer_norm", C.c_int32),
                ("icm_lr", C.c_double), ("icm_beta", C.c_double), ("icm_eta", C.c_double),
                ("virtual_env", C.c_int32), ("same_action_num", C.c_int32), ("team_size", C.c_int32), ("kernel_variant", C.c_int32), ("rn_layer_norm", C.c_int32), ("test_mode", C.c_int32), ("early_out_virtual_diff", C.c_double)]


class Td3Tapes(C.Structure):
    _fields_ = [("rand_action", C.c_void_p), ("rand_action_stride", C.c_int64), ("act_noise", C.c_void_p), ("act_noise_stride", C.c_int64),
                ("test_noise", C.c_void_p), ("test_noise_stride", C.c_int64), ("policy_noise", C.c_void_p), ("policy_noise_stride", C.c_int64),
                ("replay_idx", C.c_void_p), ("replay_idx_stride", C.c_int64), ("train_reset", C.c_void_p), ("train_reset_stride", C.c_int64),
                ("test_reset", C.c_void_p), ("test_reset_stride", C.c_int64)]


class Td3dCfg(C.Structure):
    """lenv_td3d_cfg (include/lenv_hip.h): TD3_discrete_vary on a VirtualEnv over a discrete-action real env."""
    _fields_ = [("env_id", C.c_int32), ("state_dim", C.c_int32), ("action_dim", C.c_int32), ("max_steps", C.c_int32),
                ("se_hidden", C.c_int32), ("se_layers", C.c_int32), ("se_act", C.c_int32), ("se_prelu", C.c_float),
                ("hidden", C.c_int32), ("layers", C.c_int32), ("act", C.c_int32), ("prelu", C.c_float),
                ("use_layer_norm", C.c_int32), ("gumbel_hard", C.c_int32),
                ("batch_size", C.c_int32), ("rb_size", C.c_int32), ("train_episodes", C.c_int32), ("test_episodes", C.c_int32),
                ("init_episodes", C.c_int32), ("early_out_num", C.c_int32), ("policy_delay", C.c_int32), ("rng_mode", C.c_int32),
                ("solved_reward", C.c_double), ("gamma", C.c_double), ("lr", C.c_double), ("tau", C.c_double),
                ("action_std", C.c_double), ("policy_std", C.c_double), ("policy_std_clip", C.c_double), ("max_action", C.c_double),
                ("gumbel_temp", C.c_double), ("adam_beta1", C.c_double), ("adam_beta2", C.c_double), ("adam_eps", C.c_double),
                ("step_budget", C.c_int64), ("se_layer_norm", C.c_int32), ("test_mode", C.c_int32), ("early_out_virtual_diff", C.c_double)]


TD3D_TAPE_KEYS = ("rand_action", "act_noise", "test_noise", "policy_noise", "gumbel_act", "gumbel_test", "gumbel_target", "gumbel_actor",
                  "replay_idx", "train_reset", "test_reset")


class Td3dTapes(C.Structure):
    _fields_ = [f for k in TD3D_TAPE_KEYS for f in ((k, C.c_void_p), (k + "_stride", C.c_int64))]


class Td3Out(C.Structure):
    _fields_ = [("score", C.c_void_p), ("stats", C.c_void_p), ("status", C.c_void_p), ("episode_test_mean", C.c_void_p),
                ("episode_len", C.c_void_p), ("final_returns", C.c_void_p), ("final_params", C.c_void_p), ("trace_cap", C.c_int64),
                ("trace_action", C.c_void_p), ("trace_state", C.c_void_p), ("trace_next_state", C.c_void_p), ("trace_reward", C.c_void_p)]


# lenv_struct_size(which) order (include/lenv_hip.h)
ABI_STRUCTS = [MlpDesc, DdqnCfg, QlCfg, Td3Cfg, Td3dCfg, Tapes, InnerOut, QlOut, Td3Tapes, Td3Out, Td3dTapes, ChainHp, IcmIo]

EXPORTS = ["lenv_abi_version", "lenv_error_string", "lenv_mlp_num_params", "lenv_se_step_population",
           "lenv_qnet_td_forward", "lenv_ddqn_se_workspace_bytes", "lenv_ddqn_se_lds_bytes", "lenv_ddqn_se_forward_split", "lenv_ddqn_se_team_size", "lenv_ddqn_se_inner_loop",
           "lenv_chain_key", "lenv_nes_worker_best", "lenv_nes_rank_update", "lenv_real_env_reset", "lenv_real_env_step", "lenv_ql_rn_inner_loop", "lenv_rn_shape_population", "lenv_dueling_se_workspace_bytes",
           "lenv_dueling_num_params", "lenv_dueling_se_inner_loop", "lenv_dueling_se_inner_loop_hp", "lenv_dueling_agent_init_hp", "lenv_rng_unit", "lenv_td3_rn_inner_loop_hp", "lenv_td3_agent_init_hp", "lenv_icm_num_params", "lenv_dueling_se_inner_loop_icm", "lenv_chain_uniform_init", "lenv_td3_icm_num_params", "lenv_td3_rn_inner_loop_icm", "lenv_td3_rn_workspace_bytes", "lenv_td3_num_params",
           "lenv_td3_rn_inner_loop", "lenv_mlp_forward", "lenv_cheetah_standin_reset", "lenv_cheetah_standin_step", "lenv_cont_env_reset", "lenv_cont_env_step",
           "lenv_rn_num_params", "lenv_rn_shape_rows", "lenv_nes_worker_best_multi", "lenv_nes_draw", "lenv_nes_status_fold", "lenv_nes_draw_dev", "lenv_nes_rank_update_keep",
           "lenv_td3d_workspace_bytes", "lenv_td3d_num_params", "lenv_td3d_se_num_params", "lenv_td3d_inner_loop", "lenv_td3d_agent_init", "lenv_td3_rn_team_size", "lenv_dueling_team_size", "lenv_struct_size"]


def build(force=False):
    """Compile every HIP source for gfx950 into liblenv_hip.so (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-s", "-j4"])
    return LIB_PATH


_lib = None


def lib():
    """The loaded C-ABI library.  Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LenvError("liblenv_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; "
                            "g.build()'` or `make -C learning_environments_amd/csrc`; there is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.lenv_abi_version.restype = C.c_int
        L.lenv_error_string.restype = C.c_char_p
        L.lenv_error_string.argtypes = [C.c_int]
        L.lenv_mlp_num_params.restype = C.c_int64
        L.lenv_mlp_num_params.argtypes = [C.POINTER(MlpDesc)]
        L.lenv_chain_key.restype = C.c_uint64
        L.lenv_chain_key.argtypes = [C.c_uint64] * 4
        L.lenv_ddqn_se_workspace_bytes.restype = C.c_size_t
        L.lenv_ddqn_se_workspace_bytes.argtypes = [C.POINTER(DdqnCfg), C.c_int64]
        vp = C.c_void_p
        L.lenv_se_step_population.restype = C.c_int
        L.lenv_se_step_population.argtypes = [C.POINTER(MlpDesc)] * 3 + [vp, vp, vp, vp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp, vp]
        L.lenv_qnet_td_forward.restype = C.c_int
        L.lenv_qnet_td_forward.argtypes = [C.POINTER(MlpDesc), vp, vp, vp, C.c_int64, C.c_int32, vp, C.c_int64, C.c_int32,
                                           C.c_double, vp, vp, vp]
        L.lenv_ddqn_se_inner_loop.restype = C.c_int
        L.lenv_ddqn_se_inner_loop.argtypes = [C.POINTER(DdqnCfg), vp, vp, vp, vp, vp, vp, C.POINTER(Tapes), C.c_int64, vp,
                                              C.c_size_t, C.POINTER(InnerOut), vp]
        L.lenv_ddqn_se_lds_bytes.restype = C.c_int64
        L.lenv_ddqn_se_lds_bytes.argtypes = [C.POINTER(DdqnCfg)]
        L.lenv_ddqn_se_team_size.restype = C.c_int
        L.lenv_ddqn_se_team_size.argtypes = [C.POINTER(DdqnCfg), C.c_int64]
        L.lenv_ddqn_se_forward_split.restype = C.c_int
        L.lenv_ddqn_se_forward_split.argtypes = [C.POINTER(DdqnCfg), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.lenv_real_env_reset.restype = C.c_int
        L.lenv_real_env_reset.argtypes = [C.c_int32, vp, vp, C.c_int64, vp, vp, vp, vp]
        L.lenv_real_env_step.restype = C.c_int
        L.lenv_real_env_step.argtypes = [C.c_int32, C.c_int32, C.c_int64, vp, vp, vp, vp, vp, vp, vp]
        L.lenv_ql_rn_inner_loop.restype = C.c_int
        L.lenv_ql_rn_inner_loop.argtypes = [C.POINTER(QlCfg), vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(Tapes), C.c_int64,
                                            C.POINTER(QlOut), vp]
        L.lenv_dueling_se_workspace_bytes.restype = C.c_size_t
        L.lenv_dueling_se_workspace_bytes.argtypes = [C.POINTER(DdqnCfg), C.c_int64]
        L.lenv_dueling_num_params.restype = C.c_int64
        L.lenv_dueling_num_params.argtypes = [C.POINTER(DdqnCfg)]
        L.lenv_dueling_se_inner_loop.restype = C.c_int
        L.lenv_dueling_se_inner_loop.argtypes = L.lenv_ddqn_se_inner_loop.argtypes
        L.lenv_dueling_se_inner_loop_hp.restype = C.c_int
        L.lenv_dueling_se_inner_loop_hp.argtypes = [L.lenv_ddqn_se_inner_loop.argtypes[0], C.POINTER(ChainHp)] + list(L.lenv_ddqn_se_inner_loop.argtypes[1:])
        L.lenv_dueling_agent_init_hp.restype = C.c_int
        L.lenv_dueling_agent_init_hp.argtypes = [C.POINTER(DdqnCfg), C.POINTER(ChainHp), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.lenv_icm_num_params.restype = C.c_int64
        L.lenv_icm_num_params.argtypes = [C.POINTER(DdqnCfg)]
        L.lenv_dueling_se_inner_loop_icm.restype = C.c_int
        L.lenv_dueling_se_inner_loop_icm.argtypes = [L.lenv_ddqn_se_inner_loop.argtypes[0], C.POINTER(ChainHp), C.POINTER(IcmIo)] + list(L.lenv_ddqn_se_inner_loop.argtypes[1:])
        L.lenv_chain_uniform_init.restype = C.c_int
        L.lenv_chain_uniform_init.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lenv_rng_unit.restype = C.c_double
        L.lenv_rng_unit.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
        L.lenv_td3_rn_workspace_bytes.restype = C.c_size_t
        L.lenv_td3_rn_workspace_bytes.argtypes = [C.POINTER(Td3Cfg), C.c_int64]
        L.lenv_td3_num_params.restype = C.c_int64
        L.lenv_td3_num_params.argtypes = [C.POINTER(Td3Cfg), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.lenv_td3_rn_inner_loop.restype = C.c_int
        L.lenv_td3_rn_inner_loop.argtypes = [C.POINTER(Td3Cfg), vp, vp, vp, vp, vp, vp, C.POINTER(Td3Tapes), C.c_int64, vp, C.c_size_t,
                                             C.POINTER(Td3Out), vp]
        L.lenv_td3_rn_inner_loop_hp.restype = C.c_int
        L.lenv_td3_rn_inner_loop_hp.argtypes = [C.POINTER(Td3Cfg), C.POINTER(ChainHp)] + list(L.lenv_td3_rn_inner_loop.argtypes[1:])
        L.lenv_td3_icm_num_params.restype = C.c_int64
        L.lenv_td3_icm_num_params.argtypes = [C.POINTER(Td3Cfg)]
        L.lenv_td3_rn_inner_loop_icm.restype = C.c_int
        L.lenv_td3_rn_inner_loop_icm.argtypes = [C.POINTER(Td3Cfg), C.POINTER(ChainHp), C.POINTER(IcmIo)] + list(L.lenv_td3_rn_inner_loop.argtypes[1:])
        L.lenv_td3_agent_init_hp.restype = C.c_int
        L.lenv_td3_agent_init_hp.argtypes = [C.POINTER(Td3Cfg), C.POINTER(ChainHp), vp, C.c_int64, vp, vp]
        L.lenv_rn_num_params.restype = C.c_int64
        L.lenv_rn_num_params.argtypes = [C.c_int32] * 5
        L.lenv_rn_shape_rows.restype = C.c_int
        L.lenv_rn_shape_rows.argtypes = [C.c_int32, C.POINTER(MlpDesc), C.c_int32, C.c_int32, C.c_double, vp, vp, vp, vp, vp, C.c_int64, vp, vp]
        L.lenv_mlp_forward.restype = C.c_int
        L.lenv_mlp_forward.argtypes = [C.POINTER(MlpDesc), vp, vp, C.c_int64, vp, vp]
        L.lenv_cheetah_standin_reset.restype = C.c_int
        L.lenv_cheetah_standin_reset.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp]
        L.lenv_cheetah_standin_step.restype = C.c_int
        L.lenv_cheetah_standin_step.argtypes = [C.c_int32, C.c_int64, vp, vp, vp, vp, vp, vp, vp]
        L.lenv_cont_env_reset.restype = C.c_int
        L.lenv_cont_env_reset.argtypes = [C.c_int32, vp, vp, C.c_int64, vp, vp, vp, vp]
        L.lenv_cont_env_step.restype = C.c_int
        L.lenv_cont_env_step.argtypes = [C.c_int32, C.c_int32, C.c_int64, vp, vp, vp, vp, vp, vp, vp]
        L.lenv_rn_shape_population.restype = C.c_int
        L.lenv_rn_shape_population.argtypes = [C.POINTER(QlCfg), vp, vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp]
        L.lenv_nes_worker_best_multi.restype = C.c_int
        L.lenv_nes_worker_best_multi.argtypes = [vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, vp, vp]
        L.lenv_nes_worker_best.restype = C.c_int
        L.lenv_nes_worker_best.argtypes = [vp, C.c_int64, C.c_int32, vp, vp]
        L.lenv_nes_rank_update.restype = C.c_int
        L.lenv_nes_rank_update.argtypes = [C.c_int32, vp, vp, C.c_int64, vp, vp, C.c_int64, C.c_double, C.c_int32,
                                           C.c_double, vp, vp]
        L.lenv_nes_draw.restype = C.c_int
        L.lenv_nes_draw.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_int64, C.c_float, vp, C.c_int64, C.c_int32, C.c_int64,
                                    C.c_int64, vp, vp, vp, vp]
        L.lenv_nes_status_fold.restype = C.c_int
        L.lenv_nes_status_fold.argtypes = [vp, C.c_int64, vp, C.c_int64, vp]
        L.lenv_nes_draw_dev.restype = C.c_int
        L.lenv_nes_draw_dev.argtypes = [C.c_uint64, vp, C.c_int64, C.c_int64, C.c_float, vp, C.c_int64, C.c_int32, C.c_int64,
                                        C.c_int64, vp, vp, vp, vp]
        L.lenv_nes_rank_update_keep.restype = C.c_int
        L.lenv_nes_rank_update_keep.argtypes = [C.c_int32, vp, vp, C.c_int64, vp, vp, C.c_int64, C.c_double, C.c_int32,
                                                C.c_double, vp, vp, vp, vp]
        L.lenv_dueling_team_size.restype = C.c_int
        L.lenv_dueling_team_size.argtypes = [C.POINTER(DdqnCfg), C.c_int64]
        L.lenv_td3_rn_team_size.restype = C.c_int
        L.lenv_td3_rn_team_size.argtypes = [C.POINTER(Td3Cfg), C.c_int64]
        L.lenv_td3d_workspace_bytes.restype = C.c_size_t
        L.lenv_td3d_workspace_bytes.argtypes = [C.POINTER(Td3dCfg), C.c_int64]
        L.lenv_td3d_num_params.restype = C.c_int64
        L.lenv_td3d_num_params.argtypes = [C.POINTER(Td3dCfg), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.lenv_td3d_se_num_params.restype = C.c_int64
        L.lenv_td3d_se_num_params.argtypes = [C.POINTER(Td3dCfg)]
        L.lenv_td3d_inner_loop.restype = C.c_int
        L.lenv_td3d_inner_loop.argtypes = [C.POINTER(Td3dCfg), C.POINTER(ChainHp), vp, vp, vp, vp, vp, vp, C.POINTER(Td3dTapes), C.c_int64, vp,
                                           C.c_size_t, C.POINTER(Td3Out), vp]
        L.lenv_td3d_agent_init.restype = C.c_int
        L.lenv_td3d_agent_init.argtypes = [C.POINTER(Td3dCfg), C.POINTER(ChainHp), vp, C.c_int64, vp, vp]
        if L.lenv_abi_version() != 7:
            raise LenvError("liblenv_hip.so ABI version mismatch")
        L.lenv_struct_size.restype = C.c_int64
        L.lenv_struct_size.argtypes = [C.c_int32]
        for which, cls in enumerate(ABI_STRUCTS):     # the ctypes mirrors must have the library's layout
            if L.lenv_struct_size(which) != C.sizeof(cls):
                raise LenvError("ctypes mirror of %s has %d bytes, the library's struct %d" % (cls.__name__, C.sizeof(cls), L.lenv_struct_size(which)))
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().lenv_error_string(rc).decode()
        raise ERRORS.get(rc, LenvError)("%s: %s (%d)" % (what, msg, rc))
