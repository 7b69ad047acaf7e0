"""ctypes binding of libilsx.so (include/ilsx.h).  No CPU fallback: importing the product path without
the built HIP library raises, and creating a context without an MI355X raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libilsx.so")

c_f32p = C.POINTER(C.c_float)
c_u8p = C.POINTER(C.c_uint8)
c_i64p = C.POINTER(C.c_int64)
vp = C.c_void_p


class MlpCfg(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("n_hidden", C.c_int32), ("hidden", C.c_int32),
                ("out_dim", C.c_int32), ("n_heads", C.c_int32), ("act", C.c_int32), ("hidden_sizes", C.c_int32 * 3)]


class SacCfg(C.Structure):
    _fields_ = [("reward_scale", C.c_float), ("discount", C.c_float), ("policy_lr", C.c_float),
                ("qf_lr", C.c_float), ("alpha_lr", C.c_float), ("soft_target_tau", C.c_float),
                ("alpha", C.c_float), ("train_alpha", C.c_int32),
                ("policy_mean_reg_weight", C.c_float), ("policy_std_reg_weight", C.c_float),
                ("beta_1", C.c_float), ("has_target_entropy", C.c_int32), ("target_entropy", C.c_float),
                ("max_batch", C.c_int32), ("grad_world", C.c_int32)]


class SacStats(C.Structure):
    _fields_ = [("qf1_loss", C.c_float), ("qf2_loss", C.c_float), ("policy_loss", C.c_float),
                ("alpha_loss", C.c_float), ("alpha", C.c_float), ("q1_mean", C.c_float),
                ("q2_mean", C.c_float), ("log_pi_mean", C.c_float), ("policy_mu_mean", C.c_float),
                ("policy_log_std_mean", C.c_float), ("log_alpha", C.c_double),
                ("ext_std", C.c_float * 5), ("ext_max", C.c_float * 5), ("ext_min", C.c_float * 5)]


_MB, _MG = 8, 8


class PlanarModel(C.Structure):  # ilsx_planar_model
    _fields_ = [("task", C.c_int32), ("n_body", C.c_int32), ("n_geom", C.c_int32), ("frame_skip", C.c_int32),
                ("pgs_iters", C.c_int32), ("pad0", C.c_int32),
                ("parent", C.c_int32 * _MB), ("limited", C.c_int32 * _MB), ("geom_body", C.c_int32 * _MG),
                ("anchor", (C.c_double * 2) * _MB), ("com", (C.c_double * 2) * _MB),
                ("mass", C.c_double * _MB), ("inertia", C.c_double * _MB), ("jsign", C.c_double * _MB),
                ("armature", C.c_double * _MB), ("damping", C.c_double * _MB), ("range", (C.c_double * 2) * _MB),
                ("gear", C.c_double * _MB),
                ("geom_p1", (C.c_double * 2) * _MG), ("geom_p2", (C.c_double * 2) * _MG),
                ("geom_radius", C.c_double * _MG), ("geom_friction", C.c_double * _MG),
                ("timestep", C.c_double), ("gravity", C.c_double), ("reset_noise", C.c_double),
                ("contact_margin", C.c_double), ("contact_solref", C.c_double * 2), ("contact_solimp", C.c_double * 3),
                ("limit_solref", C.c_double * 2), ("limit_solimp", C.c_double * 3),
                ("ctrl_cost", C.c_double), ("alive_bonus", C.c_double), ("z_min", C.c_double), ("z_max", C.c_double),
                ("ang_max", C.c_double), ("state_max", C.c_double), ("init_qpos", C.c_double * (_MB + 2)),
                ("stiffness", C.c_double * _MB), ("reset_noise_vel_std", C.c_double), ("qvel_clip", C.c_double),
                ("max_rows", C.c_int32), ("pad1", C.c_int32)]


_ML3, _MC3, _MB3 = 20, 32, 16


class SpatialModel(C.Structure):  # ilsx_spatial_model
    _fields_ = [("task", C.c_int32), ("n_link", C.c_int32), ("n_act", C.c_int32), ("n_contact", C.c_int32),
                ("n_body", C.c_int32), ("frame_skip", C.c_int32), ("pgs_iters", C.c_int32), ("max_rows", C.c_int32),
                ("parent", C.c_int32 * _ML3), ("limited", C.c_int32 * _ML3), ("act_link", C.c_int32 * _ML3),
                ("contact_link", C.c_int32 * _MC3), ("body_link", C.c_int32 * _MB3),
                ("anchor", (C.c_double * 3) * _ML3), ("axis", (C.c_double * 3) * _ML3), ("quat0", (C.c_double * 4) * _ML3),
                ("com", (C.c_double * 3) * _ML3), ("mass", C.c_double * _ML3), ("inertia", (C.c_double * 6) * _ML3),
                ("armature", C.c_double * _ML3), ("damping", C.c_double * _ML3), ("stiffness", C.c_double * _ML3),
                ("range", (C.c_double * 2) * _ML3), ("gear", C.c_double * _ML3),
                ("contact_pos", (C.c_double * 3) * _MC3), ("contact_radius", C.c_double * _MC3),
                ("contact_friction", C.c_double * _MC3),
                ("timestep", C.c_double), ("gravity", C.c_double), ("reset_noise", C.c_double),
                ("reset_noise_vel_std", C.c_double), ("contact_margin", C.c_double), ("ctrl_range", C.c_double),
                ("contact_solref", C.c_double * 2), ("contact_solimp", C.c_double * 3),
                ("limit_solref", C.c_double * 2), ("limit_solimp", C.c_double * 3),
                ("ctrl_cost", C.c_double), ("alive_bonus", C.c_double), ("vel_weight", C.c_double), ("z_min", C.c_double),
                ("z_max", C.c_double), ("init_qpos", C.c_double * (_ML3 + 6))]


class Td3Cfg(C.Structure):  # ilsx_td3_cfg
    _fields_ = [("reward_scale", C.c_float), ("discount", C.c_float), ("policy_lr", C.c_float), ("qf_lr", C.c_float),
                ("policy_and_target_update_period", C.c_int32), ("soft_target_tau", C.c_float),
                ("policy_noise", C.c_float), ("policy_noise_clip", C.c_float), ("max_act", C.c_float),
                ("max_batch", C.c_int32), ("her", C.c_int32), ("clip_return_l", C.c_float), ("clip_return_r", C.c_float)]


class Td3Stats(C.Structure):  # ilsx_td3_stats
    _fields_ = [("qf1_loss", C.c_float), ("qf2_loss", C.c_float), ("policy_loss", C.c_float),
                ("q1_pred", C.c_float * 4), ("q2_pred", C.c_float * 4), ("q_target", C.c_float * 4),
                ("bellman1", C.c_float * 4), ("bellman2", C.c_float * 4), ("policy_action", C.c_float * 4)]


class SacvCfg(C.Structure):  # ilsx_sacv_cfg
    _fields_ = [("reward_scale", C.c_float), ("discount", C.c_float), ("alpha", C.c_float), ("policy_lr", C.c_float),
                ("qf_lr", C.c_float), ("vf_lr", C.c_float), ("soft_target_tau", C.c_float),
                ("policy_mean_reg_weight", C.c_float), ("policy_std_reg_weight", C.c_float), ("beta_1", C.c_float),
                ("max_batch", C.c_int32)]


class SacvStats(C.Structure):  # ilsx_sacv_stats
    _fields_ = [("qf1_loss", C.c_float), ("qf2_loss", C.c_float), ("vf_loss", C.c_float), ("policy_loss", C.c_float),
                ("q1_pred", C.c_float * 4), ("q2_pred", C.c_float * 4), ("v_pred", C.c_float * 4),
                ("log_pi", C.c_float * 4), ("policy_mu", C.c_float * 4), ("policy_log_std", C.c_float * 4)]


class BcCfg(C.Structure):  # ilsx_bc_cfg
    _fields_ = [("mode", C.c_int32), ("lr", C.c_float), ("momentum", C.c_float), ("max_batch", C.c_int32)]


class PpoCfg(C.Structure):  # ilsx_ppo_cfg
    _fields_ = [("obs_dim", C.c_int32), ("act_dim", C.c_int32), ("n_hidden", C.c_int32), ("hidden", C.c_int32),
                ("reward_scale", C.c_float), ("discount", C.c_float), ("clip_eps", C.c_float),
                ("policy_lr", C.c_float), ("value_lr", C.c_float), ("gae_tau", C.c_float),
                ("value_l2_reg", C.c_float), ("mini_batch_size", C.c_int32), ("update_epoch", C.c_int32),
                ("max_samples", C.c_int32), ("use_value_clip", C.c_int32), ("conditioned_std", C.c_int32),
                ("hidden_sizes", C.c_int32 * 3), ("grad_world", C.c_int32)]


class DiscCfg(C.Structure):  # ilsx_disc_cfg
    _fields_ = [("obs_dim", C.c_int32), ("act_dim", C.c_int32), ("hid_dim", C.c_int32), ("hid_act", C.c_int32),
                ("use_grad_pen", C.c_int32), ("clamp_magnitude", C.c_float), ("disc_lr", C.c_float),
                ("disc_momentum", C.c_float), ("grad_pen_weight", C.c_float), ("max_batch", C.c_int32),
                ("state_only", C.c_int32), ("num_layer_blocks", C.c_int32), ("use_bn", C.c_int32), ("grad_world", C.c_int32)]


class OptMeta(C.Structure):  # ilsx_opt_meta
    _fields_ = [("t", C.c_int64), ("rng_step", C.c_uint64), ("n_train_steps", C.c_int64)]


class DiscStats(C.Structure):
    _fields_ = [("ce_loss", C.c_float), ("grad_pen", C.c_float), ("accuracy", C.c_float)]


# name -> (restype, argtypes); every symbol include/ilsx.h declares
PROTOTYPES = {
    "ilsx_sac_group_create": (C.c_int, [vp, C.POINTER(vp), C.c_int, C.POINTER(vp)]),
    "ilsx_sac_group_destroy": (C.c_int, [vp]),
    "ilsx_sac_group_train_from_replay": (C.c_int, [vp, C.POINTER(vp), C.c_int, C.c_int, C.c_int]),
    "ilsx_advirl_train": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                    C.c_int, C.c_float, C.POINTER(DiscStats), C.POINTER(SacStats), vp]),
    "ilsx_net_set_noise_policy": (C.c_int, [vp, C.c_float, C.c_float, C.c_float]),
    "ilsx_net_set_output_linear": (C.c_int, [vp, C.c_int]),
    "ilsx_td3_create": (C.c_int, [vp, C.POINTER(Td3Cfg), vp, vp, vp, C.POINTER(vp)]),
    "ilsx_td3_destroy": (C.c_int, [vp]),
    "ilsx_td3_train_step": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, vp, C.POINTER(Td3Stats)]),
    "ilsx_td3_train_from_replay": (C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(Td3Stats)]),
    "ilsx_td3_get_params": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
    "ilsx_td3_set_params": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
    "ilsx_sacv_create": (C.c_int, [vp, C.POINTER(SacvCfg), vp, vp, vp, vp, C.POINTER(vp)]),
    "ilsx_sacv_destroy": (C.c_int, [vp]),
    "ilsx_sacv_train_step": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, vp, C.POINTER(SacvStats)]),
    "ilsx_sacv_train_from_replay": (C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(SacvStats)]),
    "ilsx_sacv_get_params": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
    "ilsx_sacv_set_params": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
    "ilsx_bc_create": (C.c_int, [vp, C.POINTER(BcCfg), vp, C.POINTER(vp)]),
    "ilsx_bc_destroy": (C.c_int, [vp]),
    "ilsx_bc_train_step": (C.c_int, [vp, vp, vp, C.c_int, vp, C.POINTER(C.c_float)]),
    "ilsx_bc_train_from_replay": (C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "ilsx_ppo_create": (C.c_int, [vp, C.POINTER(PpoCfg), C.POINTER(vp)]),
    "ilsx_ppo_destroy": (C.c_int, [vp]),
    "ilsx_ppo_num_params": (C.c_int, [vp, C.c_int, C.POINTER(C.c_size_t)]),
    "ilsx_ppo_set_params": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
    "ilsx_ppo_get_params": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
    "ilsx_ppo_gae": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp]),
    "ilsx_ppo_values": (C.c_int, [vp, vp, C.c_int, vp]),
    "ilsx_ppo_train": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, vp, vp]),
    "ilsx_rollout_step_relabel": (C.c_int, [vp, vp, vp, C.c_int, vp, C.c_int, C.c_int]),
    "ilsx_eval_rollout": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    "ilsx_ppo_rollout": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "ilsx_vecenv_set_obs_affine": (C.c_int, [vp, vp, vp]),
    "ilsx_is_terminal": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp]),
    "ilsx_vecenv_obs_norm": (C.c_int, [vp, C.c_int, C.c_int]),
    "ilsx_vecenv_get_obs_rms": (C.c_int, [vp, vp, vp, C.POINTER(C.c_double)]),
    "ilsx_vecenv_set_obs_rms": (C.c_int, [vp, vp, vp, C.c_double]),
    "ilsx_ppo_debug_perm": (C.c_int, [vp, C.c_int, C.c_uint32, vp]),
    "ilsx_ppo_debug_grad_norm": (C.c_int, [vp, C.POINTER(C.c_float)]),
    "ilsx_ppo_policy_act": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, vp]),
    "ilsx_disc_create": (C.c_int, [vp, C.POINTER(DiscCfg), C.POINTER(vp)]),
    "ilsx_advirl_set_policy_batch_from_expert": (C.c_int, [vp, C.c_int]),
    "ilsx_her_gather": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp, vp, vp, vp]),
    "ilsx_debug_rng_stream": (C.c_int, [vp, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "ilsx_debug_philox": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_int, C.c_int, vp, vp]),
    "ilsx_disc_destroy": (C.c_int, [vp]),
    "ilsx_disc_num_params": (C.c_int, [vp, C.POINTER(C.c_size_t)]),
    "ilsx_disc_set_params": (C.c_int, [vp, vp, C.c_size_t]),
    "ilsx_disc_get_params": (C.c_int, [vp, vp, C.c_size_t]),
    "ilsx_disc_get_grads": (C.c_int, [vp, vp, C.c_size_t]),
    "ilsx_disc_get_bn_stats": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "ilsx_disc_set_bn_stats": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "ilsx_disc_train_step": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, vp, C.POINTER(DiscStats)]),
    "ilsx_disc_reward": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float, vp, vp]),
    "ilsx_vecenv_create": (C.c_int, [vp, C.POINTER(PlanarModel), C.c_int, C.c_uint64, C.POINTER(vp)]),
    "ilsx_vecenv_create_spatial": (C.c_int, [vp, C.POINTER(SpatialModel), C.c_int, C.c_uint64, C.POINTER(vp)]),
    "ilsx_vecenv_state_dims": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ilsx_vecenv_set_path_mode": (C.c_int, [vp, C.c_int]),
    "ilsx_vecenv_destroy": (C.c_int, [vp]),
    "ilsx_vecenv_dims": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ilsx_vecenv_reset": (C.c_int, [vp, vp, C.c_int, vp]),
    "ilsx_vecenv_step": (C.c_int, [vp, vp, vp, C.c_int, vp, vp, vp]),
    "ilsx_vecenv_get_state": (C.c_int, [vp, vp, vp]),
    "ilsx_vecenv_set_state": (C.c_int, [vp, vp, vp]),
    "ilsx_vecenv_cur_obs": (C.c_int, [vp, C.POINTER(vp)]),
    "ilsx_rollout_step": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ilsx_rollout_step_begin": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ilsx_rollout_step_end": (C.c_int, [vp]),
    "ilsx_eval_rollouts_lockstep": (C.c_int, [C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_double)]),
    "ilsx_rollout_steps_lockstep": (C.c_int, [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_int, C.c_int]),
    "ilsx_rollout_stats": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]),
    "ilsx_abi_version": (C.c_int, []),
    "ilsx_last_error": (C.c_char_p, []),
    "ilsx_ctx_create": (C.c_int, [C.c_int, vp, C.c_uint64, C.POINTER(vp)]),
    "ilsx_ctx_sync": (C.c_int, [vp]),
    "ilsx_ctx_destroy": (C.c_int, [vp]),
    "ilsx_ctx_stream": (vp, [vp]),
    "ilsx_ctx_rng_stream_cursor": (C.c_int, [vp, C.c_uint32, C.POINTER(C.c_uint32)]),
    "ilsx_ctx_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "ilsx_ctx_free": (C.c_int, [vp, vp]),
    "ilsx_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "ilsx_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "ilsx_comm_unique_id": (C.c_int, [vp]),
    "ilsx_comm_init": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "ilsx_comm_destroy": (C.c_int, [vp]),
    "ilsx_comm_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ilsx_comm_allreduce_sum": (C.c_int, [vp, vp, C.c_size_t]),
    "ilsx_td3_get_opt": (C.c_int, [vp, C.c_int, vp, vp, C.c_size_t, C.POINTER(OptMeta)]),
    "ilsx_td3_set_opt": (C.c_int, [vp, C.c_int, vp, vp, C.c_size_t, C.POINTER(OptMeta)]),
    "ilsx_sacv_get_opt": (C.c_int, [vp, C.c_int, vp, vp, C.c_size_t, C.POINTER(OptMeta)]),
    "ilsx_sacv_set_opt": (C.c_int, [vp, C.c_int, vp, vp, C.c_size_t, C.POINTER(OptMeta)]),
    "ilsx_bc_get_opt": (C.c_int, [vp, vp, vp, C.c_size_t, C.POINTER(OptMeta)]),
    "ilsx_bc_set_opt": (C.c_int, [vp, vp, vp, C.c_size_t, C.POINTER(OptMeta)]),
    "ilsx_ppo_get_opt": (C.c_int, [vp, C.c_int, vp, vp, C.c_size_t, C.POINTER(OptMeta)]),
    "ilsx_ppo_set_opt": (C.c_int, [vp, C.c_int, vp, vp, C.c_size_t, C.POINTER(OptMeta)]),
    "ilsx_disc_get_opt": (C.c_int, [vp, vp, vp, C.c_size_t, C.POINTER(OptMeta)]),
    "ilsx_disc_set_opt": (C.c_int, [vp, vp, vp, C.c_size_t, C.POINTER(OptMeta)]),
    "ilsx_prof_enable": (C.c_int, [vp, C.c_int]),
    "ilsx_prof_reset": (C.c_int, [vp]),
    "ilsx_prof_read": (C.c_int, [vp, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "ilsx_prof_kernel": (C.c_char_p, [vp, C.c_int]),
    "ilsx_kernel_name": (C.c_char_p, [C.c_int]),
    "ilsx_debug_set_stamp_buffer": (C.c_int, [vp, vp, C.c_int, C.POINTER(C.c_int)]),
    "ilsx_debug_dw_split": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ilsx_net_create": (C.c_int, [vp, C.POINTER(MlpCfg), C.POINTER(vp)]),
    "ilsx_net_destroy": (C.c_int, [vp]),
    "ilsx_net_num_params": (C.c_int, [vp, C.POINTER(C.c_size_t)]),
    "ilsx_net_init": (C.c_int, [vp, C.c_uint64, C.c_float, C.c_float]),
    "ilsx_net_set_params": (C.c_int, [vp, vp, C.c_size_t, C.c_int]),
    "ilsx_net_get_params": (C.c_int, [vp, vp, C.c_size_t, C.c_int]),
    "ilsx_mlp_forward": (C.c_int, [vp, vp, C.c_int, vp]),
    "ilsx_policy_act": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp, vp]),
    "ilsx_policy_log_prob": (C.c_int, [vp, vp, vp, C.c_int, vp]),
    "ilsx_replay_create": (C.c_int, [vp, C.c_int64, C.c_int, C.c_int, C.c_uint64, C.POINTER(vp)]),
    "ilsx_replay_destroy": (C.c_int, [vp]),
    "ilsx_replay_add": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, vp, C.c_int]),
    "ilsx_replay_terminate_episode": (C.c_int, [vp]),
    "ilsx_replay_set_absorbing": (C.c_int, [vp, C.c_int64, C.c_int, vp]),
    "ilsx_replay_get_absorbing": (C.c_int, [vp, vp, C.c_int, vp]),
    "ilsx_replay_sample": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "ilsx_replay_sample_many": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "ilsx_replay_record_floats": (C.c_int, [vp, C.POINTER(C.c_int)]),
    "ilsx_replay_size": (C.c_int, [vp, c_i64p, c_i64p]),
    "ilsx_replay_clear": (C.c_int, [vp]),
    "ilsx_replay_traj_endpoints": (C.c_int, [vp, c_i64p, c_i64p, C.c_int, C.POINTER(C.c_int)]),
    "ilsx_sac_create": (C.c_int, [vp, C.POINTER(SacCfg), vp, vp, vp, C.POINTER(vp)]),
    "ilsx_sac_destroy": (C.c_int, [vp]),
    "ilsx_sac_train_step": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, C.POINTER(SacStats)]),
    "ilsx_sac_train_from_replay": (C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(SacStats)]),
    "ilsx_sac_set_batch": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, vp, vp]),
    "ilsx_sac_critic_backward": (C.c_int, [vp]),
    "ilsx_sac_critic_update": (C.c_int, [vp]),
    "ilsx_sac_actor_backward": (C.c_int, [vp]),
    "ilsx_sac_actor_update": (C.c_int, [vp]),
    "ilsx_sac_last_stats": (C.c_int, [vp, C.POINTER(SacStats)]),
    "ilsx_sac_get_params": (C.c_int, [vp, C.c_int, vp, C.c_size_t, C.c_int]),
    "ilsx_sac_set_params": (C.c_int, [vp, C.c_int, vp, C.c_size_t, C.c_int]),
    "ilsx_sac_get_grads": (C.c_int, [vp, C.c_int, vp, C.c_size_t, C.c_int]),
    "ilsx_sac_grads_ptr": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]),
    "ilsx_sac_get_log_alpha": (C.c_int, [vp, C.POINTER(C.c_double)]),
    "ilsx_sac_set_log_alpha": (C.c_int, [vp, C.c_double]),
    "ilsx_sac_get_adam": (C.c_int, [vp, C.c_int, vp, vp, C.c_size_t, c_i64p]),
    "ilsx_sac_set_adam": (C.c_int, [vp, C.c_int, vp, vp, C.c_size_t, C.c_int64]),
    "ilsx_sac_get_alpha_opt": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), c_i64p, C.POINTER(C.c_uint64)]),
    "ilsx_sac_debug_batch": (C.c_int, [vp, vp, C.c_uint64, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    "ilsx_sac_debug_last_batch": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp]),
    "ilsx_sac_phase_state": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ilsx_sac_debug_break_phase": (C.c_int, [vp]),
    "ilsx_sac_set_alpha_opt": (C.c_int, [vp, C.c_double, C.c_double, C.c_int64, C.c_uint64]),
}

_lib = None


def load():
    """Load libilsx.so (once).  Raises RuntimeError if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("ILSX_LIB", LIB_PATH)   # ILSX_LIB: a measurement build of the same library (libilsx_stamps.so), never a fallback
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ilswiss_amd/csrc`.  ilswiss_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here == header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.ilsx_abi_version() != 1:
        raise RuntimeError(f"libilsx ABI {lib.ilsx_abi_version()} != 1")
    _lib = lib
    return lib


ILSX_ERR_UNSUPPORTED = -5   # include/ilsx.h


def check(rc):
    if rc != 0:
        msg = load().ilsx_last_error()
        raise RuntimeError(f"libilsx error {rc}: {msg.decode() if msg else '?'}")
