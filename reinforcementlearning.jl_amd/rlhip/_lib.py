"""ctypes loader for librlhip.so -- the C-ABI boundary (include/rlhip.h).

There is NO fallback: if the HIP library is missing this module raises at import time, and every
wrapper raises `RLHipError` on a non-zero status.  PyTorch is used by the host layer only for device
memory (`tensor.data_ptr()`), streams (`torch.cuda.current_stream().cuda_stream`) and
`torch.distributed`; no torch type ever crosses this boundary.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
# RLHIP_LIB_PATH: another build of the SAME library, e.g. lib/librlhip_bounds.so (build.py --variant=bounds: gather indices are
# validated inside every call that takes them) -- a debugging switch, not a fallback: the file must exist and export the ABI
LIB_PATH = os.environ.get("RLHIP_LIB_PATH") or os.path.join(_PKG, "lib", "librlhip.so")


class RLHipError(RuntimeError):
    pass


class RLHipTimeoutError(RLHipError, TimeoutError):
    """RLHIP_ETIMEOUT -- a rank never arrived at a gradient exchange (csrc/comm.hip)."""


class RLHipArgumentError(RLHipError, ValueError):
    """RLHIP_EINVAL -- where the reference would throw AssertionError / ArgumentError / MethodError."""


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"librlhip.so not found at {LIB_PATH}. Build it with "
        f"`python reinforcementlearning.jl_amd/build.py` (hipcc, gfx950). "
        "There is no CPU fallback for the rlhip hot path.")

# PyTorch-ROCm bundles its own libamdhip64; import torch FIRST so that librlhip.so binds to the HIP
# runtime instance that owns the tensors and streams it will be handed (two runtimes in one process
# do not share device pointers: launches fail with "no ROCm-capable device is detected").
import torch  # noqa: E402,F401

lib = C.CDLL(LIB_PATH)

i32, i64, u32, u64 = C.c_int32, C.c_int64, C.c_uint32, C.c_uint64
f32, f64, vp = C.c_float, C.c_double, C.c_void_p


class CartPoleCfg(C.Structure):
    _fields_ = [(n, f64) for n in ("gravity", "masscart", "masspole", "halflength", "forcemag", "dt",
                                   "thetathreshold_deg", "xthreshold")] + \
               [("max_steps", i64), ("continuous", i32)]


class PendulumCfg(C.Structure):
    _fields_ = [(n, f64) for n in ("max_speed", "max_torque", "g", "m", "l", "dt")] + \
               [("max_steps", i64), ("continuous", i32), ("n_actions", i32)]


class AcrobotCfg(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("link_length_a", "link_length_b", "link_mass_a", "link_mass_b",
                                          "link_com_pos_a", "link_com_pos_b", "link_moi", "max_torque_noise",
                                          "max_vel_a", "max_vel_b", "g", "dt")] + \
               [("max_steps", C.c_int64), ("nips", C.c_int32)]


class MountainCarCfg(C.Structure):
    _fields_ = [(n, f64) for n in ("min_pos", "max_pos", "max_speed", "goal_pos", "goal_velocity",
                                   "power", "gravity")] + \
               [("max_steps", i64), ("continuous", i32)]


class EnvState(C.Structure):
    _fields_ = [("s", vp * 4), ("t", vp), ("done", vp), ("reward", vp), ("episode", vp)]


class DqnStepArgs(C.Structure):
    """rlhip_dqn_step_args (include/rlhip.h)"""
    _fields_ = [("kind", i32), ("env_cfg", vp), ("st", vp), ("n", i64), ("env_seed", u64), ("env_id_base", u32),
                ("obs", vp), ("last_obs", vp), ("ring", vp), ("layers", i32), ("h", i64), ("na", i64), ("act", i32),
                ("params", vp), ("packed", vp), ("target", vp), ("target_packed", vp), ("m", vp), ("v", vp),
                ("beta_pow", vp), ("lr", f32), ("beta1", f32), ("beta2", f32), ("adam_eps", f32),
                ("max_grad_norm", f32), ("grad_scale", f32), ("eps", f64), ("explorer_seed", u64),
                ("explorer_step", u32), ("batch", i64), ("gamma", f32), ("huber_delta", f32), ("sampler_seed", u64),
                ("draw_ctr", u32), ("do_update", i32), ("do_sync", i32), ("rho", f32), ("workspace", vp), ("grad", vp),
                ("loss", vp), ("gn", vp), ("actions", vp), ("q", vp)]


class Ring(C.Structure):
    _fields_ = [(n, i64) for n in ("capacity", "n_env", "obs_dim", "head_sa", "len_sa", "head_rt",
                                   "len_rt")] + \
               [("elem_bytes", i32), ("layout", i32), ("state", vp), ("action", vp), ("reward", vp), ("terminal", vp)]


class PPOCfg(C.Structure):
    _fields_ = [(n, f32) for n in ("gamma", "lam", "clip_range", "max_grad_norm", "actor_loss_weight",
                                   "critic_loss_weight", "entropy_loss_weight", "lr", "beta1", "beta2",
                                   "adam_eps")] + \
               [(n, i32) for n in ("n_epochs", "n_microbatches", "hidden", "act", "continuous",
                                   "normalize_advantage", "layers")]


class PPOTraj(C.Structure):
    _fields_ = [(n, vp) for n in ("obs", "logp", "value", "reward", "adv", "ret", "action_f",
                                  "action_i", "terminal")]


class CommDesc(C.Structure):
    """rlhip_comm_desc (include/rlhip.h)"""
    _fields_ = [(n, i32) for n in ("rank", "world", "device", "p2p_active", "rccl_active")] + \
               [("seq", u32), ("cap", i64), ("timeout_polls", i64), ("status", vp), ("bufs", vp * 16),
                ("why", C.c_char * 256), ("rccl_path", C.c_char * 256)]


P = C.POINTER

# name -> (restype, argtypes).  Status-returning functions use restype i32 and are error-checked.
_PROTOS = {
    "rlhip_abi_version": (i32, []),
    "rlhip_last_error": (C.c_char_p, []),
    "rlhip_device_count": (i32, [P(i32)]),
    "rlhip_set_device": (i32, [i32]),
    "rlhip_device_name": (i32, [i32, C.c_char_p, i32]),
    "rlhip_malloc": (i32, [P(vp), C.c_size_t]),
    "rlhip_free": (i32, [vp]),
    "rlhip_memset": (i32, [vp, i32, C.c_size_t, vp]),
    "rlhip_memcpy_h2d": (i32, [vp, vp, C.c_size_t, vp]),
    "rlhip_memcpy_d2h": (i32, [vp, vp, C.c_size_t, vp]),
    "rlhip_memcpy_d2d": (i32, [vp, vp, C.c_size_t, vp]),
    "rlhip_stream_create": (i32, [P(vp)]),
    "rlhip_stream_destroy": (i32, [vp]),
    "rlhip_stream_sync": (i32, [vp]),
    "rlhip_event_create": (i32, [P(vp)]),
    "rlhip_event_destroy": (i32, [vp]),
    "rlhip_event_record": (i32, [vp, vp]),
    "rlhip_event_elapsed_ms": (i32, [vp, vp, P(f32)]),
    "rlhip_fill_uniform_f32": (i32, [vp, i64, u64, u32, u32, vp]),
    "rlhip_permutation": (i32, [vp, u32, u64, u32, vp]),
    "rlhip_cartpole_default": (i32, [P(CartPoleCfg)]),
    "rlhip_pendulum_default": (i32, [P(PendulumCfg)]),
    "rlhip_mountaincar_default": (i32, [P(MountainCarCfg), i32]),
    "rlhip_acrobot_default": (i32, [P(AcrobotCfg)]),
    "rlhip_env_packed_episode_capacity": (i64, [i64]),
    "rlhip_env_obs_dim": (i32, [i32]),
    "rlhip_env_state_dim": (i32, [i32]),
    "rlhip_env_reset": (i32, [i32, i32, vp, P(EnvState), i64, u64, u32, vp, vp]),
    "rlhip_env_step": (i32, [i32, i32, vp, P(EnvState), i64, vp, i32, u64, u32, vp, vp, vp]),
    "rlhip_env_obs": (i32, [i32, i32, P(EnvState), i64, vp, vp]),
    "rlhip_discount_rewards_f32": (i32, [vp, vp, i64, i64, f32, vp, vp, i32, vp]),
    "rlhip_discount_rewards_f64": (i32, [vp, vp, i64, i64, f64, vp, vp, i32, vp]),
    "rlhip_discount_rewards_reduced_f32": (i32, [vp, vp, i64, i64, f32, vp, vp, i32, vp]),
    "rlhip_discount_rewards_reduced_f64": (i32, [vp, vp, i64, i64, f64, vp, vp, i32, vp]),
    "rlhip_gae_f32": (i32, [vp, vp, vp, i64, i64, f32, f32, vp, i32, vp]),
    "rlhip_gae_f64": (i32, [vp, vp, vp, i64, i64, f64, f64, vp, i32, vp]),
    "rlhip_gae_returns_f32": (i32, [vp, vp, vp, vp, vp, i64, i64, f32, f32, vp]),
    "rlhip_eps_greedy_select_f32": (i32, [vp, i64, i64, i64, i64, vp, f64, i32, u64, u32, u32, vp, vp]),
    "rlhip_eps_greedy_prob_f32": (i32, [vp, i64, i64, i64, i64, vp, f64, i32, vp, vp]),
    "rlhip_get_eps": (f64, [i32, f64, f64, i64, i64, i64]),
    "rlhip_categorical_sample_f32": (i32, [vp, i64, i64, i64, i64, vp, u64, u32, u32, vp, vp, vp]),
    "rlhip_polyak_f32": (i32, [vp, vp, i64, f32, vp]),
    "rlhip_clip_by_global_norm_f32": (i32, [vp, i64, f32, vp, vp]),
    "rlhip_adam_f32": (i32, [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, vp]),
    "rlhip_clip_adam_f32": (i32, [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, f32, vp, vp]),
    "rlhip_normlogpdf_f32": (i32, [vp, vp, vp, vp, i64, vp]),
    "rlhip_diagnormlogpdf_f32": (i32, [vp, vp, vp, i64, i64, vp, vp]),
    "rlhip_huber_f32": (i32, [vp, vp, i64, f32, vp, vp, vp]),
    "rlhip_td_target_f32": (i32, [vp, i64, i64, i64, i64, vp, vp, f32, vp, vp]),
    "rlhip_ring_init": (i32, [P(Ring), i64, i64, i64, i32, vp, vp, vp, vp]),
    "rlhip_ring_state_bytes": (i64, [i64, i64, i64, i32]),
    "rlhip_ring_layout": (i32, [P(Ring)]),
    "rlhip_ring_push_state": (i32, [P(Ring), vp, vp]),
    "rlhip_ring_push_transition": (i32, [P(Ring), vp, vp, vp, vp, vp]),
    "rlhip_ring_length": (i64, [P(Ring)]),
    "rlhip_ring_sample_indices": (i32, [P(Ring), i64, u64, u32, vp, vp]),
    "rlhip_ring_sample_indices_nstep": (i32, [P(Ring), i64, i32, u64, u32, vp, vp]),
    "rlhip_ring_fold_nstep": (i32, [P(Ring), vp, i64, i32, f32, P(Ring), vp, vp]),
    "rlhip_td_target_n_f32": (i32, [vp, i64, i64, i64, i64, vp, vp, f32, i32, vp, vp]),
    "rlhip_gamma_pow": (f32, [f32, i32]),
    "rlhip_ring_gather_is_frame_major": (i32, [P(Ring)]),
    "rlhip_ring_check_indices": (i32, [P(Ring), vp, i64, P(i64), P(i64), vp]),
    "rlhip_ring_bounds_checked_build": (i32, []),
    "rlhip_ring_gather": (i32, [P(Ring), vp, i64, vp, vp, vp, vp, vp, vp]),
    "rlhip_mlp3_nparams": (i64, [i64, i64, i64]),
    "rlhip_mlp3_packed_elems": (i64, [i64]),
    "rlhip_mlp3_init_f32": (i32, [vp, i64, i64, i64, u64, u32, vp]),
    "rlhip_mlp3_pack_bf16": (i32, [vp, i64, i64, i64, vp, vp]),
    "rlhip_dqn3_plan_f32": (i32, [vp, vp, i64, i64, i64, i32, vp, i64, f64, u64, u32, u32, vp, vp, vp]),
    "rlhip_dqn3_workspace_bytes": (i64, [i64, i64, i64, i64]),
    "rlhip_dqn3_grad_f32": (i32, [P(Ring), i64, i64, i32, vp, vp, vp, vp, i64, vp, f32, f32, u64, u32, vp, vp, vp,
                                  vp, vp]),
    "rlhip_dqn3_grad_w_f32": (i32, [P(Ring), i64, i64, i32, vp, vp, vp, vp, i64, vp, vp, f32, f32, vp, vp, vp, vp, vp]),
    "rlhip_dqn_vec_step_f32": (i32, [vp, vp]),
    "rlhip_p2p_alloc": (i32, [i64, P(vp)]),
    "rlhip_p2p_free": (i32, [vp]),
    "rlhip_p2p_export": (i32, [vp, vp]),
    "rlhip_p2p_import": (i32, [vp, P(vp)]),
    "rlhip_p2p_close": (i32, [vp]),
    "rlhip_p2p_can_access": (i32, [i32]),
    "rlhip_p2p_probe": (i32, [vp, i64, vp]),
    "rlhip_p2p_comm_bytes": (i64, [i64]),
    "rlhip_p2p_allreduce_f32": (i32, [vp, i64, i64, i32, i32, vp, u32, i64, vp, vp]),
    "rlhip_comm_unique_id": (i32, [vp]),
    "rlhip_comm_init": (i32, [i32, i32, vp, i64, P(vp)]),
    "rlhip_comm_export": (i32, [vp, vp, P(i32)]),
    "rlhip_p2p_setup": (i32, [vp, vp, vp, P(i32)]),
    "rlhip_comm_disable_p2p": (i32, [vp, C.c_char_p]),
    "rlhip_allreduce_grads": (i32, [vp, vp, i64, vp]),
    "rlhip_comm_check": (i32, [vp]),
    "rlhip_comm_info": (i32, [vp, P(CommDesc)]),
    "rlhip_comm_set_timeout": (i32, [vp, i64]),
    "rlhip_comm_advance_seq": (i32, [vp, u32]),
    "rlhip_comm_unmap": (i32, [vp]),
    "rlhip_comm_destroy": (i32, [vp]),
    "rlhip_ppo_update_comm_f32": (i32, [i32, P(PPOCfg), i64, i64, P(PPOTraj), vp, vp, vp, vp, u64, u32, vp, vp, vp, vp,
                                        vp]),
    "rlhip_dqn_act_supported": (i32, [i32, i64, i64]),
    "rlhip_dqn_act_f32": (i32, [i32, vp, vp, i64, vp, i64, i64, i32, f64, u64, u32, u64, u32, P(Ring), vp, vp, vp, vp,
                                vp]),
    "rlhip_dqn3_act_supported": (i32, [i32, i64, i64, i64]),
    "rlhip_dqn3_act_f32": (i32, [i32, vp, vp, i64, vp, vp, i64, i64, i32, f64, u64, u32, u64, u32, P(Ring), vp, vp, vp, vp, vp]),
    "rlhip_hook_episode_stats": (i32, [vp, vp, i64, u32, vp, vp, vp, u32, vp, vp]),
    "rlhip_explorer_select_f32": (i32, [i32, vp, i64, i64, i64, i64, vp, i32, u64, u32, u32, vp, vp]),
    "rlhip_ucb_select_f32": (i32, [vp, i64, i64, i64, i64, f64, vp, i64, u64, u32, vp, vp]),
    "rlhip_ring_gather_stacked": (i32, [P(Ring), vp, i64, i32, vp, vp, vp, vp, vp, vp]),
    "rlhip_ring_push_state_maxpool": (i32, [P(Ring), vp, vp, vp]),
    "rlhip_ring_push_transition_maxpool": (i32, [P(Ring), vp, vp, vp, vp, vp, vp]),
    "rlhip_sumtree_nodes": (i64, [i64]),
    "rlhip_sumtree_fill_range": (i32, [vp, i64, i64, i64, f32, vp]),
    "rlhip_sumtree_update": (i32, [vp, i64, vp, vp, i64, vp]),
    "rlhip_sumtree_sample": (i32, [vp, i64, i64, u64, u32, vp, vp, vp]),
    "rlhip_per_priority_f32": (i32, [vp, i64, f32, f32, vp, vp]),
    "rlhip_per_is_weights_f32": (i32, [vp, i64, f32, vp, vp]),
    "rlhip_ring_push_priority": (i32, [P(Ring), vp, f32, vp]),
    "rlhip_ring_sample_prioritized": (i32, [P(Ring), vp, i64, u64, u32, vp, vp, vp, vp]),
    "rlhip_ring_sample_gather_prioritized": (i32, [P(Ring), vp, i64, u64, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "rlhip_ring_update_sample_gather_prioritized": (i32, [P(Ring), vp, vp, vp, i64, i64, u64, u32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "rlhip_mlp2_nparams": (i64, [i64, i64, i64]),
    "rlhip_mlp2_forward_f32": (i32, [vp, i64, i64, i64, i32, vp, i64, vp, vp]),
    "rlhip_mlp2_init_f32": (i32, [vp, i64, i64, i64, u64, u32, vp]),
    "rlhip_ppo_default": (i32, [P(PPOCfg)]),
    "rlhip_ppo_nparams": (i64, [i32, P(PPOCfg)]),
    "rlhip_ppo_plan_f32": (i32, [i32, P(PPOCfg), vp, vp, i64, u64, u32, u32, vp, vp, vp, vp, vp]),
    "rlhip_ppo_rollout_f32": (i32, [i32, vp, P(EnvState), i64, i64, P(PPOCfg), vp, u64, u32, u32,
                                    P(PPOTraj), vp]),
    "rlhip_ppo_gae_f32": (i32, [P(PPOCfg), i64, i64, P(PPOTraj), vp]),
    "rlhip_ppo_workspace_bytes": (i64, [i32, P(PPOCfg), i64, i64]),
    "rlhip_ppo_workspace_init": (i32, [vp, i64, vp]),
    "rlhip_ppo_workspace_release": (i32, [vp]),
    "rlhip_ppo_grad_f32": (i32, [i32, P(PPOCfg), i64, i64, P(PPOTraj), vp, u64, u32, i32, vp, vp, vp,
                                 vp]),
    "rlhip_ppo_grad_fresh_f32": (i32, [i32, P(PPOCfg), i64, i64, P(PPOTraj), vp, u64, u32, i32, vp, vp, vp,
                                 vp]),
    "rlhip_ppo_update_p2p_f32": (i32, [i32, P(PPOCfg), i64, i64, P(PPOTraj), vp, vp, vp, vp, u64, u32, vp, vp, vp, i32,
                                       i32, vp, i64, u32, i64, vp, vp]),
    "rlhip_ppo_apply_f32": (i32, [i32, P(PPOCfg), i64, i64, vp, vp, vp, vp, vp, f32, vp, vp, vp]),
    "rlhip_ppo_push_preact_f32": (i32, [P(PPOTraj), i64, i64, i64, vp, vp, vp, vp, vp, vp]),
    "rlhip_ppo_push_postact_f32": (i32, [P(PPOTraj), i64, i64, vp, vp, vp]),
    "rlhip_categorical_network_f32": (i32, [vp, i64, i64, vp, u64, u32, u32, vp, vp, vp, vp]),
    "rlhip_ppo_update_f32": (i32, [i32, P(PPOCfg), i64, i64, P(PPOTraj), vp, vp, vp, vp, u64, u32, vp,
                                   vp, vp, vp]),
    "rlhip_ppo_rollout_dc_f32": (i32, [i32, vp, P(EnvState), i64, i64, P(PPOCfg), vp, u64, u32, vp, P(PPOTraj),
                                       vp]),
    "rlhip_ppo_grad_dc_f32": (i32, [i32, P(PPOCfg), i64, i64, P(PPOTraj), vp, u64, u32, vp, i32, vp, vp, vp,
                                    vp]),
    "rlhip_ppo_update_dc_f32": (i32, [i32, P(PPOCfg), i64, i64, P(PPOTraj), vp, vp, vp, vp, u64, vp, vp, vp, vp,
                                      vp]),
    "rlhip_counters_advance": (i32, [vp, u32, u32, vp]),
    "rlhip_dqn_workspace_bytes": (i64, [i64, i64, i64, i64]),
    "rlhip_dqn_grad_f32": (i32, [P(Ring), i64, i64, i32, vp, vp, i64, f32, f32, u64, u32, vp, vp, vp,
                                 vp]),
    "rlhip_gaussian_head_sample_f32": (i32, [vp, vp, i64, i64, i64, f32, f32, i32, i32, u64, u32, u32, vp, vp, vp]),
    "rlhip_gaussian_head_logp_f32": (i32, [vp, vp, vp, i64, i64, i64, f32, f32, i32, i32, vp, vp]),
    "rlhip_env_act_push_f32": (i32, [i32, vp, P(EnvState), i64, vp, u64, u32, P(Ring), vp, vp, vp]),
    "rlhip_dqn3_update_f32": (i32, [P(Ring), i64, i64, i32, vp, vp, vp, vp, i64, f32, f32, u64, u32, vp, vp, vp, vp, vp,
                                    vp, f32, f32, f32, f32, f32, f32, vp, vp]),
    "rlhip_dqn_update_f32": (i32, [P(Ring), i64, i64, i32, vp, vp, i64, f32, f32, u64, u32, vp, vp, vp, vp, vp, vp,
                                   f32, f32, f32, f32, f32, f32, vp, vp]),
    "rlhip_dqn_grad_idx_f32": (i32, [P(Ring), i64, i64, i32, vp, vp, i64, vp, f32, f32, vp, vp, vp, vp, vp]),
    "rlhip_dqn_grad_idx_w_f32": (i32, [P(Ring), i64, i64, i32, vp, vp, i64, vp, vp, f32, f32, vp, vp, vp, vp, vp]),
    "rlhip_dqn_plan_f32": (i32, [vp, i64, i64, i64, i32, vp, i64, f64, u64, u32, u32, vp, vp, vp]),
}

_STATUS = set()
for _name, (_res, _args) in _PROTOS.items():
    _f = getattr(lib, _name)  # AttributeError here = the library does not export a declared symbol
    _f.restype = _res
    _f.argtypes = _args
    if _res is i32 and _name not in ("rlhip_abi_version", "rlhip_env_obs_dim", "rlhip_env_state_dim",
                                     "rlhip_ring_gather_is_frame_major", "rlhip_dqn_act_supported", "rlhip_dqn3_act_supported", "rlhip_p2p_can_access",
                                     "rlhip_ring_bounds_checked_build", "rlhip_ring_layout"):
        _STATUS.add(_name)

# this host is written for ABI 2 (record rings, sized PPO workspaces): a stale in-tree library must not be driven with it
EXPECTED_ABI = 2
if lib.rlhip_abi_version() != EXPECTED_ABI:
    raise ImportError(f"{LIB_PATH} reports ABI {lib.rlhip_abi_version()}, this host needs {EXPECTED_ABI}: rebuild it "
                      "(python reinforcementlearning.jl_amd/build.py)")


def last_error():
    return lib.rlhip_last_error().decode("utf-8", "replace")


def call(name, *args):
    """Invoke a status-returning ABI function; raise on error (no silent fallback)."""
    rc = getattr(lib, name)(*args)
    if name in _STATUS and rc != 0:
        msg = f"{name} failed with status {rc}: {last_error()}"
        if rc == -1:
            raise RLHipArgumentError(msg)
        if rc == -4:
            raise RLHipTimeoutError(msg)
        raise RLHipError(msg)
    return rc


def declared_symbols(header_path=None):
    """Names of all functions declared in include/rlhip.h."""
    import re

    header_path = header_path or os.path.join(os.path.dirname(_PKG), "include", "rlhip.h")
    with open(header_path) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rlhip_[a-z0-9_]+)\s*\(", src)))
