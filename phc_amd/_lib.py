"""ctypes binding of libphc_amd.so (the C ABI in include/phc_amd.h).

This is the only way the Python host side reaches the device code.  There is NO CPU
fallback: if the shared library is missing or a symbol is absent the import fails loudly.
"""
import ctypes as C
import os

# torch bundles its own libamdhip64.so: it must be in the process BEFORE libphc_amd.so is dlopen'ed, otherwise the
# system copy from /opt/rocm gets loaded next to it and kernels are registered with a runtime that owns no device
# (launches then fail with hipErrorNoDevice).  PyTorch is plumbing here: device memory, streams, torch.distributed.
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
# PHC_AMD_LIB: load another build of the SAME library (scripts/sim_phase_profile.py: the instrumented stepper); never a fallback
LIB_PATH = os.environ.get("PHC_AMD_LIB") or os.path.join(HERE, "libphc_amd.so")

c_f = C.c_float
c_i32 = C.c_int32
c_i64 = C.c_int64
c_p = C.c_void_p


class Model(C.Structure):
    _fields_ = [("num_bodies", c_i32), ("num_dof", c_i32), ("max_level", c_i32), ("num_contact_pts", c_i32),
                ("ints", c_p), ("floats", c_p), ("num_collision_pairs", c_i32),
                ("num_shapes", c_i32), ("int_stride", c_i32), ("float_stride", c_i32), ("max_body_contact_pts", c_i32)]


class MotionLib(C.Structure):
    _fields_ = [("frames", c_p), ("num_frames_total", c_i64), ("frame_stride", c_i32), ("num_bodies", c_i32),
                ("num_motions", c_i32), ("motion_lengths", c_p), ("motion_dt", c_p), ("motion_num_frames", c_p),
                ("length_starts", c_p), ("num_ext_bodies", c_i32), ("dofs_per_joint", c_i32)]


class SimState(C.Structure):
    _fields_ = [("num_envs", c_i32), ("root_states", c_p), ("dof_state", c_p), ("rigid_body_state", c_p),
                ("contact_force", c_p), ("dof_force", c_p), ("pd_target", c_p), ("force_sensor", c_p), ("env_shape", c_p), ("pd_ref", c_p)]


class SimParams(C.Structure):
    _fields_ = [("sim_dt", c_f), ("substeps", c_i32), ("control_freq_inv", c_i32), ("gravity_z", c_f),
                ("contact_stiffness", c_f), ("contact_damping", c_f), ("friction", c_f), ("friction_viscous", c_f),
                ("angular_damping", c_f), ("max_angular_velocity", c_f), ("contact_offset", c_f),
                ("control_mode", c_i32), ("limit_stiffness", c_f), ("limit_damping", c_f),
                ("self_collision", c_i32), ("self_stiffness_scale", c_f), ("self_damping_ratio", c_f), ("lane_mapping", c_i32),
                ("num_force_sensors", c_i32), ("force_sensor_body", c_i32 * 4),
                ("contact_model", c_i32), ("contact_iterations", c_i32), ("contact_impedance", c_f), ("max_depenetration_velocity", c_f),
                ("bounce_threshold_velocity", c_f), ("restitution", c_f), ("inertia_lag", c_i32), ("force_average", c_i32)]


class ColsumJob(C.Structure):
    _fields_ = [("partial", c_p), ("out", c_p), ("nchunks", c_i32), ("cols", c_i32), ("accumulate", c_i32)]


class ImParams(C.Structure):
    _fields_ = [("dt", c_f), ("max_episode_length", c_i32),
                ("k_pos", c_f), ("k_rot", c_f), ("k_vel", c_f), ("k_ang_vel", c_f),
                ("w_pos", c_f), ("w_rot", c_f), ("w_vel", c_f), ("w_ang_vel", c_f),
                ("power_reward", c_i32), ("power_coefficient", c_f),
                ("enable_early_termination", c_i32), ("use_mean_termination", c_i32), ("disable_collision_check", c_i32),
                ("local_root_obs", c_i32), ("root_height_obs", c_i32),
                ("num_track_bodies", c_i32), ("track_slot", c_p), ("reset_mask", c_p), ("num_reset_bodies", c_i32), ("first_reset_body", c_i32),
                ("termination_distances", c_p),
                ("num_key_bodies", c_i32), ("key_body_ids", c_p),
                ("num_amp_joints", c_i32), ("amp_joint_slot", c_p),
                ("num_amp_obs_steps", c_i32), ("num_amp_obs_per_step", c_i32),
                ("num_self_obs", c_i32), ("num_task_obs", c_i32),
                ("cycle_motion", c_i32), ("zero_out_far", c_i32), ("close_distance", c_f), ("far_distance", c_f),
                ("dofs_per_joint", c_i32), ("num_ext_bodies", c_i32), ("ext_parent", c_p), ("ext_offset", c_p), ("obs_v", c_i32),
                ("self_obs_v", c_i32), ("num_force_sensors", c_i32), ("amp_obs_v", c_i32),
                ("remove_base_rot", c_i32), ("num_self_obs_extra", c_i32), ("num_amp_obs_extra", c_i32),
                ("num_traj_samples", c_i32), ("traj_sample_timestep", c_f), ("track_body_reward", c_i32), ("num_self_obs_hist", c_i32), ("zero_out_far_train", c_i32), ("zero_out_far_steps", c_i32), ("cycle_motion_xp", c_i32),
                ("self_obs_extra", c_p), ("amp_obs_extra", c_p), ("amp_ref_table", c_p)]


class ImBuffers(C.Structure):
    _fields_ = [("progress_buf", c_p), ("reset_buf", c_p), ("terminate_buf", c_p), ("rew_buf", c_p), ("reward_raw", c_p),
                ("obs_buf", c_p), ("amp_obs_in", c_p), ("amp_obs_out", c_p), ("sampled_motion_ids", c_p),
                ("motion_start_times", c_p), ("motion_start_times_offset", c_p), ("global_offset", c_p),
                ("ref_body_pos", c_p), ("ref_body_rot", c_p), ("ref_body_vel", c_p), ("ref_dof_pos", c_p),
                ("cycle_counter", c_p), ("recovery_counter", c_p), ("point_goal", c_p), ("cycle_phase", c_p),
                ("reset_list", c_p), ("reset_count", c_p), ("reset_slot", c_i32), ("reset_sublist_cap", c_i32), ("body_state_hist", c_p), ("offset_rand", c_p), ("occl_mask", c_p), ("amp_env_stride", C.c_int64), ("reset_rng_counter", c_p)]


class PpoParams(C.Structure):
    _fields_ = [("e_clip", c_f), ("critic_coef", c_f), ("entropy_coef", c_f), ("bounds_loss_coef", c_f), ("clip_value", c_i32)]


P = C.POINTER
_SIGNATURES = {
    "phc_abi_version": ([], c_i32),
    "phc_motion_state": ([P(MotionLib), c_i32] + [c_p] * 15, c_i32),
    "phc_sample_time_interval": ([P(MotionLib), c_i32, c_p, c_p, c_p, c_p], c_i32),
    "phc_sim_step": ([P(Model), P(SimParams), P(SimState), c_p, c_p, c_p, c_p, c_i32, c_p], c_i32),
    "phc_refresh_body_state": ([P(Model), P(SimState), c_p], c_i32),
    "phc_im_post_physics": ([P(Model), P(MotionLib), P(ImParams), P(SimState), P(ImBuffers), c_p], c_i32),
    "phc_amp_ref_table": ([P(Model), P(MotionLib), P(ImParams), c_i64, c_p, c_p, c_p], c_i32),
    "phc_im_reset": ([P(Model), P(MotionLib), P(ImParams), P(SimState), P(ImBuffers), c_i32, c_p, c_p, c_i32, c_p], c_i32),
    "phc_im_reset_done": ([P(Model), P(MotionLib), P(ImParams), P(SimState), P(ImBuffers), C.c_uint64, C.c_uint64, c_i32, c_p], c_i32),
    "phc_im_reset_from_state": ([P(Model), P(MotionLib), P(ImParams), P(SimState), P(ImBuffers), c_i32, c_p, c_i32, c_p], c_i32),
    "phc_refresh_body_state_indexed": ([P(Model), P(SimState), c_i32, c_p, c_p], c_i32),
    "phc_amp_obs_demo": ([P(Model), P(MotionLib), P(ImParams), c_i32, c_p, c_p, c_p, c_p], c_i32),
    "phc_gae": ([c_i32, c_i32, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p], c_i32),
    "phc_fk": ([P(Model), c_i64, c_p, c_p, c_p, c_p, c_p], c_i32),
    "phc_running_norm_workspace": ([c_i64, c_i32], c_i64),
    "phc_running_norm": ([c_p, c_p, c_i64, c_i32, c_p, c_p, c_f, c_f, c_p, c_i32, c_i32, c_p, c_p, c_p, c_p, c_p], c_i32),
    "phc_colsum_workspace": ([c_i64, c_i32], c_i64),
    "phc_colsum_bf16": ([c_p, c_i64, c_i32, c_p, c_p, c_p], c_i32),
    "phc_colsum_relu_bf16": ([c_p, c_p, c_i64, c_i32, c_p, c_p, c_p, c_p], c_i32),
    "phc_sum_slabs_bf16": ([c_p, c_i32, c_i64, c_p, c_i32, c_p], c_i32),
    "phc_split3_bf16": ([c_p, c_i64, c_p, c_i64, c_i64, c_i32, c_i64, c_i32, c_p, c_i32, c_p, c_i64, c_i64, c_i32, c_p], c_i32),
    "phc_colsum_chunks": ([c_i64], c_i32),
    "phc_linear1_chunks": ([c_i64], c_i32),
    "phc_colsum_finish_batch": ([c_i32, P(ColsumJob), c_p], c_i32),
    "phc_rollout_bookkeeping": ([c_p, c_f, c_p, c_p, c_p, c_i32, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p], c_i32),
    "phc_disc_bce": ([c_p, c_i32, c_i32, c_i32, c_f, c_p, c_p, c_p], c_i32),
    "phc_sumsq_workspace": ([], c_i64),
    "phc_weighted_sumsq": ([c_i32, c_p, c_p, c_p, c_i32, c_p, c_p, c_p], c_i32),
    "phc_policy_sample": ([c_p, c_p, c_i32, c_p, c_p, c_p, c_p, c_f, c_p, c_i64, c_i32, c_p, c_p, c_p, c_p, c_p, c_p], c_i32),
    "phc_linear1_workspace": ([c_i64, c_i32], c_i64),
    "phc_linear1_forward": ([c_p, c_p, c_p, c_i64, c_i32, c_p, c_p], c_i32),
    "phc_linear1_backward": ([c_p, c_p, c_p, c_i64, c_i32, c_p, c_p, c_p, c_p], c_i32),
    "phc_adam_workspace": ([], c_i64),
    "phc_adam_clip_step": ([c_p, c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_f, c_f, c_i64, c_f, c_p, c_p, c_p, c_p, c_p], c_i32),
    "phc_ppo_loss_workspace": ([], c_i64),
    "phc_ppo_loss": ([c_p, c_p, c_i32] + [c_p] * 9 + [c_i64, c_i32, P(PpoParams), c_p, c_p, c_p, c_p, c_p], c_i32),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def load():
    """dlopen libphc_amd.so and type every entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build the HIP extension first "
                          "(python -m phc_amd.build or __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    for name, (argtypes, restype) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export the symbol
        fn.argtypes = argtypes
        fn.restype = restype
    if lib.phc_abi_version() != 37:
        raise ImportError("libphc_amd.so ABI version mismatch")
    _lib = lib
    return lib


class PhcError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        kind = {-1: "invalid argument", -2: "unsupported configuration"}.get(rc, f"hipError {rc}")
        raise PhcError(f"{what} failed: {kind}")
