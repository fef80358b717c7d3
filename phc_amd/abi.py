"""Helpers that fill the C-ABI structs of include/phc_amd.h from array objects.

Works for torch tensors (product: device pointers) and for numpy arrays (oracle/hostemu: host
pointers) -- the structs only carry raw addresses and sizes.
"""
import ctypes as C

import numpy as np

from . import _lib as L

MAX_BODIES = 64  # PHC_MAX_BODIES
NUM_INT_TABLES = 21  # PHC_NTAB


def ptr(x):
    """Raw address of a torch tensor / numpy array (must be contiguous), or None."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        assert x.flags["C_CONTIGUOUS"], "numpy array must be C-contiguous"
        return x.ctypes.data
    assert x.is_contiguous(), "tensor must be contiguous"
    return x.data_ptr()


def model_struct(ints, floats, num_bodies, num_dof, max_level, num_contact_pts, num_shapes=1):
    """`num_shapes` > 1: ints / floats are [K, ...] stacks of K packed models of one topology (model.py pack_shapes())."""
    m = L.Model()
    m.num_shapes = int(num_shapes)
    if num_shapes > 1:
        assert ints.ndim == 2 and floats.ndim == 2 and ints.shape[0] == floats.shape[0] == num_shapes
        m.int_stride, m.float_stride = int(ints.shape[1]), int(floats.shape[1])
        m.num_collision_pairs = int(ints[:, 4 + NUM_INT_TABLES * MAX_BODIES].max())
    else:
        m.num_collision_pairs = int(ints.reshape(-1)[4 + NUM_INT_TABLES * MAX_BODIES])   # count stored right after the int tables (model.py pack())
    m.num_bodies, m.num_dof, m.max_level, m.num_contact_pts = num_bodies, num_dof, max_level, num_contact_pts
    per_body = ints.reshape(max(1, int(num_shapes)), -1)[:, 4 + 9 * MAX_BODIES:4 + 10 * MAX_BODIES]   # int table 9: contact points per body
    m.max_body_contact_pts = int(per_body.max())
    m.ints, m.floats = ptr(ints), ptr(floats)
    return m


def motion_lib_struct(frames, frame_stride, num_bodies, motion_lengths, motion_dt, motion_num_frames, length_starts,
                      num_ext_bodies=0, dofs_per_joint=3):
    s = L.MotionLib()
    s.num_ext_bodies, s.dofs_per_joint = int(num_ext_bodies), int(dofs_per_joint)
    s.frames = ptr(frames)
    s.num_frames_total = int(frames.shape[0])
    s.frame_stride = int(frame_stride)
    s.num_bodies = int(num_bodies)
    s.num_motions = int(motion_lengths.shape[0])
    s.motion_lengths, s.motion_dt = ptr(motion_lengths), ptr(motion_dt)
    s.motion_num_frames, s.length_starts = ptr(motion_num_frames), ptr(length_starts)
    return s


def sim_state_struct(num_envs, root_states, dof_state, rigid_body_state, contact_force, dof_force, pd_target, force_sensor=None, env_shape=None, pd_ref=None):
    s = L.SimState()
    s.pd_ref = ptr(pd_ref)
    s.force_sensor = ptr(force_sensor)
    s.env_shape = ptr(env_shape)
    s.num_envs = int(num_envs)
    s.root_states, s.dof_state, s.rigid_body_state = ptr(root_states), ptr(dof_state), ptr(rigid_body_state)
    s.contact_force, s.dof_force, s.pd_target = ptr(contact_force), ptr(dof_force), ptr(pd_target)
    return s


def sim_params_struct(sim_dt=1 / 60, substeps=2, control_freq_inv=2, gravity_z=-9.81, contact_stiffness=1.0e5,
                      contact_damping=1.0e3, friction=1.0, friction_viscous=2.0e3, angular_damping=0.01,
                      max_angular_velocity=100.0, contact_offset=0.02, control_mode=0, limit_stiffness=0.0, limit_damping=0.0, lane_mapping=0,
                      self_collision=0, self_stiffness_scale=0.25, self_damping_ratio=0.5, force_sensor_bodies=(), contact_model=0,
                      contact_iterations=4, contact_impedance=1.0e5, max_depenetration_velocity=10.0, bounce_threshold_velocity=0.2, restitution=0.0,
                      inertia_lag=0, force_average=0):
    p = L.SimParams()
    if isinstance(contact_model, str) and contact_model not in ("penalty", "tgs", "rigid"):
        raise ValueError(f"solver.contact must be 'penalty' or 'tgs' (alias 'rigid'), not {contact_model!r}")
    p.contact_model = {"penalty": 0, "tgs": 1, "rigid": 1}.get(contact_model, contact_model)
    p.inertia_lag, p.force_average = int(bool(inertia_lag)), int(bool(force_average))
    if p.contact_model == 1 and p.inertia_lag:
        raise ValueError("solver.inertia_lag needs the penalty contact model (the rigid model re-solves every sub-step with fresh impedances)")
    if p.contact_model == 1 and int(contact_iterations) < 2:
        raise ValueError("solver.contact_iterations must be >= 2 for the rigid contact model")
    p.contact_iterations, p.contact_impedance = int(contact_iterations), float(contact_impedance)
    p.max_depenetration_velocity, p.bounce_threshold_velocity, p.restitution = float(max_depenetration_velocity), float(bounce_threshold_velocity), float(restitution)
    assert len(force_sensor_bodies) <= 4, "at most 4 force sensors"
    p.num_force_sensors = len(force_sensor_bodies)
    for i, b in enumerate(force_sensor_bodies):
        p.force_sensor_body[i] = int(b)
    p.sim_dt, p.substeps, p.control_freq_inv, p.gravity_z = sim_dt, substeps, control_freq_inv, gravity_z
    p.contact_stiffness, p.contact_damping, p.friction, p.friction_viscous = contact_stiffness, contact_damping, friction, friction_viscous
    p.angular_damping, p.max_angular_velocity, p.contact_offset = angular_damping, max_angular_velocity, contact_offset
    p.control_mode, p.limit_stiffness, p.limit_damping = int(control_mode), float(limit_stiffness), float(limit_damping)
    p.lane_mapping = int(lane_mapping)
    p.self_collision, p.self_stiffness_scale, p.self_damping_ratio = int(self_collision), float(self_stiffness_scale), float(self_damping_ratio)
    return p


def frame_stride_for(num_bodies, num_ext_bodies=0, dofs_per_joint=3):
    """Floats per frame record (layout in include/phc_amd.h): pos 3 | rot 4 per body incl. the extended ones, vel 3 | angvel 3
    per simulated body, then local_rot 4 per body + dof_vel 3 per joint (spherical models) or dof_pos + dof_vel per DoF
    (revolute models); padded to a multiple of 4 floats (16 B) so every record starts float4-aligned."""
    nbe = num_bodies + num_ext_bodies
    n = nbe * 7 + num_bodies * 6 + ((num_bodies - 1) * 2 if dofs_per_joint == 1 else num_bodies * 4 + (num_bodies - 1) * 3)
    return (n + 3) // 4 * 4


def pack_frames(gts, grs, gvs, gavs, lrs, dvs, xp=np, gts_ext=None, grs_ext=None, dof_pos=None):
    """[F,NB,3/4] field tensors -> [F, stride] records (layout of phc_motion_lib_t).  Spherical models pass `lrs` (local
    rotations); revolute models pass `dof_pos` [F,ND] instead (and `dvs` [F,ND]); `gts_ext / grs_ext` [F,E,3/4] are the
    extended reference bodies."""
    F_, nb = gts.shape[0], gts.shape[1]
    ne = 0 if gts_ext is None else gts_ext.shape[1]
    dpj = 1 if dof_pos is not None else 3
    stride = frame_stride_for(nb, ne, dpj)
    if xp is np:
        out = np.zeros((F_, stride), dtype=np.float32)
        cat = lambda a, b: np.concatenate([a, b], axis=1)
    else:
        out = xp.zeros((F_, stride), dtype=xp.float32, device=gts.device)
        cat = lambda a, b: xp.cat([a, b], dim=1)
    pos = gts if ne == 0 else cat(gts, gts_ext)
    rot = grs if ne == 0 else cat(grs, grs_ext)
    o = 0
    fields = [(pos, (nb + ne) * 3), (rot, (nb + ne) * 4), (gvs, nb * 3), (gavs, nb * 3)]
    fields += [(dof_pos, nb - 1), (dvs, nb - 1)] if dpj == 1 else [(lrs, nb * 4), (dvs, (nb - 1) * 3)]
    for arr, w in fields:
        out[:, o:o + w] = arr.reshape(F_, w)
        o += w
    return out


def im_params_struct(dt, max_episode_length, reward_specs, power_reward, power_coefficient, enable_early_termination,
                     use_mean_termination, disable_collision_check, local_root_obs, root_height_obs, num_track_bodies,
                     track_slot, reset_mask, num_reset_bodies, first_reset_body, termination_distances, num_key_bodies, key_body_ids,
                     num_amp_joints, amp_joint_slot, num_amp_obs_steps, num_amp_obs_per_step, num_self_obs, num_task_obs,
                     cycle_motion=False, zero_out_far=False, close_distance=0.25, far_distance=3.0,
                     dofs_per_joint=3, ext_parent=None, ext_offset=None, obs_v=6, self_obs_v=1, num_force_sensors=0, amp_obs_v=1,
                     remove_base_rot=False, self_obs_extra=None, amp_obs_extra=None, zero_out_far_train=False, zero_out_far_steps=90,
                     cycle_motion_xp=False, num_self_obs_hist=0, track_body_reward=False, num_traj_samples=1, traj_sample_timestep=1 / 30, amp_ref_table=None):
    """`self_obs_extra` / `amp_obs_extra`: fp32 [N, E] per-env constant observation columns (shape parameters, limb weights) or None."""
    p = L.ImParams()
    p.remove_base_rot = int(bool(remove_base_rot))
    p.num_self_obs_hist = int(num_self_obs_hist)
    p.track_body_reward = int(bool(track_body_reward))
    p.num_traj_samples, p.traj_sample_timestep = int(num_traj_samples), float(np.float32(traj_sample_timestep))
    p.zero_out_far_train, p.zero_out_far_steps, p.cycle_motion_xp = int(bool(zero_out_far_train)), int(zero_out_far_steps), int(bool(cycle_motion_xp))
    p.num_self_obs_extra = 0 if self_obs_extra is None else int(self_obs_extra.shape[1])
    p.num_amp_obs_extra = 0 if amp_obs_extra is None else int(amp_obs_extra.shape[1])
    p.self_obs_extra, p.amp_obs_extra = ptr(self_obs_extra), ptr(amp_obs_extra)
    p.amp_ref_table = ptr(amp_ref_table)
    p.amp_obs_v = int(amp_obs_v)
    p.obs_v = int(obs_v)
    p.self_obs_v, p.num_force_sensors = int(self_obs_v), int(num_force_sensors)
    p.dofs_per_joint = int(dofs_per_joint)
    p.num_ext_bodies = 0 if ext_parent is None else int(ext_parent.shape[0])
    p.ext_parent, p.ext_offset = ptr(ext_parent), ptr(ext_offset)
    p.dt = float(np.float32(dt))
    p.max_episode_length = int(max_episode_length)
    for k in ("k_pos", "k_rot", "k_vel", "k_ang_vel", "w_pos", "w_rot", "w_vel", "w_ang_vel"):
        setattr(p, k, float(reward_specs[k]))
    p.power_reward, p.power_coefficient = int(bool(power_reward)), float(power_coefficient)
    p.enable_early_termination = int(bool(enable_early_termination))
    p.use_mean_termination = int(bool(use_mean_termination))
    p.disable_collision_check = int(bool(disable_collision_check))
    p.local_root_obs, p.root_height_obs = int(bool(local_root_obs)), int(bool(root_height_obs))
    p.num_track_bodies, p.track_slot = int(num_track_bodies), ptr(track_slot)
    p.reset_mask, p.num_reset_bodies = ptr(reset_mask), int(num_reset_bodies)
    p.first_reset_body = int(first_reset_body)
    p.termination_distances = ptr(termination_distances)
    p.num_key_bodies, p.key_body_ids = int(num_key_bodies), ptr(key_body_ids)
    p.num_amp_joints, p.amp_joint_slot = int(num_amp_joints), ptr(amp_joint_slot)
    p.num_amp_obs_steps, p.num_amp_obs_per_step = int(num_amp_obs_steps), int(num_amp_obs_per_step)
    p.num_self_obs, p.num_task_obs = int(num_self_obs), int(num_task_obs)
    p.cycle_motion, p.zero_out_far = int(bool(cycle_motion)), int(bool(zero_out_far))
    p.close_distance, p.far_distance = float(close_distance), float(far_distance)
    return p


RESET_SUBLISTS = 16  # PHC_RESET_SUBLISTS
RESET_COUNT_STRIDE = 32  # PHC_RESET_COUNT_STRIDE


def reset_sublist_cap(num_envs):
    """Capacity of one device reset sub-list (include/phc_amd.h: phc_im_buffers_t.reset_sublist_cap)."""
    blocks = (num_envs + 7) // 8
    return 8 * ((blocks + RESET_SUBLISTS - 1) // RESET_SUBLISTS)


def im_buffers_struct(progress_buf, reset_buf, terminate_buf, rew_buf, reward_raw, obs_buf, amp_obs_in, amp_obs_out,
                      sampled_motion_ids, motion_start_times, motion_start_times_offset, global_offset,
                      ref_body_pos=None, ref_body_rot=None, ref_body_vel=None, ref_dof_pos=None,
                      cycle_counter=None, recovery_counter=None, point_goal=None, cycle_phase=None, reset_list=None, reset_count=None,
                      reset_slot=0, offset_rand=None, body_state_hist=None, occl_mask=None, amp_env_stride=0, reset_rng_counter=None):
    b = L.ImBuffers()
    b.amp_env_stride = int(amp_env_stride)
    b.reset_rng_counter = ptr(reset_rng_counter)
    b.occl_mask = ptr(occl_mask)
    b.offset_rand = ptr(offset_rand)
    b.body_state_hist = ptr(body_state_hist)
    b.reset_list, b.reset_count, b.reset_slot = ptr(reset_list), ptr(reset_count), int(reset_slot)
    b.reset_sublist_cap = 0 if reset_list is None else int(reset_list.shape[0]) // RESET_SUBLISTS
    b.progress_buf, b.reset_buf, b.terminate_buf = ptr(progress_buf), ptr(reset_buf), ptr(terminate_buf)
    b.rew_buf, b.reward_raw, b.obs_buf = ptr(rew_buf), ptr(reward_raw), ptr(obs_buf)
    # (windows inside longer per-env strips come as non-contiguous views: their first element is what the kernels need)
    first = lambda t: t.ctypes.data if isinstance(t, np.ndarray) else t.data_ptr()
    b.amp_obs_in, b.amp_obs_out = (ptr(t) if amp_env_stride == 0 else first(t) for t in (amp_obs_in, amp_obs_out))
    b.sampled_motion_ids = ptr(sampled_motion_ids)
    b.motion_start_times, b.motion_start_times_offset = ptr(motion_start_times), ptr(motion_start_times_offset)
    b.global_offset = ptr(global_offset)
    b.ref_body_pos, b.ref_body_rot, b.ref_body_vel, b.ref_dof_pos = ptr(ref_body_pos), ptr(ref_body_rot), ptr(ref_body_vel), ptr(ref_dof_pos)
    b.cycle_counter, b.recovery_counter = ptr(cycle_counter), ptr(recovery_counter)
    b.point_goal, b.cycle_phase = ptr(point_goal), ptr(cycle_phase)
    return b


def task_index_tables(model, track_bodies, reset_bodies, key_bodies, amp_remove_names=("L_Hand", "R_Hand", "L_Toe", "R_Toe"),
                      has_dof_subset=True):
    """Per-body index tables the task kernels use (all int32 numpy, length PHC_MAX_BODIES).

    track_slot / reset_mask follow `_build_key_body_ids_tensor` (humanoid.py:1674-1691) on cfg.env.trackBodies /
    reset_bodies; amp_joint_slot follows the dof_subset construction (humanoid.py:388-413)."""
    MB = MAX_BODIES
    names = model.body_names
    track_slot = np.full(MB, -1, dtype=np.int32)
    for s, n in enumerate(track_bodies):
        track_slot[names.index(n)] = s
    reset_mask = np.zeros(MB, dtype=np.int32)
    for n in reset_bodies:
        reset_mask[names.index(n)] = 1
    key_ids = np.array([names.index(n) for n in key_bodies], dtype=np.int32)
    amp_slot = np.full(MB, -1, dtype=np.int32)
    s = 0
    for j in range(1, len(names)):
        if (not has_dof_subset) or names[j] not in amp_remove_names:
            amp_slot[j] = s
            s += 1
    return track_slot, reset_mask, key_ids, amp_slot, s
