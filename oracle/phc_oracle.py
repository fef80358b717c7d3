"""TEST INFRASTRUCTURE ONLY -- CPU (numpy) restatement of the reference hot path.

This module restates, in plain numpy, the reference's algorithm for the hot
path of SURVEY.md section 8 so that the HIP kernels can be checked on the GPU
box, where ``/root/reference`` does not exist.  Every function cites the
reference file:line it follows.  It is pinned against golden vectors produced
by running the reference's *own* code in the build container
(``oracle/gen_golden.py`` -> ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``phc_amd``) never does.

Conventions: quaternions are xyzw (w last), fp32 everywhere except the
motion-loading FK which the reference runs in fp64 (poselib on float64 pkl data).
"""
import numpy as np

F = np.float32


# --------------------------------------------------------------------------
# R12 -- quaternion primitives (phc/utils/isaacgym_torch_utils.py)
# --------------------------------------------------------------------------
def quat_mul(a, b):
    """isaacgym_torch_utils.py:25-45 (8-multiplication form)."""
    a = np.asarray(a)
    b = np.asarray(b)
    shape = a.shape
    a = a.reshape(-1, 4)
    b = b.reshape(-1, 4)
    x1, y1, z1, w1 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    x2, y2, z2, w2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    ww = (z1 + x1) * (x2 + y2)
    yy = (w1 - y1) * (w2 + z2)
    zz = (w1 + y1) * (w2 - z2)
    xx = ww + yy + zz
    qq = a.dtype.type(0.5) * (xx + (z1 - x1) * (x2 - y2))
    w = qq - ww + (z1 - y1) * (y2 - z2)
    x = qq - xx + (x1 + w1) * (x2 + w2)
    y = qq - yy + (w1 - x1) * (y2 + z2)
    z = qq - zz + (z1 + y1) * (w2 - x2)
    return np.stack([x, y, z, w], axis=-1).reshape(shape)


def quat_conjugate(a):
    """isaacgym_torch_utils.py:90-93."""
    a = np.asarray(a)
    return np.concatenate([-a[..., :3], a[..., 3:]], axis=-1)


def normalize(x, eps=1e-9):
    """isaacgym_torch_utils.py:49-50."""
    n = np.sqrt((x * x).sum(-1, keepdims=True))
    return x / np.maximum(n, x.dtype.type(eps))


def quat_from_angle_axis(angle, axis):
    """isaacgym_torch_utils.py:102-106."""
    theta = (angle / angle.dtype.type(2))[..., None]
    xyz = normalize(axis) * np.sin(theta)
    w = np.cos(theta)
    return normalize(np.concatenate([xyz, w], axis=-1))


def normalize_angle(x):
    """isaacgym_torch_utils.py:110-111."""
    return np.arctan2(np.sin(x), np.cos(x))


def my_quat_rotate(q, v):
    """isaacgym_torch_utils.py:238-247."""
    q_w = q[..., 3:4]
    q_vec = q[..., :3]
    two = q.dtype.type(2.0)
    a = v * (two * q_w * q_w - q.dtype.type(1.0))
    b = np.cross(q_vec, v) * q_w * two
    c = q_vec * (q_vec * v).sum(-1, keepdims=True) * two
    return a + b + c


def quat_to_angle_axis(q):
    """isaacgym_torch_utils.py:250-271."""
    min_theta = 1e-5
    qw = q[..., 3]
    one = q.dtype.type(1)
    with np.errstate(invalid="ignore", divide="ignore"):
        sin_theta = np.sqrt(one - qw * qw)
        angle = q.dtype.type(2) * np.arccos(qw)
        angle = normalize_angle(angle)
        axis = q[..., 0:3] / sin_theta[..., None]
    mask = np.abs(sin_theta) > min_theta
    default_axis = np.zeros_like(axis)
    default_axis[..., -1] = 1
    angle = np.where(mask, angle, np.zeros_like(angle))
    axis = np.where(mask[..., None], axis, default_axis)
    return angle, axis


def quat_to_exp_map(q):
    """isaacgym_torch_utils.py:284-290."""
    angle, axis = quat_to_angle_axis(q)
    return angle[..., None] * axis


def quat_to_tan_norm(q):
    """isaacgym_torch_utils.py:294-306."""
    ref_tan = np.zeros(q.shape[:-1] + (3,), dtype=q.dtype)
    ref_tan[..., 0] = 1
    tan = my_quat_rotate(q, ref_tan)
    ref_norm = np.zeros_like(ref_tan)
    ref_norm[..., -1] = 1
    norm = my_quat_rotate(q, ref_norm)
    return np.concatenate([tan, norm], axis=-1)


def exp_map_to_angle_axis(exp_map):
    """isaacgym_torch_utils.py:342-358."""
    min_theta = 1e-5
    angle = np.sqrt((exp_map * exp_map).sum(-1))
    with np.errstate(invalid="ignore", divide="ignore"):
        axis = exp_map / angle[..., None]
    angle = normalize_angle(angle)
    default_axis = np.zeros_like(exp_map)
    default_axis[..., -1] = 1
    mask = np.abs(angle) > min_theta
    angle = np.where(mask, angle, np.zeros_like(angle))
    axis = np.where(mask[..., None], axis, default_axis)
    return angle, axis


def exp_map_to_quat(exp_map):
    """isaacgym_torch_utils.py:362-365."""
    angle, axis = exp_map_to_angle_axis(exp_map)
    return quat_from_angle_axis(angle, axis)


def slerp(q0, q1, t):
    """isaacgym_torch_utils.py:369-390 (thresholds 0.001 and >=1)."""
    cos_half_theta = (q0 * q1).sum(-1)
    neg = cos_half_theta < 0
    q1 = q1.copy()
    q1[neg] = -q1[neg]
    cos_half_theta = np.abs(cos_half_theta)[..., None]
    one = q0.dtype.type(1.0)
    with np.errstate(invalid="ignore", divide="ignore"):
        half_theta = np.arccos(cos_half_theta)
        sin_half_theta = np.sqrt(one - cos_half_theta * cos_half_theta)
        ratioA = np.sin((one - t) * half_theta) / sin_half_theta
        ratioB = np.sin(t * half_theta) / sin_half_theta
        new_q = ratioA * q0 + ratioB * q1
    half = q0.dtype.type(0.5)
    new_q = np.where(np.abs(sin_half_theta) < 0.001, half * q0 + half * q1, new_q)
    new_q = np.where(np.abs(cos_half_theta) >= 1, q0, new_q)
    return new_q


def calc_heading(q):
    """isaacgym_torch_utils.py:394-405."""
    ref_dir = np.zeros(q.shape[:-1] + (3,), dtype=q.dtype)
    ref_dir[..., 0] = 1
    rot_dir = my_quat_rotate(q, ref_dir)
    return np.arctan2(rot_dir[..., 1], rot_dir[..., 0])


def _z_axis_like(q):
    axis = np.zeros(q.shape[:-1] + (3,), dtype=q.dtype)
    axis[..., 2] = 1
    return axis


def calc_heading_quat(q):
    """isaacgym_torch_utils.py:409-419."""
    return quat_from_angle_axis(calc_heading(q), _z_axis_like(q))


def calc_heading_quat_inv(q):
    """isaacgym_torch_utils.py:423-433."""
    return quat_from_angle_axis(-calc_heading(q), _z_axis_like(q))


# --------------------------------------------------------------------------
# M7/M8/M9 -- motion library lookups (phc/utils/motion_lib_base.py)
# --------------------------------------------------------------------------
def sample_time_interval(phase, motion_len):
    """motion_lib_base.py:414-423 -- ``phase`` is the torch.rand draw (fp32)."""
    curr_fps = F(1 / 30)
    t = (phase.astype(F) * motion_len.astype(F)) / curr_fps
    return t.astype(np.int64).astype(F) * curr_fps


def calc_frame_blend(time, length, num_frames, dt):
    """motion_lib_base.py:549-559.  Indices are the bit-exact part of the contract."""
    time = time.astype(F).copy()
    phase = time / length.astype(F)
    phase = np.clip(phase, F(0.0), F(1.0))
    time[time < 0] = 0
    nfm1 = (num_frames - 1)
    frame_idx0 = (phase * nfm1.astype(F)).astype(np.int64)
    frame_idx1 = np.minimum(frame_idx0 + 1, nfm1)
    blend = np.clip((time - frame_idx0.astype(F) * dt.astype(F)) / dt.astype(F), F(0.0), F(1.0))
    return frame_idx0, frame_idx1, blend


def get_motion_state(lib, motion_ids, motion_times, offset=None):
    """motion_lib_base.py:437-520.  ``lib`` is a dict of the flat frame tensors
    (gts grs lrs gvs gavs dvs, fp32) + per-motion arrays (motion_lengths,
    motion_num_frames, motion_dt, length_starts)."""
    motion_len = lib["motion_lengths"][motion_ids]
    num_frames = lib["motion_num_frames"][motion_ids]
    dt = lib["motion_dt"][motion_ids]
    idx0, idx1, blend = calc_frame_blend(motion_times, motion_len, num_frames, dt)
    f0l = idx0 + lib["length_starts"][motion_ids]
    f1l = idx1 + lib["length_starts"][motion_ids]
    b1 = blend[:, None]
    b2 = b1[:, :, None]
    one = F(1.0)
    rg_pos = (one - b2) * lib["gts"][f0l] + b2 * lib["gts"][f1l]
    if offset is not None:
        rg_pos = rg_pos + offset[:, None, :].astype(F)
    body_vel = (one - b2) * lib["gvs"][f0l] + b2 * lib["gvs"][f1l]
    body_ang_vel = (one - b2) * lib["gavs"][f0l] + b2 * lib["gavs"][f1l]
    dof_vel = (one - b2) * lib["dvs"][f0l] + b2 * lib["dvs"][f1l]
    local_rot = slerp(lib["lrs"][f0l], lib["lrs"][f1l], b2)
    dof_pos = quat_to_exp_map(local_rot[:, 1:]).reshape(len(motion_ids), -1)  # :564-567
    rb_rot = slerp(lib["grs"][f0l], lib["grs"][f1l], b2)
    return {
        "root_pos": rg_pos[:, 0].copy(), "root_rot": rb_rot[:, 0].copy(), "dof_pos": dof_pos,
        "root_vel": body_vel[:, 0].copy(), "root_ang_vel": body_ang_vel[:, 0].copy(),
        "dof_vel": dof_vel.reshape(len(motion_ids), -1),
        "rg_pos": rg_pos, "rb_rot": rb_rot, "body_vel": body_vel, "body_ang_vel": body_ang_vel,
        "f0l": f0l, "f1l": f1l, "blend": blend,
    }


# --------------------------------------------------------------------------
# M5/M6 -- motion loading: poselib FK + finite-difference velocities (fp64)
# --------------------------------------------------------------------------
def _pl_quat_mul(a, b):
    """poselib/poselib/core/rotation3d.py:15-28."""
    x1, y1, z1, w1 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    x2, y2, z2, w2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    w = w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2
    x = w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2
    y = w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2
    z = w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2
    return np.stack([x, y, z, w], axis=-1)


def _pl_quat_normalize(q):
    """rotation3d.py:31-98: positive real part, unit norm."""
    q = np.where(q[..., 3:] < 0, -q, q)
    n = np.sqrt((q * q).sum(-1, keepdims=True))
    return q / np.maximum(n, 1e-9)


def _pl_quat_mul_norm(a, b):
    return _pl_quat_normalize(_pl_quat_mul(a, b))


def _pl_quat_rotate(rot, vec):
    """rotation3d.py:206-211."""
    other = np.concatenate([vec, np.zeros_like(vec[..., :1])], axis=-1)
    return _pl_quat_mul(_pl_quat_mul(rot, other), quat_conjugate(rot))[..., :3]


def _pl_quat_angle_axis(x):
    """rotation3d.py:231-240."""
    s = 2 * (x[..., 3] ** 2) - 1
    angle = np.arccos(np.clip(s, -1, 1))
    axis = x[..., :3]
    axis = axis / np.maximum(np.sqrt((axis * axis).sum(-1, keepdims=True)), 1e-9)
    return angle, axis


def poselib_fk_from_global(parents, local_translation, pose_quat_global, root_trans):
    """SkeletonState.from_rotation_and_root_translation(is_local=False) followed by
    .local_rotation / .global_translation (skeleton3d.py:444-462, 390-426).
    Returns (local_rot [T,J,4], global_trans [T,J,3]) in fp64."""
    g = np.asarray(pose_quat_global, dtype=np.float64)
    T, J, _ = g.shape
    lt = np.asarray(local_translation, dtype=np.float64)
    local_rot = np.zeros_like(g)
    for j in range(J):
        p = parents[j]
        if p == -1:
            local_rot[:, j] = g[:, j]
        else:
            local_rot[:, j] = _pl_quat_mul_norm(quat_conjugate(g[:, p]), g[:, j])
    grot = np.zeros_like(g)
    gpos = np.zeros((T, J, 3))
    for j in range(J):
        p = parents[j]
        if p == -1:
            grot[:, j] = local_rot[:, j]
            gpos[:, j] = np.asarray(root_trans, dtype=np.float64)
        else:
            grot[:, j] = _pl_quat_mul_norm(grot[:, p], local_rot[:, j])  # transform_mul :318-326
            gpos[:, j] = _pl_quat_rotate(grot[:, p], np.broadcast_to(lt[j], (T, 3))) + gpos[:, p]
    return local_rot, gpos


def poselib_velocities(global_trans, global_rot, fps):
    """SkeletonMotion._compute_velocity / _compute_angular_velocity
    (skeleton3d.py:1100-1118): np.gradient + gaussian_filter1d(sigma=2, nearest)."""
    from scipy.ndimage import gaussian_filter1d
    dt = 1 / fps
    vel = np.gradient(global_trans, axis=-3) / dt
    vel = gaussian_filter1d(vel, 2, axis=-3, mode="nearest")
    diff = np.zeros_like(global_rot)
    diff[..., 3] = 1
    diff[:-1] = _pl_quat_mul_norm(global_rot[1:], quat_conjugate(global_rot[:-1]))
    angle, axis = _pl_quat_angle_axis(diff)
    ang_vel = axis * angle[..., None] / dt
    ang_vel = gaussian_filter1d(ang_vel, 2, axis=-3, mode="nearest")
    return vel, ang_vel


def compute_motion_dof_vels(local_rot, fps):
    """motion_lib_base.py:47-70 (fp64 in, like the reference's fp64 local_rotation)."""
    dt = 1.0 / fps
    q0 = local_rot[:-1]
    q1 = local_rot[1:]
    diff = quat_mul(quat_conjugate(q0), q1)
    angle, axis = quat_to_angle_axis(diff)
    dv = axis * angle[..., None] / dt
    dv = dv[:, 1:, :]
    return np.concatenate([dv, dv[-1:]], axis=0)


def load_motion_clip(parents, local_translation, clip):
    """MotionLibSMPL.load_motion_with_skeleton (motion_lib_smpl.py:101-180) without the
    random heading (flags.im_eval / flags.test path) and without mesh height fix
    (data/smpl absent -> mesh_parsers None, :66-68,148-151)."""
    g = np.asarray(clip["pose_quat_global"], dtype=np.float64)
    trans = np.asarray(clip["root_trans_offset"], dtype=np.float64)
    fps = clip.get("fps", 30)
    lrs, gts = poselib_fk_from_global(parents, local_translation, g, trans)
    gvs, gavs = poselib_velocities(gts, g, fps)
    dvs = compute_motion_dof_vels(lrs, fps)
    return {"gts": gts.astype(F), "grs": g.astype(F), "lrs": lrs.astype(F), "gvs": gvs.astype(F),
            "gavs": gavs.astype(F), "dvs": dvs.astype(F), "fps": fps, "num_frames": g.shape[0]}


def build_motion_lib(parents, local_translation, clips):
    """MotionLibBase.load_motions concatenation (motion_lib_base.py:257-319) for an
    ordered list of clips (one per env)."""
    per = [load_motion_clip(parents, local_translation, c) for c in clips]
    lib = {k: np.concatenate([p[k] for p in per], axis=0) for k in ("gts", "grs", "lrs", "gvs", "gavs", "dvs")}
    nf = np.array([p["num_frames"] for p in per], dtype=np.int64)
    fps = np.array([p["fps"] for p in per], dtype=F)
    lib["motion_num_frames"] = nf
    lib["motion_fps"] = fps
    lib["motion_dt"] = np.array([1.0 / p["fps"] for p in per], dtype=F)
    lib["motion_lengths"] = np.array([1.0 / p["fps"] * (p["num_frames"] - 1) for p in per], dtype=F)
    shifted = np.roll(nf, 1)
    shifted[0] = 0
    lib["length_starts"] = np.cumsum(shifted)
    return lib


# --------------------------------------------------------------------------
# R1/R2/R5 -- reward and reset (phc/env/tasks/humanoid_im.py)
# --------------------------------------------------------------------------
DEFAULT_REWARD_SPECS = {"k_pos": 100, "k_rot": 10, "k_vel": 0.1, "k_ang_vel": 0.1,
                        "w_pos": 0.5, "w_rot": 0.3, "w_vel": 0.1, "w_ang_vel": 0.1}  # humanoid_im.py:57


def compute_imitation_reward(body_pos, body_rot, body_vel, body_ang_vel,
                             ref_body_pos, ref_body_rot, ref_body_vel, ref_body_ang_vel, specs=DEFAULT_REWARD_SPECS):
    """humanoid_im.py:1524-1554."""
    s = {k: F(v) for k, v in specs.items()}
    d = ref_body_pos - body_pos
    r_pos = np.exp(-s["k_pos"] * (d * d).mean(-1).mean(-1))
    dq = quat_mul(ref_body_rot, quat_conjugate(body_rot))
    ang = quat_to_angle_axis(dq)[0]
    r_rot = np.exp(-s["k_rot"] * (ang * ang).mean(-1))
    d = ref_body_vel - body_vel
    r_vel = np.exp(-s["k_vel"] * (d * d).mean(-1).mean(-1))
    d = ref_body_ang_vel - body_ang_vel
    r_ang = np.exp(-s["k_ang_vel"] * (d * d).mean(-1).mean(-1))
    reward = s["w_pos"] * r_pos + s["w_rot"] * r_rot + s["w_vel"] * r_vel + s["w_ang_vel"] * r_ang
    return reward.astype(F), np.stack([r_pos, r_rot, r_vel, r_ang], axis=-1).astype(F)


def power_reward(dof_force, dof_vel, progress_buf, coef=0.0005):
    """humanoid_im.py:939-946."""
    power = np.abs(dof_force * dof_vel).sum(-1)
    pr = -F(coef) * power
    pr[progress_buf <= 3] = 0
    return pr.astype(F)


def compute_humanoid_im_reset(progress_buf, body_pos_sub, ref_body_pos_sub, pass_time, termination_distance,
                              enable_early_termination=True, use_mean=False, disable_collision=False):
    """humanoid_im.py:1581-1608 (reset_buf/contact args are unused there)."""
    terminated = np.zeros(progress_buf.shape, dtype=np.int64)
    if enable_early_termination:
        dist = np.sqrt(((body_pos_sub - ref_body_pos_sub) ** 2).sum(-1))
        if use_mean:
            fallen = (dist.mean(-1, keepdims=True) > termination_distance[..., 0:1]).any(-1)
        else:
            fallen = (dist > termination_distance).any(-1)
        fallen = fallen & (progress_buf > 1)
        if disable_collision:
            fallen[:] = False
        terminated = np.where(fallen, 1, terminated)
    reset = np.where(pass_time, 1, terminated)
    return reset.astype(np.int64), terminated.astype(np.int64)


# --------------------------------------------------------------------------
# R6/R7/R9 -- observations
# --------------------------------------------------------------------------
def compute_humanoid_observations_smpl_max(body_pos, body_rot, body_vel, body_ang_vel,
                                           local_root_obs=True, root_height_obs=True):
    """humanoid.py:1995-2050 with upright=True, no shape / limb-weight params."""
    N, J, _ = body_pos.shape
    root_pos = body_pos[:, 0]
    root_rot = body_rot[:, 0]
    root_h = root_pos[:, 2:3]
    hinv = calc_heading_quat_inv(root_rot)
    hinv_e = np.repeat(hinv[:, None], J, axis=1).reshape(-1, 4)
    lp = my_quat_rotate(hinv_e, (body_pos - root_pos[:, None]).reshape(-1, 3)).reshape(N, J * 3)[:, 3:]
    lr = quat_to_tan_norm(quat_mul(hinv_e, body_rot.reshape(-1, 4))).reshape(N, J * 6)
    if not local_root_obs:
        lr[:, 0:6] = quat_to_tan_norm(root_rot)
    lv = my_quat_rotate(hinv_e, body_vel.reshape(-1, 3)).reshape(N, J * 3)
    lav = my_quat_rotate(hinv_e, body_ang_vel.reshape(-1, 3)).reshape(N, J * 3)
    parts = ([root_h] if root_height_obs else []) + [lp, lr, lv, lav]
    return np.concatenate(parts, axis=-1).astype(F)


def compute_imitation_observations_v6(root_pos, root_rot, body_pos, body_rot, body_vel, body_ang_vel,
                                      ref_body_pos, ref_body_rot, ref_body_vel, ref_body_ang_vel, time_steps=1):
    """humanoid_im.py:1309-1358 with upright=True."""
    B, J, _ = body_pos.shape
    T = time_steps
    hinv = calc_heading_quat_inv(root_rot)
    h = calc_heading_quat(root_rot)
    hinv_e = np.repeat(np.repeat(hinv[:, None], J, axis=1), T, axis=0).reshape(-1, 4)
    h_e = np.repeat(np.repeat(h[:, None], J, axis=1), T, axis=0).reshape(-1, 4)
    dpos = ref_body_pos.reshape(B, T, J, 3) - body_pos.reshape(B, 1, J, 3)
    dpos_l = my_quat_rotate(hinv_e, dpos.reshape(-1, 3))
    drot = quat_mul(ref_body_rot.reshape(B, T, J, 4), quat_conjugate(np.repeat(body_rot[:, None], T, axis=1)))
    drot_l = quat_mul(quat_mul(hinv_e, drot.reshape(-1, 4)), h_e)
    dvel = ref_body_vel.reshape(B, T, J, 3) - body_vel.reshape(B, 1, J, 3)
    dvel_l = my_quat_rotate(hinv_e, dvel.reshape(-1, 3))
    dav = ref_body_ang_vel.reshape(B, T, J, 3) - body_ang_vel.reshape(B, 1, J, 3)
    dav_l = my_quat_rotate(hinv_e, dav.reshape(-1, 3))
    lref = ref_body_pos.reshape(B, T, J, 3) - root_pos.reshape(B, 1, 1, 3)
    lref = my_quat_rotate(hinv_e, lref.reshape(-1, 3))
    lrefrot = quat_to_tan_norm(quat_mul(hinv_e, ref_body_rot.reshape(-1, 4)))
    obs = [dpos_l.reshape(B, T, -1), quat_to_tan_norm(drot_l).reshape(B, T, -1), dvel_l.reshape(B, T, -1),
           dav_l.reshape(B, T, -1), lref.reshape(B, T, -1), lrefrot.reshape(B, T, -1)]
    return np.concatenate(obs, axis=-1).reshape(B, -1).astype(F)


def remove_base_rot(quat):
    """humanoid.py:1936-1939."""
    base = np.array([[-0.5, -0.5, -0.5, 0.5]], dtype=quat.dtype)   # quat_conjugate((0.5, 0.5, 0.5, 0.5))
    return quat_mul(quat, np.repeat(base, quat.shape[0], axis=0))


def build_amp_observations_smpl(root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel, key_body_pos,
                                dof_subset, local_root_obs=True, root_height_obs=True, upright=True, shape_params=None, limb_weight_params=None):
    """humanoid_amp.py:967-1011 with has_dof_subset=True; shape_params / limb_weight_params: the has_shape_obs_disc / has_limb_weight_obs
    columns (:1005-1008)."""
    N = root_pos.shape[0]
    root_h = root_pos[:, 2:3]
    if not upright:
        root_rot = remove_base_rot(root_rot)
    hinv = calc_heading_quat_inv(root_rot)
    root_rot_obs = quat_mul(hinv, root_rot) if local_root_obs else root_rot
    root_rot_obs = quat_to_tan_norm(root_rot_obs)
    lrv = my_quat_rotate(hinv, root_vel)
    lrav = my_quat_rotate(hinv, root_ang_vel)
    K = key_body_pos.shape[1]
    lk = key_body_pos - root_pos[:, None]
    hinv_e = np.repeat(hinv[:, None], K, axis=1).reshape(-1, 4)
    lk = my_quat_rotate(hinv_e, lk.reshape(-1, 3)).reshape(N, K * 3)
    dv = dof_vel[:, dof_subset]
    dp = dof_pos[:, dof_subset]
    dof_obs = quat_to_tan_norm(exp_map_to_quat(dp.reshape(-1, 3))).reshape(N, -1)  # humanoid.py:1756-1765
    parts = ([root_h] if root_height_obs else []) + [root_rot_obs, lrv, lrav, dof_obs, dv, lk]
    parts += [p for p in (shape_params, limb_weight_params) if p is not None]
    return np.concatenate(parts, axis=-1).astype(F)


# --------------------------------------------------------------------------
# A1 -- PD action offset / scale (humanoid.py:1331-1409), 3-DoF SMPL joints
# --------------------------------------------------------------------------
def build_pd_action_offset_scale_smpl(lim_low, lim_high, dof_names):
    lim_low = np.array(lim_low, dtype=np.float32).copy()
    lim_high = np.array(lim_high, dtype=np.float32).copy()
    nj = len(dof_names)
    for j in range(nj):
        lo = np.max(np.abs(lim_low[3 * j:3 * j + 3]))
        hi = np.max(np.abs(lim_high[3 * j:3 * j + 3]))
        s = min([1.2 * max([lo, hi]), np.pi])
        lim_low[3 * j:3 * j + 3] = -s
        lim_high[3 * j:3 * j + 3] = s
    offset = 0.5 * (lim_high + lim_low)
    scale = 0.5 * (lim_high - lim_low)
    scale[dof_names.index("L_Knee") * 3 + 1] = 5
    scale[dof_names.index("R_Knee") * 3 + 1] = 5
    return offset.astype(F), scale.astype(F)


# --------------------------------------------------------------------------
# P5 -- GAE (phc/learning/common_agent.py:493-505)
# --------------------------------------------------------------------------
def discount_values(mb_fdones, mb_values, mb_rewards, mb_next_values, gamma=0.99, tau=0.95):
    lastgaelam = 0
    mb_advs = np.zeros_like(mb_rewards)
    for t in reversed(range(mb_rewards.shape[0])):
        not_done = 1.0 - mb_fdones[t]
        not_done = not_done[..., None] if not_done.ndim < mb_rewards[t].ndim else not_done
        delta = mb_rewards[t] + gamma * mb_next_values[t] - mb_values[t]
        lastgaelam = delta + gamma * tau * not_done * lastgaelam
        mb_advs[t] = lastgaelam
    return mb_advs


# --------------------------------------------------------------------------
# Robot path (Unitree H1 / G1): MotionLibReal lookup, extended bodies, robot AMP observation, explicit `pd` torque
# --------------------------------------------------------------------------
def get_motion_state_robot(lib, motion_ids, motion_times, offset=None):
    """motion_lib_real.py:236-361 (`"dof_pos" in self.__dict__` branch): joint angles and rates blend linearly, the
    (NB + E)-wide `gts_t / grs_t` carry the extended bodies.  ``lib``: flat frame tensors gts grs gvs gavs dvs dof_pos
    gts_t grs_t (fp32) + the per-motion arrays of get_motion_state."""
    motion_len = lib["motion_lengths"][motion_ids]
    num_frames = lib["motion_num_frames"][motion_ids]
    dt = lib["motion_dt"][motion_ids]
    idx0, idx1, blend = calc_frame_blend(motion_times, motion_len, num_frames, dt)
    f0l = idx0 + lib["length_starts"][motion_ids]
    f1l = idx1 + lib["length_starts"][motion_ids]
    b1 = blend[:, None]
    b2 = b1[:, :, None]
    one = F(1.0)
    off = F(0) if offset is None else offset[:, None, :].astype(F)
    rg_pos = (one - b2) * lib["gts"][f0l] + b2 * lib["gts"][f1l] + off
    rg_pos_t = (one - b2) * lib["gts_t"][f0l] + b2 * lib["gts_t"][f1l] + off
    body_vel = (one - b2) * lib["gvs"][f0l] + b2 * lib["gvs"][f1l]
    body_ang_vel = (one - b2) * lib["gavs"][f0l] + b2 * lib["gavs"][f1l]
    dof_vel = (one - b1) * lib["dvs"][f0l] + b1 * lib["dvs"][f1l]
    dof_pos = (one - b1) * lib["dof_pos"][f0l] + b1 * lib["dof_pos"][f1l]
    rb_rot = slerp(lib["grs"][f0l], lib["grs"][f1l], b2)
    rg_rot_t = slerp(lib["grs_t"][f0l], lib["grs_t"][f1l], b2)
    return {"root_pos": rg_pos[:, 0].copy(), "root_rot": rb_rot[:, 0].copy(), "dof_pos": dof_pos.astype(F), "dof_vel": dof_vel.astype(F),
            "root_vel": body_vel[:, 0].copy(), "root_ang_vel": body_ang_vel[:, 0].copy(),
            "rg_pos": rg_pos, "rb_rot": rb_rot, "body_vel": body_vel, "body_ang_vel": body_ang_vel,
            "rg_pos_t": rg_pos_t, "rg_rot_t": rg_rot_t, "f0l": f0l, "f1l": f1l, "blend": blend}


def extend_bodies(body_pos, body_rot, ext_parent, ext_pos):
    """humanoid_im.py:917-919: the extended bodies (hands / head) ride on their parent link; appended behind the NB simulated bodies."""
    ext_parent = np.asarray(ext_parent, np.int64)
    N, E = body_pos.shape[0], len(ext_parent)
    pr = body_rot[:, ext_parent].reshape(-1, 4)
    ep = np.broadcast_to(np.asarray(ext_pos, F)[None], (N, E, 3)).reshape(-1, 3)
    cur = my_quat_rotate(pr, ep).reshape(N, E, 3) + body_pos[:, ext_parent]
    return np.concatenate([body_pos, cur], 1).astype(F), np.concatenate([body_rot, body_rot[:, ext_parent]], 1).astype(F)


def build_amp_observations_robot(root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel, key_body_pos,
                                 local_root_obs=True, root_height_obs=True, upright=True):
    """humanoid_amp.py:1063-1104 without shape / limb-weight columns: the joint angles themselves are the dof observation (:1091)."""
    N = root_pos.shape[0]
    root_h = root_pos[:, 2:3]
    if not upright:
        root_rot = remove_base_rot(root_rot)
    hinv = calc_heading_quat_inv(root_rot)
    root_rot_obs = quat_to_tan_norm(quat_mul(hinv, root_rot) if local_root_obs else root_rot)
    lrv = my_quat_rotate(hinv, root_vel)
    lrav = my_quat_rotate(hinv, root_ang_vel)
    K = key_body_pos.shape[1]
    lk = key_body_pos - root_pos[:, None]
    hinv_e = np.repeat(hinv[:, None], K, axis=1).reshape(-1, 4)
    lk = my_quat_rotate(hinv_e, lk.reshape(-1, 3)).reshape(N, K * 3)
    parts = ([root_h] if root_height_obs else []) + [root_rot_obs, lrv, lrav, dof_pos, dof_vel, lk]
    return np.concatenate(parts, axis=-1).astype(F)


def compute_torques_pd(actions, dof_pos, dof_vel, p_gains, d_gains, default_dof_pos, torque_limits, action_scale=1.0):
    """humanoid.py:1575-1599, control_type "P": clip(p (a s + q0 - q) - d qd, +-limit) in fp32, torch's operation order."""
    a = actions.astype(F) * F(action_scale)
    t = p_gains.astype(F) * (a + default_dof_pos.astype(F) - dof_pos.astype(F)) - d_gains.astype(F) * dof_vel.astype(F)
    lim = np.broadcast_to(np.asarray(torque_limits, F), t.shape)
    return np.clip(t, -lim, lim).astype(F)
