"""Synthetic AMASS-shaped reference motions (SURVEY.md section 8d "Synthetic inputs").

There is no AMASS data in the build container or on the GPU box, so the bench
and the tests drive the path with smooth random joint trajectories that have the
*schema* of the reference's motion pkl (written by the reference's
``scripts/data_process/convert_amass_isaac.py:129-138`` and read by
``phc/utils/motion_lib_smpl.py:123-135``):

    key -> {pose_quat_global [T,J,4] xyzw f64, pose_quat [T,J,4], root_trans_offset [T,3] f64,
            pose_aa [T,J*3], trans_orig [T,3], beta [10], gender "neutral", fps 30}

Pure numpy; the generator is deterministic in ``seed``.
"""
import numpy as np
from scipy.ndimage import gaussian_filter1d


def _exp_map_to_quat(e):
    ang = np.linalg.norm(e, axis=-1, keepdims=True)
    small = ang < 1e-8
    axis = np.where(small, np.array([0.0, 0.0, 1.0]), e / np.maximum(ang, 1e-12))
    half = 0.5 * ang
    return np.concatenate([axis * np.sin(half), np.cos(half)], axis=-1)


def _quat_mul(a, b):
    x1, y1, z1, w1 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    x2, y2, z2, w2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
                     w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2,
                     w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], axis=-1)


BASE_ROT = np.array([0.5, 0.5, 0.5, 0.5])   # root rotation of a y-up SMPL asset standing upright (robot.has_upright_start False, humanoid.py:1937)


def make_clip(rng, parents, num_frames, body_names=None, fps=30, base_rot=None):
    """One smooth random clip.  Per-joint exp-map = low-pass filtered random walk
    (step N(0,0.05^2), sigma=3 frames), clipped to +-1 rad (knee y/z +-0.1);
    root height 0.9+-0.05 m; root xy random walk ~1 m/s; random initial yaw.
    `base_rot` (xyzw): rest rotation of the root for assets that are not modelled upright (the root's world rotation is
    yaw * tilt * base_rot)."""
    J = len(parents)
    T = int(num_frames)
    steps = rng.normal(0.0, 0.05, size=(T, J, 3))
    e = gaussian_filter1d(np.cumsum(steps, axis=0), 3, axis=0, mode="nearest")
    e = np.clip(e, -1.0, 1.0)
    if body_names is not None:
        for j, n in enumerate(body_names):
            if n.endswith("Knee"):
                e[:, j, 1:] = np.clip(e[:, j, 1:], -0.1, 0.1)
    # root: mostly upright with a yaw walk
    yaw = rng.uniform(-np.pi, np.pi) + gaussian_filter1d(np.cumsum(rng.normal(0, 0.03, size=T)), 3, mode="nearest")
    e[:, 0, :2] *= 0.15
    e[:, 0, 2] = 0.0
    q_local = _exp_map_to_quat(e)
    yaw_q = np.stack([np.zeros(T), np.zeros(T), np.sin(0.5 * yaw), np.cos(0.5 * yaw)], axis=-1)
    q_local[:, 0] = _quat_mul(yaw_q, q_local[:, 0])
    if base_rot is not None:
        q_local[:, 0] = _quat_mul(q_local[:, 0], np.broadcast_to(np.asarray(base_rot, dtype=np.float64), (T, 4)))
    q_global = np.zeros_like(q_local)
    for j in range(J):
        p = parents[j]
        q_global[:, j] = q_local[:, j] if p < 0 else _quat_mul(q_global[:, p], q_local[:, j])
    q_global /= np.linalg.norm(q_global, axis=-1, keepdims=True)
    vel_xy = gaussian_filter1d(rng.normal(0, 1.0, size=(T, 2)), 5, axis=0, mode="nearest") * 2.0
    trans = np.zeros((T, 3))
    trans[:, :2] = np.cumsum(vel_xy, axis=0) / fps
    trans[:, 2] = 0.9 + np.clip(gaussian_filter1d(rng.normal(0, 0.05, size=T), 5, mode="nearest") * 3, -0.05, 0.05)
    # axis-angle of local rotations (mujoco joint order) -- only carried through as "motion_aa"
    w = np.clip(q_local[..., 3], -1, 1)
    ang = 2 * np.arccos(np.abs(w))
    s = np.sqrt(np.maximum(1 - w * w, 1e-16))
    aa = q_local[..., :3] / s[..., None] * (ang * np.sign(w + 1e-30))[..., None]
    return {
        "pose_quat_global": q_global.astype(np.float64),
        "pose_quat": q_local.astype(np.float64),
        "root_trans_offset": trans.astype(np.float64),
        "trans_orig": trans.astype(np.float64),
        "pose_aa": aa.reshape(T, J * 3).astype(np.float64),
        "beta": np.zeros(10),
        "gender": "neutral",
        "fps": fps,
    }


def make_motion_dict(parents, num_clips, seed=0, min_frames=30, max_frames=1800, mean_seconds=8.0,
                     body_names=None, fps=30, lengths=None, base_rot=None):
    """AMASS-shaped dict of ``num_clips`` clips.  Lengths are log-normal around
    ``mean_seconds`` clipped to [min_frames, max_frames] unless given."""
    rng = np.random.default_rng(seed)
    out = {}
    for i in range(num_clips):
        if lengths is not None:
            T = int(lengths[i])
        else:
            T = int(np.clip(rng.lognormal(np.log(mean_seconds), 0.5) * fps, min_frames, max_frames))
        out[f"synthetic_{i:05d}"] = make_clip(rng, parents, T, body_names=body_names, fps=fps, base_rot=base_rot)
    return out


def make_robot_clip(rng, model, num_frames, num_extend=3, fps=30, root_height=1.0):
    """One smooth random clip in the schema of the reference's retargeted robot pkls (read by
    ``phc/utils/motion_lib_real.py:381-389``): ``pose_aa [T, NB+E, 3]`` = root axis-angle, then one axis*angle row per
    revolute joint, then zero rows for the extended bodies; ``root_trans_offset [T,3]``; ``dof [T,ND]``; ``fps``."""
    T, nd = int(num_frames), model.num_dof
    lo, hi = model.dof_limits()
    mid, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
    walk = gaussian_filter1d(np.cumsum(rng.normal(0.0, 0.05, size=(T, nd)), axis=0), 3, axis=0, mode="nearest")
    dof = mid + np.clip(walk, -1.0, 1.0) * np.minimum(half, 0.8) * 0.6
    pose_aa = np.zeros((T, model.num_bodies + num_extend, 3))
    for i in range(1, model.num_bodies):
        s = model.dof_start[i]
        pose_aa[:, i] = model.dof_axis[s][None] * dof[:, s:s + 1]
    yaw = rng.uniform(-np.pi, np.pi) + gaussian_filter1d(np.cumsum(rng.normal(0, 0.03, size=T)), 3, mode="nearest")
    tilt = gaussian_filter1d(np.cumsum(rng.normal(0, 0.01, size=(T, 2)), axis=0), 3, axis=0, mode="nearest")
    e = np.concatenate([np.clip(tilt, -0.15, 0.15), np.zeros((T, 1))], axis=-1)
    yaw_q = np.stack([np.zeros(T), np.zeros(T), np.sin(0.5 * yaw), np.cos(0.5 * yaw)], axis=-1)
    q_root = _quat_mul(yaw_q, _exp_map_to_quat(e))
    w = np.clip(q_root[..., 3], -1, 1)
    ang = 2 * np.arccos(np.abs(w))
    s_ = np.sqrt(np.maximum(1 - w * w, 1e-16))
    pose_aa[:, 0] = q_root[..., :3] / s_[..., None] * (ang * np.sign(w + 1e-30))[..., None]
    vel_xy = gaussian_filter1d(rng.normal(0, 1.0, size=(T, 2)), 5, axis=0, mode="nearest") * 1.5
    trans = np.zeros((T, 3))
    trans[:, :2] = np.cumsum(vel_xy, axis=0) / fps
    trans[:, 2] = root_height + np.clip(gaussian_filter1d(rng.normal(0, 0.05, size=T), 5, mode="nearest") * 3, -0.04, 0.04)
    return {"pose_aa": pose_aa.astype(np.float32), "root_trans_offset": trans.astype(np.float64), "dof": dof.astype(np.float32), "fps": fps}


def make_robot_stand_clip(model, default_dof_pos, seconds=10.0, fps=30, num_extend=3, arm_swing=0.0, freq=0.4):
    """A physically feasible clip for a revolute-joint robot (round 5: policy-level acceptance of BASELINE configs[4]): the reference's default joint pose
    (`humanoid.py:1121,1181`, robots.py) standing still -- with `arm_swing` > 0 the shoulder-pitch joints swing in antiphase by that many radians at `freq` Hz,
    ramped in over the first second.  Same schema as `make_robot_clip`; the motion library's height fix puts the soles on the ground."""
    T, nd = int(round(seconds * fps)) + 1, model.num_dof
    dof = np.tile(np.asarray(default_dof_pos, np.float64)[None], (T, 1))
    if arm_swing > 0:
        t = np.arange(T) / fps
        ramp = np.clip(t / 1.0, 0.0, 1.0)
        for i, name in enumerate(model.body_names):
            if "shoulder_pitch" in name:
                sgn = 1.0 if name.startswith("left") else -1.0
                dof[:, model.dof_start[i]] += sgn * arm_swing * np.sin(2 * np.pi * freq * t) * ramp
    pose_aa = np.zeros((T, model.num_bodies + num_extend, 3))
    for i in range(1, model.num_bodies):
        s_ = model.dof_start[i]
        pose_aa[:, i] = model.dof_axis[s_][None] * dof[:, s_:s_ + 1]
    trans = np.zeros((T, 3))
    trans[:, 2] = 1.0
    return {"pose_aa": pose_aa.astype(np.float32), "root_trans_offset": trans.astype(np.float64), "dof": dof.astype(np.float32), "fps": fps}


def make_robot_motion_dict(model, num_clips, seed=0, mean_seconds=8.0, fps=30, lengths=None, num_extend=3, min_frames=30):
    """`make_motion_dict` for a revolute-joint robot (H1)."""
    rng = np.random.default_rng(seed)
    out = {}
    for i in range(num_clips):
        if lengths is not None:
            T = int(lengths[i])
        else:
            sec = float(np.clip(rng.lognormal(np.log(mean_seconds), 0.5), 1.0, 60.0))
            T = max(int(round(sec * fps)), min_frames)
        out[f"synthetic_{i:05d}"] = make_robot_clip(rng, model, T, num_extend=num_extend, fps=fps)
    return out


def make_stand_clip(model, seconds=10.0, fps=30):
    """A physically feasible clip for end-to-end sanity runs: the SMPL rest pose standing still, soles on the ground."""
    T, J = int(round(seconds * fps)) + 1, model.num_bodies
    origin = np.zeros((J, 3))
    for j in range(1, J):
        origin[j] = origin[model.parent[j]] + model.local_translation[j]
    lowest = (origin[model.contact_body, 2] + model.contact_pos[:, 2] - model.contact_radius).min()
    q = np.zeros((T, J, 4))
    q[..., 3] = 1.0
    trans = np.zeros((T, 3))
    trans[:, 2] = -lowest + 0.002
    return {"pose_quat_global": q, "pose_quat": q.copy(), "root_trans_offset": trans, "trans_orig": trans.copy(),
            "pose_aa": np.zeros((T, J * 3)), "beta": np.zeros(10), "gender": "neutral", "fps": fps}


def make_armswing_clip(model, seconds=10.0, fps=30, swing=0.6, freq=0.4):
    """A second physically feasible sanity clip: standing on the spot while both arms swing horizontally (shoulder yaw +-`swing` rad
    at `freq` Hz, in antiphase), elbows flexing with them, the torso counter-twisting slightly.  Built from joint rotations through
    the model's own tree (FK of the local rotations), soles on the ground as in `make_stand_clip`."""
    T, J = int(round(seconds * fps)) + 1, model.num_bodies
    names = list(model.body_names)
    t = np.arange(T) / fps
    ph = 2 * np.pi * freq * t
    ramp = np.clip(t / 1.0, 0.0, 1.0)               # start from the rest pose (the reset imposes the clip's first frame)
    e = np.zeros((T, J, 3))
    for side, sgn in (("L", 1.0), ("R", -1.0)):
        e[:, names.index(f"{side}_Shoulder"), 2] = sgn * swing * np.sin(ph) * ramp           # yaw: arm forward / backward
        e[:, names.index(f"{side}_Shoulder"), 0] = -sgn * 0.35 * ramp                        # lowered a little from the T pose
        e[:, names.index(f"{side}_Elbow"), 2] = sgn * 0.4 * (0.5 + 0.5 * np.sin(ph)) * ramp  # elbows flex on the forward swing
    e[:, names.index("Torso"), 2] = -0.08 * np.sin(ph) * ramp
    e[:, names.index("Spine"), 2] = -0.06 * np.sin(ph) * ramp
    q_local = _exp_map_to_quat(e)
    q_global = np.zeros_like(q_local)
    for j in range(J):
        p = model.parent[j]
        q_global[:, j] = q_local[:, j] if p < 0 else _quat_mul(q_global[:, p], q_local[:, j])
    q_global /= np.linalg.norm(q_global, axis=-1, keepdims=True)
    origin = np.zeros((J, 3))
    for j in range(1, J):
        origin[j] = origin[model.parent[j]] + model.local_translation[j]
    lowest = (origin[model.contact_body, 2] + model.contact_pos[:, 2] - model.contact_radius).min()
    trans = np.zeros((T, 3))
    trans[:, 2] = -lowest + 0.002
    w = np.clip(q_local[..., 3], -1, 1)
    ang = 2 * np.arccos(np.abs(w))
    s_ = np.sqrt(np.maximum(1 - w * w, 1e-16))
    aa = q_local[..., :3] / s_[..., None] * (ang * np.sign(w + 1e-30))[..., None]
    return {"pose_quat_global": q_global, "pose_quat": q_local, "root_trans_offset": trans, "trans_orig": trans.copy(),
            "pose_aa": aa.reshape(T, J * 3), "beta": np.zeros(10), "gender": "neutral", "fps": fps}


# ------------------------------------------------------------------------------------------------------------------
# Locomotion-class sanity clips (round 3): squat, step in place, walk.  Built by inverse kinematics of the legs on prescribed pelvis and
# ankle trajectories, so that a stance foot stays where it was put (no sliding), a swing foot leaves the ground and comes back, and -- for
# the walk -- the centre of mass travels.  Kinematic, not dynamically optimised: what a retargeted mocap clip is to the tracker.
def _quat_rotate(q, v):
    qv, w = q[..., :3], q[..., 3:4]
    t = 2.0 * np.cross(qv, v)
    return v + w * t + np.cross(qv, t)


def _fk(model, q_local, root_pos):
    """-> (q_global [T,J,4], origin [T,J,3]) of local rotations q_local [T,J,4] (xyzw) with the root at root_pos [T,3]."""
    T, J = q_local.shape[:2]
    qg, org = np.zeros_like(q_local), np.zeros((T, J, 3))
    for j in range(J):
        p = model.parent[j]
        if p < 0:
            qg[:, j], org[:, j] = q_local[:, j], root_pos
        else:
            qg[:, j] = _quat_mul(qg[:, p], q_local[:, j])
            org[:, j] = org[:, p] + _quat_rotate(qg[:, p], np.broadcast_to(model.local_translation[j], (T, 3)))
    return qg / np.linalg.norm(qg, axis=-1, keepdims=True), org


def _lowest_point(model, qg, org):
    """Height of the lowest contact point of the whole body per frame."""
    pts = org[:, model.contact_body] + _quat_rotate(qg[:, model.contact_body], np.broadcast_to(model.contact_pos, (len(org),) + model.contact_pos.shape))
    return (pts[..., 2] - model.contact_radius).min(axis=1)


def _leg_ik(model, side, pelvis, ankle_target, lateral_tilt):
    """Hip / knee / ankle exp-maps (about y: flexion; about x: the lateral tilt, undone at the ankle) of one leg such that the ankle joint sits
    at `ankle_target` [T,3] (world, sagittal x / z solved, y given by the tilt) for the pelvis origin at `pelvis` [T,3], pelvis upright.
    Newton iterations on the leg's own forward kinematics (the asset's link offsets are not exactly vertical)."""
    names = list(model.body_names)
    hip, knee, ankle = (names.index(f"{side}_{n}") for n in ("Hip", "Knee", "Ankle"))
    T = len(pelvis)

    def ankle_pos(a, b):   # hip flexion a (thigh forward), knee flexion b: rotations about y by -a and +b, after the tilt about x
        e_h = np.stack([lateral_tilt, -a, np.zeros(T)], -1)
        q_h = _exp_map_to_quat(e_h)
        q_k = _quat_mul(q_h, _exp_map_to_quat(np.stack([np.zeros(T), b, np.zeros(T)], -1)))
        p_h = pelvis + model.local_translation[hip]
        p_k = p_h + _quat_rotate(q_h, np.broadcast_to(model.local_translation[knee], (T, 3)))
        return p_k + _quat_rotate(q_k, np.broadcast_to(model.local_translation[ankle], (T, 3)))
    a, b = np.full(T, 0.3), np.full(T, 0.6)
    for _ in range(25):
        r = (ankle_pos(a, b) - ankle_target)[:, [0, 2]]
        h = 1e-5
        ja = ((ankle_pos(a + h, b) - ankle_target)[:, [0, 2]] - r) / h
        jb = ((ankle_pos(a, b + h) - ankle_target)[:, [0, 2]] - r) / h
        det = ja[:, 0] * jb[:, 1] - ja[:, 1] * jb[:, 0]
        det = np.where(np.abs(det) < 1e-9, 1e-9, det)
        da = (r[:, 0] * jb[:, 1] - r[:, 1] * jb[:, 0]) / det
        db = (ja[:, 0] * r[:, 1] - ja[:, 1] * r[:, 0]) / det
        a, b = a - np.clip(da, -0.3, 0.3), np.clip(b - np.clip(db, -0.3, 0.3), 0.02, 2.4)
    assert np.abs((ankle_pos(a, b) - ankle_target)[:, [0, 2]]).max() < 2e-3, "leg IK did not converge (target out of reach?)"
    e = {hip: np.stack([lateral_tilt, -a, np.zeros(T)], -1), knee: np.stack([np.zeros(T), b, np.zeros(T)], -1),
         ankle: np.stack([-lateral_tilt, a - b, np.zeros(T)], -1)}   # the foot stays flat: the ankle undoes thigh + shank pitch and the tilt
    return e


def make_gait_clip(model, kind, seconds=10.0, fps=30, speed=None, period=None, lift=None, sway=None, arm=None, heading=0.0, depth=None):
    """`squat`, `stepinplace` or `walk` for the SMPL humanoid (facing +x, z up).  The first second ramps out of the rest pose (the reset
    imposes a clip frame; the evaluation starts every clip at t = 0).  Round 5: the gait's parameters can be given -- walking `speed` (m/s), cycle
    `period` (s, two steps), foot `lift` (m), lateral `sway` (m), `arm` swing factor, squat `depth` (m) -- and the whole clip turned by `heading`
    (rad about +z): the members of `make_locomotion_library`."""
    assert kind in ("squat", "stepinplace", "walk")
    T, J = int(round(seconds * fps)) + 1, model.num_bodies
    names = list(model.body_names)
    t = np.arange(T) / fps
    ramp = np.clip(t / 1.0, 0.0, 1.0) ** 2 * (3 - 2 * np.clip(t / 1.0, 0.0, 1.0))
    # rest pose: pelvis height with the soles on the ground, ankle joints' rest positions relative to the pelvis
    q0 = np.zeros((1, J, 4)); q0[..., 3] = 1.0
    qg0, org0 = _fk(model, q0, np.zeros((1, 3)))
    h0 = float(-_lowest_point(model, qg0, org0)[0]) + 0.002
    ank = {s: org0[0, names.index(f"{s}_Ankle")] for s in "LR"}
    pelvis = np.zeros((T, 3))
    drop = 0.03 if kind != "walk" else 0.05
    pelvis[:, 2] = h0 - drop                          # knees slightly bent throughout (a straight leg has no IK margin; the reset imposes the clip's frame)
    foot = {s: np.tile(ank[s] + np.array([0.0, 0.0, h0]), (T, 1)) for s in "LR"}
    tilt = np.zeros(T)
    e = np.zeros((T, J, 3))
    if kind == "squat":
        period = 2.5 if period is None else float(period)
        depth = 0.20 if depth is None else float(depth)
        pelvis[:, 2] -= depth * ramp * 0.5 * (1 - np.cos(2 * np.pi * np.clip(t - 1.0, 0, None) / period))
        lean = (h0 - drop - pelvis[:, 2]) / 0.20
        e[:, names.index("Torso"), 1] = 0.20 * lean
        e[:, names.index("Spine"), 1] = 0.15 * lean
        for s, sg in (("L", 1.0), ("R", -1.0)):
            e[:, names.index(f"{s}_Shoulder"), 0] = -sg * 0.9 * ramp
            e[:, names.index(f"{s}_Shoulder"), 1] = -0.5 * lean
    else:
        period = (1.2 if kind == "stepinplace" else 1.0) if period is None else float(period)       # one gait cycle = two steps
        speed = 0.0 if kind == "stepinplace" else (0.7 if speed is None else float(speed))
        lift = (0.10 if kind == "stepinplace" else 0.07) if lift is None else float(lift)
        ph = np.clip(t - 1.0, 0, None) / period               # gait phase in cycles, 0 during the ramp
        go = (t >= 1.0).astype(float)
        def pelvis_x(tt):   # the speed ramps up over the first second of the gait
            tau = np.clip(tt - 1.0, 0, None)
            tr = np.clip(tau, 0, 1)
            return speed * np.where(tau < 1, tr ** 3 - 0.5 * tr ** 4, 0.5 + (tau - 1))
        pelvis[:, 0] = pelvis_x(t)
        sway = (0.045 if kind == "stepinplace" else 0.025) if sway is None else float(sway)
        for s, off in (("L", 0.0), ("R", 0.5)):               # the left foot swings in the first half of a cycle, the right one in the second
            c = ph + off
            k = np.floor(c)
            u = c - k                                         # phase within this foot's own cycle: swing for u < 0.4, stance after
            sw = np.clip(u / 0.4, 0, 1)
            first = 0.0 if off == 0.0 else 1.0                # the cycle in which this foot leaves its rest position
            swing = (u < 0.4) & (go > 0) & (k >= first)
            smooth = sw * sw * (3 - 2 * sw)
            # a foot is planted where the pelvis will be at the middle of the stance that follows; before its first swing it rests at x = 0
            plant = lambda kk: np.where(kk >= first, pelvis_x(1.0 + period * (kk + 0.7 - off)), 0.0)
            x = np.where(go > 0, np.where(swing, plant(k - 1) + (plant(k) - plant(k - 1)) * smooth, plant(k)), 0.0) + ank[s][0]
            foot[s][:, 0] = x
            foot[s][:, 2] = ank[s][2] + h0 + np.where(swing, lift * np.sin(np.pi * sw) ** 2, 0.0)
        # lateral sway towards the stance foot (left foot swings first -> weight on the right, y < 0)
        pelvis[:, 1] = -sway * np.sin(2 * np.pi * (ph + 0.05)) * go * np.clip((t - 1.0) / 0.6, 0, 1)
        tilt = -pelvis[:, 1] / max(h0 - 0.1, 0.5)
        for s, sg in (("L", 1.0), ("R", -1.0)):
            sh = names.index(f"{s}_Shoulder")
            e[:, sh, 0] = -sg * 1.0 * ramp
            e[:, sh, 1] = sg * 0.25 * np.sin(2 * np.pi * ph) * go * ((1.0 if kind == "walk" else 0.4) if arm is None else float(arm))
            e[:, names.index(f"{s}_Elbow"), 2] = sg * 0.3 * ramp
    for s in "LR":
        target = foot[s].copy()
        for j, v in _leg_ik(model, s, pelvis, target, tilt).items():
            e[:, j] = v
    q_local = _exp_map_to_quat(e)
    if heading != 0.0:      # the whole clip turned about +z: root rotation and root path
        yq = np.array([0.0, 0.0, np.sin(0.5 * heading), np.cos(0.5 * heading)])
        q_local[:, 0] = _quat_mul(np.broadcast_to(yq, (T, 4)), q_local[:, 0])
        c, s_h = np.cos(heading), np.sin(heading)
        pelvis = np.stack([c * pelvis[:, 0] - s_h * pelvis[:, 1], s_h * pelvis[:, 0] + c * pelvis[:, 1], pelvis[:, 2]], -1)
    qg, org = _fk(model, q_local, pelvis)
    trans = pelvis.copy()
    trans[:, 2] += 0.002 - _lowest_point(model, qg, org)       # the lowest sole corner on the ground in every frame
    w = np.clip(q_local[..., 3], -1, 1)
    ang = 2 * np.arccos(np.abs(w))
    s_ = np.sqrt(np.maximum(1 - w * w, 1e-16))
    aa = q_local[..., :3] / s_[..., None] * (ang * np.sign(w + 1e-30))[..., None]
    return {"pose_quat_global": qg, "pose_quat": q_local, "root_trans_offset": trans, "trans_orig": trans.copy(),
            "pose_aa": aa.reshape(T, J * 3), "beta": np.zeros(10), "gender": "neutral", "fps": fps}


def make_locomotion_library(model, num_clips=64, seed=0, seconds=8.0, fps=30, with_squats=True):
    """A small multi-clip motion set of physically feasible clips for policy-level acceptance runs (round 5): standing, standing with swinging arms,
    stepping in place and walking at 0.3 .. 0.9 m/s with different cadences, foot lifts, arm swings and headings -- the members are `make_stand_clip`,
    `make_armswing_clip` and parametrised `make_gait_clip`s, drawn with a seeded generator.  Keys sort by kind.  `with_squats`: two squat clips (the
    class no policy of rounds 3-4 learned) ride along at the end."""
    rng = np.random.default_rng(seed)
    out = {}
    n_stand, n_arm = max(1, num_clips // 32), max(2, num_clips // 10)
    n_squat = 2 if (with_squats and num_clips >= 16) else 0
    n_step = max(2, num_clips // 5)
    n_walk = max(1, num_clips - n_stand - n_arm - n_squat - n_step)
    for i in range(n_stand):
        out[f"loco_a_stand_{i:03d}"] = make_stand_clip(model, seconds + 2.0 * i, fps)
    for i in range(n_arm):
        out[f"loco_b_armswing_{i:03d}"] = make_armswing_clip(model, seconds, fps, swing=float(rng.uniform(0.3, 0.8)), freq=float(rng.uniform(0.3, 0.6)))
    for i in range(n_step):
        out[f"loco_c_stepinplace_{i:03d}"] = make_gait_clip(model, "stepinplace", seconds, fps, period=float(rng.uniform(1.0, 1.5)), lift=float(rng.uniform(0.06, 0.12)),
                                                          sway=float(rng.uniform(0.035, 0.05)), arm=float(rng.uniform(0.2, 0.8)), heading=float(rng.uniform(-np.pi, np.pi)))
    for i in range(n_walk):
        while True:   # (stride = speed x period must stay within the legs' reach at the clip's pelvis height: re-draw the few that do not)
            per = float(rng.uniform(0.9, 1.2))
            sp = float(rng.uniform(0.3, min(0.9, 0.74 / per)))
            try:
                out[f"loco_d_walk_{i:03d}"] = make_gait_clip(model, "walk", seconds, fps, speed=sp, period=per, lift=float(rng.uniform(0.05, 0.09)),
                                                            sway=float(rng.uniform(0.02, 0.03)), arm=float(rng.uniform(0.5, 1.2)), heading=float(rng.uniform(-np.pi, np.pi)))
                break
            except AssertionError:
                continue
    for i in range(n_squat):
        out[f"loco_e_squat_{i:03d}"] = make_gait_clip(model, "squat", seconds, fps, period=float(rng.uniform(2.2, 3.0)), depth=float(rng.uniform(0.12, 0.2)))
    return out
