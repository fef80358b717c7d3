"""Reference-motion library: host-side mirror of the reference's `MotionLibBase` / `MotionLibSMPL`
(phc/utils/motion_lib_base.py, phc/utils/motion_lib_smpl.py) backed by the HIP lookup kernels.

Same public surface (names, argument meaning, returned dict keys) so code written against the
reference's motion lib keeps working:

    lib = MotionLibSMPL(cfg)                   # cfg.motion_file / device / min_length / im_eval ...
    lib.load_motions(skeleton_trees=..., gender_betas=..., limb_weights=..., random_sample=True, start_idx=0)
    res = lib.get_motion_state(motion_ids, motion_times, offset=None)   # dict of tensors
    t   = lib.sample_time_interval(motion_ids)

What is different inside (MI355X-first):
  * the eight per-field frame tensors (gts grs lrs gvs gavs dvs ...) are stored as ONE buffer of
    contiguous per-frame records in HBM (`frames[F, stride]`, layout in include/phc_amd.h), so a
    lookup touches one contiguous run per frame; `lib.gts` etc. are strided views of that buffer;
  * lookups (frame index arithmetic, lerp / slerp, exp-map) run in `phc_motion_state`
    (phc_amd/csrc/phc_kernels.hip) -- one lane per body;
  * clip loading (FK + finite-difference velocities, fp64 like poselib) is vectorised numpy over
    frames; the random heading is applied per env *after* the per-clip FK (rotation about z commutes
    with FK, np.gradient and the gaussian filter), so a clip shared by many envs is processed once.
"""
import os
from enum import Enum

import joblib
import numpy as np
import torch
from scipy.ndimage import gaussian_filter1d

from . import _lib as L
from . import abi
from .utils.flags import flags


class FixHeightMode(Enum):  # motion_lib_base.py:27-30
    no_fix = 0
    full_fix = 1
    ankle_fix = 2


def _q_mul(a, b):
    x1, y1, z1, w1 = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    x2, y2, z2, w2 = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 + y1 * w2 + z1 * x2 - x1 * z2,
                     w1 * z2 + z1 * w2 + x1 * y2 - y1 * x2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], axis=-1)


def _q_conj(a):
    return a * np.array([-1.0, -1.0, -1.0, 1.0])


def _q_pos_unit(q):
    """poselib quat_normalize: positive real part, unit norm (rotation3d.py:31-98)."""
    q = np.where(q[..., 3:] < 0, -q, q)
    return q / np.maximum(np.linalg.norm(q, axis=-1, keepdims=True), 1e-9)


def _q_rot(q, v):
    qv = q[..., :3]
    t = 2.0 * np.cross(qv, v)
    return v + q[..., 3:] * t + np.cross(qv, t)


def process_clip(parents, local_translation, pose_quat_global, root_trans, fps):
    """FK + velocities of one clip, fp64 (poselib semantics: skeleton3d.py:390-462,1100-1118;
    motion_lib_base.py:47-70).  Returns dict of [T,...] float64 arrays."""
    g = np.asarray(pose_quat_global, dtype=np.float64)
    T, J, _ = g.shape
    lt = np.asarray(local_translation, dtype=np.float64)
    par = np.asarray(parents)
    has_par = par >= 0
    # local rotations from the given global ones
    lrs = g.copy()
    lrs[:, has_par] = _q_pos_unit(_q_mul(_q_conj(g[:, par[has_par]]), g[:, has_par]))
    # FK with the normalised local rotations
    grot = np.zeros_like(g)
    gts = np.zeros((T, J, 3))
    for j in range(J):
        p = par[j]
        if p < 0:
            grot[:, j] = lrs[:, j]
            gts[:, j] = np.asarray(root_trans, dtype=np.float64)
        else:
            grot[:, j] = _q_pos_unit(_q_mul(grot[:, p], lrs[:, j]))
            gts[:, j] = _q_rot(grot[:, p], lt[j][None]) + gts[:, p]
    dt = 1.0 / fps
    gvs = gaussian_filter1d(np.gradient(gts, axis=0) / dt, 2, axis=0, mode="nearest")
    dq = np.zeros_like(g)
    dq[..., 3] = 1.0
    dq[:-1] = _q_pos_unit(_q_mul(g[1:], _q_conj(g[:-1])))
    ang = np.arccos(np.clip(2 * dq[..., 3] ** 2 - 1, -1, 1))
    ax = dq[..., :3] / np.maximum(np.linalg.norm(dq[..., :3], axis=-1, keepdims=True), 1e-9)
    gavs = gaussian_filter1d(ax * ang[..., None] / dt, 2, axis=0, mode="nearest")
    # dof velocities: axis*angle of conj(q_t) * q_{t+1} (isaacgym quat_to_angle_axis semantics), last frame repeated
    d = _q_mul(_q_conj(lrs[:-1]), lrs[1:])
    w = d[..., 3]
    sin_t = np.sqrt(np.maximum(1 - w * w, 0.0))
    angle = 2 * np.arccos(np.clip(w, -1, 1))
    angle = np.arctan2(np.sin(angle), np.cos(angle))
    mask = sin_t > 1e-5
    axis = np.where(mask[..., None], d[..., :3] / np.where(mask, sin_t, 1.0)[..., None], np.array([0.0, 0.0, 1.0]))
    angle = np.where(mask, angle, 0.0)
    dv = (axis * angle[..., None] / dt)[:, 1:]
    dvs = np.concatenate([dv, dv[-1:]], axis=0)
    return dict(gts=gts, grs=g, lrs=lrs, gvs=gvs, gavs=gavs, dvs=dvs)


def apply_heading(clip, yaw):
    """Rotate a processed clip about +z by `yaw` (motion_lib_smpl.py:137-146 applied after FK)."""
    c, s = np.cos(0.5 * yaw), np.sin(0.5 * yaw)
    qh = np.array([0.0, 0.0, s, c])
    out = dict(clip)
    for k in ("gts", "gvs", "gavs"):
        out[k] = _q_rot(qh, clip[k])
    out["grs"] = _q_mul(np.broadcast_to(qh, clip["grs"].shape), clip["grs"])
    lrs = clip["lrs"].copy()
    lrs[:, 0] = out["grs"][:, 0]
    out["lrs"] = lrs
    return out


def _aa_to_quat_xyzw(aa):
    """pytorch3d axis_angle_to_quaternion (rotation_conversions.py:468-500), returned as xyzw."""
    ang = np.linalg.norm(aa, axis=-1, keepdims=True)
    half = 0.5 * ang
    small = np.abs(ang) < 1e-6
    k = np.where(small, 0.5 - ang * ang / 48.0, np.sin(half) / np.where(small, 1.0, ang))
    return np.concatenate([aa * k, np.cos(half)], axis=-1)


def _quat_xyzw_to_mat(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    two_s = 2.0 / (q * q).sum(-1)
    m = np.stack([1 - two_s * (y * y + z * z), two_s * (x * y - z * w), two_s * (x * z + y * w),
                  two_s * (x * y + z * w), 1 - two_s * (x * x + z * z), two_s * (y * z - x * w),
                  two_s * (x * z - y * w), two_s * (y * z + x * w), 1 - two_s * (x * x + y * y)], axis=-1)
    return m.reshape(q.shape[:-1] + (3, 3))


def _mat_to_quat_xyzw(m):
    """rotation_conversions.matrix_to_quaternion (:106-153): of the four candidates the best-conditioned one is taken AS IS
    (its own component positive, no sign standardisation); returned as xyzw (`wxyz_to_xyzw`, :12-13)."""
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = [m[..., i, j] for i in range(3) for j in range(3)]
    q_abs = np.sqrt(np.maximum(np.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1), 0.0))
    cand = np.stack([np.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], -1),
                     np.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], -1),
                     np.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], -1),
                     np.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], -1)], -2)
    cand = cand / (2.0 * np.maximum(q_abs[..., None], 0.1))
    best = np.take_along_axis(cand, q_abs.argmax(-1)[..., None, None], axis=-2)[..., 0, :]
    return best[..., [1, 2, 3, 0]]


def _angular_velocity(g, dt):
    """poselib finite-difference angular velocity of a [T,J,4] xyzw rotation track (skeleton3d.py:1100-1118 /
    torch_humanoid_batch.py:270-279): axis*angle of r_{t+1} * r_t^-1 over dt, gaussian sigma 2."""
    dq = np.zeros_like(g)
    dq[..., 3] = 1.0
    dq[:-1] = _q_pos_unit(_q_mul(g[1:], _q_conj(g[:-1])))
    ang = np.arccos(np.clip(2 * dq[..., 3] ** 2 - 1, -1, 1))
    ax = dq[..., :3] / np.maximum(np.linalg.norm(dq[..., :3], axis=-1, keepdims=True), 1e-9)
    return gaussian_filter1d(ax * ang[..., None] / dt, 2, axis=0, mode="nearest")


def robot_fk(parents, local_translation, local_rotation_wxyz, ext_parent, ext_pos, ext_rot_wxyz, pose_aa, root_trans):
    """`Humanoid_Batch.forward_kinematics_batch` (torch_humanoid_batch.py:224-257) in fp64 numpy: world position / rotation
    matrix of the NB simulated + E extended bodies; also returns the per-body pose quaternions (xyzw)."""
    par = list(np.asarray(parents)) + list(np.asarray(ext_parent))
    off = np.concatenate([np.asarray(local_translation, np.float64), np.asarray(ext_pos, np.float64).reshape(-1, 3)], axis=0)
    rest = np.concatenate([np.asarray(local_rotation_wxyz, np.float64), np.asarray(ext_rot_wxyz, np.float64).reshape(-1, 4)], axis=0)
    rest_m = _quat_xyzw_to_mat(rest[:, [1, 2, 3, 0]])
    J = len(par)
    pose = np.asarray(pose_aa, np.float64)[:, :J]
    T = pose.shape[0]
    pq = _aa_to_quat_xyzw(pose)
    pm = _quat_xyzw_to_mat(pq)
    wpos = np.zeros((T, J, 3))
    wmat = np.zeros((T, J, 3, 3))
    for i in range(J):
        if par[i] < 0:
            wpos[:, i] = np.asarray(root_trans, np.float64)
            wmat[:, i] = pm[:, 0]
        else:
            wpos[:, i] = wmat[:, par[i]] @ off[i] + wpos[:, par[i]]
            wmat[:, i] = wmat[:, par[i]] @ (rest_m[i] @ pm[:, i])   # :248: parent * rest rotation * joint rotation
    return wpos, wmat, pq, pose


def process_clip_real(parents, local_translation, local_rotation_wxyz, ext_parent, ext_pos, ext_rot_wxyz, pose_aa, root_trans, fps):
    """Robot clips (H1 / G1): `Humanoid_Batch.fk_batch(..., return_full=True)` (torch_humanoid_batch.py:163-257) in fp64 numpy.
    pose_aa [T, NB+E, 3]: root axis-angle, one axis*angle row per joint, (zero) rows for the E extended bodies.
    Returns NB-wide gts/grs/gvs/gavs, (NB+E)-wide gts_t/grs_t, dof_pos / dvs [T,ND], lrs [T,NB+E,4]."""
    nb = len(parents)
    wpos, wmat, pq, pose = robot_fk(parents, local_translation, local_rotation_wxyz, ext_parent, ext_pos, ext_rot_wxyz, pose_aa, root_trans)
    wrot = _mat_to_quat_xyzw(wmat)
    dt = 1.0 / fps
    vel = lambda p: gaussian_filter1d(np.gradient(p, axis=0) / dt, 2, axis=0, mode="nearest")
    dof_pos = pose.sum(-1)[:, 1:nb]                                # :211 "you can sum it up since each joint has 1 dof"
    dv = (dof_pos[1:] - dof_pos[:-1]) / dt
    dvs = np.concatenate([dv, dv[-2:-1]], axis=0)                  # :218 appends the SECOND-to-last difference (kept as is)
    return dict(gts=wpos[:, :nb], grs=wrot[:, :nb], gvs=vel(wpos[:, :nb]), gavs=_angular_velocity(wrot[:, :nb], dt),
                gts_t=wpos, grs_t=wrot, gvs_t=vel(wpos), gavs_t=_angular_velocity(wrot, dt), dof_pos=dof_pos, dvs=dvs, lrs=pq)


# ---- clip workers (load_motions pass 2) ------------------------------------------------------------------------------------------------
# FK + finite-difference velocities of one clip crop are numpy fp64 on the host cores; a library draw for thousands of envs (BASELINE
# configs[2]: ~5 800 distinct clips for 8 192 envs, 17 ms each) spreads them over a process pool.  The pool's workers come from a FORKSERVER
# (round 4; it was `fork`): load_motions runs -- and `resample_motions()` re-runs it in the middle of training -- in a process that holds a HIP
# context, torch's thread pools and, with several ranks, RCCL's proxy / watchdog threads; forking THAT process copies locked mutexes and a device
# context into the children (the classic multi-GPU hang).  The fork server is a fresh interpreter started before any of this matters for it: it
# never touches the device, workers forked from it inherit nothing of the caller.  What a worker needs is therefore PICKLED: per worker once
# the constants of the clip family (`_clip_consts`: skeleton arrays / robot model arrays -- no motion data), per job the cropped clip arrays.
_WORKER_CONSTS = None


def _set_worker_consts(consts):
    global _WORKER_CONSTS
    _WORKER_CONSTS = consts


def _run_clip_job(payload, consts=None):
    """One distinct (clip crop, skeleton) pair -> (packed fp32 frame records [T, stride], fps, T).  Pure numpy."""
    c = _WORKER_CONSTS if consts is None else consts
    if c["kind"] == "smpl":
        t, g, trans, fps = payload
        parents, local_translation = c["trees"][t]
        proc = process_clip(parents, local_translation, g, trans, fps)
        f = {k: proc[k].astype(np.float32) for k in ("gts", "grs", "gvs", "gavs", "lrs", "dvs")}
        return abi.pack_frames(f["gts"], f["grs"], f["gvs"], f["gavs"], f["lrs"], f["dvs"]), fps, g.shape[0]
    t, pose_aa, trans, fps = payload
    if c["fix_height"]:
        # motion_lib_real.py:61-72: shift the clip so that the lowest point of the robot in its FIRST frame touches z = 0.  The reference takes the
        # minimum over all mesh vertices; here the links' convex-hull support points (the points the stepper uses for ground contact) stand in
        # for the meshes, which are not shipped with the package.
        wpos, wmat, _, _ = robot_fk(c["parent"], c["local_translation"], c["local_rotation"], c["ext_parent"], c["ext_pos"], c["ext_rot"], pose_aa[:1], trans[:1])
        z = wpos[0][c["contact_body"], 2] + np.einsum("kj,kj->k", wmat[0][c["contact_body"]][:, 2, :], c["contact_pos"]) - c["contact_radius"]
        trans = trans.copy()
        trans[:, 2] -= float(z.min())
    proc = process_clip_real(c["parent"], c["local_translation"], c["local_rotation"], c["ext_parent"], c["ext_pos"], c["ext_rot"], pose_aa, trans, fps)
    nb = c["num_bodies"]
    f = {k: proc[k].astype(np.float32) for k in ("gts", "grs", "gvs", "gavs", "dof_pos", "dvs", "gts_t", "grs_t")}
    packed = abi.pack_frames(f["gts"], f["grs"], f["gvs"], f["gavs"], None, f["dvs"], gts_ext=f["gts_t"][:, nb:], grs_ext=f["grs_t"][:, nb:], dof_pos=f["dof_pos"])
    return packed, int(fps), trans.shape[0]   # fk_batch returns fps = int(1 / dt) (torch_humanoid_batch.py:219)


_POOL = {}
_POOL_STATE = {"atexit": False, "timer": None, "lock": None, "busy": 0}


def default_pool_workers():
    """Clip workers per process: half the usable cores, shared between the ranks of this node (8 ranks of an 8-GPU launch must not start 8 x 32
    workers), at most 32."""
    cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")) or 1))
    return max(1, min(32, cpus // (2 * local_world)))


def _clip_pool(workers):
    """The process pool behind load_motions: created on first use, re-used by calls that follow closely (load + the first `resample_motions()`,
    evaluation libraries), and shut down after `PHC_MOTION_POOL_IDLE_S` (default 60) seconds without work -- a `resample_motions()` comes every few
    hundred epochs; up to 32 idle torch-importing processes per rank for the whole training in between were pure cost.  The next
    call simply starts a new pool (the forkserver itself stays: a worker start is a fork of the preloaded server, ~10 ms each).
    Context: forkserver (see above); `PHC_MOTION_POOL_CONTEXT=spawn` selects spawn."""
    import multiprocessing as mp
    import threading
    if _POOL_STATE["lock"] is None:
        _POOL_STATE["lock"] = threading.Lock()
    with _POOL_STATE["lock"]:
        _POOL_STATE["busy"] += 1          # (the idle timer's shutdown leaves a pool alone while a call is using it; _arm_idle_shutdown() ends the use)
        if _POOL_STATE["timer"] is not None:
            _POOL_STATE["timer"].cancel()
        key = (os.getpid(), workers)
        if key not in _POOL:
            for k in list(_POOL):            # (a pool inherited through somebody else's fork belongs to the parent)
                if k[0] == os.getpid():
                    _POOL.pop(k).terminate()
                else:
                    _POOL.pop(k)
            ctx = mp.get_context(os.environ.get("PHC_MOTION_POOL_CONTEXT", "forkserver"))
            if ctx.get_start_method() == "forkserver":
                # numpy / torch / this module are imported once, in the server.  (Ignored by multiprocessing if the host application has started a
                # forkserver already: the workers then import them themselves -- slower to start, same results.)
                ctx.set_forkserver_preload(["phc_amd.motion_lib"])
            _POOL[key] = ctx.Pool(workers)
            if not _POOL_STATE["atexit"]:
                import atexit
                atexit.register(_close_pools)
                _POOL_STATE["atexit"] = True
        return _POOL[key]


def _arm_idle_shutdown():
    """End of a pooled call: (re)start the idle timer."""
    import threading
    idle = float(os.environ.get("PHC_MOTION_POOL_IDLE_S", "60"))
    with _POOL_STATE["lock"]:
        _POOL_STATE["busy"] = max(0, _POOL_STATE["busy"] - 1)
        t = _POOL_STATE["timer"]
        if t is not None:
            t.cancel()
        if idle <= 0:
            return
        t = threading.Timer(idle, _close_pools, kwargs={"only_if_idle": True})
        t.daemon = True
        t.start()
        _POOL_STATE["timer"] = t


def _close_pools(only_if_idle=False):
    lock = _POOL_STATE["lock"]
    if lock is not None:
        lock.acquire()
    try:
        if only_if_idle and _POOL_STATE["busy"] > 0:
            return
        for k in list(_POOL):
            pool = _POOL.pop(k)
            if k[0] == os.getpid():
                pool.terminate()
    finally:
        if lock is not None:
            lock.release()


def _run_clip_chunk(args):
    consts, payloads = args
    return [_run_clip_job(p, consts) for p in payloads]


class MotionLibBase:
    """See module docstring.  Mirrors reference MotionLibBase (motion_lib_base.py:114-567)."""

    def __init__(self, motion_lib_cfg):
        self.m_cfg = motion_lib_cfg
        self._sim_fps = 1 / self.m_cfg.get("step_dt", 1 / 30)
        self._device = torch.device(self.m_cfg.device)
        self.mesh_parsers = None
        self.load_data(self.m_cfg.motion_file, min_length=self.m_cfg.get("min_length", -1), im_eval=self.m_cfg.get("im_eval", False))
        self.setup_constants(fix_height=self.m_cfg.get("fix_height", FixHeightMode.full_fix), multi_thread=self.m_cfg.get("multi_thread", False))
        self._struct = None

    # ---- M2: load_data (motion_lib_base.py:131-158) ----
    def load_data(self, motion_file, min_length=-1, im_eval=False):
        if isinstance(motion_file, dict):
            data = motion_file  # in-memory AMASS-shaped dict (synthetic benches / tests)
        elif os.path.isfile(motion_file):
            data = joblib.load(motion_file)
        else:
            raise FileNotFoundError(f"motion_file {motion_file!r} not found (directory mode is not supported)")
        self._motion_data_load = data
        if min_length != -1:
            data_list = {k: v for k, v in data.items() if len(v["pose_quat_global"]) >= min_length}
        elif im_eval:
            data_list = {k: v for k, v in sorted(data.items(), key=lambda e: len(e[1]["pose_quat_global"]), reverse=True)}
        else:
            data_list = data
        self._motion_data_list = list(data_list.values())
        self._motion_data_keys = np.array(list(data_list.keys()))
        self._num_unique_motions = len(self._motion_data_list)
        if self._num_unique_motions == 0:
            raise ValueError("no motion clip left after the min_length filter")

    def setup_constants(self, fix_height=FixHeightMode.full_fix, multi_thread=True):
        self.fix_height = fix_height
        self.multi_thread = multi_thread
        U = self._num_unique_motions
        self._curr_motion_ids = None
        self._termination_history = torch.zeros(U, device=self._device)
        self._success_rate = torch.zeros(U, device=self._device)
        self._sampling_history = torch.zeros(U, device=self._device)
        self._sampling_prob = torch.ones(U, device=self._device) / U
        self._sampling_batch_prob = None

    # ---- M3: load_motions (motion_lib_base.py:181-326) ----
    def load_motions(self, skeleton_trees, gender_betas=None, limb_weights=None, random_sample=True, start_idx=0, max_len=-1):
        num_to_load = len(skeleton_trees)
        tree = skeleton_trees[0]
        self.num_joints = len(tree.node_names)
        # per-env body shapes (humanoid.py:824-866): env i's clip is run through env i's skeleton (motion_lib_smpl.py:153
        # `skeleton_trees[f]`); the envs share a few tree OBJECTS, so the unit of work below is the distinct (clip, tree) pair
        trees, tree_of = [], []
        for t in skeleton_trees:
            k = next((i for i, o in enumerate(trees) if o is t), None)
            if k is None:
                trees.append(t)
                k = len(trees) - 1
            tree_of.append(k)
        if random_sample:
            sample_idxes = torch.multinomial(self._sampling_prob, num_samples=num_to_load, replacement=True).to(self._device)
        else:
            sample_idxes = torch.remainder(torch.arange(num_to_load) + start_idx, self._num_unique_motions).to(self._device)
        self._curr_motion_ids = sample_idxes
        self.curr_motion_keys = self._motion_data_keys[sample_idxes.cpu().numpy()]
        sp = self._sampling_prob[self._curr_motion_ids]
        self._sampling_batch_prob = sp / sp.sum()

        idx_np = sample_idxes.cpu().numpy()
        cache = {}
        per = []
        # the reference seeds numpy with randint(5000) * pid (motion_lib_smpl.py:106): in its single-process path pid == 0, i.e. seed 0
        # on EVERY call (`heading_rng: seed0_per_call`, what the golden fixtures were generated with); its default multi-worker path
        # gives every worker and call a different seed -> default here: one persistent stream per library (seed + rank), advanced
        # across calls, so that `resample_motions()` draws new headings / crops
        if self.m_cfg.get("heading_rng", "persistent") == "seed0_per_call":
            rs = np.random.RandomState(0)
        else:
            if getattr(self, "_heading_rs", None) is None:
                self._heading_rs = np.random.RandomState((int(torch.initial_seed()) + 7919 * int(self.m_cfg.get("rank", 0))) % (2 ** 32))
            rs = self._heading_rs
        # pass 1 -- every random draw, in the reference's order (per env: the crop of a not-yet-seen clip, then the heading), so that
        # the stream does not depend on how the work below is scheduled
        uniq, crop, yaws, seen = [], {}, [], set()
        for e, u in enumerate(idx_np):
            u = int(u)
            if u not in crop:
                crop[u] = self._draw_crop(self._motion_data_list[u], max_len, rs)
            if (u, tree_of[e]) not in seen:
                seen.add((u, tree_of[e]))
                uniq.append((u, tree_of[e]))
            yaws.append(self._draw_heading(rs))
        # pass 2 -- FK + velocities ONCE per distinct clip (fp64 on the host cores; a fork pool when there are many: cfg 3 samples
        # ~5 800 distinct clips of the 11 313 for 8 192 envs, 17 ms each)
        consts = self._clip_consts(trees)
        jobs = [self._clip_payload(self._motion_data_list[u], t, max_len, crop[u]) for u, t in uniq]
        workers = int(self.m_cfg.get("num_workers", 0)) or default_pool_workers()
        if len(jobs) >= int(self.m_cfg.get("pool_min_jobs", 256)) and workers > 1:
            # chunks of jobs, each carrying the (small) constants: no per-worker state to set up or to go stale between calls
            per = max(1, -(-len(jobs) // (workers * 8)))
            chunks = [(consts, jobs[i:i + per]) for i in range(0, len(jobs), per)]
            try:
                done = [r for part in _clip_pool(workers).map(_run_clip_chunk, chunks, chunksize=1) for r in part]
            finally:
                _arm_idle_shutdown()
        else:
            done = [_run_clip_job(j, consts) for j in jobs]
        slot = {ut: k for k, ut in enumerate(uniq)}
        # pass 3 -- one packed fp32 record array per DISTINCT clip goes to the device once; the per-env copies (the reference keeps one
        # clip copy per env, motion_lib_base.py:300-307) are a device-side gather, and the per-env heading a device-side rotation
        dev = self._device
        self.num_bodies = self.num_joints
        packed = [rec for rec, _, _ in done]
        u_nf = np.array([p.shape[0] for p in packed], dtype=np.int64)
        u_start = np.concatenate([[0], np.cumsum(u_nf)[:-1]])
        uniq_frames = torch.from_numpy(np.concatenate(packed, axis=0)).to(dev)
        e_slot = np.array([slot[(int(u), tree_of[e])] for e, u in enumerate(idx_np)], dtype=np.int64)
        nfs = [int(u_nf[k]) for k in e_slot]
        fpss = [done[k][1] for k in e_slot]
        src = torch.from_numpy(np.concatenate([np.arange(u_nf[k], dtype=np.int64) + u_start[k] for k in e_slot])).to(dev)
        self.frames = uniq_frames.index_select(0, src)
        del uniq_frames, src
        if yaws[0] is not None:
            yaw_f = torch.repeat_interleave(torch.tensor(yaws, dtype=torch.float32, device=dev), torch.tensor(nfs, device=dev))
            self._apply_heading_device(yaw_f)
        self._make_views()
        aa_list = [self._clip_pose_aa(self._motion_data_list[int(u)], nf, crop[int(u)]) for u, nf in zip(idx_np, nfs)]
        per = None
        self.grvs, self.gravs = self.gvs[:, 0], self.gavs[:, 0]
        self._motion_aa = torch.from_numpy(np.concatenate(aa_list)).to(dev)
        nf_t = torch.tensor(nfs, dtype=torch.int64)
        self._motion_num_frames = nf_t.to(dev)
        self._motion_fps = torch.tensor(fpss, dtype=torch.float32, device=dev)
        self._motion_dt = torch.tensor([1.0 / f for f in fpss], dtype=torch.float32, device=dev)
        self._motion_lengths = torch.tensor([1.0 / f * (n - 1) for f, n in zip(fpss, nfs)], dtype=torch.float32, device=dev)
        gb = torch.zeros(num_to_load, 17) if gender_betas is None else torch.as_tensor(gender_betas)
        self._motion_bodies = gb.to(dev).float()
        lw = np.zeros((num_to_load, 10), dtype=np.float32) if limb_weights is None else np.asarray(limb_weights, dtype=np.float32)
        self._motion_limb_weights = torch.from_numpy(lw).to(dev)
        self._num_motions = num_to_load
        shifted = nf_t.roll(1)
        shifted[0] = 0
        self.length_starts = shifted.cumsum(0).to(dev)
        self.motion_ids = torch.arange(num_to_load, dtype=torch.long, device=dev)
        self._struct = abi.motion_lib_struct(self.frames, self.frames.shape[1], self.num_bodies, self._motion_lengths, self._motion_dt,
                                             self._motion_num_frames, self.length_starts, num_ext_bodies=self.num_ext_bodies,
                                             dofs_per_joint=self.dofs_per_joint)
        self.frames_epoch = getattr(self, "frames_epoch", 0) + 1   # (consumers that derive tables from `frames` key them on this)
        return per

    # ---- per-family hooks (SMPL here; MotionLibReal overrides) ----
    num_ext_bodies = 0
    dofs_per_joint = 3

    def _draw_crop(self, clip, max_len, rs):
        """Random crop start of a clip longer than max_len (motion_lib_smpl.py:124-128), or None."""
        T = np.asarray(clip["pose_quat_global"]).shape[0]
        return int(rs.randint(0, T - max_len + 1)) if (max_len != -1 and T > max_len) else None

    def _draw_heading(self, rs):
        """Random heading about +z per env (motion_lib_smpl.py:137-146), or None when headings are off."""
        randomize = (not flags.im_eval) and (not flags.test) and self.m_cfg.get("randomrize_heading", True)
        return float(np.pi * (2 * rs.random_sample() - 1.0)) if randomize else None

    def _clip_consts(self, trees):
        """Per-family constants of the clip workers (picklable, numpy only; no motion data)."""
        return {"kind": "smpl", "trees": [(np.asarray(t.parent_indices), np.asarray(t.local_translation)) for t in trees]}

    def _clip_payload(self, clip, tree_index, max_len, start):
        """One job of the clip workers: the crop of the clip's arrays (motion_lib_smpl.py:101-180 up to, not including, the per-env heading)."""
        trans = clip["root_trans_offset"]
        trans = trans.numpy() if isinstance(trans, torch.Tensor) else np.asarray(trans)
        g = np.asarray(clip["pose_quat_global"])
        if start is not None:
            g, trans = g[start:start + max_len], trans[start:start + max_len]
        return tree_index, np.ascontiguousarray(g), np.ascontiguousarray(trans), clip.get("fps", 30)

    def _process_unique_clip(self, clip, tree, max_len, start):
        """(kept for tools) FK + velocities of one clip on skeleton `tree`: (proc dict, fps, frames)."""
        _, g, trans, fps = self._clip_payload(clip, 0, max_len, start)
        return process_clip(np.asarray(tree.parent_indices), np.asarray(tree.local_translation), g, trans, fps), fps, g.shape[0]

    def _clip_pose_aa(self, clip, nf, start=None):
        if "pose_aa" in clip:
            aa = np.asarray(clip["pose_aa"], dtype=np.float32).reshape(-1, self.num_joints * 3)
            return aa[:nf] if start is None else aa[start:start + nf]
        return np.zeros((nf, self.num_joints * 3), dtype=np.float32)

    def _pack_one(self, proc):
        f = {k: proc[k].astype(np.float32) for k in ("gts", "grs", "gvs", "gavs", "lrs", "dvs")}
        return abi.pack_frames(f["gts"], f["grs"], f["gvs"], f["gavs"], f["lrs"], f["dvs"])

    def _make_views(self):
        nb = self.num_bodies
        F_ = self.frames.shape[0]
        o = 0
        views = {}
        for name, w in (("gts", 3), ("grs", 4), ("gvs", 3), ("gavs", 3), ("lrs", 4)):
            views[name] = self.frames[:, o:o + nb * w].view(F_, nb, w)
            o += nb * w
        views["dvs"] = self.frames[:, o:o + (nb - 1) * 3].view(F_, nb - 1, 3)
        self.gts, self.grs, self.gvs, self.gavs, self.lrs, self.dvs = (views[k] for k in ("gts", "grs", "gvs", "gavs", "lrs", "dvs"))

    def _apply_heading_device(self, yaw):
        """`apply_heading` on the packed per-env records, on the device: yaw [F] (one value per frame).  Positions and velocities are
        rotated about +z, rotations left-multiplied by the yaw quaternion, the root's local rotation follows its global one
        (motion_lib_smpl.py:137-146 commutes with FK, np.gradient and the gaussian filter: tests/test_abi_and_host.py heading golden)."""
        self._make_views()
        c, s_ = torch.cos(yaw)[:, None], torch.sin(yaw)[:, None]
        for v in (self.gts, self.gvs, self.gavs):
            x, y = v[..., 0].clone(), v[..., 1].clone()
            v[..., 0] = c * x - s_ * y
            v[..., 1] = s_ * x + c * y
        hz, hw = torch.sin(0.5 * yaw)[:, None], torch.cos(0.5 * yaw)[:, None]
        q = self.grs
        x, y, z, w = (q[..., k].clone() for k in range(4))
        # (0, 0, hz, hw) * (x, y, z, w)
        q[..., 0] = hw * x - hz * y
        q[..., 1] = hw * y + hz * x
        q[..., 2] = hw * z + hz * w
        q[..., 3] = hw * w - hz * z
        self.lrs[:, 0] = self.grs[:, 0]

    # ---- small accessors (motion_lib_base.py:328-435) ----
    def num_motions(self):
        return self._num_motions

    def get_total_length(self):
        return float(self._motion_lengths.sum())

    def get_motion_length(self, motion_ids=None):
        return self._motion_lengths if motion_ids is None else self._motion_lengths[motion_ids]

    def get_motion_num_steps(self, motion_ids=None):
        nf = self._motion_num_frames if motion_ids is None else self._motion_num_frames[motion_ids]
        fps = self._motion_fps if motion_ids is None else self._motion_fps[motion_ids]
        return (nf * self._sim_fps / fps).ceil().int()

    def sample_motions(self, n):
        return torch.multinomial(self._sampling_batch_prob, num_samples=n, replacement=True).to(self._device)

    def sample_time(self, motion_ids, truncate_time=None):
        phase = torch.rand(motion_ids.shape, device=self._device)
        motion_len = self._motion_lengths[motion_ids]
        if truncate_time is not None:
            assert truncate_time >= 0.0
            motion_len = motion_len - truncate_time
        return phase * motion_len

    def sample_time_interval(self, motion_ids, truncate_time=None):
        """motion_lib_base.py:414-423; the rand draw stays a torch.rand so the RNG stream matches."""
        if truncate_time is not None:
            raise NotImplementedError("truncate_time is not used on the imitation path")
        phase = torch.rand(motion_ids.shape, device=self._device)
        return self.sample_time_interval_from_phase(motion_ids, phase)

    def sample_time_interval_from_phase(self, motion_ids, phase):
        self._require_gpu()
        out = torch.empty(motion_ids.shape, dtype=torch.float32, device=self._device)
        ids = motion_ids.to(torch.int64).contiguous()
        L.check(L.load().phc_sample_time_interval(self._struct, ids.numel(), ids.data_ptr(), phase.contiguous().data_ptr(),
                                                  out.data_ptr(), _stream()), "phc_sample_time_interval")
        return out

    def update_hard_sampling_weight(self, failed_keys):
        if len(failed_keys) > 0:
            all_keys = self._motion_data_keys.tolist()
            indexes = [all_keys.index(k) for k in failed_keys]
            self._sampling_prob[:] = 0
            self._sampling_prob[indexes] = 1 / len(indexes)
        else:
            self._sampling_prob = torch.ones(self._num_unique_motions, device=self._device) / self._num_unique_motions

    def update_soft_sampling_weight(self, failed_keys):
        if len(failed_keys) > 0:
            all_keys = self._motion_data_keys.tolist()
            indexes = [all_keys.index(k) for k in failed_keys]
            self._termination_history[indexes] += 1
            self.update_sampling_prob(self._termination_history)
        else:
            self._sampling_prob = torch.ones(self._num_unique_motions, device=self._device) / self._num_unique_motions

    def update_sampling_prob(self, termination_history):
        if len(termination_history) == len(self._termination_history) and termination_history.sum() > 0:
            self._sampling_prob[:] = termination_history / termination_history.sum()
            self._termination_history = termination_history
            return True
        return False

    # ---- M9: get_motion_state (motion_lib_base.py:437-520) ----
    def _require_gpu(self):
        if self._device.type != "cuda":
            raise RuntimeError("MotionLib lookups run on the HIP device only (no CPU fallback); construct it with device='cuda:N'")
        if self._struct is None:
            raise RuntimeError("load_motions() has not been called")

    def get_motion_state(self, motion_ids, motion_times, offset=None):
        self._require_gpu()
        n = len(motion_ids)
        nb = self.num_bodies
        dev = self._device
        ids = motion_ids.to(torch.int64).contiguous()
        times = motion_times.to(torch.float32).contiguous()
        off = None if offset is None else offset.to(torch.float32).contiguous()
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        rg_pos, rb_rot, body_vel, body_ang_vel = f(n, nb, 3), f(n, nb, 4), f(n, nb, 3), f(n, nb, 3)
        nd = (nb - 1) * (1 if self.dofs_per_joint == 1 else 3)
        dof_pos, dof_vel = f(n, nd), f(n, nd)
        ne = self.num_ext_bodies
        pos_e, rot_e = (f(n, ne, 3), f(n, ne, 4)) if ne else (None, None)
        idx0 = torch.empty(n, dtype=torch.int64, device=dev)
        L.check(L.load().phc_motion_state(self._struct, n, ids.data_ptr(), times.data_ptr(), abi.ptr(off), rg_pos.data_ptr(),
                                          rb_rot.data_ptr(), body_vel.data_ptr(), body_ang_vel.data_ptr(), dof_pos.data_ptr(),
                                          dof_vel.data_ptr(), idx0.data_ptr(), None, None, abi.ptr(pos_e), abi.ptr(rot_e), _stream()),
                "phc_motion_state")
        f0l = idx0 + self.length_starts[ids]
        res = {
            "root_pos": rg_pos[:, 0].clone(), "root_rot": rb_rot[:, 0].clone(), "dof_pos": dof_pos,
            "root_vel": body_vel[:, 0].clone(), "root_ang_vel": body_ang_vel[:, 0].clone(), "dof_vel": dof_vel,
            "motion_aa": self._motion_aa[f0l], "rg_pos": rg_pos, "rb_rot": rb_rot, "body_vel": body_vel,
            "body_ang_vel": body_ang_vel, "motion_bodies": self._motion_bodies[ids], "motion_limb_weights": self._motion_limb_weights[ids],
        }
        if ne:  # motion_lib_real.py:300-312,356-359 (the extended velocities are not kept: nothing on the path reads them)
            res["rg_pos_t"] = torch.cat([rg_pos, pos_e], dim=1)
            res["rg_rot_t"] = torch.cat([rb_rot, rot_e], dim=1)
        return res

    def get_root_pos_smpl(self, motion_ids, motion_times):
        """motion_lib_base.py:522-547."""
        return {"root_pos": self.get_motion_state(motion_ids, motion_times)["root_pos"]}

    def _get_num_bodies(self):
        return self.num_bodies

    @property
    def struct(self):
        return self._struct


class MotionLibSMPL(MotionLibBase):
    """phc/utils/motion_lib_smpl.py:45-180.  The SMPL mesh height fix needs the licensed SMPL model
    files (`data/smpl`); like the reference when they are absent (:66-68) it is skipped."""

    def __init__(self, motion_lib_cfg):
        super().__init__(motion_lib_cfg)
        self.mesh_parsers = None


class MotionLibReal(MotionLibBase):
    """Robot motion library (reference phc/utils/motion_lib_real.py:54-430): clips are `pose_aa` per body + root translation;
    FK is `Humanoid_Batch.fk_batch` (restated in `process_clip_real`); joint coordinates are stored and blended as scalars;
    the `extend_config` bodies (hands / head) exist in the reference only.  No heading randomisation (commented out in the
    reference, :395-404).  `cfg.robot_model` is the compiled articulation (phc_amd.model), `cfg.robot` the robot yaml."""

    dofs_per_joint = 1

    def __init__(self, motion_lib_cfg):
        self.robot_model = motion_lib_cfg["robot_model"]
        ext = list(motion_lib_cfg["robot"].get("extend_config", []))
        names = self.robot_model.body_names
        self.ext_names = [e["joint_name"] for e in ext]
        self.ext_parent = np.array([names.index(e["parent_name"]) for e in ext], dtype=np.int32)
        self.ext_pos = np.array([e["pos"] for e in ext], dtype=np.float64).reshape(-1, 3)
        self.ext_rot = np.array([e["rot"] for e in ext], dtype=np.float64).reshape(-1, 4)   # wxyz
        self.num_ext_bodies = len(ext)
        super().__init__(motion_lib_cfg)

    def load_data(self, motion_file, min_length=-1, im_eval=False):
        if isinstance(motion_file, dict):
            data = motion_file
        elif os.path.isfile(motion_file):
            data = joblib.load(motion_file)
        else:
            raise FileNotFoundError(f"motion_file {motion_file!r} not found (directory mode is not supported)")
        self._motion_data_load = data
        n = lambda v: len(v["root_trans_offset"])
        if min_length != -1:
            data_list = {k: v for k, v in data.items() if n(v) >= min_length}
        elif im_eval:
            data_list = {k: v for k, v in sorted(data.items(), key=lambda e: n(e[1]), reverse=True)}
        else:
            data_list = data
        self._motion_data_list = list(data_list.values())
        self._motion_data_keys = np.array(list(data_list.keys()))
        self._num_unique_motions = len(self._motion_data_list)
        if self._num_unique_motions == 0:
            raise ValueError("no motion clip left after the min_length filter")

    def fix_trans_height(self, pose_aa, trans):
        """motion_lib_real.py:61-72: shift the clip so that the lowest point of the robot in its FIRST frame touches z = 0.
        The reference takes the minimum over all mesh vertices; here the links' convex-hull support points (the same
        points the stepper uses for ground contact) stand in for the meshes, which are not shipped with the package."""
        if self.fix_height == FixHeightMode.no_fix:
            return trans, 0.0
        m = self.robot_model
        wpos, wmat, _, _ = robot_fk(m.parent, m.local_translation, m.local_rotation, self.ext_parent, self.ext_pos, self.ext_rot, pose_aa[:1], trans[:1])
        z = wpos[0][m.contact_body, 2] + np.einsum("kj,kj->k", wmat[0][m.contact_body][:, 2, :], m.contact_pos) - m.contact_radius
        diff = float(z.min())
        trans = trans.copy()
        trans[:, 2] -= diff
        return trans, diff

    def _draw_crop(self, clip, max_len, rs):
        T = np.asarray(clip["root_trans_offset"]).shape[0]
        return int(rs.randint(0, T - max_len + 1)) if (max_len != -1 and T >= max_len) else None       # motion_lib_real.py:381-386

    def _draw_heading(self, rs):
        return None

    def _clip_consts(self, trees):
        m = self.robot_model
        return {"kind": "robot", "parent": np.asarray(m.parent), "local_translation": np.asarray(m.local_translation),
                "local_rotation": np.asarray(m.local_rotation), "ext_parent": self.ext_parent, "ext_pos": self.ext_pos, "ext_rot": self.ext_rot,
                "contact_body": np.asarray(m.contact_body), "contact_pos": np.asarray(m.contact_pos), "contact_radius": np.asarray(m.contact_radius),
                "fix_height": self.fix_height != FixHeightMode.no_fix, "num_bodies": int(self.num_joints)}

    def _clip_payload(self, clip, tree_index, max_len, start):
        trans = clip["root_trans_offset"]
        trans = (trans.numpy() if isinstance(trans, torch.Tensor) else np.asarray(trans)).astype(np.float64)
        pose_aa = clip["pose_aa"]
        pose_aa = (pose_aa.numpy() if isinstance(pose_aa, torch.Tensor) else np.asarray(pose_aa)).astype(np.float64)
        if start is not None:
            trans, pose_aa = trans[start:start + max_len], pose_aa[start:start + max_len]
        return tree_index, np.ascontiguousarray(pose_aa), np.ascontiguousarray(trans), clip.get("fps", 30)

    def _process_unique_clip(self, clip, tree, max_len, start):
        """(kept for tools) (proc dict, fps, frames) of one clip."""
        _, pose_aa, trans, fps = self._clip_payload(clip, 0, max_len, start)
        trans, _ = self.fix_trans_height(pose_aa, trans)
        m = self.robot_model
        proc = process_clip_real(m.parent, m.local_translation, m.local_rotation, self.ext_parent, self.ext_pos, self.ext_rot, pose_aa, trans, fps)
        return proc, int(fps), trans.shape[0]

    def _clip_pose_aa(self, clip, nf, start=None):
        return np.zeros((nf, self.num_joints * 3), dtype=np.float32)   # motion_lib_real.py:171-173: no "beta" -> zeros

    def _pack_one(self, proc):
        nb = self.num_bodies
        f = {k: proc[k].astype(np.float32) for k in ("gts", "grs", "gvs", "gavs", "dof_pos", "dvs", "gts_t", "grs_t")}
        return abi.pack_frames(f["gts"], f["grs"], f["gvs"], f["gavs"], None, f["dvs"], gts_ext=f["gts_t"][:, nb:], grs_ext=f["grs_t"][:, nb:],
                               dof_pos=f["dof_pos"])

    def _make_views(self):
        nb, ne = self.num_bodies, self.num_ext_bodies
        F_, nbe = self.frames.shape[0], nb + ne
        self.gts_t = self.frames[:, 0:nbe * 3].view(F_, nbe, 3)
        self.grs_t = self.frames[:, nbe * 3:nbe * 7].view(F_, nbe, 4)
        self.gts, self.grs = self.gts_t[:, :nb], self.grs_t[:, :nb]
        o = nbe * 7
        self.gvs = self.frames[:, o:o + nb * 3].view(F_, nb, 3)
        self.gavs = self.frames[:, o + nb * 3:o + nb * 6].view(F_, nb, 3)
        o += nb * 6
        self.dof_pos = self.frames[:, o:o + nb - 1]
        self.dvs = self.frames[:, o + nb - 1:o + 2 * (nb - 1)]
        self.lrs = None


def _stream():
    return torch.cuda.current_stream().cuda_stream
