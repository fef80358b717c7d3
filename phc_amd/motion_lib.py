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
        if random_sample:
            sample_idxes = torch.multinomial(self._sampling_prob, num_samples=num_to_load, replacement=True).to(self._device)
        else:
            sample_idxes = torch.remainder(torch.arange(num_to_load) + start_idx, self._num_unique_motions).to(self._device)
        self._curr_motion_ids = sample_idxes
        self.curr_motion_keys = self._motion_data_keys[sample_idxes.cpu().numpy()]
        sp = self._sampling_prob[self._curr_motion_ids]
        self._sampling_batch_prob = sp / sp.sum()

        idx_np = sample_idxes.cpu().numpy()
        parents = np.asarray(tree.parent_indices)
        local_t = np.asarray(tree.local_translation)
        cache = {}
        per = []
        # the reference seeds numpy with randint(5000) * pid and pid == 0 in the single-process path
        # (motion_lib_smpl.py:106) -> RandomState(0) for the heading draws
        rs = np.random.RandomState(0)
        randomize = (not flags.im_eval) and (not flags.test) and self.m_cfg.get("randomrize_heading", True)
        aa_list, nfs, fpss = [], [], []
        for i, u in enumerate(idx_np):
            clip = self._motion_data_list[u]
            if u not in cache:
                trans = clip["root_trans_offset"]
                trans = trans.numpy() if isinstance(trans, torch.Tensor) else np.asarray(trans)
                g = np.asarray(clip["pose_quat_global"])
                if max_len != -1 and g.shape[0] > max_len:
                    start = rs.randint(0, g.shape[0] - max_len + 1)
                    g, trans = g[start:start + max_len], trans[start:start + max_len]
                cache[u] = (process_clip(parents, local_t, g, trans, clip.get("fps", 30)), clip.get("fps", 30), g.shape[0])
            proc, fps, nf = cache[u]
            if randomize:
                proc = apply_heading(proc, np.pi * (2 * rs.random_sample() - 1.0))
            per.append(proc)
            nfs.append(nf)
            fpss.append(fps)
            if "pose_aa" in clip:
                aa_list.append(np.asarray(clip["pose_aa"], dtype=np.float32).reshape(-1, self.num_joints * 3)[:nf])
            else:
                aa_list.append(np.zeros((nf, self.num_joints * 3), dtype=np.float32))
        dev = self._device
        fields = {k: np.concatenate([p[k] for p in per], axis=0).astype(np.float32) for k in ("gts", "grs", "gvs", "gavs", "lrs", "dvs")}
        frames = abi.pack_frames(fields["gts"], fields["grs"], fields["gvs"], fields["gavs"], fields["lrs"], fields["dvs"])
        self.frames = torch.from_numpy(frames).to(dev)
        self.num_bodies = self.num_joints
        nb = self.num_bodies
        F_ = self.frames.shape[0]
        o = 0
        views = {}
        for name, w in (("gts", 3), ("grs", 4), ("gvs", 3), ("gavs", 3), ("lrs", 4)):
            views[name] = self.frames[:, o:o + nb * w].view(F_, nb, w)
            o += nb * w
        views["dvs"] = self.frames[:, o:o + (nb - 1) * 3].view(F_, nb - 1, 3)
        self.gts, self.grs, self.gvs, self.gavs, self.lrs, self.dvs = (views[k] for k in ("gts", "grs", "gvs", "gavs", "lrs", "dvs"))
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
        self._struct = abi.motion_lib_struct(self.frames, self.frames.shape[1], nb, self._motion_lengths, self._motion_dt,
                                             self._motion_num_frames, self.length_starts)
        return per

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
        dof_pos, dof_vel = f(n, (nb - 1) * 3), f(n, (nb - 1) * 3)
        idx0 = torch.empty(n, dtype=torch.int64, device=dev)
        L.check(L.load().phc_motion_state(self._struct, n, ids.data_ptr(), times.data_ptr(), abi.ptr(off), rg_pos.data_ptr(),
                                          rb_rot.data_ptr(), body_vel.data_ptr(), body_ang_vel.data_ptr(), dof_pos.data_ptr(),
                                          dof_vel.data_ptr(), idx0.data_ptr(), None, None, _stream()), "phc_motion_state")
        f0l = idx0 + self.length_starts[ids]
        return {
            "root_pos": rg_pos[:, 0].clone(), "root_rot": rb_rot[:, 0].clone(), "dof_pos": dof_pos,
            "root_vel": body_vel[:, 0].clone(), "root_ang_vel": body_ang_vel[:, 0].clone(), "dof_vel": dof_vel,
            "motion_aa": self._motion_aa[f0l], "rg_pos": rg_pos, "rb_rot": rb_rot, "body_vel": body_vel,
            "body_ang_vel": body_ang_vel, "motion_bodies": self._motion_bodies[ids], "motion_limb_weights": self._motion_limb_weights[ids],
        }

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


def _stream():
    return torch.cuda.current_stream().cuda_stream
