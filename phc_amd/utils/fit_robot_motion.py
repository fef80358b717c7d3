"""Retargeting fit: SMPL joint trajectories -> robot (H1 / G1) motion clips (f-4; reference scripts/data_process/fit_smpl_motion.py:56-186).

The reference's `process_motion` has two halves:
  1. AMASS pose + the robot-fitted SMPL shape -> SMPL joint positions (`SMPL_Parser.get_joints_verts`, :78-99).  That half needs the SMPL
     model files and smpl_sim; neither ships with the reference nor exists on this machine, so it stays OUTSIDE: this module takes the
     joint trajectories `smpl_joints [T, 24, 3]` (SMPL_BONE_ORDER_NAMES order, already scaled / grounded as :86-89 does) as its input;
  2. the fit itself (:101-150): per frame the robot's joint angles, root rotation (initialised with the SMPL root's heading, :97-98) and one
     global root offset are optimised with Adam (lr 0.02, 500 iterations) so that the robot's matched bodies (`robot.joint_matches`, incl.
     the extended hand / head bodies) follow the matched SMPL joints: loss = mean |p_robot - p_smpl| + 0.01 mean(dof^2); after every step
     the angles are clamped to the joint ranges and smoothed along time by a 5-tap gaussian (sigma 0.75); finally the clip is moved to the
     ground by the lowest point of the FIRST frame (:164-170: mesh vertices there; here the links' convex-hull support points stand in
     for the meshes, as in MotionLibReal.fix_trans_height).  That half is what this module runs -- differentiable forward kinematics of
     `Humanoid_Batch.fk_batch` (torch_humanoid_batch.py:163-257) in torch, on the CPU or on the device -- and its output has the schema the
     reference dumps (:172-179) and MotionLibReal reads.

`gaussian_filter_1d_batch` comes from smpl_sim (not vendored: restated as a replicate-padded 1-d convolution with the normalised kernel).

    python -m phc_amd.utils.fit_robot_motion --robot unitree_h1 --joints joints.npz --out h1_clips.pkl
where joints.npz holds, per clip key, `<key>/smpl_joints [T,24,3]`, `<key>/root_trans [T,3]` and `<key>/root_aa [T,3]` (SMPL root axis-angle).
"""
import math

import numpy as np
import torch

SMPL_BONE_ORDER_NAMES = ["Pelvis", "L_Hip", "R_Hip", "Torso", "L_Knee", "R_Knee", "Spine", "L_Ankle", "R_Ankle", "Chest", "L_Toe", "R_Toe",
                         "Neck", "L_Thorax", "R_Thorax", "Head", "L_Shoulder", "R_Shoulder", "L_Elbow", "R_Elbow", "L_Wrist", "R_Wrist", "L_Hand",
                         "R_Hand"]


def _aa_to_mat(aa):
    """Rodrigues, batched [..., 3] -> [..., 3, 3]: R = I + A [aa]x + B [aa]x^2 with A = sin(t) / t, B = (1 - cos t) / t^2 and their series near
    t = 0 -- the fit starts from all-zero joint angles, where a formula through the normalised axis has a vanishing gradient
    (pytorch3d.axis_angle_to_matrix, which the reference uses, is smooth there too)."""
    t2 = (aa * aa).sum(-1, keepdim=True)
    small = t2 < 1e-8
    t2s = torch.where(small, torch.ones_like(t2), t2)
    t = torch.sqrt(t2s)
    A = torch.where(small, 1.0 - t2 / 6.0, torch.sin(t) / t)
    B = torch.where(small, 0.5 - t2 / 24.0, (1.0 - torch.cos(t)) / t2s)
    x, y, z = aa[..., 0], aa[..., 1], aa[..., 2]
    zero = torch.zeros_like(x)
    K = torch.stack([zero, -z, y, z, zero, -x, -y, x, zero], dim=-1).reshape(aa.shape[:-1] + (3, 3))
    eye = torch.eye(3, dtype=aa.dtype, device=aa.device).expand(K.shape)
    return eye + A[..., None] * K + B[..., None] * (K @ K)


def _quat_wxyz_to_mat(q):
    q = q / q.norm(dim=-1, keepdim=True)
    w, x, y, z = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(q.shape[:-1] + (3, 3))


class RobotFK:
    """Differentiable `Humanoid_Batch.forward_kinematics_batch` (torch_humanoid_batch.py:224-257) for the NB simulated + E extended bodies:
    world position = parent rotation * offset + parent position; world rotation = parent * rest rotation * joint rotation."""

    def __init__(self, model, extend_config, device="cpu", dtype=torch.float32):
        names = list(model.body_names)
        self.names = names + [e["joint_name"] for e in extend_config]
        self.parents = list(np.asarray(model.parent)) + [names.index(e["parent_name"]) for e in extend_config]
        t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), dtype=dtype, device=device)
        self.offsets = torch.cat([t(model.local_translation), t([e["pos"] for e in extend_config]).reshape(-1, 3)], dim=0)
        self.rest = _quat_wxyz_to_mat(torch.cat([t(model.local_rotation), t([e["rot"] for e in extend_config]).reshape(-1, 4)], dim=0))
        self.num_bodies, self.num_ext = len(names), len(extend_config)
        axis = np.zeros((model.num_bodies - 1, 3))
        for i in range(1, model.num_bodies):
            axis[i - 1] = model.dof_axis[model.dof_start[i]]
        self.dof_axis = t(axis)                                      # [ND, 3] (one revolute joint per body)
        lo, hi = model.dof_limits()
        self.joints_range = torch.stack([t(lo), t(hi)], dim=-1)      # [ND, 2]

    def __call__(self, pose_aa, root_trans):
        """pose_aa [T, NB+E, 3] (root axis-angle, axis * angle per joint, zeros for the extended bodies), root_trans [T, 3]
        -> world positions [T, NB+E, 3], world rotation matrices [T, NB+E, 3, 3]."""
        pose_aa, root_trans = pose_aa.to(self.offsets), root_trans.to(self.offsets)
        R = _aa_to_mat(pose_aa)
        pos, rot = [None] * len(self.parents), [None] * len(self.parents)
        for i, p in enumerate(self.parents):
            if p < 0:
                pos[i], rot[i] = root_trans, R[:, 0]
            else:
                pos[i] = (rot[p] @ self.offsets[i]) + pos[p]
                rot[i] = rot[p] @ (self.rest[i] @ R[:, i])
        return torch.stack(pos, dim=1), torch.stack(rot, dim=1)


def gaussian_filter_time(x, kernel_size=5, sigma=0.75):
    """[T, D] smoothed along T (fit_smpl_motion.py:103-105,134: smpl_sim's gaussian_filter_1d_batch; restated, see module docstring)."""
    half = kernel_size // 2
    k = torch.exp(-0.5 * (torch.arange(-half, half + 1, dtype=x.dtype, device=x.device) / sigma) ** 2)
    k = (k / k.sum()).view(1, 1, -1)
    xt = x.t().unsqueeze(1)                                          # [D, 1, T]
    xt = torch.nn.functional.pad(xt, (half, half), mode="replicate")
    return torch.nn.functional.conv1d(xt, k).squeeze(1).t()


def heading_rotvec(root_aa):
    """:97-98: the SMPL root rotation with the y-up base rotation removed, reduced to its heading (rotation about z), as a rotation vector."""
    from scipy.spatial.transform import Rotation as sRot
    q = (sRot.from_rotvec(np.asarray(root_aa, dtype=np.float64)) * sRot.from_quat([0.5, 0.5, 0.5, 0.5]).inv())
    d = q.apply(np.array([1.0, 0.0, 0.0]))
    yaw = np.arctan2(d[:, 1], d[:, 0])                               # torch_utils.calc_heading
    out = np.zeros((len(yaw), 3))
    out[:, 2] = yaw
    return out


def fit_clip(model, robot_cfg, smpl_joints, root_trans, root_aa=None, iterations=500, lr=0.02, device="cpu", fps=30, log=None):
    """One clip.  smpl_joints [T, 24, 3], root_trans [T, 3] (the SMPL root position the joints were computed with), root_aa [T, 3] SMPL root
    axis-angle (None: heading 0).  Returns the reference's dump dict (fit_smpl_motion.py:172-179)."""
    ext = list(robot_cfg.get("extend_config", []))
    fk = RobotFK(model, ext, device=device)
    pick_r = [fk.names.index(m[0]) for m in robot_cfg["joint_matches"]]
    pick_s = [SMPL_BONE_ORDER_NAMES.index(m[1]) for m in robot_cfg["joint_matches"]]
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32), device=device)
    joints = t(smpl_joints)
    T = joints.shape[0]
    trans = t(root_trans)
    root_trans_offset = trans + (joints[:, 0] - trans)                 # :91-92
    gt_root = t(heading_rotvec(root_aa) if root_aa is not None else np.zeros((T, 3)))
    nd, ne = model.num_bodies - 1, len(ext)
    dof = torch.zeros((T, nd, 1), device=device, requires_grad=True)
    root_rot = gt_root.clone().requires_grad_(True)
    root_off = torch.zeros((1, 3), device=device, requires_grad=True)
    opt = torch.optim.Adam([dof, root_rot, root_off], lr=lr)
    lo, hi = fk.joints_range[:, 0, None], fk.joints_range[:, 1, None]
    zeros_ext = torch.zeros((T, ne, 3), device=device)
    target = joints[:, pick_s]

    def pose():
        return torch.cat([root_rot[:, None], fk.dof_axis * dof, zeros_ext], dim=1)
    loss_val = float("nan")
    for it in range(iterations):
        pos, _ = fk(pose(), root_trans_offset + root_off)
        loss = (pos[:, pick_r] - target).norm(dim=-1).mean() + 0.01 * torch.mean(torch.square(dof))      # :116-121
        opt.zero_grad()
        loss.backward()
        opt.step()
        with torch.no_grad():
            dof.clamp_(lo, hi)                                                                           # :131
            dof.copy_(gaussian_filter_time(dof[..., 0])[..., None])                                       # :134
        loss_val = float(loss.detach())
        if log is not None and (it % 50 == 0 or it == iterations - 1):
            log(f"iter {it}: {loss_val * 1000:.3f}")
    with torch.no_grad():
        dof.clamp_(lo, hi)
        pose_aa = pose()
        root_dump = (root_trans_offset + root_off).clone()
        # move to the ground by the lowest point of the first frame (:164-170; support points instead of mesh vertices)
        pos0, rot0 = fk(pose_aa[:1], root_dump[:1])
        cb = torch.as_tensor(np.asarray(model.contact_body), device=device, dtype=torch.long)
        cp = t(model.contact_pos)
        z = pos0[0, cb, 2] + (rot0[0, cb][:, 2, :] * cp).sum(-1) - t(model.contact_radius)
        h = float(z.min())
        root_dump[:, 2] -= h
        joints_dump = joints.clone()
        joints_dump[..., 2] -= h
        pos, _ = fk(pose_aa, root_dump)
        err = (pos[:, pick_r] - joints_dump[:, pick_s]).norm(dim=-1).mean()
    from scipy.spatial.transform import Rotation as sRot
    return {"root_trans_offset": root_dump.cpu().numpy(), "pose_aa": pose_aa.cpu().numpy(), "dof": dof[..., 0].detach().cpu().numpy(),
            "root_rot": sRot.from_rotvec(root_rot.detach().cpu().numpy()).as_quat(), "smpl_joints": joints_dump.cpu().numpy(), "fps": fps,
            "fit_loss": loss_val, "fit_keypoint_error": float(err)}


def main(argv=None):
    import argparse
    import joblib
    from ..cfg_defaults import GROUPS
    from ..model import load_model
    from .. import robots
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--robot", default="unitree_h1", choices=["unitree_h1", "unitree_h1_nohead", "unitree_g1"])
    ap.add_argument("--joints", required=True, help="npz: <key>/smpl_joints, <key>/root_trans, <key>/root_aa")
    ap.add_argument("--out", required=True)
    ap.add_argument("--iterations", type=int, default=500)
    ap.add_argument("--device", default="cpu")
    a = ap.parse_args(argv)
    rc = GROUPS["robot"][a.robot]
    model = load_model(f"{rc['humanoid_type']}_humanoid")
    robots.apply_robot_gains(model, robots.ROBOTS[rc["humanoid_type"]])
    data = np.load(a.joints)
    keys = sorted({k.split("/")[0] for k in data.files})
    out = {}
    for k in keys:
        root_aa = data[f"{k}/root_aa"] if f"{k}/root_aa" in data.files else None
        out[k] = fit_clip(model, rc, data[f"{k}/smpl_joints"], data[f"{k}/root_trans"], root_aa, a.iterations, device=a.device,
                          log=lambda s, k=k: print(k, s, flush=True))
    joblib.dump(out, a.out)
    print(f"wrote {len(out)} clips to {a.out}")


if __name__ == "__main__":
    main()
