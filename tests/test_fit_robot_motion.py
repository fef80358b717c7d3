"""f-4 retargeting fit (phc_amd/utils/fit_robot_motion.py <- reference scripts/data_process/fit_smpl_motion.py:56-186), CPU.
The differentiable FK equals the numpy `robot_fk` (itself pinned to the reference's Humanoid_Batch by tests/test_h1.py goldens); the fit
recovers a ground-truth robot clip from the positions of its matched bodies handed over as "SMPL joints"; its output loads into MotionLibReal."""
import numpy as np
import pytest
import torch

from phc_amd import robots
from phc_amd.cfg_defaults import GROUPS
from phc_amd.model import load_model
from phc_amd.motion_lib import robot_fk
from phc_amd.utils.fit_robot_motion import SMPL_BONE_ORDER_NAMES, RobotFK, fit_clip, gaussian_filter_time
from phc_amd.utils.synthetic_motion import make_robot_clip


def _robot(name):
    rc = GROUPS["robot"][name]
    model = load_model(f"{rc['humanoid_type']}_humanoid")
    robots.apply_robot_gains(model, robots.ROBOTS[rc["humanoid_type"]])
    return rc, model


@pytest.mark.parametrize("name", ["unitree_h1", "unitree_g1"])
def test_torch_fk_equals_numpy_fk(name):
    rc, model = _robot(name)
    ext = list(rc["extend_config"])
    clip = make_robot_clip(np.random.default_rng(1), model, 40, num_extend=len(ext))
    fk = RobotFK(model, ext, dtype=torch.float64)
    pos, rot = fk(torch.from_numpy(clip["pose_aa"]), torch.from_numpy(clip["root_trans_offset"]))
    names = model.body_names
    wpos, wmat, _, _ = robot_fk(model.parent, model.local_translation, model.local_rotation, [names.index(e["parent_name"]) for e in ext],
                                [e["pos"] for e in ext], [e["rot"] for e in ext], clip["pose_aa"], clip["root_trans_offset"])
    np.testing.assert_allclose(pos.numpy(), wpos, atol=1e-10)
    np.testing.assert_allclose(rot.numpy(), wmat, atol=1e-10)


def test_gaussian_filter_keeps_constants_and_smooths():
    x = torch.ones(30, 4) * 0.3
    assert torch.allclose(gaussian_filter_time(x), x, atol=1e-6)
    y = torch.zeros(31, 1)
    y[15] = 1.0
    f = gaussian_filter_time(y)
    assert abs(float(f.sum()) - 1.0) < 1e-6 and float(f[15]) < 0.6 and float(f[14]) > 0.15 and float(f[12]) == 0.0   # 5 taps


def test_fit_recovers_a_robot_clip_and_loads_into_the_motion_library():
    rc, model = _robot("unitree_h1")
    ext = list(rc["extend_config"])
    T = 48
    gt = make_robot_clip(np.random.default_rng(5), model, T, num_extend=len(ext))
    fk = RobotFK(model, ext, dtype=torch.float64)
    pos, _ = fk(torch.from_numpy(gt["pose_aa"]), torch.from_numpy(gt["root_trans_offset"]))
    # "SMPL joints": the matched SMPL slots carry the positions of the robot bodies they are matched to
    joints = np.zeros((T, 24, 3))
    for rname, sname in rc["joint_matches"]:
        joints[:, SMPL_BONE_ORDER_NAMES.index(sname)] = pos[:, fk.names.index(rname)].numpy()
    # SMPL root axis-angle whose heading is the clip's: base rotation (0.5, 0.5, 0.5, 0.5) behind the robot's root rotation
    from scipy.spatial.transform import Rotation as sRot
    root_aa = (sRot.from_rotvec(gt["pose_aa"][:, 0]) * sRot.from_quat([0.5, 0.5, 0.5, 0.5])).as_rotvec()
    torch.manual_seed(0)
    out = fit_clip(model, rc, joints, joints[:, 0], root_aa, iterations=300)
    assert set(out) >= {"root_trans_offset", "pose_aa", "dof", "root_rot", "smpl_joints", "fps"}          # fit_smpl_motion.py:172-179
    assert out["pose_aa"].shape == (T, model.num_bodies + len(ext), 3) and out["dof"].shape == (T, model.num_dof)
    assert out["fit_keypoint_error"] < 0.03, out["fit_keypoint_error"]                                    # metres, mean over 16 matched bodies
    lo, hi = model.dof_limits()
    assert (out["dof"] >= lo - 1e-6).all() and (out["dof"] <= hi + 1e-6).all()
    # grounded: lowest support point of the first frame at z = 0
    pos0, rot0 = RobotFK(model, ext)(torch.from_numpy(out["pose_aa"][:1]), torch.from_numpy(out["root_trans_offset"][:1]))
    z = pos0[0, model.contact_body, 2].numpy() + np.einsum("kj,kj->k", rot0[0].numpy()[model.contact_body][:, 2, :], model.contact_pos) - model.contact_radius
    assert abs(float(z.min())) < 1e-4
    # the dump is a MotionLibReal clip
    from phc_amd.config import EasyDict
    from phc_amd.motion_lib import FixHeightMode, MotionLibReal
    from phc_amd.model import ArticulationModel  # noqa: F401
    from phc_amd.env.tasks.humanoid_im import SkeletonTree
    cfg = EasyDict({"motion_file": {"fit_00000": out}, "device": "cpu", "fix_height": FixHeightMode.full_fix, "min_length": -1, "max_length": -1,
                    "im_eval": False, "multi_thread": False, "smpl_type": "h1", "randomrize_heading": False, "step_dt": 1 / 50, "robot": rc,
                    "robot_model": model})
    try:
        lib = MotionLibReal(cfg)
        lib.load_motions(skeleton_trees=[SkeletonTree(model.body_names, model.parent, model.local_translation)] * 2, random_sample=False)
    except RuntimeError as e:     # the library packs its records for the device; on a CPU-only box the host part above is what is checked
        if "HIP" not in str(e) and "cuda" not in str(e).lower():
            raise
        return
    assert lib._motion_num_frames.tolist() == [T, T]


@pytest.mark.parametrize("name", ["unitree_h1", "unitree_g1"])
def test_robot_stand_and_armswing_clips(name):
    """`env.motion_file=stand | armswing` on a robot (round 5, profiles/r05_robots/): the default joint pose held still; with `arm_swing` only the two shoulder-pitch
    joints move, in antiphase, ramped in from rest; `pose_aa` is the per-joint axis times the joint angle (what the motion library's FK consumes)."""
    from phc_amd.utils.synthetic_motion import make_robot_stand_clip
    rc, model = _robot(name)
    q0 = np.asarray(robots.ROBOTS[rc["humanoid_type"]]["default_dof_pos"])
    ne = len(rc["extend_config"])
    still = make_robot_stand_clip(model, q0, seconds=2.0, num_extend=ne)
    T = 61
    assert still["dof"].shape == (T, model.num_dof) and still["pose_aa"].shape == (T, model.num_bodies + ne, 3) and still["fps"] == 30
    np.testing.assert_allclose(still["dof"], np.tile(q0[None], (T, 1)), atol=1e-7)
    swing = make_robot_stand_clip(model, q0, seconds=2.0, num_extend=ne, arm_swing=0.5)
    moved = np.flatnonzero(np.abs(swing["dof"] - still["dof"]).max(0) > 0)
    names = [n for i, n in enumerate(model.body_names) if i > 0 and model.dof_start[i] in moved]
    assert len(moved) == 2 and all("shoulder_pitch" in n for n in names), names
    d = swing["dof"][:, moved] - still["dof"][:, moved]
    np.testing.assert_allclose(d[:, 0], -d[:, 1], atol=1e-7)
    assert d[0].max() == 0 and 0.3 < np.abs(d).max() <= 0.5
    for i in range(1, model.num_bodies):
        s = model.dof_start[i]
        np.testing.assert_allclose(swing["pose_aa"][:, i], model.dof_axis[s][None] * swing["dof"][:, s:s + 1], atol=1e-6)
