"""TEST INFRASTRUCTURE ONLY -- golden for the AMASS -> motion-pkl conversion: the reference's own poselib calls driven as
scripts/data_process/convert_amass_isaac.py:85-137 drives them (the script itself needs the licensed SMPL model files for
`SMPL_Robot`; it forces the neutral zero-beta body, whose skeleton is the shipped smpl_0_humanoid.xml).

    python oracle/gen_golden_amass.py  -> tests/golden/amass_convert.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_shim  # noqa: E402

ref_shim.install()
import torch  # noqa: E402
from scipy.spatial.transform import Rotation as sRot  # noqa: E402

from gen_golden import MJCF, OUT  # noqa: E402
from phc_amd.utils.convert_amass import SMPL_BONE_ORDER_NAMES  # noqa: E402  (smpl_sim's table is not installed here)


def main():
    from poselib.poselib.skeleton.skeleton3d import SkeletonState, SkeletonTree
    tree = SkeletonTree.from_mjcf(MJCF)
    mujoco_joint_names = list(tree.node_names)
    rng = np.random.default_rng(42)
    B = 37
    pose_aa_in = rng.normal(0, 0.4, (B, 72))
    pose_aa_in[:, :3] = rng.normal(0, 1.0, (B, 3)) + np.array([1.2, 1.2, 1.2])   # AMASS roots are z-up-to-y-up style rotations
    root_trans = np.cumsum(rng.normal(0, 0.02, (B, 3)), axis=0) + np.array([0, 0, 0.9])
    smpl_2_mujoco = [SMPL_BONE_ORDER_NAMES.index(q) for q in mujoco_joint_names if q in SMPL_BONE_ORDER_NAMES]
    pose_aa = np.concatenate([pose_aa_in[:, :66], np.zeros((B, 6))], axis=1)
    pose_aa_mj = pose_aa.reshape(-1, 24, 3)[..., smpl_2_mujoco, :].copy()
    pose_quat = sRot.from_rotvec(pose_aa_mj.reshape(-1, 3)).as_quat().reshape(B, 24, 4)
    root_trans_offset = torch.from_numpy(root_trans) + tree.local_translation[0]
    st = SkeletonState.from_rotation_and_root_translation(tree, torch.from_numpy(pose_quat), root_trans_offset, is_local=True)
    pose_quat_global = (sRot.from_quat(st.global_rotation.reshape(-1, 4).numpy()) * sRot.from_quat([0.5, 0.5, 0.5, 0.5]).inv()).as_quat().reshape(B, -1, 4)
    st2 = SkeletonState.from_rotation_and_root_translation(tree, torch.from_numpy(pose_quat_global), root_trans_offset, is_local=False)
    np.savez_compressed(os.path.join(OUT, "amass_convert.npz"), pose_aa_in=pose_aa_in, trans=root_trans, pose_quat_global=pose_quat_global,
                        pose_quat=st2.local_rotation.numpy(), root_trans_offset=root_trans_offset.numpy(), pose_aa=pose_aa)
    print("amass_convert golden written")


if __name__ == "__main__":
    main()
