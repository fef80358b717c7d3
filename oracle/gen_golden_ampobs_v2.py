"""TEST INFRASTRUCTURE -- golden for amp_obs_v 2: the reference's `build_amp_observations_smpl_v2` (phc/env/tasks/humanoid_amp.py:1015-1059)
on the states of tests/golden/task_fns.npz (key bodies R_Ankle, L_Ankle, R_Wrist, L_Wrist).   python oracle/gen_golden_ampobs_v2.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
hamp = ref_shim.ref_module("phc.env.tasks.humanoid_amp")
from poselib.poselib.skeleton.skeleton3d import SkeletonTree  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "task_fns.npz"))
t = lambda k: torch.from_numpy(g[k])
names = list(SkeletonTree.from_mjcf(os.path.join(ref_shim.REFERENCE_ROOT, "phc/data/assets/mjcf/smpl_0_humanoid.xml")).node_names)
kid = torch.tensor([names.index(b) for b in ["R_Ankle", "L_Ankle", "R_Wrist", "L_Wrist"]])
dof_names = names[1:]
remove = ["L_Hand", "R_Hand", "L_Toe", "R_Toe"]
dof_subset = torch.from_numpy(np.concatenate([np.arange(i * 3, i * 3 + 3) for i, nm in enumerate(dof_names) if nm not in remove]))
N = g["body_pos"].shape[0]
bp, br, bv, bav = t("body_pos"), t("body_rot"), t("body_vel"), t("body_ang_vel")
v2 = hamp.build_amp_observations_smpl_v2(bp[:, 0], br[:, 0], bv[:, 0], bav[:, 0], t("dof_pos"), t("dof_vel"), bp[:, kid], bv[:, kid], torch.zeros(N, 11),
                                         torch.zeros(N, 10), dof_subset, True, True, True, False, False, True)
v1 = hamp.build_amp_observations_smpl(bp[:, 0], br[:, 0], bv[:, 0], bav[:, 0], t("dof_pos"), t("dof_vel"), bp[:, kid], torch.zeros(N, 11), torch.zeros(N, 10),
                                      dof_subset, True, True, True, False, False, True)
assert v2.shape[1] == v1.shape[1] + 12 and torch.equal(v2[:, :196], v1)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "amp_obs_v2.npz"), amp_obs_v2=v2.numpy())
print("wrote amp_obs_v2.npz", v2.shape)
