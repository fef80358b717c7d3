"""TEST INFRASTRUCTURE -- golden for self_obs_v 3 (S6 force sensors in the observation): the reference's
`compute_humanoid_observations_smpl_max_v3` (phc/env/tasks/humanoid.py:2113-2169) on the body states of tests/golden/task_fns.npz and a
seeded sensor tensor [N, 2*6] (L_Ankle, R_Ankle: force then torque, sensor frame).   python oracle/gen_golden_selfobs_v3.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
hum = ref_shim.ref_module("phc.env.tasks.humanoid")
g = np.load(os.path.join(ROOT, "tests", "golden", "task_fns.npz"))
t = lambda k: torch.from_numpy(g[k])
N = g["body_pos"].shape[0]
rng = np.random.default_rng(606)
sensors = (rng.standard_normal((N, 12)) * np.array([30, 30, 400, 10, 10, 3] * 2)).astype(np.float32)
out = hum.compute_humanoid_observations_smpl_max_v3(t("body_pos"), t("body_rot"), t("body_vel"), t("body_ang_vel"), torch.from_numpy(sensors),
                                                    torch.zeros(N, 11), torch.zeros(N, 10), True, True, True, False, False)
ref_v1 = hum.compute_humanoid_observations_smpl_max(t("body_pos"), t("body_rot"), t("body_vel"), t("body_ang_vel"), torch.zeros(N, 11), torch.zeros(N, 10),
                                                   True, True, True, False, False)
assert out.shape[1] == ref_v1.shape[1] + 12
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "self_obs_v3.npz"), sensors=sensors, self_obs_v3=out.numpy())
print("wrote self_obs_v3.npz", out.shape)
