"""TEST INFRASTRUCTURE -- goldens for the task-observation versions other than 6 / 7 (env.obs_v = 1, 2, 3, 8, 9; humanoid_im.py:1203-1306,
1395-1515) at time_steps = 1: the reference's own jit functions on the states of tests/golden/task_fns.npz.  `_v2` also takes joint
positions: the simulator's are task_fns' `dof_pos`; the reference's at t + dt are looked up with oracle/phc_oracle.get_motion_state (itself
pinned to the reference's MotionLibSMPL.get_motion_state by tests/test_oracle_golden.py) from the motion_lib_eval fixture.
python oracle/gen_golden_task_obs_versions.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
import phc_oracle as po  # noqa: E402

ref_shim.install()
him = ref_shim.ref_module("phc.env.tasks.humanoid_im")

g = np.load(os.path.join(ROOT, "tests", "golden", "task_fns.npz"))
gl = dict(np.load(os.path.join(ROOT, "tests", "golden", "motion_lib_eval.npz")))
t = lambda k: torch.from_numpy(g[k])
bp, br, bv, bav = t("body_pos"), t("body_rot"), t("body_vel"), t("body_ang_vel")
r1 = (t("ref1_pos"), t("ref1_rot"), t("ref1_vel"), t("ref1_ang_vel"))
N = bp.shape[0]
dt = np.float32(1 / 30)
t1 = ((g["progress"] + 1).astype(np.float32) * dt + g["start_times"]).astype(np.float32)     # humanoid_im.py:752
ref1_dof = po.get_motion_state(gl, g["env_motion"], t1)["dof_pos"].astype(np.float32)
assert np.abs(po.get_motion_state(gl, g["env_motion"], t1)["rg_pos"] - g["ref1_pos"]).max() < 2e-6
out = dict(ref1_dof_pos=ref1_dof)
for upright in (True, False):
    u = f"u{int(upright)}"
    out[f"v1_{u}"] = him.compute_imitation_observations(bp[:, 0], br[:, 0], bp, br, bv, bav, *r1, 1, upright).numpy()
    out[f"v2_{u}"] = him.compute_imitation_observations_v2(bp[:, 0], br[:, 0], bp, br, bv, bav, t("dof_pos").reshape(N, 23, 3), *r1,
                                                          torch.from_numpy(ref1_dof).reshape(N, 23, 3), 1, upright).numpy()
    out[f"v3_{u}"] = him.compute_imitation_observations_v3(bp[:, 0], br[:, 0], bp, br, bv, bav, *r1, 1, upright).numpy()
    out[f"v8_{u}"] = him.compute_imitation_observations_v8(bp[:, 0], br[:, 0], bp, br, bv, bav, *r1, 1, upright).numpy()
    out[f"v9_{u}"] = him.compute_imitation_observations_v9(bp[:, 0], br[:, 0], bp, br, bv, bav, r1[0], r1[1], r1[2][:, 0], r1[3][:, 0], 1, upright).numpy()
J = 24
assert out["v1_u1"].shape[1] == 15 * J and out["v2_u1"].shape[1] == 15 * J + 3 * (J - 1) and out["v3_u1"].shape[1] == 9 * J
assert out["v8_u1"].shape[1] == 30 * J and out["v9_u1"].shape[1] == 18 * J + 6
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "task_obs_versions.npz"), **out)
print("wrote task_obs_versions.npz", {k: v.shape for k, v in out.items()})

# ---- env.full_body_reward False (humanoid_im.py:925-936): compute_imitation_reward on the tracked-body subsets ----
TRACK = ["Pelvis", "L_Ankle", "R_Ankle", "Head", "L_Hand", "R_Hand"]
names = ['Pelvis', 'L_Hip', 'L_Knee', 'L_Ankle', 'L_Toe', 'R_Hip', 'R_Knee', 'R_Ankle', 'R_Toe', 'Torso', 'Spine', 'Chest', 'Neck', 'Head', 'L_Thorax',
         'L_Shoulder', 'L_Elbow', 'L_Wrist', 'L_Hand', 'R_Thorax', 'R_Shoulder', 'R_Elbow', 'R_Wrist', 'R_Hand']
tid = torch.tensor([names.index(b) for b in TRACK])
specs = {"k_pos": 100., "k_rot": 10., "k_vel": 0.1, "k_ang_vel": 0.1, "w_pos": 0.5, "w_rot": 0.3, "w_vel": 0.1, "w_ang_vel": 0.1}
r0 = (t("ref_pos"), t("ref_rot"), t("ref_vel"), t("ref_ang_vel"))
rew, raw = him.compute_imitation_reward(bp[:, 0], br[:, 0], bp[:, tid], br[:, tid], bv[:, tid], bav[:, tid], r0[0][:, tid], r0[1][:, tid], r0[2][:, tid],
                                        r0[3][:, tid], specs)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "reward_track_bodies.npz"), track_ids=tid.numpy(), reward=rew.numpy(), reward_raw=raw.numpy())
print("wrote reward_track_bodies.npz", rew.shape, raw.shape)
