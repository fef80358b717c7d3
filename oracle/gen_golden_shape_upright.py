"""TEST INFRASTRUCTURE -- goldens for (a) robot.has_upright_start False (`remove_base_rot`, phc/env/tasks/humanoid.py:1936-1939, and the
`if not upright:` branch of every observation function) and (b) the shape / limb-weight observation columns (has_shape_obs, has_weight_obs,
has_shape_obs_disc, has_weight_obs_disc: humanoid.py:2043-2047, humanoid_amp.py:1005-1008): the reference's own jit functions on the
states of tests/golden/task_fns.npz, for both values of local_root_obs.   python oracle/gen_golden_shape_upright.py"""
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
him = ref_shim.ref_module("phc.env.tasks.humanoid_im")
hamp = ref_shim.ref_module("phc.env.tasks.humanoid_amp")

g = np.load(os.path.join(ROOT, "tests", "golden", "task_fns.npz"))
t = lambda k: torch.from_numpy(g[k])
N = g["body_pos"].shape[0]
rng = np.random.default_rng(4242)
shape = np.concatenate([rng.integers(0, 3, (N, 1)).astype(np.float32), rng.standard_normal((N, 10)).astype(np.float32)], axis=1)   # gender + 10 betas
limb = (np.abs(rng.standard_normal((N, 10))) * np.array([0.9] * 5 + [12.0] * 5)).astype(np.float32)                               # 5 lengths + 5 masses
bp, br, bv, bav = t("body_pos"), t("body_rot"), t("body_vel"), t("body_ang_vel")
kid = torch.from_numpy(g["key_body_ids"]).long()
sub = torch.from_numpy(g["dof_subset"]).long()
out = dict(shape=shape, limb=limb)
for local_root in (True, False):
    for upright in (True, False):
        tag = f"l{int(local_root)}u{int(upright)}"
        out[f"self_{tag}"] = hum.compute_humanoid_observations_smpl_max(bp, br, bv, bav, torch.from_numpy(shape), torch.from_numpy(limb), local_root, True,
                                                                         upright, True, True).numpy()
        out[f"amp_{tag}"] = hamp.build_amp_observations_smpl(bp[:, 0], br[:, 0], bv[:, 0], bav[:, 0], t("dof_pos"), t("dof_vel"), bp[:, kid],
                                                              torch.from_numpy(shape), torch.from_numpy(limb), sub, local_root, True, True, True, True,
                                                              upright).numpy()
    # (task observations do not depend on local_root_obs)
for upright in (True, False):
    out[f"task_v6_u{int(upright)}"] = him.compute_imitation_observations_v6(bp[:, 0], br[:, 0], bp, br, bv, bav, t("ref1_pos"), t("ref1_rot"), t("ref1_vel"),
                                                                           t("ref1_ang_vel"), 1, upright).numpy()
    out[f"task_v7_u{int(upright)}"] = him.compute_imitation_observations_v7(bp[:, 0], br[:, 0], bp, bv, t("ref1_pos"), t("ref1_vel"), 1, upright).numpy()
assert np.array_equal(out["self_l1u1"][:, :358], g["self_obs"]) and np.array_equal(out["task_v6_u1"], g["task_obs"])
assert np.array_equal(out["amp_l1u1"][:, :196], g["amp_obs"]) and out["self_l1u1"].shape[1] == 358 + 21 and out["amp_l1u1"].shape[1] == 196 + 21
assert np.abs(out["self_l1u0"][:, :358] - g["self_obs"]).max() > 0.1
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "obs_shape_upright.npz"), **out)
print("wrote obs_shape_upright.npz", {k: v.shape for k, v in out.items()})
