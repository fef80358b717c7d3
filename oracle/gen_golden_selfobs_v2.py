"""TEST INFRASTRUCTURE -- golden for self_obs_v 2 (body-state history in the policy observation): the reference's
`compute_humanoid_observations_smpl_max_v2` (phc/env/tasks/humanoid.py:2054-2108) on the current body states of tests/golden/task_fns.npz and
five seeded "previous" states per env (the states of other envs of the fixture, shifted a little), past_track_steps = 5, local_root_obs True
(the only value the reference function runs with), both values of upright.   python oracle/gen_golden_selfobs_v2.py"""
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
N, P = g["body_pos"].shape[0], 5
rng = np.random.default_rng(77)
hist = np.zeros((N, P, 24, 13), dtype=np.float32)
for k in range(P):
    src = np.roll(np.arange(N), k + 1)
    hist[:, k, :, 0:3] = g["body_pos"][src] + rng.normal(0, 0.05, (N, 1, 3)).astype(np.float32)
    hist[:, k, :, 3:7] = g["body_rot"][src]
    hist[:, k, :, 7:10] = g["body_vel"][src]
    hist[:, k, :, 10:13] = g["body_ang_vel"][src]
cur = np.concatenate([g["body_pos"], g["body_rot"], g["body_vel"], g["body_ang_vel"]], axis=-1)[:, None]
allst = torch.from_numpy(np.concatenate([hist, cur], axis=1))                 # [N, P + 1, 24, 13], oldest first, current last (:1443-1447)
out = dict(hist=hist)
for local_root in (True,):   # (local_root_obs False: the reference's own shapes do not match at humanoid.py:2085-2087 -- it raises)
    for upright in (True, False):
        o = hum.compute_humanoid_observations_smpl_max_v2(allst[..., 0:3].contiguous(), allst[..., 3:7].contiguous(), allst[..., 7:10].contiguous(),
                                                          allst[..., 10:13].contiguous(), torch.zeros(N, 11), torch.zeros(N, 10), local_root, True, upright,
                                                          False, False, P + 1)
        out[f"l{int(local_root)}u{int(upright)}"] = o.numpy()
assert out["l1u1"].shape == (N, 6 * 358) and np.array_equal(out["l1u1"][:, 5 * 358:], g["self_obs"])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "self_obs_v2.npz"), **out)
print("wrote self_obs_v2.npz", {k: v.shape for k, v in out.items()})
