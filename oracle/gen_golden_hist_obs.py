"""TEST INFRASTRUCTURE -- golden for env.enableHistObs: the reference's own `HumanoidAMP._compute_humanoid_obs` (phc/env/tasks/humanoid_amp.py:546-557,
on top of `Humanoid._compute_humanoid_obs`, humanoid.py:1435-1485) run on a `__new__`-made task that carries exactly the attributes the two methods
read: the body states of tests/golden/task_fns.npz and a seeded AMP history buffer [N, 10, 196].  Stored: the method's output for env_ids = None and
for a subset of envs.   python oracle/gen_golden_hist_obs.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
amp = ref_shim.ref_module("phc.env.tasks.humanoid_amp")
g = np.load(os.path.join(ROOT, "tests", "golden", "task_fns.npz"))
N = 12                                            # (the first envs of the fixture: keeps this one small)
t = lambda k: torch.from_numpy(g[k][:N])
rng = np.random.default_rng(707)
S, P = 10, 196
amp_buf = rng.standard_normal((N, S, P)).astype(np.float32)
task = amp.HumanoidAMP.__new__(amp.HumanoidAMP)
task._rigid_body_pos, task._rigid_body_rot = t("body_pos"), t("body_rot")
task._rigid_body_vel, task._rigid_body_ang_vel = t("body_vel"), t("body_ang_vel")
task.self_obs_v = 1
task.humanoid_type = "smpl"
task.humanoid_shapes = torch.zeros(N, 17)
task.humanoid_limb_and_weights = torch.zeros(N, 10)
task._local_root_obs, task._root_height_obs, task._has_upright_start = True, True, True
task._has_shape_obs, task._has_limb_weight_obs = False, False
task._enable_hist_obs = True
task._amp_obs_buf = torch.from_numpy(amp_buf)
task._num_amp_obs_steps, task._num_amp_obs_per_step = S, P
full = amp.HumanoidAMP._compute_humanoid_obs(task)
ids = torch.tensor(sorted(rng.choice(N, size=max(2, N // 3), replace=False).tolist()), dtype=torch.long)
sub = amp.HumanoidAMP._compute_humanoid_obs(task, ids)
task._enable_hist_obs = False
plain = amp.HumanoidAMP._compute_humanoid_obs(task)
assert full.shape == (N, plain.shape[1] + S * P) and sub.shape == (len(ids), full.shape[1])
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hist_obs.npz"), amp_obs_buf=amp_buf, env_ids=ids.numpy(), obs_all=full.numpy(), obs_subset=sub.numpy(),
                    obs_without_hist=plain.numpy())
print("wrote hist_obs.npz", full.shape, sub.shape)
