"""TEST INFRASTRUCTURE -- golden for env.fut_tracks (humanoid_im.py:39-47,741-747): the task observation against T = numTrajSamples = 3
reference frames trajSampleTimestepInv = 30 apart, obs_v 6, 7 and 9 (time-major blocks).  The reference's own jit functions
compute_imitation_observations_v6 / _v7 / _v9 at time_steps = 3 on the simulator states of tests/golden/task_fns.npz; the T reference frames per env
are looked up with oracle/phc_oracle.get_motion_state (pinned to the reference's MotionLibSMPL.get_motion_state by tests/test_oracle_golden.py)
at the reference's motion times, (progress + 1) * dt + k / 30 + start + offset evaluated in torch fp32 exactly as :744-745 writes it.
python oracle/gen_golden_fut_tracks.py"""
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
N, J, T = bp.shape[0], 24, 3
dt, ts = 1 / 30, 1 / 30
rng = np.random.default_rng(5)
start_off = torch.from_numpy(rng.uniform(0, 0.2, N).astype(np.float32))
progress = torch.from_numpy(g["progress"].astype(np.int64)) - 1          # progress_buf when _compute_task_obs runs in the golden's step
start = t("start_times")
time_internals = torch.arange(T).repeat(N).view(-1, T) * ts                                                          # :743
times = ((progress[:, None] + 1 + 1) * dt + time_internals + start[:, None] + start_off[:, None]).flatten()          # :744 (buf already incremented)
assert times.dtype == torch.float32
mids = np.repeat(g["env_motion"].astype(np.int64), T)
ms = po.get_motion_state(gl, mids, times.numpy())
ref = [torch.from_numpy(ms[k].astype(np.float32)) for k in ("rg_pos", "rb_rot", "body_vel", "body_ang_vel")]
out = dict(start_off=start_off.numpy(), times=times.numpy().reshape(N, T))
for upright in (True, False):
    u = f"u{int(upright)}"
    out[f"v6_{u}"] = him.compute_imitation_observations_v6(bp[:, 0], br[:, 0], bp, br, bv, bav, *ref, T, upright).numpy()
    out[f"v7_{u}"] = him.compute_imitation_observations_v7(bp[:, 0], br[:, 0], bp, bv, ref[0], ref[2], T, upright).numpy()
    out[f"v9_{u}"] = him.compute_imitation_observations_v9(bp[:, 0], br[:, 0], bp, br, bv, bav, ref[0], ref[1], ref[2][:, 0], ref[3][:, 0], T, upright).numpy()
assert out["v6_u1"].shape == (N, T * 24 * J) and out["v9_u1"].shape == (N, T * (18 * J + 6))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "task_obs_fut_tracks.npz"), **out)
print("wrote task_obs_fut_tracks.npz", {k: v.shape for k, v in out.items()})
