"""dev tool: follow a clip with PD targets = the reference pose of the next frame (no policy) from t = 0 and print, per step, the largest
body-position error and its body -- where does a clip become untrackable?   python scripts/probes/track_probe.py squat:10 [steps]"""
import sys

import torch

sys.path.insert(0, ".")
from phc_amd.config import compose  # noqa: E402
from phc_amd.env.tasks.vec_task import parse_task  # noqa: E402
from phc_amd.utils.flags import flags  # noqa: E402

clip = sys.argv[1] if len(sys.argv) > 1 else "squat:10"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
flags.test = True
task, env = parse_task(compose(["env.num_envs=64", f"env.motion_file={clip}", "env.enableEarlyTermination=False"] + sys.argv[3:]))
env.reset()
names = task._body_names
for k in range(steps):
    a = (task.ref_dof_pos - task._pd_action_offset) / task._pd_action_scale
    obs, r, done, info = env.step(a.clamp(-1, 1))
    t = task.progress_buf[0].item() * task.dt
    res = task._motion_lib.get_motion_state(task._sampled_motion_ids, task.progress_buf * task.dt + task._motion_start_times)
    err = (task._rigid_body_pos - res["rg_pos"]).norm(dim=-1)[0]
    j = int(err.argmax())
    if k % 3 == 0 or err.max() > 0.2:
        print(f"step {k:3d} t {t:5.2f}  reward {float(r[0]):.3f}  max err {float(err.max()):.3f} m at {names[j]}  root z {float(task._rigid_body_pos[0, 0, 2]):.3f} (ref {float(res['rg_pos'][0, 0, 2]):.3f})  "
              f"root x {float(task._rigid_body_pos[0, 0, 0]):.3f} (ref {float(res['rg_pos'][0, 0, 0]):.3f})  contact fz {float(task._contact_forces[0, :, 2].sum()):.0f} N")
