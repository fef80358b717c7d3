"""How does the stepper behave under a FIXED action on the standing clip?  Prints per-step statistics over 300 control steps (10 s) with
early termination off: root height / tilt / horizontal drift, contact force vs weight, the four imitation-reward terms, max body deviation.

    python scripts/stand_probe.py [zero|track] [num_envs]
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from phc_amd.config import compose  # noqa: E402
from phc_amd.env.tasks.vec_task import parse_task  # noqa: E402


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "zero"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    torch.manual_seed(0)
    cfg = compose([f"env.num_envs={n}", "env.motion_file=stand:12", "env.enableEarlyTermination=False"] + sys.argv[3:])
    task, env = parse_task(cfg)
    env.reset()
    task._motion_start_times[:] = 0          # every env from the start of the clip
    task.progress_buf[:] = 0
    weight = float(task.model.mass.sum()) * 9.81
    inv = 1.0 / task._pd_action_scale
    print(f"mode {mode}: {n} envs, total mass {task.model.mass.sum():.1f} kg, weight {weight:.0f} N, dt {task.dt:.4f}")
    print("step  root_z   tilt_deg  drift_xy   Fz/weight   r_pos  r_rot  r_vel  r_angvel   max_dev   reward")
    for t in range(300):
        a = torch.zeros(n, task.num_actions, device=task.device) if mode == "zero" else (task.ref_dof_pos - task._pd_action_offset) * inv
        obs, rew, done, info = env.step(a)
        if t < 10 or t % 20 == 19:
            root = task._root_states
            q = root[:, 3:7]
            up_z = 1 - 2 * (q[:, 0] ** 2 + q[:, 1] ** 2)        # z component of the body z axis
            tilt = torch.rad2deg(torch.arccos(up_z.clamp(-1, 1)))
            dev = (task._rigid_body_pos - task.ref_body_pos).norm(dim=-1).max(dim=-1).values
            fz = task._contact_forces[..., 2].sum(-1)
            rr = info["reward_raw"].mean(0).tolist()
            print(f"{t + 1:4d}  {root[:, 2].mean():.4f}  {tilt.mean():8.3f}  {root[:, :2].norm(dim=-1).mean():8.4f}  {float(fz.mean()) / weight:9.3f}   "
                  f"{rr[0]:.3f}  {rr[1]:.3f}  {rr[2]:.3f}  {rr[3]:.3f}   {dev.mean():8.4f}  {rew.mean():.4f}")
    print("final: fraction with max body deviation < 0.25 m:", float((dev < 0.25).float().mean()))


if __name__ == "__main__":
    main()
