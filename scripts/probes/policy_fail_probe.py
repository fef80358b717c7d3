"""dev tool: train the imitation policy on one clip for a few hundred epochs, then follow the DETERMINISTIC policy from chosen clip times and print, per
step, the largest body-position error and its body, the root pose and the contact force -- how does an episode of a plateaued policy end?
    python scripts/probes/policy_fail_probe.py squat:10 700 """
import sys

import torch

sys.path.insert(0, ".")
from phc_amd.config import compose  # noqa: E402
from phc_amd.env.tasks.vec_task import parse_task  # noqa: E402
from phc_amd.learning.amp_agent import IMAmpAgent  # noqa: E402
from phc_amd.utils.flags import flags  # noqa: E402

clip = sys.argv[1] if len(sys.argv) > 1 else "squat:10"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 700
starts = ["zero", "random", "random"]
torch.manual_seed(0)
cfg = compose(["env.num_envs=4096", f"env.motion_file={clip}"])
task, env = parse_task(cfg)
agent = IMAmpAgent(env, cfg)
agent.init_train()
for ep in range(epochs):
    info = agent.train_epoch()
    if (ep + 1) % 100 == 0:
        print(f"epoch {ep + 1}: task_r {info['mean_task_reward']:.3f} ep_len {agent.batch_size / max(float(agent.exp['dones'].float().sum()), 1.0):.1f}", flush=True)
# where in the clip do training episodes end?
t_end = (task.progress_buf.float() * task.dt + task._motion_start_times)[task._terminate_buf.bool()]
agent.set_eval()
names = task._body_names
for t0 in starts:
    flags.test = t0 == "zero"      # episodes start at t = 0 (humanoid_im.py:1000-1023); otherwise at a random clip time
    task.reset_buf[:] = 1
    obs = env.reset()
    flags.test = False
    print(f"--- deterministic policy from clip time {t0} ---")
    with torch.no_grad():
        for k in range(40):
            res_a = agent.get_action_values(obs)
            obs, r, done, info = env.step(agent.preprocess_actions(res_a["mus"]))
            ref = task._motion_lib.get_motion_state(task._sampled_motion_ids, task.progress_buf * task.dt + task._motion_start_times)
            err = (task._rigid_body_pos - ref["rg_pos"]).norm(dim=-1)[0]
            j = int(err.argmax())
            print(f"step {k:3d} t {float(task.progress_buf[0]) * task.dt + float(task._motion_start_times[0]):5.2f} reward {float(r[0]):.3f} max err {float(err.max()):.3f} m at {names[j]:10s} "
                  f"root z {float(task._rigid_body_pos[0, 0, 2]):.3f} (ref {float(ref['rg_pos'][0, 0, 2]):.3f}) root xy {float(task._rigid_body_pos[0, 0, 0]):+.3f} {float(task._rigid_body_pos[0, 0, 1]):+.3f} "
                  f"fz {float(task._contact_forces[0, :, 2].sum()):5.0f} N  |a|max {float(res_a['mus'][0].abs().max()):.2f}  terminate {int(info['terminate'][0])}")
            if bool(info["terminate"][0]):
                break
