"""Round 6: how large is the power term of the task reward (humanoid_im.py:939-946: -0.0005 * sum |dof_force * dof_vel|), and how much of it is the exploration noise?

Trains the 64-clip library WITHOUT the power term for `train_s` seconds (every seed tracks after ~250 epochs), then rolls the policy out for `steps` env steps three ways --
deterministic (mu), sampled as a rollout samples (mu + sigma * noise, sigma = exp(-2.9)), and with zero actions -- and prints sum |dof_force * dof_vel| per env step
(watts; x 0.0005 = the reward term), by clip class, for steps with progress > 3 (the reference zeroes the first three).

    python scripts/probes/power_probe.py [train_s=40] [seed=1] [envs=3072] [more overrides, e.g. +solver.force_average=1 sim.substeps=8]
"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from phc_amd.config import compose  # noqa: E402
from phc_amd.env.tasks.vec_task import parse_task  # noqa: E402
from phc_amd.learning.amp_agent import IMAmpAgent  # noqa: E402

train_s = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
envs = int(sys.argv[3]) if len(sys.argv) > 3 else 3072
torch.manual_seed(seed)
cfg = compose(["learning=im_pnn", "env=env_im_pnn", f"env.num_envs={envs}", "env.motion_file=locomotion:64:0", "env.num_prim=2", "env.training_prim=0",
               "env.auto_pmcp=False", "env.auto_pmcp_soft=True", "env.power_reward=False"] + sys.argv[4:])
task, env = parse_task(cfg)
agent = IMAmpAgent(env, cfg)
agent.init_train()
t0, n = time.time(), 0
while time.time() - t0 < train_s:
    agent.train_epoch()
    n += 1
ep_len = agent.batch_size / max(float(agent.exp["dones"].float().sum()), 1.0)
print(f"trained {n} epochs ({time.time() - t0:.0f} s) without the power term: mean episode length {ep_len:.1f}", flush=True)
agent.set_eval()
net = agent.model.a2c_network
keys = task._motion_lib.curr_motion_keys if hasattr(task._motion_lib, "curr_motion_keys") else None


def rollout(mode, steps=150):
    agent.obs = agent.env_reset()
    rows = []
    for _ in range(steps):
        with torch.no_grad(), agent._autocast():
            mu, logstd = net.eval_actor(agent._preproc_obs(task.obs_buf))
        mu = mu.float()
        if mode == "sampled":
            a = mu + torch.exp(logstd.float()) * torch.randn_like(mu)
        elif mode == "zero":
            a = torch.zeros_like(mu)
        else:
            a = mu
        task.reset_done()
        env.step(torch.clamp(a, -1.0, 1.0))
        power = (task.dof_force_tensor * task._dof_vel).abs().sum(-1)
        ok = task.progress_buf > 3
        rows.append((float(power[ok].mean()) if bool(ok.any()) else float("nan"), float(task.dof_force_tensor.abs().mean()), float(task._dof_vel.abs().mean()),
                     float(task.reset_buf.float().mean())))
    r = np.array(rows[20:])
    return {"power_W": float(np.nanmean(r[:, 0])), "reward_term": float(-0.0005 * np.nanmean(r[:, 0])), "mean_abs_dof_force": float(r[:, 1].mean()),
            "mean_abs_dof_vel": float(r[:, 2].mean()), "resets_per_step": float(r[:, 3].mean())}


out = {"train_epochs": n, "mean_episode_length": ep_len}
for mode in ("deterministic", "sampled", "zero"):
    out[mode] = rollout(mode)
    print(mode, json.dumps(out[mode]), flush=True)
print("POWER_JSON" + json.dumps(out))
