"""Train the imitation policy on one synthetic clip for a few hundred PPO epochs on the MI355X and record the learning curve
(task reward, episode length) -- an end-to-end sanity check of stepper + reward + learner together.

    python scripts/learning_curve.py [epochs] [num_envs] [out.json]
"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from phc_amd.config import compose  # noqa: E402
from phc_amd.env.tasks.vec_task import parse_task  # noqa: E402
from phc_amd.learning.amp_agent import IMAmpAgent  # noqa: E402


def main():
    epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    num_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/learning_curve.json"
    extra = sys.argv[4:]
    torch.manual_seed(0)
    cfg = compose([f"env.num_envs={num_envs}", "env.motion_file=synthetic:1:0"] + extra)   # extra may override env.motion_file
    task, env = parse_task(cfg)
    agent = IMAmpAgent(env, cfg)
    agent.init_train()
    rows, t0 = [], time.time()
    for ep in range(epochs):
        done0 = 0
        info = agent.train_epoch()
        dones = float(agent.exp["dones"].float().sum())
        steps = agent.batch_size
        rows.append({"epoch": ep + 1, "task_reward": info["mean_task_reward"], "disc_reward": info["mean_disc_reward"],
                     "mean_episode_length": steps / max(dones, 1.0), "actor_loss": info["actor_loss"], "critic_loss": info["critic_loss"],
                     "total_fps": info["total_fps"]})
        if (ep + 1) % 25 == 0 or ep == 0:
            r = rows[-1]
            print(f"epoch {r['epoch']:4d}  task_r {r['task_reward']:.4f}  ep_len {r['mean_episode_length']:6.1f}  disc_r {r['disc_reward']:.3f}  "
                  f"fps {r['total_fps']:.0f}  ({time.time() - t0:.0f} s)", flush=True)
    json.dump({"config": {"num_envs": num_envs, "epochs": epochs, "extra": extra}, "rows": rows}, open(out, "w"))


if __name__ == "__main__":
    main()
