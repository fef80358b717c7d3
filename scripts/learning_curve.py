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
    bf16 = "--fp32-gemm" not in extra          # A/B of the GEMM precision (the reference trains in fp32, im.yaml:51; the product runs bf16 MFMA GEMMs)
    extra = [e for e in extra if e != "--fp32-gemm"]
    seed = next((int(e.split("=")[1]) for e in extra if e.startswith("--seed=")), 0)      # (round 6: outcomes of early training depend on the seed, profiles/r06_multi_clip/)
    extra = [e for e in extra if not e.startswith("--seed=")]
    torch.manual_seed(seed)
    cfg = compose([f"env.num_envs={num_envs}", "env.motion_file=synthetic:1:0"] + extra)   # extra may override env.motion_file
    task, env = parse_task(cfg)
    agent = IMAmpAgent(env, cfg, bf16=bf16)
    agent.init_train()
    rows, t0 = [], time.time()
    for ep in range(epochs):
        done0 = 0
        info = agent.train_epoch()
        dones = float(agent.exp["dones"].float().sum())
        steps = agent.batch_size
        rows.append({"epoch": ep + 1, "task_reward": info["mean_task_reward"], "disc_reward": info["mean_disc_reward"],
                     "mean_episode_length": steps / max(dones, 1.0), "actor_loss": info["actor_loss"], "critic_loss": info["critic_loss"],
                     "total_fps": info["total_fps"], "reward_raw": info["reward_raw"], "kl": info["kl"],
                     "terminated_per_step": float(task._terminate_buf.float().mean())})
        if (ep + 1) % 25 == 0 or ep == 0:
            r = rows[-1]
            print(f"epoch {r['epoch']:4d}  task_r {r['task_reward']:.4f}  ep_len {r['mean_episode_length']:6.1f}  disc_r {r['disc_reward']:.3f}  "
                  f"raw {[round(x, 3) for x in r['reward_raw']]} fps {r['total_fps']:.0f}  ({time.time() - t0:.0f} s)", flush=True)
        if (ep + 1) % 250 == 0:
            json.dump({"config": {"num_envs": num_envs, "epochs": epochs, "extra": extra}, "rows": rows}, open(out, "w"))
    # acceptance: the evaluation sweep (IMAmpAgent.eval == im_amp.py:136-242): every clip from t = 0 with the DETERMINISTIC policy,
    # termination distance 0.5 m on the mean body distance -> success rate and MPJPE; then one full-length episode from t = 0 with
    # training's 0.25 m per-body termination, deterministic actions: steps survived and the mean reward terms
    eval_info, failed = agent.eval(output_dir=None, log=print)
    from phc_amd.utils.flags import flags
    flags.test = True                       # episodes start at t = 0 (humanoid_im.py:1000-1023)
    agent.set_eval()
    obs = env.reset()
    alive = torch.ones(task.num_envs, dtype=torch.bool, device=task.device)
    survived = torch.zeros(task.num_envs, device=task.device)
    raw_sum, n_steps = None, int(task._motion_lib._motion_lengths.min().item() * 30) - 2
    with torch.no_grad():
        for t in range(n_steps):
            res = agent.get_action_values(obs)
            obs, r, done, info = env.step(agent.preprocess_actions(res["mus"]))
            alive &= ~info["terminate"].bool()
            survived += alive.float()
            rr = info["reward_raw"][alive].mean(0) if alive.any() else torch.zeros(info["reward_raw"].shape[1], device=task.device)
            raw_sum = rr if raw_sum is None else raw_sum + rr
    flags.test = False
    acc = {"deterministic_rollout_steps": n_steps, "mean_steps_survived": float(survived.mean()), "fraction_surviving_whole_clip": float(alive.float().mean()),
           "mean_reward_raw_terms [pos, rot, vel, ang_vel, power]": (raw_sum / n_steps).tolist(), **eval_info}
    print("acceptance:", json.dumps(acc))
    json.dump({"config": {"num_envs": num_envs, "epochs": epochs, "extra": extra, "gemm_dtype": "bf16" if bf16 else "f32"}, "acceptance": acc, "rows": [r for r in rows if r["epoch"] == 1 or r["epoch"] % 10 == 0]}, open(out, "w"))


if __name__ == "__main__":
    main()
