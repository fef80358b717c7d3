"""Where does the rollout (`play_steps`, 32 steps) spend its time?  Re-runs the fused rollout loop of IMAmpAgent.play_steps with HIP events
around its segments (device time) and perf_counter around the whole loop (wall): python scripts/profile_rollout.py [num_envs]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from phc_amd.config import compose  # noqa: E402
from phc_amd.env.tasks.vec_task import parse_task  # noqa: E402
from phc_amd.learning.amp_agent import IMAmpAgent  # noqa: E402
from phc_amd.learning.fast_ops import policy_sample  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = compose([f"env.num_envs={n}", "env.motion_file=synthetic:1:0", "+learning.params.config.hip_graph=True"])
task, env = parse_task(cfg)
agent = IMAmpAgent(env, cfg)
agent.init_train()
for _ in range(3):
    agent.train_epoch()
torch.cuda.synchronize()
SEG = ["reset_done + obs row", "normalise + actor/critic + sample", "env.step", "buffer writes", "next-value critic", "bookkeeping"]
acc = np.zeros(len(SEG))
e = agent.exp
net = agent.model.a2c_network
agent.set_eval()
reps = 3
wall = 0.0
with torch.no_grad(), agent.grads.shadow_scope():
    for rep in range(reps):
        evs = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        terminated_flags = torch.zeros(agent.num_actors, device=agent.device)
        for k in range(agent.horizon_length):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(SEG) + 1)]
            ev[0].record()
            task.reset_done()
            agent.obs = task.obs_buf
            e["obses"][k].copy_(agent.obs)
            ev[1].record()
            processed = agent._preproc_obs(agent.obs)
            with agent._autocast():
                mu, logstd = net.eval_actor(processed)
                value = net.eval_critic(processed)
            policy_sample(mu.contiguous(), value.contiguous(), (logstd[0] if logstd.dim() == 2 else logstd).float().contiguous(), agent.value_mean_std,
                          e["actions"][k], e["mus"][k], e["sigmas"][k], e["neglogpacs"][k], e["values"][k])
            ev[2].record()
            agent.obs, rewards, agent.dones, infos = agent.vec_env.step(e["actions"][k])
            ev[3].record()
            e["rewards"][k].copy_(rewards.unsqueeze(1))
            e["next_obses"][k].copy_(agent.obs)
            e["dones"][k].copy_(agent.dones)
            e["amp_obs"][k].copy_(infos["amp_obs"])
            ev[4].record()
            terminated = infos["terminate"].float()
            with agent._autocast():
                value = net.eval_critic(agent._preproc_obs(agent.obs))
            policy_sample(None, value.contiguous(), None, agent.value_mean_std, None, None, None, None, e["next_values"][k], mask=terminated)
            ev[5].record()
            terminated_flags += terminated
            rr = infos["reward_raw"].mean(dim=0)
            agent.current_rewards += rewards.unsqueeze(1)
            agent.current_lengths += 1
            not_dones = 1.0 - agent.dones.float()
            agent.current_rewards = agent.current_rewards * not_dones.unsqueeze(1)
            agent.current_lengths = agent.current_lengths * not_dones
            ev[6].record()
            evs.append(ev)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall += time.perf_counter() - t0
        for ev in evs:
            for i in range(len(SEG)):
                acc[i] += ev[i].elapsed_time(ev[i + 1])
        print(f"rep {rep}: host issue time {t_issue * 1e3:.2f} ms, wall {1e3 * (time.perf_counter() - t0):.2f} ms")
acc /= reps
print(f"rollout of {agent.horizon_length} steps, {n} envs: wall {wall / reps * 1e3:.2f} ms; event time per segment (ms per rollout, includes launch gaps):")
for nm, v in zip(SEG, acc):
    print(f"  {nm:36s} {v:7.3f}  ({v / agent.horizon_length * 1e3:6.1f} us / step)")
print(f"  {'sum':36s} {acc.sum():7.3f}")
