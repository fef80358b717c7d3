"""Per-epoch PPO timings from a cold start (dev tool): does the update time drift as the GPU warms up?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from phc_amd.config import compose
from phc_amd.env.tasks.vec_task import parse_task
from phc_amd.learning.amp_agent import IMAmpAgent
cfg = compose(["env.num_envs=4096", "env.motion_file=synthetic:1:0", "+learning.params.config.hip_graph=True"])
task, env = parse_task(cfg)
env.reset()
agent = IMAmpAgent(env, cfg)
agent.init_train()
t0 = time.perf_counter()
for e in range(40):
    info = agent.train_epoch()
    if e < 6 or e % 4 == 3:
        print(f"epoch {e + 1:3d}  t={time.perf_counter() - t0:6.2f}s  play {info['play_time'] * 1e3:6.1f} ms  update {info['update_time'] * 1e3:6.1f} ms", flush=True)
