"""Breakdown of one PPO epoch with torch.profiler (dev tool): python scripts/profile_ppo.py [num_envs]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from phc_amd.config import compose
from phc_amd.env.tasks.vec_task import parse_task
from phc_amd.learning.amp_agent import IMAmpAgent

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = compose([f"env.num_envs={n}", "env.motion_file=synthetic:1:0"])
task, env = parse_task(cfg)
agent = IMAmpAgent(env, cfg)
agent.init_train()
agent.train_epoch()
torch.cuda.synchronize()
# GEMM alignment probe
for K in (934, 960, 1960, 1984):
    a = torch.randn(16384, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(1024, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): torch.nn.functional.linear(a, w)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): torch.nn.functional.linear(a, w)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print(f"linear 16384x{K} -> 1024 bf16: {dt*1e6:.1f} us  {2*16384*K*1024/dt/1e12:.1f} TFLOP/s")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    info = agent.train_epoch()
    torch.cuda.synchronize()
print({k: round(v, 4) if isinstance(v, float) else v for k, v in info.items() if k in ("play_time", "update_time", "total_fps")})
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
