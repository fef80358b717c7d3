"""Which torch op launches which kernel in ONE optimizer step (dev tool): eager `_fwd_bwd` of one minibatch under torch.profiler, ops in launch
order with their kernels.  python scripts/probes/profile_update_ops.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from phc_amd.config import compose
from phc_amd.env.tasks.vec_task import parse_task
from phc_amd.learning.amp_agent import IMAmpAgent

cfg = compose(["env.num_envs=4096", "env.motion_file=synthetic:1:0", "+learning.params.config.hip_graph=True"])
task, env = parse_task(cfg)
agent = IMAmpAgent(env, cfg)
agent.init_train()
agent.train_epoch()
agent.train_epoch()
torch.cuda.synchronize()
agent.set_train()
agent._graph_static_dataset()
body = lambda: agent._graph_step_body()
for _ in range(2):
    body()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    body()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.kernels]
evs.sort(key=lambda e: e.time_range.start)
tot = 0.0
for e in evs:
    ks = ", ".join(f"{k.name.split('(')[0].split('<')[0][-40:]}:{k.duration:.1f}" for k in e.kernels)
    tot += sum(k.duration for k in e.kernels)
    print(f"{e.name[:44]:44s} {str(e.input_shapes)[:70]:70s} {ks}")
print("total kernel us", tot, "ops", len(evs))
