"""dev tool: where one PPO optimizer step spends its GPU time -- per phase (record_function ranges) device time and kernel count.
python scripts/profile_step.py [num_envs] [learning]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile, record_function
from phc_amd.config import compose
from phc_amd.env.tasks.vec_task import parse_task
from phc_amd.learning import amp_agent as A

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
learning = sys.argv[2] if len(sys.argv) > 2 else "im"
cfg = compose([f"env.num_envs={n}", "env.motion_file=synthetic:1:0", f"learning={learning}"])
task, env = parse_task(cfg)
agent = A.IMAmpAgent(env, cfg)
agent.init_train()
agent.train_epoch()
torch.cuda.synchronize()


def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        with record_function(label):
            return f(*a, **k)
    setattr(obj, name, g)


wrap(agent, "_get_item", "PH:get_item")
wrap(agent, "_fwd_bwd", "PH:fwd_bwd")
wrap(agent, "_clip_and_step", "PH:clip_step")
wrap(agent, "_preproc_obs", "PH:preproc")
wrap(agent, "_preproc_amp_obs", "PH:preproc")
wrap(agent, "_disc_loss", "PH:disc_loss")
wrap(agent, "play_steps", "PH:rollout")
wrap(agent, "prepare_dataset", "PH:prepare_dataset")
wrap(agent.model, "forward", "PH:model_forward") if hasattr(agent.model, "forward") else None
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    info = agent.train_epoch()
    torch.cuda.synchronize()
print({k: round(v, 4) if isinstance(v, float) else v for k, v in info.items() if k in ("play_time", "update_time", "total_fps")})
ev = prof.events()
ranges = [(e.name, e.time_range.start, e.time_range.end) for e in ev if e.name.startswith("PH:")]
# device kernels attributed to the innermost enclosing PH range of the CPU op that launched them (via correlation: kernels carry
# the launching op's time range in e.cpu_parent chain is not exposed -> use launch timestamps of the cuda runtime events)
import collections
tot, cnt = collections.Counter(), collections.Counter()
launches = [e for e in ev if e.device_type.name == "CPU" and e.kernels]
for e in launches:
    t = e.time_range.start
    lab, best = "other", None
    for name, s, f in ranges:
        if s <= t <= f and (best is None or f - s < best):
            lab, best = name, f - s
    for k in e.kernels:
        tot[lab] += k.duration
        cnt[lab] += 1
for lab in sorted(tot, key=lambda l: -tot[l]):
    print(f"{lab:22s} {tot[lab] / 1e3:9.2f} ms  {cnt[lab]:6d} kernels  {tot[lab] / max(cnt[lab], 1):7.1f} us avg")
print("total device time %.2f ms in %d kernels" % (sum(tot.values()) / 1e3, sum(cnt.values())))
