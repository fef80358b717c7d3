"""dev tool: where does the AMP history filled from the per-frame table differ from the full builds?"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from phc_amd.config import compose  # noqa: E402
from phc_amd.env.tasks.vec_task import parse_task  # noqa: E402


def make_task(num_envs, motion="synthetic:3:1", seed=0, **over):
    torch.manual_seed(seed)
    return parse_task(compose([f"env.num_envs={num_envs}", f"env.motion_file={motion}"] + [f"{k}={v}" for k, v in over.items()]))


ta, ea = make_task(256, motion="synthetic:5:1")
tb, eb = make_task(256, motion="synthetic:5:1", **{"+env.amp_ref_table": False})
torch.manual_seed(3); ea.reset()
torch.manual_seed(3); eb.reset()
A = ta._amp_obs_buf.cpu().numpy(); B = tb._amp_obs_buf.cpu().numpy()
d = np.abs(A - B)
print("max diff", d.max(), "mean", d.mean(), "frac > 2e-6", (d > 2e-6).mean())
print("per history step k max:", d.max(axis=(0, 2)))
print("per column block max: root", d[:, :, :13].max(), "joints", d[:, :, 13:184].max(), "key", d[:, :, 184:].max())
cols = d.max(axis=(0, 1)); print("worst columns", np.argsort(-cols)[:12], cols[np.argsort(-cols)[:12]])
e, k, c = np.unravel_index(d.argmax(), d.shape)
print("worst element env", e, "k", k, "col", c, A[e, k, c], B[e, k, c])
lib = ta._motion_lib
f = np.float32
def frame_blend(t, len_, nf, dt):
    t = f(t); phase = f(t / len_); phase = min(max(phase, f(0)), f(1))
    if t < 0: t = f(0)
    prod = f(phase * f(nf - 1)); i0 = int(prod); sub = f(f(i0) * dt); bl = f(f(t - sub) / dt)
    return i0, float(min(max(bl, f(0)), f(1)))
t0 = ta._motion_start_times.cpu().numpy(); lens = lib._motion_lengths.cpu().numpy(); nfs = lib._motion_num_frames.cpu().numpy(); dts = lib._motion_dt.cpu().numpy()
mids = ta._sampled_motion_ids.cpu().numpy()
dt = f(ta.dt)
off = np.zeros((256, 10)); bls = np.zeros((256, 10))
for env in range(256):
    m = mids[env]
    for kk in range(10):
        t = f(t0[env] + f(f(-dt) * f(kk)))
        i0, b = frame_blend(t, lens[m], int(nfs[m]), dts[m]); bls[env, kk] = b; off[env, kk] = min(b, 1 - b)
print("task dt", repr(ta.dt), "blend distance from 0/1: max", off.max(), "frac > 1e-4", (off > 1e-4).mean(), "frac > 1e-5", (off > 1e-5).mean())
print("worst element: start", t0[e], "blend", bls[e, k], "len", lens[mids[e]], "nf", nfs[mids[e]])
per_env = d.max(axis=(1, 2))
print("corr: envs with large diff vs their max blend offset:", [(int(i), float(per_env[i]), float(off[i].max())) for i in np.argsort(-per_env)[:6]])
