"""The launch lasts as long as its SLOWEST wavefront: per-wavefront phase cycles of one stepper launch (instrumented library, scripts/probes/build_prof.sh),
their spread over the launch, and the phase table of the slowest wavefronts next to the median one.    python scripts/probes/sim_wave_spread.py [num_envs]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PROF = os.path.join(ROOT, "phc_amd", "_obj", "libphc_amd_prof.so")
os.environ["PHC_AMD_LIB"] = PROF
import torch  # noqa: E402
from phc_amd.config import compose  # noqa: E402
from phc_amd.env.tasks.vec_task import parse_task  # noqa: E402

NAMES = ["load + initial FK", "body-body contact", "velocity products + per-body init", "drive exchange", "backward sweep", "acceleration sweep",
         "joint integration", "kinematics", "store + publish"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    torch.manual_seed(0)
    task, env = parse_task(compose([f"env.num_envs={n}", "env.motion_file=synthetic:1:0"] + sys.argv[2:]))
    raw = C.CDLL(PROF)
    env.reset()
    a = (torch.rand(n, task.num_actions, device=task.device) * 2 - 1) * 0.1
    for _ in range(70):
        task.reset_done(); env.step(a)
    torch.cuda.synchronize()
    # round 5: how long does the INSTRUMENTED launch itself last (HIP events), next to the product library's launch on the same state?
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    task.reset_done(); task.pre_physics_step(a)
    ev[0].record(); task._physics_step(); ev[1].record()
    task.post_physics_step()
    torch.cuda.synchronize()
    launch_us = ev[0].elapsed_time(ev[1]) * 1e3
    nwg = (n + 1) // 2
    buf = (C.c_ulonglong * (10 * nwg))()
    raw.phc_debug_profile_wg(buf, nwg)
    where = (C.c_ulonglong * (2 * nwg))()
    raw.phc_debug_profile_where(where, nwg)
    wh = np.array(where, dtype=np.uint64).reshape(nwg, 2)
    t = np.array(buf, dtype=np.float64).reshape(nwg, 10)[:, :9]
    tot = t.sum(1)
    order = np.argsort(tot)
    q = lambda p: tot[order[int(p * (nwg - 1))]]
    print(f"{n} envs = {nwg} wavefronts; cycles per wavefront and launch: min {tot.min():.0f}  median {q(0.5):.0f}  p90 {q(0.9):.0f}  p99 {q(0.99):.0f}  max {tot.max():.0f}")
    cols = {"median": order[nwg // 2], "p90": order[int(0.9 * (nwg - 1))], "p99": order[int(0.99 * (nwg - 1))], "max": order[-1]}
    print(f"{'phase':36s}" + "".join(f"{k:>10s}" for k in cols) + f"{'mean':>10s}{'max/wf':>10s}")
    for i, nm in enumerate(NAMES):
        print(f"{nm:36s}" + "".join(f"{t[w, i]:10.0f}" for w in cols.values()) + f"{t[:, i].mean():10.0f}{t[:, i].max():10.0f}")
    prog = task.progress_buf.cpu().numpy()
    fz = (task._contact_forces[..., 2].abs().sum(-1) > 0).cpu().numpy()
    for k, w in cols.items():
        e = [2 * w, min(2 * w + 1, n - 1)]
        print(f"{k}: workgroup {w}, envs {e}, progress {prog[e].tolist()}, ground contact {fz[e].tolist()}")
    # ---- round 5 (VERDICT r4 "weak" 8a): where and when did the slow wavefronts run? ----
    start = wh[:, 0].astype(np.float64)
    start -= start.min()
    end = start + tot
    xcc = (wh[:, 1] >> np.uint64(32)).astype(np.int64) & 0xf
    hw = (wh[:, 1] & np.uint64(0xffffffff)).astype(np.int64)
    cu, se, simd = (hw >> 8) & 0xf, (hw >> 13) & 0x7, (hw >> 4) & 0x3     # gfx9 HW_ID: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13
    print(f"instrumented launch: {launch_us:.1f} us by HIP events; last wavefront ends {end.max():.0f} cycles after the first one starts "
          f"(= {end.max() / launch_us / 1e3:.2f} GHz if the two agree); wavefront starts spread over {start.max():.0f} cycles")
    slow = tot > 2.0 * np.median(tot)
    print(f"wavefronts slower than 2 x the median: {int(slow.sum())} of {nwg}; with ground contact in either env: "
          f"{int((slow & (fz[0::2][:nwg] | fz[1::2][:nwg])).sum())}; ground-contact wavefronts in all: {int((fz[0::2][:nwg] | fz[1::2][:nwg]).sum())}")
    print("per XCD: wavefronts, median total, slow ones, median start")
    for x in sorted(set(xcc.tolist())):
        m = xcc == x
        print(f"  xcc {x}: {int(m.sum()):5d}  {np.median(tot[m]):9.0f}  {int((slow & m).sum()):4d}  {np.median(start[m]):9.0f}")
    key = xcc * 1000 + se * 100 + cu
    worst = sorted(((int(slow[key == k].sum()), int((key == k).sum()), int(k)) for k in set(key.tolist())), reverse=True)[:8]
    print("CUs with the most slow wavefronts (slow, resident, xcc*1000 + se*100 + cu):", worst)
    late = start > np.percentile(start, 90)
    print(f"correlation of a wavefront's total with its start time: {np.corrcoef(start, tot)[0, 1]:.2f}; late starters (last 10 %): median total {np.median(tot[late]):.0f}")
    ph = t[:, 2]
    print("per-body init cycles by number of envs of the wavefront in ground contact:",
          {int(k): (int(((fz[0::2][:nwg].astype(int) + fz[1::2][:nwg].astype(int)) == k).sum()), float(np.median(ph[(fz[0::2][:nwg].astype(int) + fz[1::2][:nwg].astype(int)) == k]))) for k in (0, 1, 2)})


if __name__ == "__main__":
    main()
