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
    nwg = (n + 1) // 2
    buf = (C.c_ulonglong * (10 * nwg))()
    raw.phc_debug_profile_wg(buf, nwg)
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


if __name__ == "__main__":
    main()
