"""One lane group's timeline through k_im_reset (instrumented library; s_memtime + s_waitcnt(0) at five points).
    python scripts/probes/reset_timeline.py [num_envs]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PROF = os.path.join(ROOT, "phc_amd", "_obj", "libphc_amd_prof.so")
os.environ["PHC_AMD_LIB"] = PROF
import torch  # noqa: E402
from phc_amd.config import compose  # noqa: E402
from phc_amd.env.tasks.vec_task import parse_task  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    torch.manual_seed(0)
    task, env = parse_task(compose([f"env.num_envs={n}", "env.motion_file=synthetic:1:0"] + sys.argv[2:]))
    raw = C.CDLL(PROF)
    raw.phc_debug_set_skip(1 << 15)
    env.reset()
    a = (torch.rand(n, task.num_actions, device=task.device) * 2 - 1) * 0.1
    for _ in range(10):
        task.reset_done(); env.step(a)
    torch.cuda.synchronize()
    print(f"{n} envs, {int((task.reset_buf != 0).sum())} envs reset this step")
    names = ["list entry -> env", "motion id -> start time", "state + self obs (y = 0) / task obs (y = 1)", "AMP history frame (y >= 2)"]
    for r, k in ((3, 0), (200, 0), (3, 1), (200, 1), (3, 2), (3, 7), (200, 11)):
        buf = (C.c_ulonglong * 64)()
        raw.phc_debug_reset_timeline(buf, r * 16 + k)
        task.reset_done(); env.step(a)
        torch.cuda.synchronize()
        raw.phc_debug_reset_timeline(buf, -1)
        t = [int(buf[i]) for i in range(5)]
        if not all(t):
            print(f"group r={r} y={k}: not an active group this step {t}")
            continue
        print(f"group r={r} y={k}: total {t[4] - t[0]} cycles: " + ", ".join(f"{nm} {b - a}" for nm, a, b in zip(names, t[:-1], t[1:])))


if __name__ == "__main__":
    main()
