"""One wavefront's timeline through sub-step 1 of the stepper (instrumented library, s_memtime stamps at the phase / level boundaries;
~40 stamps of ~50 cycles each on a ~35 k cycle sub-step).     python scripts/probes/sim_timeline.py [num_envs] [block]"""
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

NAMES = {1: "sub-step start", 2: "after body-body contact", 3: "after velocity products", 4: "after per-body init", 5: "after drive exchange",
         6: "after acceleration sweep (+ finish)", 7: "after joint integration", 8: "after kinematics"}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    block = int(sys.argv[2]) if len(sys.argv) > 2 else 777
    torch.manual_seed(0)
    task, env = parse_task(compose([f"env.num_envs={n}", "env.motion_file=synthetic:1:0"] + sys.argv[3:]))
    raw = C.CDLL(PROF)
    raw.phc_debug_set_skip((1 << 15) | int(os.environ.get("PHC_TL_SKIP", "0"), 0))
    env.reset()
    a = (torch.rand(n, task.num_actions, device=task.device) * 2 - 1) * 0.1
    for _ in range(10):
        task.reset_done(); env.step(a)
    torch.cuda.synchronize()
    if os.environ.get("PHC_TL_LIFT"):      # lift every humanoid off the ground: the same sub-step without a single ground contact
        task._root_states[:, 2] += float(os.environ["PHC_TL_LIFT"])
    buf = (C.c_ulonglong * 512)()
    raw.phc_debug_timeline(buf, block)
    task.reset_done(); env.step(a)
    torch.cuda.synchronize()
    raw.phc_debug_timeline(buf, -1)
    pts = [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(256) if buf[2 * i]]
    print(f"{n} envs, workgroup {block}: {len(pts)} stamps, sub-step 1 = {pts[-1][1] - pts[0][1]} cycles")
    for (i0, t0), (i1, t1) in zip(pts[:-1], pts[1:]):
        nm = NAMES.get(i1) or (f"backward level {i1 - 100}: shuffles done" if 100 <= i1 < 120 else f"backward level {i1 - 120}: bodies done" if 120 <= i1 < 140
                               else f"acceleration level {i1 - 140} done" if 140 <= i1 < 160 else f"kinematics jump step {i1 - 160} done")
        print(f"  {t1 - t0:7d}  {nm}")


if __name__ == "__main__":
    main()
