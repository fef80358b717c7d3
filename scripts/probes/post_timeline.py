"""One env's timeline through k_im_post_physics (instrumented library; s_memtime + s_waitcnt(0) at twelve points: the SERIALISED cost of each section).
    python scripts/probes/post_timeline.py [num_envs]"""
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
    raw.phc_debug_post_timeline.argtypes = [C.c_void_p, C.c_longlong]
    env.reset()
    a = (torch.rand(n, task.num_actions, device=task.device) * 2 - 1) * 0.1
    for _ in range(10):
        task.reset_done(); env.step(a)
    torch.cuda.synchronize()
    names = {1: "requests at the top + per-env context (prologue)", 2: "frame indices of both lookups", 3: "frame-record requests (drained here by the stamp)",
             4: "self observation", 5: "AMP observation frame", 6: "blends + reward partials + power", 7: "task observation", 8: "ref_* side buffers",
             9: "(end of the lane function)", 10: "AMP window shift (1 step in 10) + 32-lane sums", 11: "finalize (lane 0)"}
    for e in (5, 1000, n - 3):
        buf = (C.c_ulonglong * 32)()
        raw.phc_debug_post_timeline(None, e)
        task.reset_done(); env.step(a)
        torch.cuda.synchronize()
        raw.phc_debug_post_timeline(buf, -1)
        t = [int(buf[i]) for i in range(12)]
        print(f"env {e}: total {t[11] - t[0]} cycles")
        for i in range(1, 12):
            print(f"   {t[i] - t[i - 1]:7d}  {names[i]}")


if __name__ == "__main__":
    main()
