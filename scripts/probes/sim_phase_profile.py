"""Where do the cycles of the stepper go?  Builds an instrumented copy of the library (-DPHC_SIM_PROFILE: s_memtime deltas per
phase, accumulated per wavefront) next to the product one, runs the bench's env step with it and prints the phase table.

    python scripts/sim_phase_profile.py build      # here (hipcc cross-compiles): phc_amd/_obj/libphc_amd_prof.so
    python scripts/sim_phase_profile.py run        # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PROF = os.path.join(ROOT, "phc_amd", "_obj", "libphc_amd_prof.so")
NAMES = ["load + initial FK", "body-body contact (capsules, pairs, collect)", "velocity products + per-body init (ground contact, drive)",
         "drive exchange of the re-rooted tree", "backward sweep", "acceleration sweep", "joint integration", "kinematics by pointer jumping",
         "store + publish"]


def build():
    from phc_amd import build as b
    hipcc = b._hipcc()
    objs = []
    os.makedirs(os.path.dirname(PROF), exist_ok=True)
    for src, extra in b.SOURCES.items():
        obj = os.path.join(ROOT, "phc_amd", "_obj", "prof_" + src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-DPHC_SIM_PROFILE", *extra, "-c", os.path.join(b.CSRC, src), "-o", obj]
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", PROF])
    print(PROF)


def run():
    os.environ["PHC_AMD_LIB"] = PROF
    import torch
    from phc_amd import _lib
    from phc_amd.config import compose
    from phc_amd.env.tasks.vec_task import parse_task
    lib = _lib.load()
    raw = C.CDLL(PROF)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    extra = sys.argv[3:]
    torch.manual_seed(0)
    task, env = parse_task(compose([f"env.num_envs={n}", "env.motion_file=synthetic:1:0"] + extra))
    env.reset()
    a = (torch.rand(n, task.num_actions, device=task.device) * 2 - 1) * 0.1
    for _ in range(20):
        task.reset_done(); env.step(a)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 16)()
    raw.phc_debug_profile(buf, 1)
    steps = 100
    for _ in range(steps):
        task.reset_done(); env.step(a)
    torch.cuda.synchronize()
    raw.phc_debug_profile(buf, 0)
    waves = buf[15]
    tot = sum(buf[i] for i in range(9))
    print(f"{n} envs, {steps} steps, {waves} wavefront executions; s_memtime cycles per wavefront per launch (4 sub-steps):")
    for i, nm in enumerate(NAMES):
        print(f"  {nm:52s} {buf[i] / waves:10.0f}  {100.0 * buf[i] / tot:5.1f} %")
    print(f"  {'total':52s} {tot / waves:10.0f}")


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
