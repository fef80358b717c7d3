"""Build libphc_amd.so (HIP, gfx950) in-tree with hipcc.  `python -m phc_amd.build`."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libphc_amd.so")
# source -> extra flags.
#   task kernels: parity with the reference's torch ops is pinned at 1e-5 and torch does not fuse multiply-add, so
#     -ffp-contract=off (e.g. sqrt(1 - w*w) in quat_to_angle_axis is cancellation-prone: a contracted fma moves exp-map
#     outputs by 1e-3); -fno-slp-vectorize avoids v_pk_* register marshalling (reset 68 -> 49 us, post-physics 35 -> 27 us).
#   stepper: see the header of phc_sim.hip (-ffast-math -fno-slp-vectorize: 158 -> 109 us).
#   learner kernels: bandwidth-bound passes; IEEE division / no contraction so the normalised values equal torch's.
SOURCES = {"phc_kernels.hip": ["-fno-slp-vectorize", "-ffp-contract=off"], "phc_sim.hip": ["-ffast-math", "-fno-slp-vectorize"],
           "phc_learn.hip": ["-ffp-contract=off"]}
HEADERS = ["phc_math.h", "phc_task.h", "phc_im.h", "phc_aba.h", os.path.join("..", "..", "include", "phc_amd.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in list(SOURCES) + HEADERS)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into phc_amd/libphc_amd.so."""
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    common = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment"]
    objs = []
    os.makedirs(os.path.join(HERE, "_obj"), exist_ok=True)
    for src, extra in SOURCES.items():
        obj = os.path.join(HERE, "_obj", src.replace(".hip", ".o"))
        cmd = [hipcc, *common, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n" + r.stdout + r.stderr)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
