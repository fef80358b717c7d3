"""TEST INFRASTRUCTURE ONLY -- a travel copy of the reference modules the GPU-box legs need.

The reference is a Python checkout under /root/reference that exists in the build container only; the GPU box gets a snapshot of /root/repo.  This script
copies EXACTLY the reference files that `oracle/time_reference.py` and `tests/test_reference_direct_gpu.py` import (found by importing them through
`oracle/ref_shim.py` in a child interpreter and reading `sys.modules`), plus the one MJCF they parse, into `oracle/_ref/` -- which is listed in `.gitignore`
(it never enters the history; like a built `.so` it only rides along with `gpurun` snapshots).  `__graft_entry__.build()` runs it whenever
/root/reference is present.  `ref_shim.REFERENCE_ROOT` falls back to `oracle/_ref` where /root/reference does not exist.

    python oracle/make_ref.py            # -> oracle/_ref/{phc,poselib}/... + MANIFEST.txt
"""
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.environ.get("PHC_REFERENCE_SOURCE", "/root/reference")
DST = os.path.join(HERE, "_ref")
MODULES = ["phc.env.tasks.humanoid_im", "phc.env.tasks.humanoid", "phc.env.tasks.humanoid_amp", "phc.utils.flags", "phc.utils.motion_lib_base",
           "phc.utils.motion_lib_smpl", "phc.utils.torch_utils", "poselib.poselib.skeleton.skeleton3d", "poselib.poselib.core.rotation3d"]
DATA = ["phc/data/assets/mjcf/smpl_0_humanoid.xml"]

_PROBE = r"""
import json, os, sys
sys.path.insert(0, %(here)r)
os.environ["PHC_REFERENCE_ROOT"] = %(src)r
import ref_shim
ref_shim.install()
import importlib
for m in %(mods)r:
    importlib.import_module(m)
root = os.path.realpath(%(src)r) + os.sep
files = sorted({os.path.realpath(m.__file__) for m in list(sys.modules.values()) if getattr(m, "__file__", None) and os.path.realpath(m.__file__).startswith(root)})
print("MANIFEST" + json.dumps([f[len(root):] for f in files]))
"""


def main():
    if not os.path.isdir(SRC):
        print(f"make_ref: {SRC} not present -- nothing to do (the GPU box uses the copy that travelled with the snapshot)")
        return 0
    out = subprocess.run([sys.executable, "-c", _PROBE % dict(here=HERE, src=SRC, mods=MODULES)], capture_output=True, text=True)
    line = next((l for l in out.stdout.splitlines() if l.startswith("MANIFEST")), None)
    if out.returncode != 0 or line is None:
        sys.stderr.write(out.stdout[-2000:] + out.stderr[-2000:])
        raise SystemExit("make_ref: importing the reference failed")
    files = json.loads(line[len("MANIFEST"):]) + DATA
    if os.path.isdir(DST):
        shutil.rmtree(DST)
    n = 0
    for rel in files:
        dst = os.path.join(DST, rel)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        shutil.copy2(os.path.join(SRC, rel), dst)
        n += os.path.getsize(dst)
        d = os.path.dirname(rel)
        while d:     # package markers along the way (the reference's own, where it has them)
            init = os.path.join(d, "__init__.py")
            if os.path.exists(os.path.join(SRC, init)) and not os.path.exists(os.path.join(DST, init)):
                shutil.copy2(os.path.join(SRC, init), os.path.join(DST, init))
            d = os.path.dirname(d)
    with open(os.path.join(DST, "MANIFEST.txt"), "w") as f:
        f.write("# travel copy of reference files for the GPU-box legs (oracle/make_ref.py); git-ignored, never committed\n" + "\n".join(files) + "\n")
    print(f"make_ref: {len(files)} files, {n / 1024:.0f} KiB -> {DST}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
