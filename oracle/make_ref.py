"""TEST INFRASTRUCTURE ONLY -- a travel archive of the reference modules the GPU-box legs need.

The reference is a Python checkout under /root/reference that exists in the build container only; the GPU box gets a snapshot of /root/repo.  This script
packs EXACTLY the reference files that `oracle/time_reference.py` and `tests/test_reference_direct_gpu.py` import (found by importing them through
`oracle/ref_shim.py` in a child interpreter and reading `sys.modules`), plus the one MJCF they parse, into ONE archive `oracle/_ref/reference_modules.zip` --
a built artefact like `oracle/_build/libphc_hostemu.so`: `oracle/_ref/` is listed in `.gitignore` (it never enters the history and holds no source tree;
it only rides along with `gpurun` snapshots).  Python imports straight from the archive (zipimport).  `__graft_entry__.build()` runs this whenever
/root/reference is present.  `ref_shim.REFERENCE_ROOT` falls back to the archive where /root/reference does not exist.

    python oracle/make_ref.py            # -> oracle/_ref/reference_modules.zip + MANIFEST.txt
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
    import zipfile
    if os.path.isdir(DST):
        shutil.rmtree(DST)
    os.makedirs(DST)
    members = set(files)
    for rel in files:     # package markers along the way (the reference's own, where it has them)
        d = os.path.dirname(rel)
        while d:
            if os.path.exists(os.path.join(SRC, d, "__init__.py")):
                members.add(os.path.join(d, "__init__.py"))
            d = os.path.dirname(d)
    n = 0
    with zipfile.ZipFile(os.path.join(DST, "reference_modules.zip"), "w", zipfile.ZIP_DEFLATED) as z:
        for rel in sorted(members):
            z.write(os.path.join(SRC, rel), rel)
            n += os.path.getsize(os.path.join(SRC, rel))
        # directories the reference imports as implicit namespace packages (no __init__.py of their own, e.g. phc/env/util): zipimport needs a marker
        dirs = set()
        for rel in members:
            d = os.path.dirname(rel)
            while d:
                dirs.add(d)
                d = os.path.dirname(d)
        for d in sorted(dirs):
            if os.path.join(d, "__init__.py") not in members and any(m.endswith(".py") and os.path.dirname(m) == d for m in members):
                z.writestr(os.path.join(d, "__init__.py"), "")
    with open(os.path.join(DST, "MANIFEST.txt"), "w") as f:
        f.write("# members of reference_modules.zip (oracle/make_ref.py): the reference files the GPU-box legs import; git-ignored, never committed\n" + "\n".join(sorted(members)) + "\n")
    print(f"make_ref: {len(members)} files, {n / 1024:.0f} KiB -> {os.path.join(DST, 'reference_modules.zip')} ({os.path.getsize(os.path.join(DST, 'reference_modules.zip')) / 1024:.0f} KiB)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
