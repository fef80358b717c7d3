"""TEST INFRASTRUCTURE ONLY -- import shim for the read-only Python reference.

Makes the *unmodified* reference modules under ``/root/reference`` importable in
the build container (no isaacgym / rl_games / smpl_sim / hydra ... installed) so
that ``oracle/gen_golden.py`` can run the reference's own functions and dump
golden vectors into ``tests/golden``.  Nothing in the product (``phc_amd``),
``bench.py`` or the ``-m gpu`` tests may import this file: ``/root/reference``
does not exist on the GPU box.

What is stubbed:
  * ``isaacgym``           -> MagicMock, except ``isaacgym.torch_utils`` which is
                              the reference's own ``phc/utils/isaacgym_torch_utils.py``
  * ``easydict.EasyDict``  -> 15-line attribute dict
  * ``smpl_sim.utils.torch_ext.to_torch`` -> tensor passthrough / from_numpy
  * ``rl_games``           -> oracle/rl_games_stub.py: a real ObjectFactory, torch_ext.policy_kl, ModelA2CContinuousLogStd and
                              EMPTY agent / player base classes, so that the reference's learner modules import and their
                              method bodies can be called on `__new__`-made instances (oracle/gen_golden_learner.py)
  * every other missing third-party package -> auto-mocked on import
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest.mock import MagicMock

# /root/reference in the build container; on the GPU box the travel archive `oracle/_ref/reference_modules.zip` that oracle/make_ref.py packed (git-ignored,
# rides along with the gpurun snapshot like a built .so): only the files oracle/time_reference.py and tests/test_reference_direct_gpu.py import.  Python
# imports straight from the archive (zipimport: "<zip>" and "<zip>/phc" are valid sys.path entries).
_TRAVEL_COPY = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "reference_modules.zip")
REFERENCE_ROOT = os.environ.get("PHC_REFERENCE_ROOT") or ("/root/reference" if os.path.isdir("/root/reference/phc") else _TRAVEL_COPY)


def is_archive():
    return REFERENCE_ROOT.endswith(".zip")


def available():
    """Is some copy of the reference importable here?"""
    return os.path.isfile(REFERENCE_ROOT) if is_archive() else os.path.isdir(os.path.join(REFERENCE_ROOT, "phc"))


def data_path(rel):
    """A real file path for a data file of the reference (an MJCF): the file itself, or its member of the travel archive extracted to a temp dir."""
    if not is_archive():
        return os.path.join(REFERENCE_ROOT, rel)
    import atexit
    import shutil
    import tempfile
    import zipfile
    if _EXTRACT_DIR[0] is None:   # ONE scratch directory per process, removed at exit (a fresh mkdtemp per call was never cleaned up)
        _EXTRACT_DIR[0] = tempfile.mkdtemp(prefix="phc_ref_")
        atexit.register(shutil.rmtree, _EXTRACT_DIR[0], ignore_errors=True)
    out = os.path.join(_EXTRACT_DIR[0], rel)
    if os.path.isfile(out):
        return out
    with zipfile.ZipFile(REFERENCE_ROOT) as z:
        return z.extract(rel, _EXTRACT_DIR[0])


_EXTRACT_DIR = [None]


_MOCK_TOPLEVEL = (
    "isaacgym", "smpl_sim", "smplx", "open3d", "imageio", "aiohttp", "cv2", "gym",
    "skimage", "lxml", "stl", "hydra", "omegaconf", "termcolor", "mujoco", "wandb",
    "tensorboardX", "pyvirtualdisplay", "rl_games", "vtk", "chumpy", "mujoco_py",
    "autograd", "pytorch3d", "ipdb", "gdown",
)


class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        elif isinstance(v, (list, tuple)):  # easydict converts dicts inside sequences too
            v = type(v)(_EasyDict(x) if isinstance(x, dict) and not isinstance(x, _EasyDict) else x for x in v)
        super().__setitem__(k, v)
        self.__dict__[k] = v   # the real EasyDict mirrors items as attributes; motion_lib_real.py:211,217 tests `in m.__dict__`

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


class _MockModule(MagicMock):
    """A MagicMock that import machinery accepts as a package."""
    __path__ = []
    __all__ = []


class _MockFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _MOCK_TOPLEVEL:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _MockModule(name=spec.name)
        m.__name__ = spec.name
        m.__spec__ = spec
        m.__loader__ = self
        return m

    def exec_module(self, module):
        pass


_installed = False


def install():
    """Idempotently put the reference on sys.path and register all stubs."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}; the shim needs /root/reference (build container) or the travel archive oracle/_ref/reference_modules.zip (oracle/make_ref.py)")
    import numpy as np
    import torch

    for p in (os.path.join(REFERENCE_ROOT, "phc"), REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)

    ed = types.ModuleType("easydict")
    ed.EasyDict = _EasyDict
    sys.modules.setdefault("easydict", ed)

    # rl_games: real minimal stand-ins for the pieces the reference's learner executes (oracle/rl_games_stub.py); every other
    # rl_games.* submodule stays an auto-mock
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import rl_games_stub
    rl_games_stub.register()

    sys.meta_path.append(_MockFinder())

    # isaacgym.torch_utils must be the real (reference) quaternion library
    itu = importlib.import_module("phc.utils.isaacgym_torch_utils")
    importlib.import_module("isaacgym")
    sys.modules["isaacgym.torch_utils"] = itu
    sys.modules["isaacgym"].torch_utils = itu

    def to_torch(x, *a, **k):
        return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x))

    te = importlib.import_module("smpl_sim.utils.torch_ext")
    te.to_torch = to_torch

    # smpl_sim.poselib is the same poselib the reference vendors at /root/reference/poselib (torch_humanoid_batch.py:18)
    try:
        pr = importlib.import_module("poselib.poselib.core.rotation3d")
        sys.modules["smpl_sim.poselib.core.rotation3d"] = pr
        importlib.import_module("smpl_sim.poselib.core").rotation3d = pr
    except Exception:
        pass
    _install_lxml_stand_in()

    # numpy-2 removed aliases that the reference still uses at import/run time
    for name, val in (("Inf", np.inf), ("int", int), ("float", float), ("bool", bool)):
        if not hasattr(np, name):
            setattr(np, name, val)
    _installed = True


def _install_lxml_stand_in():
    """`lxml` is not installed here; Humanoid_Batch (torch_humanoid_batch.py:19,39-45) only needs parse / find / findall /
    getchildren, which a PRIVATE pure-python copy of xml.etree.ElementTree provides (its Element is a Python class, so the
    lxml-only `getchildren()` can be added without touching the interpreter's own ElementTree)."""
    import importlib.util
    saved = sys.modules.get("_elementtree", "absent")
    sys.modules["_elementtree"] = None
    try:
        spec = importlib.util.find_spec("xml.etree.ElementTree")
        pyet = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(pyet)
    finally:
        if saved == "absent":
            del sys.modules["_elementtree"]
        else:
            sys.modules["_elementtree"] = saved
    pyet.Element.getchildren = lambda self: list(self)
    m = types.ModuleType("lxml.etree")
    m.XMLParser = lambda **kw: pyet.XMLParser()
    m.parse = lambda f, parser=None: pyet.parse(f)
    m.ElementTree, m.Element, m.SubElement = pyet.ElementTree, pyet.Element, pyet.SubElement
    pkg = types.ModuleType("lxml")
    pkg.__path__ = []
    pkg.etree = m
    sys.modules["lxml"] = pkg
    sys.modules["lxml.etree"] = m


def ref_module(name):
    install()
    return importlib.import_module(name)
