"""TEST INFRASTRUCTURE ONLY: ctypes access to oracle/hostemu -- the g++ (OpenMP) build of the very per-lane functions the
HIP kernels are made of (phc_amd/csrc/*.h are host+device).  Used by `pytest -m "not gpu"` to check the kernel math on a
machine without a GPU, and by bench.py's `cpu_baseline` leg as the CPU port of the path.  The product never loads it."""
import ctypes as C
import os
import subprocess

import numpy as np

from phc_amd import _lib as L
from phc_amd import abi
from phc_amd.model import load_model

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostemu", "hostemu.cpp")
OUT = os.path.join(HERE, "_build", "libphc_hostemu.so")
CSRC = os.path.join(os.path.dirname(HERE), "phc_amd", "csrc")


# PHC_HOSTEMU_SANITIZE=1 (scripts/sanitize_hostemu.sh): the same sources under AddressSanitizer + UndefinedBehaviorSanitizer -- the per-lane code of the kernels, their
# LDS-array stand-ins and the table indexing checked for out-of-bounds accesses and undefined arithmetic on the CPU (GPU sanitizers are not available on the pool).
SANITIZE = os.environ.get("PHC_HOSTEMU_SANITIZE", "") not in ("", "0")
if SANITIZE:
    OUT = os.path.join(HERE, "_build", "libphc_hostemu_asan.so")
_FLAGS = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] if SANITIZE else ["-O2"]


def build():
    deps = [SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(os.path.dirname(HERE), "include", "phc_amd.h")]
    if os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(["g++"] + _FLAGS + ["-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", SRC, "-o", OUT], check=True)
    return OUT


_emu = None


def emu():
    global _emu
    if _emu is None:
        lib = C.CDLL(build())
        vp, i32 = C.c_void_p, C.c_int32
        PS = C.POINTER
        lib.emu_motion_state.argtypes = [PS(L.MotionLib), i32] + [vp] * 14
        lib.emu_sample_time_interval.argtypes = [PS(L.MotionLib), i32, vp, vp, vp]
        lib.emu_im_post_physics.argtypes = [PS(L.Model), PS(L.MotionLib), PS(L.ImParams), PS(L.SimState), PS(L.ImBuffers)]
        lib.emu_im_reset.argtypes = [PS(L.Model), PS(L.MotionLib), PS(L.ImParams), PS(L.SimState), PS(L.ImBuffers), i32, vp, vp, i32]
        lib.emu_im_reset_from_state.argtypes = [PS(L.Model), PS(L.MotionLib), PS(L.ImParams), PS(L.SimState), PS(L.ImBuffers), i32, vp, i32]
        lib.emu_amp_obs_demo.argtypes = [PS(L.Model), PS(L.MotionLib), PS(L.ImParams), i32, vp, vp, vp]
        lib.emu_amp_ref_table.argtypes = [PS(L.Model), PS(L.MotionLib), PS(L.ImParams), C.c_int64, vp, vp]
        lib.emu_sim_step.argtypes = [PS(L.Model), PS(L.SimParams), PS(L.SimState), vp, vp, vp, vp, i32, i32]
        _emu = lib
    return _emu


def P(s):
    return C.byref(s)


def np_model(name="smpl_humanoid", kp_scale=1.0, kd_scale=1.0):
    m = load_model(name)
    from phc_amd.robots import apply_collision_filter
    apply_collision_filter(m, name.split("_")[0] if name in ("h1_humanoid", "g1_humanoid") else "smpl")
    ints, floats = m.pack(kp_scale, kd_scale)
    return m, abi.model_struct(ints, floats, m.num_bodies, m.num_dof, m.max_level, len(m.contact_body)), (ints, floats)


def np_motion_lib(lib):
    """dict with gts grs gvs gavs lrs dvs + per-motion arrays (numpy) -> (struct, keepalive)."""
    frames = abi.pack_frames(lib["gts"], lib["grs"], lib["gvs"], lib["gavs"], lib["lrs"], lib["dvs"])
    nb = lib["gts"].shape[1]
    keep = dict(frames=frames, ml=np.ascontiguousarray(lib["motion_lengths"], dtype=np.float32),
                mdt=np.ascontiguousarray(lib["motion_dt"], dtype=np.float32),
                mnf=np.ascontiguousarray(lib["motion_num_frames"], dtype=np.int64),
                ls=np.ascontiguousarray(lib["length_starts"], dtype=np.int64))
    s = abi.motion_lib_struct(frames, frames.shape[1], nb, keep["ml"], keep["mdt"], keep["mnf"], keep["ls"])
    return s, keep


# ---- the stepper at DOUBLE precision (oracle/hostemu/hostemu64.cpp): the exact-arithmetic statement of the kernel's recursion, incl. the lagged scheme -------------------
SRC64 = os.path.join(HERE, "hostemu", "hostemu64.cpp")
OUT64 = os.path.join(HERE, "_build", "libphc_hostemu64_asan.so" if SANITIZE else "libphc_hostemu64.so")


def build64():
    deps = [SRC, SRC64] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(os.path.dirname(HERE), "include", "phc_amd.h")]
    if os.path.exists(OUT64) and all(os.path.getmtime(d) <= os.path.getmtime(OUT64) for d in deps):
        return OUT64
    os.makedirs(os.path.dirname(OUT64), exist_ok=True)
    subprocess.run(["g++"] + _FLAGS + ["-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", SRC64, "-o", OUT64], check=True)
    return OUT64


def _clone64(cls):
    """The ctypes mirror of a struct of include/phc_amd.h as the `-Dfloat=double` build lays it out."""
    return type(cls.__name__ + "64", (C.Structure,), {"_fields_": [(n, C.c_double if t is C.c_float else t) for n, t in cls._fields_]})


Model64, SimParams64, SimState64 = _clone64(L.Model), _clone64(L.SimParams), _clone64(L.SimState)
_emu64 = None


def emu64():
    global _emu64
    if _emu64 is None:
        lib = C.CDLL(build64())
        lib.emu_sim_step.argtypes = [C.POINTER(Model64), C.POINTER(SimParams64), C.POINTER(SimState64), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        _emu64 = lib
    return _emu64


def _to64(s, cls64):
    d = cls64()
    for n, _ in s._fields_:
        setattr(d, n, getattr(s, n))
    return d


def sim_step_f64(model, params, root_states, dof_state, pd_target, num_sim_calls=2, kp_scale=1.0, kd_scale=1.0):
    """`phc_sim_step` for n envs through the double-precision build: `model` an ArticulationModel, `params` the fp32 `phc_sim_params_t` (abi.sim_params_struct),
    states as float arrays [n, 13] / [n, D, 2] / [n, D].  -> dict(root, dof, rbs, cf, df) of float64 arrays."""
    ints, floats = model.pack(kp_scale, kd_scale, float_dtype=np.float64)
    ms = _to64(abi.model_struct(ints, floats, model.num_bodies, model.num_dof, model.max_level, len(model.contact_body)), Model64)
    n, nb, nd = np.asarray(root_states).shape[0], model.num_bodies, model.num_dof
    a = dict(root=np.array(root_states, dtype=np.float64, order="C"), dof=np.array(dof_state, dtype=np.float64, order="C"), rbs=np.zeros((n, nb, 13)),
             cf=np.zeros((n, nb, 3)), df=np.zeros((n, nd)), pd=np.array(pd_target, dtype=np.float64, order="C"))
    sim = _to64(abi.sim_state_struct(n, a["root"], a["dof"], a["rbs"], a["cf"], a["df"], a["pd"]), SimState64)
    rc = emu64().emu_sim_step(C.byref(ms), C.byref(_to64(params, SimParams64)), C.byref(sim), None, None, None, None, int(num_sim_calls), 1)
    assert rc == 0, rc
    return a
