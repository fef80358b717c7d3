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


def build():
    deps = [SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
           [os.path.join(os.path.dirname(HERE), "include", "phc_amd.h")]
    if os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", SRC, "-o", OUT], check=True)
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
