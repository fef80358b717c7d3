"""Two ways to run the path's kernels with identical arguments:

  HostEmu -- TEST INFRASTRUCTURE: g++ build of the per-lane functions (oracle/hostemu), numpy arrays, CPU.
  Hip     -- the product: libphc_amd.so through its C ABI, torch tensors in HBM, needs an MI355X.

Parity tests are written once and parametrised over the two (`hip` carries the `gpu` marker).
"""
import ctypes as C

import numpy as np
import pytest

from hostemu_util import P, emu
from phc_amd import abi


class HostEmu:
    name = "hostemu"

    def arr(self, x):
        return np.array(x, order="C", copy=True)  # always a private copy, like a host->device upload

    def zeros(self, shape, dtype=np.float32):
        return np.zeros(shape, dtype=dtype)

    def np(self, x):
        return np.asarray(x)

    def sync(self):
        pass

    def motion_state(self, lib, n, ids, times, off, rg_pos, rb_rot, bv, bav, dp, dv, i0, i1, bl, pe=None, re=None):
        return emu().emu_motion_state(P(lib), n, *[abi.ptr(a) for a in (ids, times, off, rg_pos, rb_rot, bv, bav, dp, dv, i0, i1, bl, pe, re)])

    def sample_time_interval(self, lib, n, ids, phase, out):
        return emu().emu_sample_time_interval(P(lib), n, abi.ptr(ids), abi.ptr(phase), abi.ptr(out))

    def im_post_physics(self, model, lib, prm, sim, buf):
        return emu().emu_im_post_physics(P(model), P(lib), P(prm), P(sim), P(buf))

    def im_reset(self, model, lib, prm, sim, buf, n, env_ids, phase, start_at_zero):
        return emu().emu_im_reset(P(model), P(lib), P(prm), P(sim), P(buf), n, abi.ptr(env_ids), abi.ptr(phase), start_at_zero)

    def im_reset_from_state(self, model, lib, prm, sim, buf, n, env_ids, fill_history):
        return emu().emu_im_reset_from_state(P(model), P(lib), P(prm), P(sim), P(buf), n, abi.ptr(env_ids), fill_history)

    def amp_obs_demo(self, model, lib, prm, n, ids, t0, out):
        return emu().emu_amp_obs_demo(P(model), P(lib), P(prm), n, abi.ptr(ids), abi.ptr(t0), abi.ptr(out))

    def amp_ref_table(self, model, lib, prm, num_frames, next_frame, table):
        return emu().emu_amp_ref_table(P(model), P(lib), P(prm), num_frames, abi.ptr(next_frame), abi.ptr(table))

    def sim_step(self, model, params, sim, actions, off, scale, freeze, num_sim_calls):
        return emu().emu_sim_step(P(model), P(params), P(sim), abi.ptr(actions), abi.ptr(off), abi.ptr(scale), abi.ptr(freeze), num_sim_calls, 1)

    def refresh_body_state(self, model, sim):
        prm = abi.sim_params_struct()
        return emu().emu_sim_step(P(model), P(prm), P(sim), None, None, None, None, 0, 0)


class Hip:
    name = "hip"

    def __init__(self):
        import torch
        from phc_amd import _lib
        self.torch = torch
        self.lib = _lib.load()
        assert torch.cuda.is_available(), "gpu-marked test started without a HIP device"
        self.dev = "cuda:0"

    def arr(self, x):
        return self.torch.from_numpy(np.ascontiguousarray(x)).to(self.dev)

    def zeros(self, shape, dtype=np.float32):
        return self.torch.zeros(shape, dtype=getattr(self.torch, np.dtype(dtype).name), device=self.dev)

    def np(self, x):
        return x.detach().cpu().numpy()

    def sync(self):
        self.torch.cuda.synchronize()

    def _s(self):
        return self.torch.cuda.current_stream().cuda_stream

    def motion_state(self, lib, n, ids, times, off, rg_pos, rb_rot, bv, bav, dp, dv, i0, i1, bl, pe=None, re=None):
        return self.lib.phc_motion_state(lib, n, *[abi.ptr(a) for a in (ids, times, off, rg_pos, rb_rot, bv, bav, dp, dv, i0, i1, bl, pe, re)], self._s())

    def sample_time_interval(self, lib, n, ids, phase, out):
        return self.lib.phc_sample_time_interval(lib, n, abi.ptr(ids), abi.ptr(phase), abi.ptr(out), self._s())

    def im_post_physics(self, model, lib, prm, sim, buf):
        return self.lib.phc_im_post_physics(model, lib, prm, sim, buf, self._s())

    def im_reset(self, model, lib, prm, sim, buf, n, env_ids, phase, start_at_zero):
        return self.lib.phc_im_reset(model, lib, prm, sim, buf, n, abi.ptr(env_ids), abi.ptr(phase), start_at_zero, self._s())

    def im_reset_from_state(self, model, lib, prm, sim, buf, n, env_ids, fill_history):
        return self.lib.phc_im_reset_from_state(model, lib, prm, sim, buf, n, abi.ptr(env_ids), fill_history, self._s())

    def amp_obs_demo(self, model, lib, prm, n, ids, t0, out):
        return self.lib.phc_amp_obs_demo(model, lib, prm, n, abi.ptr(ids), abi.ptr(t0), abi.ptr(out), self._s())

    def amp_ref_table(self, model, lib, prm, num_frames, next_frame, table):
        return self.lib.phc_amp_ref_table(model, lib, prm, num_frames, abi.ptr(next_frame), abi.ptr(table), self._s())

    def sim_step(self, model, params, sim, actions, off, scale, freeze, num_sim_calls):
        return self.lib.phc_sim_step(model, params, sim, abi.ptr(actions), abi.ptr(off), abi.ptr(scale), abi.ptr(freeze), num_sim_calls, self._s())

    def refresh_body_state(self, model, sim):
        return self.lib.phc_refresh_body_state(model, sim, self._s())


_CACHE = {}


def get_backend(name):
    if name not in _CACHE:
        _CACHE[name] = HostEmu() if name == "hostemu" else Hip()
    return _CACHE[name]


BACKENDS = ["hostemu", pytest.param("hip", marks=pytest.mark.gpu)]


def model_on(be, name="smpl_humanoid", kp_scale=1.0, kd_scale=1.0, zero_armature=False, anisotropic=False, reroot=True):
    from phc_amd.model import load_model
    m = load_model(name)
    m.reroot = reroot   # False: the solver tree is the kinematic tree (model.py solver_tree())
    if anisotropic:   # different gains / armature on the three axes of every joint (the SMPL asset's are equal: the stepper's scalar-D path)
        w = np.tile(np.array([1.0, 0.6, 1.5]), m.num_dof // 3)
        m.dof_kp, m.dof_kd, m.dof_armature = m.dof_kp * w, m.dof_kd * w[::-1], m.dof_armature * w
    from phc_amd.robots import apply_collision_filter
    apply_collision_filter(m, name.split("_")[0] if name in ("h1_humanoid", "g1_humanoid") else "smpl")
    if name in ("h1_humanoid", "g1_humanoid"):
        from phc_amd.robots import ROBOTS, apply_robot_gains
        apply_robot_gains(m, ROBOTS[name.split("_")[0]])
    if zero_armature:
        m.dof_armature[:] = 0
    ints, floats = m.pack(kp_scale, kd_scale)
    ints_d, floats_d = be.arr(ints), be.arr(floats)
    return m, abi.model_struct(ints_d, floats_d, m.num_bodies, m.num_dof, m.max_level, len(m.contact_body)), (ints_d, floats_d)


def motion_lib_on(be, lib):
    nb = lib["gts"].shape[1]
    if "dof_pos" in lib:   # robot library (H1): extended bodies + scalar joint coordinates
        ne = lib["gts_t"].shape[1] - nb
        frames = be.arr(abi.pack_frames(lib["gts"], lib["grs"], lib["gvs"], lib["gavs"], None, lib["dvs"], gts_ext=lib["gts_t"][:, nb:],
                                        grs_ext=lib["grs_t"][:, nb:], dof_pos=lib["dof_pos"]))
        keep = dict(frames=frames, ml=be.arr(lib["motion_lengths"].astype(np.float32)), mdt=be.arr(lib["motion_dt"].astype(np.float32)),
                    mnf=be.arr(lib["motion_num_frames"].astype(np.int64)), ls=be.arr(lib["length_starts"].astype(np.int64)))
        return abi.motion_lib_struct(frames, frames.shape[1], nb, keep["ml"], keep["mdt"], keep["mnf"], keep["ls"], num_ext_bodies=ne,
                                     dofs_per_joint=1), keep
    frames = be.arr(abi.pack_frames(lib["gts"], lib["grs"], lib["gvs"], lib["gavs"], lib["lrs"], lib["dvs"]))
    keep = dict(frames=frames, ml=be.arr(lib["motion_lengths"].astype(np.float32)), mdt=be.arr(lib["motion_dt"].astype(np.float32)),
                mnf=be.arr(lib["motion_num_frames"].astype(np.int64)), ls=be.arr(lib["length_starts"].astype(np.int64)))
    s = abi.motion_lib_struct(frames, frames.shape[1], nb, keep["ml"], keep["mdt"], keep["mnf"], keep["ls"])
    return s, keep
