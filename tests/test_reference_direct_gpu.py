"""BASELINE configs[1] at its own size (4096 envs) against the REFERENCE'S OWN FUNCTIONS, imported -- not through the numpy restatement.

Round 5: `oracle/make_ref.py` packs exactly the reference files these legs import into the git-ignored archive `oracle/_ref/reference_modules.zip`
(it rides along with the gpurun snapshot like a built .so); `oracle/ref_shim.py` imports /root/reference in the build container and that archive (zipimport) on the GPU
box.  Every output of `phc_im_post_physics` for one env step of the HIP task is recomputed here by the reference: `MotionLibSMPL.get_motion_state`
(motion_lib_base.py:437-520, on a `__new__`-made library that holds the task's own clip tensors), `compute_imitation_reward`, `compute_humanoid_im_reset`
(humanoid_im.py:1524-1608), `compute_humanoid_observations_smpl_max` (humanoid.py:1995-2052), `compute_imitation_observations_v6` (humanoid_im.py:1309-1360),
`build_amp_observations_smpl` (humanoid_amp.py:967-1012), driven in the order of `post_physics_step` (humanoid.py:1634-1650).
Flags bit-exact, floats <= 1e-4 (the north star's bar; measured ~1e-5)."""
import numpy as np
import pytest
import torch

import ref_shim

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not ref_shim.available(), reason="no copy of the reference here (oracle/make_ref.py packs the travel archive)")]

N = 4096


def _reference_library(task):
    """The reference's MotionLibSMPL, `__new__`-made, holding the task's own per-env clip tensors (CPU): exactly the attributes get_motion_state reads."""
    ref_shim.install()
    from phc.utils.motion_lib_smpl import MotionLibSMPL
    ml = task._motion_lib
    lib = MotionLibSMPL.__new__(MotionLibSMPL)
    for k in ("gts", "grs", "gvs", "gavs", "lrs", "dvs"):
        setattr(lib, k, getattr(ml, k).cpu().contiguous())
    lib._motion_lengths, lib._motion_dt = ml._motion_lengths.cpu(), ml._motion_dt.cpu()
    lib._motion_num_frames, lib.length_starts = ml._motion_num_frames.cpu(), ml.length_starts.cpu()
    F_, M = lib.gts.shape[0], lib._motion_lengths.shape[0]
    lib._motion_aa = torch.zeros(F_, 1)
    lib._motion_bodies, lib._motion_limb_weights = torch.zeros(M, 17), torch.zeros(M, 10)
    lib._device = torch.device("cpu")
    lib.mesh_parsers = None
    nb = lib.gts.shape[1]
    lib._get_num_bodies = lambda: nb
    return lib


def test_post_physics_equals_the_imported_reference_functions_at_4096_envs():
    from phc_amd.config import compose
    from phc_amd.env.tasks.vec_task import parse_task
    ref_shim.install()
    him = ref_shim.ref_module("phc.env.tasks.humanoid_im")
    hum = ref_shim.ref_module("phc.env.tasks.humanoid")
    hamp = ref_shim.ref_module("phc.env.tasks.humanoid_amp")
    from phc.utils.flags import flags
    flags.test = flags.im_eval = flags.real_traj = False
    torch.manual_seed(0)
    task, env = parse_task(compose([f"env.num_envs={N}", "env.motion_file=synthetic:1:0"]))
    env.reset()
    act = lambda: (torch.rand(N, task.num_actions, device=task.device) * 2 - 1) * 0.3
    for _ in range(7):       # a few steps with resets in between: envs at different progress, some about to terminate
        task.reset_done()
        env.step(act())
    task.reset_done()
    amp_before = task._amp_obs_buf.clone().cpu()
    prog_before = task.progress_buf.cpu().clone()
    obs, rew, done, info = env.step(act())
    torch.cuda.synchronize()
    lib = _reference_library(task)
    c = lambda x: x.detach().cpu()
    prog = c(task.progress_buf)
    assert torch.equal(prog, prog_before + 1)
    mids, st, so, goff = c(task._sampled_motion_ids), c(task._motion_start_times), c(task._motion_start_times_offset), c(task._global_offset)
    dt = task.dt
    t0 = prog * dt + st + so                       # humanoid_im.py:886-888 (`motion_times`, fp32 torch arithmetic)
    t1 = (prog + 1) * dt + st + so                 # humanoid_im.py:720 (the observation's lookup, one control step ahead)
    r0 = lib.get_motion_state(mids, t0, offset=goff)
    r1 = lib.get_motion_state(mids, t1, offset=goff)
    bp, br, bv, bav = c(task._rigid_body_pos), c(task._rigid_body_rot), c(task._rigid_body_vel), c(task._rigid_body_ang_vel)
    assert torch.isfinite(bp).all()
    # ---- reward (humanoid_im.py:925-947) ----
    specs = {k: float(v) for k, v in task.reward_specs.items()}
    rw, raw = him.compute_imitation_reward(bp[:, 0], br[:, 0], bp, br, bv, bav, r0["rg_pos"], r0["rb_rot"], r0["body_vel"], r0["body_ang_vel"], specs)
    power = torch.abs(torch.multiply(c(task.dof_force_tensor), c(task._dof_vel))).sum(dim=-1)
    power_reward = -task.power_coefficient * power
    power_reward[prog <= 3] = 0                    # humanoid_im.py:942 ("First 3 frame power reward should not be counted")
    np.testing.assert_allclose(c(info["reward_raw"])[:, :4].numpy(), raw.numpy(), atol=1e-4)
    np.testing.assert_allclose(c(rew).numpy(), (rw + power_reward).numpy(), atol=1e-4, rtol=1e-4)
    # ---- reset / terminate (humanoid_im.py:1150-1190), bit-exact ----
    rid = c(task._reset_bodies_id)
    pass_time = t0 >= lib._motion_lengths[mids]
    reset, term = him.compute_humanoid_im_reset(torch.zeros(N, dtype=torch.long), prog, torch.zeros(N, task.num_bodies, 3), torch.zeros(2, dtype=torch.long),
                                                bp[:, rid], r0["rg_pos"][:, rid], pass_time, True, c(task._termination_distances)[rid].expand(N, -1).contiguous(),
                                                False, False)
    assert torch.equal(c(done), reset) and torch.equal(c(info["terminate"]), term)
    assert int(term.sum()) > 0 and int(term.sum()) < N, "the step must contain terminating and surviving envs"
    # ---- observations (humanoid.py:1995-2052, humanoid_im.py:1309-1360) ----
    z17, z10 = torch.zeros(N, 17), torch.zeros(N, 10)
    so_ = hum.compute_humanoid_observations_smpl_max(bp, br, bv, bav, z17, z10, True, True, True, False, False)
    tid = c(task._track_bodies_id)
    to = him.compute_imitation_observations_v6(bp[:, 0], br[:, 0], bp[:, tid], br[:, tid], bv[:, tid], bav[:, tid], r1["rg_pos"][:, tid], r1["rb_rot"][:, tid],
                                               r1["body_vel"][:, tid], r1["body_ang_vel"][:, tid], 1, True)
    np.testing.assert_allclose(c(obs).numpy(), torch.cat([so_, to], dim=-1).numpy(), atol=1e-4)
    # ---- AMP observation + history shift (humanoid_amp.py:662-691, 967-1012) ----
    kid = c(task._key_body_ids)
    amp = hamp.build_amp_observations_smpl(bp[:, 0], br[:, 0], bv[:, 0], bav[:, 0], c(task._dof_pos), c(task._dof_vel), bp[:, kid], z17, z10, task.dof_subset.cpu(),
                                           True, True, True, False, False, True)
    S = task._num_amp_obs_steps
    a = c(info["amp_obs"]).reshape(N, S, -1)
    np.testing.assert_allclose(a[:, 0].numpy(), amp.numpy(), atol=1e-4)
    assert torch.equal(a[:, 1:], amp_before[:, :-1])
    err = float((c(obs) - torch.cat([so_, to], dim=-1)).abs().max())
    print(f"4096 envs vs the imported reference functions: max |obs diff| {err:.2e}, max |reward diff| {float((c(rew) - rw - power_reward).abs().max()):.2e}, "
          f"{int(term.sum())} terminated / {int(reset.sum())} reset envs agree bit for bit")
