"""Kernel math on the CPU: the per-lane functions the HIP kernels are built from (phc_amd/csrc/*.h),
compiled with g++ by tests/hostemu and driven lane by lane, checked against
  * golden vectors produced by the reference's own code (tests/golden), and
  * the numpy oracle for compositions the reference cannot run without Isaac Gym.
The same checks run against the real HIP kernels in tests/test_gpu_parity.py (-m gpu)."""
import numpy as np
import pytest

import phc_oracle as po
from hostemu_util import P, emu, np_model, np_motion_lib
from phc_amd import abi

F = np.float32
ENV_IM = dict(
    key_bodies=["R_Ankle", "L_Ankle", "R_Wrist", "L_Wrist"],
    reset_bodies=['Pelvis', 'L_Hip', 'L_Knee', 'R_Hip', 'R_Knee', 'Torso', 'Spine', 'Chest', 'Neck', 'Head', 'L_Thorax',
                  'L_Shoulder', 'L_Elbow', 'L_Wrist', 'L_Hand', 'R_Thorax', 'R_Shoulder', 'R_Elbow', 'R_Wrist', 'R_Hand'],
)
SPECS = po.DEFAULT_REWARD_SPECS


def make_im_params(model, n_envs, use_mean=False, power_reward=True):
    track_slot, reset_mask, key_ids, amp_slot, n_amp = abi.task_index_tables(model, model.body_names, ENV_IM["reset_bodies"], ENV_IM["key_bodies"])
    td = np.full((n_envs, model.num_bodies), 0.25, dtype=F)
    prm = abi.im_params_struct(dt=2 * (1 / 60), max_episode_length=300, reward_specs=SPECS, power_reward=power_reward,
                               power_coefficient=0.0005, enable_early_termination=True, use_mean_termination=use_mean,
                               disable_collision_check=False, local_root_obs=True, root_height_obs=True,
                               num_track_bodies=model.num_bodies, track_slot=track_slot, reset_mask=reset_mask,
                               num_reset_bodies=len(ENV_IM["reset_bodies"]), termination_distances=td,
                               num_key_bodies=len(key_ids), key_body_ids=key_ids, num_amp_joints=n_amp, amp_joint_slot=amp_slot,
                               num_amp_obs_steps=10, num_amp_obs_per_step=196, num_self_obs=358, num_task_obs=576)
    return prm, (track_slot, reset_mask, key_ids, amp_slot, td)


def test_motion_state_vs_reference_golden(golden):
    g = golden("motion_lib_eval")
    lib, keep = np_motion_lib(g)
    n = len(g["ms_ids"])
    nb = 24
    out = {k: np.zeros(s, dtype=F) for k, s in dict(rg_pos=(n, nb, 3), rb_rot=(n, nb, 4), body_vel=(n, nb, 3), body_ang_vel=(n, nb, 3),
                                                    dof_pos=(n, 69), dof_vel=(n, 69), blend=(n,)).items()}
    i0 = np.zeros(n, dtype=np.int64)
    i1 = np.zeros(n, dtype=np.int64)
    ids = np.ascontiguousarray(g["ms_ids"], dtype=np.int64)
    times = np.ascontiguousarray(g["ms_times"], dtype=F)
    off = np.ascontiguousarray(g["ms_offset"], dtype=F)
    emu().emu_motion_state(P(lib), n, abi.ptr(ids), abi.ptr(times), abi.ptr(off), *[abi.ptr(out[k]) for k in
                           ("rg_pos", "rb_rot", "body_vel", "body_ang_vel", "dof_pos", "dof_vel")], abi.ptr(i0), abi.ptr(i1), abi.ptr(out["blend"]))
    np.testing.assert_array_equal(i0, g["ms_idx0"])   # bit-exact indexing
    np.testing.assert_array_equal(i1, g["ms_idx1"])
    np.testing.assert_array_equal(out["blend"], g["ms_blend"])
    for k in ("rg_pos", "rb_rot", "body_vel", "body_ang_vel", "dof_pos", "dof_vel"):
        np.testing.assert_allclose(out[k], g["ms_" + k], atol=2e-5, rtol=0, err_msg=k)
    t = np.zeros(n, dtype=F)
    ph = np.ascontiguousarray(g["sti_phase"], dtype=F)
    emu().emu_sample_time_interval(P(lib), n, abi.ptr(ids), abi.ptr(ph), abi.ptr(t))
    np.testing.assert_array_equal(t, g["sti_time"])


def _sim_arrays(g, N, nb=24, nd=69):
    rbs = np.concatenate([g["body_pos"], g["body_rot"], g["body_vel"], g["body_ang_vel"]], axis=-1).astype(F)
    dof_state = np.stack([g["dof_pos"], g["dof_vel"]], axis=-1).astype(F)
    root = np.ascontiguousarray(rbs[:, 0, :])
    arrs = dict(root=root, dof=np.ascontiguousarray(dof_state), rbs=np.ascontiguousarray(rbs), cf=np.zeros((N, nb, 3), F),
                df=np.ascontiguousarray(g["dof_force"], dtype=F), pd=np.zeros((N, nd), F))
    return arrs, abi.sim_state_struct(N, arrs["root"], arrs["dof"], arrs["rbs"], arrs["cf"], arrs["df"], arrs["pd"])


@pytest.mark.parametrize("use_mean", [False, True])
def test_post_physics_vs_reference_golden(golden, use_mean):
    """reward / reset / self obs / task obs v6 / AMP obs of one post_physics_step == reference jit functions."""
    g = golden("task_fns")
    gl = golden("motion_lib_eval")
    model, mstruct, keepm = np_model()
    lib, keep = np_motion_lib(gl)
    N = g["body_pos"].shape[0]
    prm, keepp = make_im_params(model, N, use_mean=use_mean)
    arrs, sim = _sim_arrays(g, N)
    rng = np.random.default_rng(0)
    amp_in = rng.standard_normal((N, 10, 196)).astype(F)
    amp_out = np.zeros_like(amp_in)
    b = dict(progress=(g["progress"] - 1).astype(np.int64), reset=np.zeros(N, np.int64), term=np.zeros(N, np.int64), rew=np.zeros(N, F),
             raw=np.zeros((N, 5), F), obs=np.zeros((N, 934), F), mids=np.ascontiguousarray(g["env_motion"], dtype=np.int64),
             st=np.ascontiguousarray(g["start_times"], dtype=F), so=np.zeros(N, F), goff=np.zeros((N, 3), F),
             rbp=np.zeros((N, 24, 3), F), rbr=np.zeros((N, 24, 4), F), rbv=np.zeros((N, 24, 3), F), rdp=np.zeros((N, 69), F))
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp_in, amp_out, b["mids"], b["st"],
                                b["so"], b["goff"], b["rbp"], b["rbr"], b["rbv"], b["rdp"])
    assert emu().emu_im_post_physics(P(mstruct), P(lib), P(prm), P(sim), P(buf)) == 0
    np.testing.assert_array_equal(b["progress"], g["progress"])
    np.testing.assert_allclose(b["raw"][:, :4], g["reward_raw"], atol=1e-5)
    np.testing.assert_allclose(b["raw"][:, 4], g["power_reward"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(b["rew"], g["reward"] + g["power_reward"], atol=1e-5)
    np.testing.assert_array_equal(b["reset"], g["reset_mean" if use_mean else "reset"])       # bit-exact flags
    np.testing.assert_array_equal(b["term"], g["terminate_mean" if use_mean else "terminate"])
    np.testing.assert_allclose(b["obs"][:, :358], g["self_obs"], atol=1e-5)
    np.testing.assert_allclose(b["obs"][:, 358:], g["task_obs"], atol=1e-5)
    np.testing.assert_allclose(amp_out[:, 0], g["amp_obs"], atol=1e-5)
    np.testing.assert_array_equal(amp_out[:, 1:], amp_in[:, :-1])                              # history shift
    np.testing.assert_allclose(b["rbp"], g["ref1_pos"], atol=2e-5)                              # side-effect buffers (:855-868)
    np.testing.assert_allclose(b["rbv"], g["ref1_vel"], atol=2e-5)


def test_amp_demo_and_reset_vs_oracle(golden):
    """build_amp_obs_demo and the reset composition, against the numpy oracle driven the reference's way."""
    gl = golden("motion_lib_eval")
    model, mstruct, keepm = np_model()
    lib, keep = np_motion_lib(gl)
    N = 6
    prm, keepp = make_im_params(model, N)
    track_slot, reset_mask, key_ids, amp_slot, td = keepp
    dof_subset = np.concatenate([np.arange(3 * (j - 1), 3 * j) for j in range(1, 24) if amp_slot[j] >= 0])
    rng = np.random.default_rng(3)
    n = 16
    ids = rng.integers(0, N, n).astype(np.int64)
    t0 = (rng.random(n).astype(F) * gl["motion_lengths"][ids]).astype(F)
    out = np.zeros((n, 10, 196), F)
    assert emu().emu_amp_obs_demo(P(mstruct), P(lib), P(prm), n, abi.ptr(ids), abi.ptr(t0), abi.ptr(out)) == 0
    dt = F(2 * (1 / 60))
    times = (t0[:, None] + (-dt) * np.arange(10, dtype=F)[None]).astype(F)       # humanoid_amp.py:258-260
    ms = po.get_motion_state(gl, np.repeat(ids, 10), times.reshape(-1))
    want = po.build_amp_observations_smpl(ms["root_pos"], ms["root_rot"], ms["root_vel"], ms["root_ang_vel"], ms["dof_pos"], ms["dof_vel"],
                                          ms["rg_pos"][:, key_ids], dof_subset).reshape(n, 10, 196)
    np.testing.assert_allclose(out, want, atol=2e-5)

    # ---- reset of a subset of envs ----
    nb, nd = 24, 69
    arrs = dict(root=np.zeros((N, 13), F), dof=np.zeros((N, nd, 2), F), rbs=np.zeros((N, nb, 13), F), cf=np.ones((N, nb, 3), F),
                df=np.ones((N, nd), F), pd=np.zeros((N, nd), F))
    sim = abi.sim_state_struct(N, arrs["root"], arrs["dof"], arrs["rbs"], arrs["cf"], arrs["df"], arrs["pd"])
    amp = np.zeros((N, 10, 196), F)
    b = dict(progress=np.full(N, 7, np.int64), reset=np.ones(N, np.int64), term=np.ones(N, np.int64), rew=np.zeros(N, F), raw=np.zeros((N, 5), F),
             obs=np.zeros((N, 934), F), mids=np.arange(N, dtype=np.int64), st=np.full(N, -1, F), so=np.full(N, 3, F), goff=np.ones((N, 3), F))
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp, amp, b["mids"], b["st"], b["so"], b["goff"])
    env_ids = np.array([4, 1, 2], dtype=np.int64)
    phase = rng.random(3).astype(F)
    assert emu().emu_im_reset(P(mstruct), P(lib), P(prm), P(sim), P(buf), 3, abi.ptr(env_ids), abi.ptr(phase), 0) == 0
    t = po.sample_time_interval(phase, gl["motion_lengths"][env_ids])
    np.testing.assert_array_equal(b["st"][env_ids], t)
    np.testing.assert_array_equal(b["st"][[0, 3, 5]], F(-1))      # untouched envs
    assert (b["progress"][env_ids] == 0).all() and (b["reset"][env_ids] == 0).all() and (b["term"][env_ids] == 0).all()
    assert (b["so"][env_ids] == 0).all() and (b["goff"][env_ids] == 0).all() and (b["progress"][[0, 3, 5]] == 7).all()
    ms = po.get_motion_state(gl, env_ids, t, np.zeros((3, 3), F))
    np.testing.assert_allclose(arrs["root"][env_ids, 0:3], ms["root_pos"], atol=2e-5)
    np.testing.assert_allclose(arrs["root"][env_ids, 3:7], ms["root_rot"], atol=2e-5)
    np.testing.assert_allclose(arrs["root"][env_ids, 7:10], ms["root_vel"], atol=2e-5)
    np.testing.assert_allclose(arrs["dof"][env_ids, :, 0], ms["dof_pos"], atol=2e-5)
    np.testing.assert_allclose(arrs["dof"][env_ids, :, 1], ms["dof_vel"], atol=2e-5)
    np.testing.assert_allclose(arrs["pd"][env_ids], ms["dof_pos"], atol=2e-5)
    np.testing.assert_allclose(arrs["rbs"][env_ids, :, 0:3], ms["rg_pos"], atol=2e-5)
    np.testing.assert_allclose(arrs["rbs"][env_ids, :, 3:7], ms["rb_rot"], atol=2e-5)
    assert (arrs["cf"][env_ids] == 0).all() and (arrs["cf"][[0, 3, 5]] == 1).all()
    ms1 = po.get_motion_state(gl, env_ids, (np.int64(1) * dt + t + F(0)).astype(F), np.zeros((3, 3), F))
    so = po.compute_humanoid_observations_smpl_max(ms["rg_pos"], ms["rb_rot"], ms["body_vel"], ms["body_ang_vel"])
    to = po.compute_imitation_observations_v6(ms["rg_pos"][:, 0], ms["rb_rot"][:, 0], ms["rg_pos"], ms["rb_rot"], ms["body_vel"],
                                              ms["body_ang_vel"], ms1["rg_pos"], ms1["rb_rot"], ms1["body_vel"], ms1["body_ang_vel"])
    np.testing.assert_allclose(b["obs"][env_ids, :358], so, atol=2e-5)
    np.testing.assert_allclose(b["obs"][env_ids, 358:], to, atol=2e-5)
    assert (b["obs"][[0, 3, 5]] == 0).all()
    times = (t[:, None] + (-dt) * np.arange(10, dtype=F)[None]).astype(F)
    msh = po.get_motion_state(gl, np.repeat(env_ids, 10), times.reshape(-1))
    want = po.build_amp_observations_smpl(msh["root_pos"], msh["root_rot"], msh["root_vel"], msh["root_ang_vel"], msh["dof_pos"],
                                          msh["dof_vel"], msh["rg_pos"][:, key_ids], dof_subset).reshape(3, 10, 196)
    np.testing.assert_allclose(amp[env_ids], want, atol=2e-5)
