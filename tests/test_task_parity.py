"""Parity of the task kernels (reference-motion lookup, reward, reset, observations, AMP obs, env reset) against
  * golden vectors produced by the reference's own code (tests/golden), and
  * the numpy oracle for compositions the reference cannot run without Isaac Gym.
Each test runs on two backends (tests/backends.py): `hostemu` = the kernels' per-lane functions compiled
with g++ (CPU, test infrastructure) and `hip` = the real kernels through the C ABI on an MI355X (-m gpu)."""
import numpy as np
import pytest

import phc_oracle as po
from backends import BACKENDS, get_backend, model_on, motion_lib_on
from phc_amd import abi

F = np.float32
ENV_IM = dict(
    key_bodies=["R_Ankle", "L_Ankle", "R_Wrist", "L_Wrist"],
    reset_bodies=['Pelvis', 'L_Hip', 'L_Knee', 'R_Hip', 'R_Knee', 'Torso', 'Spine', 'Chest', 'Neck', 'Head', 'L_Thorax',
                  'L_Shoulder', 'L_Elbow', 'L_Wrist', 'L_Hand', 'R_Thorax', 'R_Shoulder', 'R_Elbow', 'R_Wrist', 'R_Hand'],
)
SPECS = po.DEFAULT_REWARD_SPECS


def make_im_params(be, model, n_envs, use_mean=False, power_reward=True, power_coefficient=0.0005, track_bodies=None, reset_bodies=None,
                   num_self_obs=358, num_amp_obs_per_step=196, **extra):
    track_bodies = track_bodies or model.body_names
    reset_bodies = reset_bodies or ENV_IM["reset_bodies"]
    tabs = abi.task_index_tables(model, track_bodies, reset_bodies, ENV_IM["key_bodies"])
    n_amp = tabs[4]
    amp_slot_np = tabs[3]
    track_slot, reset_mask, key_ids, amp_slot = (be.arr(t) for t in tabs[:4])
    td = be.arr(np.full(64, 0.25, dtype=F))
    prm = abi.im_params_struct(dt=2 * (1 / 60), max_episode_length=300, reward_specs=SPECS, power_reward=power_reward,
                               power_coefficient=power_coefficient, enable_early_termination=True, use_mean_termination=use_mean,
                               disable_collision_check=False, local_root_obs=True, root_height_obs=True,
                               num_track_bodies=len(track_bodies), track_slot=track_slot, reset_mask=reset_mask,
                               num_reset_bodies=len(reset_bodies), first_reset_body=model.body_names.index(reset_bodies[0]),
                               termination_distances=td,
                               num_key_bodies=len(ENV_IM["key_bodies"]), key_body_ids=key_ids, num_amp_joints=n_amp, amp_joint_slot=amp_slot,
                               num_amp_obs_steps=10, num_amp_obs_per_step=num_amp_obs_per_step, num_self_obs=num_self_obs, num_task_obs=24 * len(track_bodies), **extra)
    prm._keepalive = (track_slot, reset_mask, key_ids, amp_slot, td)  # the struct only holds raw addresses
    return prm, (track_slot, reset_mask, be.np(key_ids), amp_slot_np, td)


@pytest.mark.parametrize("backend", BACKENDS)
def test_motion_state_vs_reference_golden(golden, backend):
    be = get_backend(backend)
    g = golden("motion_lib_eval")
    lib, keep = motion_lib_on(be, g)
    n = len(g["ms_ids"])
    nb = 24
    out = {k: be.zeros(s) for k, s in dict(rg_pos=(n, nb, 3), rb_rot=(n, nb, 4), body_vel=(n, nb, 3), body_ang_vel=(n, nb, 3),
                                           dof_pos=(n, 69), dof_vel=(n, 69), blend=(n,)).items()}
    i0, i1 = be.zeros(n, np.int64), be.zeros(n, np.int64)
    ids = be.arr(g["ms_ids"].astype(np.int64))
    times = be.arr(g["ms_times"].astype(F))
    off = be.arr(g["ms_offset"].astype(F))
    assert be.motion_state(lib, n, ids, times, off, *[out[k] for k in ("rg_pos", "rb_rot", "body_vel", "body_ang_vel", "dof_pos", "dof_vel")],
                           i0, i1, out["blend"]) == 0
    be.sync()
    np.testing.assert_array_equal(be.np(i0), g["ms_idx0"])   # bit-exact indexing
    np.testing.assert_array_equal(be.np(i1), g["ms_idx1"])
    np.testing.assert_array_equal(be.np(out["blend"]), g["ms_blend"])
    for k in ("rg_pos", "rb_rot", "body_vel", "body_ang_vel", "dof_pos", "dof_vel"):
        np.testing.assert_allclose(be.np(out[k]), g["ms_" + k], atol=2e-5, rtol=0, err_msg=k)
    t = be.zeros(n)
    assert be.sample_time_interval(lib, n, ids, be.arr(g["sti_phase"].astype(F)), t) == 0
    be.sync()
    np.testing.assert_array_equal(be.np(t), g["sti_time"])


def _sim_arrays(be, g, N, nb=24, nd=69):
    rbs = np.concatenate([g["body_pos"], g["body_rot"], g["body_vel"], g["body_ang_vel"]], axis=-1).astype(F)
    dof_state = np.stack([g["dof_pos"], g["dof_vel"]], axis=-1).astype(F)
    arrs = dict(root=be.arr(rbs[:, 0, :]), dof=be.arr(dof_state), rbs=be.arr(rbs), cf=be.zeros((N, nb, 3)),
                df=be.arr(g["dof_force"].astype(F)), pd=be.zeros((N, nd)))
    return arrs, abi.sim_state_struct(N, arrs["root"], arrs["dof"], arrs["rbs"], arrs["cf"], arrs["df"], arrs["pd"])


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("use_mean", [False, True])
def test_post_physics_vs_reference_golden(golden, backend, use_mean):
    """reward / reset / self obs / task obs v6 / AMP obs of one post_physics_step == reference jit functions."""
    be = get_backend(backend)
    g = golden("task_fns")
    gl = golden("motion_lib_eval")
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, gl)
    N = g["body_pos"].shape[0]
    prm, keepp = make_im_params(be, model, N, use_mean=use_mean)
    arrs, sim = _sim_arrays(be, g, N)
    rng = np.random.default_rng(0)
    amp_in_np = rng.standard_normal((N, 10, 196)).astype(F)
    amp_in, amp_out = be.arr(amp_in_np), be.zeros((N, 10, 196))
    b = dict(progress=be.arr((g["progress"] - 1).astype(np.int64)), reset=be.zeros(N, np.int64), term=be.zeros(N, np.int64), rew=be.zeros(N),
             raw=be.zeros((N, 5)), obs=be.zeros((N, 934)), mids=be.arr(g["env_motion"].astype(np.int64)),
             st=be.arr(g["start_times"].astype(F)), so=be.zeros(N), goff=be.zeros((N, 3)),
             rbp=be.zeros((N, 24, 3)), rbr=be.zeros((N, 24, 4)), rbv=be.zeros((N, 24, 3)), rdp=be.zeros((N, 69)))
    cap = abi.reset_sublist_cap(N)
    rl, rc = be.zeros(abi.RESET_SUBLISTS * cap, np.int32), be.zeros((3, abi.RESET_SUBLISTS, abi.RESET_COUNT_STRIDE), np.int32)
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp_in, amp_out, b["mids"], b["st"],
                                b["so"], b["goff"], b["rbp"], b["rbr"], b["rbv"], b["rdp"], reset_list=rl, reset_count=rc, reset_slot=1)
    assert be.im_post_physics(mstruct, lib, prm, sim, buf) == 0
    be.sync()
    o = {k: be.np(v) for k, v in b.items()}
    amp_out = be.np(amp_out)
    # device-built lists of the finished envs: sub-list (env / 8) % 16, counted in slot 1, together exactly the envs with the flag set
    rl, rc = be.np(rl).reshape(abi.RESET_SUBLISTS, cap), be.np(rc)[:, :, 0]
    assert (rc[0] == 0).all() and (rc[2] == 0).all()
    listed = np.concatenate([rl[s_, :rc[1, s_]] for s_ in range(abi.RESET_SUBLISTS)])
    np.testing.assert_array_equal(np.sort(listed), np.nonzero(g["reset_mean" if use_mean else "reset"])[0])
    assert all(((rl[s_, :rc[1, s_]] >> 3) % abi.RESET_SUBLISTS == s_).all() for s_ in range(abi.RESET_SUBLISTS))
    np.testing.assert_array_equal(o["progress"], g["progress"])
    np.testing.assert_allclose(o["raw"][:, :4], g["reward_raw"], atol=1e-5)
    np.testing.assert_allclose(o["raw"][:, 4], g["power_reward"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(o["rew"], g["reward"] + g["power_reward"], atol=1e-5)
    np.testing.assert_array_equal(o["reset"], g["reset_mean" if use_mean else "reset"])       # bit-exact flags
    np.testing.assert_array_equal(o["term"], g["terminate_mean" if use_mean else "terminate"])
    np.testing.assert_allclose(o["obs"][:, :358], g["self_obs"], atol=1e-5)
    np.testing.assert_allclose(o["obs"][:, 358:], g["task_obs"], atol=1e-5)
    np.testing.assert_allclose(amp_out[:, 0], g["amp_obs"], atol=1e-5)
    np.testing.assert_array_equal(amp_out[:, 1:], amp_in_np[:, :-1])                           # history shift
    np.testing.assert_allclose(o["rbp"], g["ref1_pos"], atol=2e-5)                              # side-effect buffers (:855-868)
    np.testing.assert_allclose(o["rbv"], g["ref1_vel"], atol=2e-5)


@pytest.mark.parametrize("backend", BACKENDS)
def test_self_obs_v3_force_sensors_vs_reference_golden(golden, backend):
    """S6 + R6 `_v3`: with env.self_obs_v=3 the self observation is compute_humanoid_observations_smpl_max_v3 (humanoid.py:2113-2169) =
    the v1 block followed by the force-sensor readings (2 x 6); the task observation moves behind it.  Sensor tensor given here."""
    be = get_backend(backend)
    g, g3, gl = golden("task_fns"), golden("self_obs_v3"), golden("motion_lib_eval")
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, gl)
    N = g["body_pos"].shape[0]
    prm, keepp = make_im_params(be, model, N, num_self_obs=370, self_obs_v=3, num_force_sensors=2)
    arrs, sim = _sim_arrays(be, g, N)
    sens = be.arr(g3["sensors"].astype(F))
    sim.force_sensor = abi.ptr(sens)
    amp_in, amp_out = be.zeros((N, 10, 196)), be.zeros((N, 10, 196))
    b = dict(progress=be.arr((g["progress"] - 1).astype(np.int64)), reset=be.zeros(N, np.int64), term=be.zeros(N, np.int64), rew=be.zeros(N),
             raw=be.zeros((N, 5)), obs=be.zeros((N, 946)), mids=be.arr(g["env_motion"].astype(np.int64)),
             st=be.arr(g["start_times"].astype(F)), so=be.zeros(N), goff=be.zeros((N, 3)))
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp_in, amp_out, b["mids"], b["st"],
                                b["so"], b["goff"])
    assert be.im_post_physics(mstruct, lib, prm, sim, buf) == 0
    be.sync()
    obs = be.np(b["obs"])
    np.testing.assert_allclose(obs[:, :370], g3["self_obs_v3"], atol=1e-5)
    np.testing.assert_array_equal(obs[:, 358:370], g3["sensors"])                 # readings pass through untouched
    np.testing.assert_allclose(obs[:, 370:], g["task_obs"], atol=1e-5)            # the task block follows the longer self block
    np.testing.assert_allclose(be.np(b["rew"]), g["reward"] + g["power_reward"], atol=1e-5)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("local_root,upright", [(True, True), (True, False), (False, True), (False, False)])
def test_upright_start_and_shape_columns_vs_reference_golden(golden, backend, local_root, upright):
    """robot.has_upright_start False (`remove_base_rot` in front of the heading in every observation function, humanoid.py:1936-1939) and
    the per-env constant columns of has_shape_obs / has_weight_obs (+ `_disc`): self observation = reference
    compute_humanoid_observations_smpl_max(..., has_smpl_params, has_limb_weight_params), AMP step = build_amp_observations_smpl(...,
    has_shape_obs_disc, has_limb_weight_obs), task observation v6 -- all from ONE post-physics launch, against the reference's outputs
    (oracle/gen_golden_shape_upright.py)."""
    be = get_backend(backend)
    g, gs, gl = golden("task_fns"), golden("obs_shape_upright"), golden("motion_lib_eval")
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, gl)
    N = g["body_pos"].shape[0]
    extra = be.arr(np.concatenate([gs["shape"], gs["limb"]], axis=1).astype(F))
    prm, keepp = make_im_params(be, model, N, num_self_obs=358 + 21, num_amp_obs_per_step=196 + 21, remove_base_rot=not upright,
                                self_obs_extra=extra, amp_obs_extra=extra)
    prm.local_root_obs = int(local_root)
    arrs, sim = _sim_arrays(be, g, N)
    amp_in, amp_out = be.zeros((N, 10, 217)), be.zeros((N, 10, 217))
    b = dict(progress=be.arr((g["progress"] - 1).astype(np.int64)), reset=be.zeros(N, np.int64), term=be.zeros(N, np.int64), rew=be.zeros(N),
             raw=be.zeros((N, 5)), obs=be.zeros((N, 379 + 576)), mids=be.arr(g["env_motion"].astype(np.int64)),
             st=be.arr(g["start_times"].astype(F)), so=be.zeros(N), goff=be.zeros((N, 3)))
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp_in, amp_out, b["mids"], b["st"],
                                b["so"], b["goff"])
    assert be.im_post_physics(mstruct, lib, prm, sim, buf) == 0
    be.sync()
    tag = f"l{int(local_root)}u{int(upright)}"
    obs = be.np(b["obs"])
    np.testing.assert_allclose(obs[:, :379], gs[f"self_{tag}"], atol=1e-5)
    np.testing.assert_array_equal(obs[:, 358:379], np.concatenate([gs["shape"], gs["limb"]], axis=1))
    np.testing.assert_allclose(obs[:, 379:], gs[f"task_v6_u{int(upright)}"], atol=1e-5)
    np.testing.assert_allclose(be.np(amp_out)[:, 0], gs[f"amp_{tag}"], atol=1e-5)
    # keypoint task observation (obs_v 7) under the same flag
    prm7, keep7 = make_im_params(be, model, N, obs_v=7, remove_base_rot=not upright)
    prm7.num_task_obs = 9 * 24
    b["obs7"] = be.zeros((N, 358 + 216))
    b["progress"] = be.arr((g["progress"] - 1).astype(np.int64))
    a7i, a7o = be.zeros((N, 10, 196)), be.zeros((N, 10, 196))
    buf7 = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs7"], a7i, a7o, b["mids"], b["st"], b["so"], b["goff"])
    assert be.im_post_physics(mstruct, lib, prm7, sim, buf7) == 0
    be.sync()
    np.testing.assert_allclose(be.np(b["obs7"])[:, 358:], gs[f"task_v7_u{int(upright)}"], atol=1e-5)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("obs_v,upright", [(1, True), (2, True), (3, True), (8, True), (9, True), (2, False), (8, False), (9, False)])
def test_task_obs_versions_vs_reference_golden(golden, backend, obs_v, upright):
    """env.obs_v = 1 / 2 / 3 / 8 / 9: the task block of the post-physics launch == the reference's compute_imitation_observations{,_v2,_v3,
    _v8,_v9} (humanoid_im.py:1203-1306,1395-1515; oracle/gen_golden_task_obs_versions.py), self observation and reward untouched."""
    be = get_backend(backend)
    g, gv, gl = golden("task_fns"), golden("task_obs_versions"), golden("motion_lib_eval")
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, gl)
    N = g["body_pos"].shape[0]
    want = gv[f"v{obs_v}_u{int(upright)}"]
    prm, keepp = make_im_params(be, model, N, obs_v=obs_v, remove_base_rot=not upright)
    prm.num_task_obs = want.shape[1]
    arrs, sim = _sim_arrays(be, g, N)
    amp_in, amp_out = be.zeros((N, 10, 196)), be.zeros((N, 10, 196))
    b = dict(progress=be.arr((g["progress"] - 1).astype(np.int64)), reset=be.zeros(N, np.int64), term=be.zeros(N, np.int64), rew=be.zeros(N),
             raw=be.zeros((N, 5)), obs=be.zeros((N, 358 + want.shape[1])), mids=be.arr(g["env_motion"].astype(np.int64)),
             st=be.arr(g["start_times"].astype(F)), so=be.zeros(N), goff=be.zeros((N, 3)))
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp_in, amp_out, b["mids"], b["st"],
                                b["so"], b["goff"])
    assert be.im_post_physics(mstruct, lib, prm, sim, buf) == 0
    be.sync()
    obs = be.np(b["obs"])
    np.testing.assert_allclose(obs[:, 358:], want, atol=2e-5)
    if upright:
        np.testing.assert_allclose(obs[:, :358], g["self_obs"], atol=1e-5)
    np.testing.assert_allclose(be.np(b["rew"]), g["reward"] + g["power_reward"], atol=1e-5)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("head", [10, 4, 0])
def test_amp_history_window_in_a_strip_equals_the_shifted_buffers(golden, backend, head):
    """phc_im_buffers_t.amp_env_stride: the AMP history as a window of S frames in a per-env strip of 2 S.  Writing the new frame in the row
    before the window (heads 10, 4) or moving the window back to the lower half with the ordinary shift (head 0) leaves the same history,
    newest first, as the reference's shift into a second buffer (humanoid_amp.py:662-670) -- and touches nothing outside the new window."""
    be = get_backend(backend)
    g, gl = golden("task_fns"), golden("motion_lib_eval")
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, gl)
    N, S, A = g["body_pos"].shape[0], 10, 196
    prm, keepp = make_im_params(be, model, N)
    arrs, sim = _sim_arrays(be, g, N)
    rng = np.random.default_rng(2)
    hist = rng.standard_normal((N, S, A)).astype(F)

    def run(amp_in, amp_out, **kw):
        b = dict(progress=be.arr((g["progress"] - 1).astype(np.int64)), reset=be.zeros(N, np.int64), term=be.zeros(N, np.int64), rew=be.zeros(N),
                 raw=be.zeros((N, 5)), obs=be.zeros((N, 934)), mids=be.arr(g["env_motion"].astype(np.int64)), st=be.arr(g["start_times"].astype(F)),
                 so=be.zeros(N), goff=be.zeros((N, 3)))
        buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp_in, amp_out, b["mids"], b["st"], b["so"],
                                    b["goff"], **kw)
        assert be.im_post_physics(mstruct, lib, prm, sim, buf) == 0
        be.sync()
        return b
    plain_in, plain_out = be.arr(hist), be.zeros((N, S, A))
    run(plain_in, plain_out)
    want = be.np(plain_out)
    np.testing.assert_array_equal(want[:, 1:], hist[:, :-1])
    strip_np = rng.standard_normal((N, 2 * S, A)).astype(F)
    strip_np[:, head:head + S] = hist
    strip = be.arr(strip_np)
    new_head = head - 1 if head > 0 else S
    run(strip[:, head:head + S], strip[:, new_head:new_head + S], amp_env_stride=2 * S * A)
    got = be.np(strip)
    np.testing.assert_array_equal(got[:, new_head:new_head + S], want)
    untouched = np.ones(2 * S, bool)
    untouched[new_head:new_head + S] = False
    np.testing.assert_array_equal(got[:, untouched], strip_np[:, untouched])


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("obs_v,upright", [(6, True), (7, True), (9, True), (6, False), (9, False)])
def test_task_obs_fut_tracks_vs_reference_golden(golden, backend, obs_v, upright):
    """env.fut_tracks with numTrajSamples = 3: the task block holds one standard obs_v 6 / 7 / 9 block per future reference sample, time-major,
    sampled at (progress + 1) * dt + k / 30 + start + offset (humanoid_im.py:741-747,1309-1358,1467-1520; oracle/gen_golden_fut_tracks.py)."""
    be = get_backend(backend)
    g, gv, gl = golden("task_fns"), golden("task_obs_fut_tracks"), golden("motion_lib_eval")
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, gl)
    N = g["body_pos"].shape[0]
    want = gv[f"v{obs_v}_u{int(upright)}"]
    prm, keepp = make_im_params(be, model, N, obs_v=obs_v, remove_base_rot=not upright, num_traj_samples=3, traj_sample_timestep=1 / 30)
    prm.num_task_obs = want.shape[1]
    arrs, sim = _sim_arrays(be, g, N)
    amp_in, amp_out = be.zeros((N, 10, 196)), be.zeros((N, 10, 196))
    b = dict(progress=be.arr((g["progress"] - 1).astype(np.int64)), reset=be.zeros(N, np.int64), term=be.zeros(N, np.int64), rew=be.zeros(N),
             raw=be.zeros((N, 5)), obs=be.zeros((N, 358 + want.shape[1])), mids=be.arr(g["env_motion"].astype(np.int64)),
             st=be.arr(g["start_times"].astype(F)), so=be.arr(gv["start_off"].astype(F)), goff=be.zeros((N, 3)))
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp_in, amp_out, b["mids"], b["st"],
                                b["so"], b["goff"])
    assert be.im_post_physics(mstruct, lib, prm, sim, buf) == 0
    be.sync()
    obs = be.np(b["obs"])
    np.testing.assert_allclose(obs[:, 358:], want, atol=3e-5)
    if upright:
        np.testing.assert_allclose(obs[:, :358], g["self_obs"], atol=1e-5)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("obs_v,mask,use_mean", [(6, "fixed", False), (6, "random", True), (7, "random", False), (8, "fixed", True), (8, "random", False)])
def test_occlusion_mask_vs_reference_golden(golden, backend, obs_v, mask, use_mean):
    """env.occl_training: occluded tracked bodies take the simulated state as their reference in the task observation (obs_v 6 / 8: all fields,
    7: position) and in the early-termination distance (humanoid_im.py:796-804,845-851,1180-1181; oracle/gen_golden_occl.py); reward untouched."""
    be = get_backend(backend)
    g, go, gl = golden("task_fns"), golden("task_occl"), golden("motion_lib_eval")
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, gl)
    N = g["body_pos"].shape[0]
    want = go[f"v{obs_v}_{mask}"]
    prm, keepp = make_im_params(be, model, N, obs_v=obs_v, use_mean=use_mean)
    prm.num_task_obs = want.shape[1]
    td = be.arr(np.full(64, go["term_dist"][int(use_mean)], dtype=F))
    prm.termination_distances = abi.ptr(td)
    arrs, sim = _sim_arrays(be, g, N)
    amp_in, amp_out = be.zeros((N, 10, 196)), be.zeros((N, 10, 196))
    occl = be.arr(go[f"mask_{mask}"].astype(np.uint8))
    b = dict(progress=be.arr((g["progress"] - 1).astype(np.int64)), reset=be.zeros(N, np.int64), term=be.zeros(N, np.int64), rew=be.zeros(N),
             raw=be.zeros((N, 5)), obs=be.zeros((N, 358 + want.shape[1])), mids=be.arr(g["env_motion"].astype(np.int64)),
             st=be.arr(g["start_times"].astype(F)), so=be.zeros(N), goff=be.zeros((N, 3)))
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp_in, amp_out, b["mids"], b["st"],
                                b["so"], b["goff"], occl_mask=occl)
    assert be.im_post_physics(mstruct, lib, prm, sim, buf) == 0
    be.sync()
    obs = be.np(b["obs"])
    np.testing.assert_allclose(obs[:, 358:], want, atol=2e-5)
    np.testing.assert_allclose(obs[:, :358], g["self_obs"], atol=1e-5)
    np.testing.assert_array_equal(be.np(b["term"]), go[f"terminate_{mask}_mean{int(use_mean)}"])
    np.testing.assert_array_equal(be.np(b["reset"]), go[f"reset_{mask}_mean{int(use_mean)}"])
    np.testing.assert_allclose(be.np(b["rew"]), g["reward"] + g["power_reward"], atol=1e-5)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("upright", [True, False])
def test_self_obs_v2_history_vs_reference_golden(golden, backend, upright):
    """env.self_obs_v = 2: the policy observation carries the v1 block of the past_track_steps = 5 previous body states and of the current one,
    all relative to the current root (compute_humanoid_observations_smpl_max_v2, humanoid.py:2054-2108; oracle/gen_golden_selfobs_v2.py); the
    launch then advances the history (`_update_tensor_history`), and a reset fills it with the reset state (`_init_tensor_history`)."""
    be = get_backend(backend)
    g, g2, gl = golden("task_fns"), golden("self_obs_v2"), golden("motion_lib_eval")
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, gl)
    N, P = g["body_pos"].shape[0], 5
    prm, keepp = make_im_params(be, model, N, num_self_obs=358 * (P + 1), self_obs_v=2, num_self_obs_hist=P, remove_base_rot=not upright)
    arrs, sim = _sim_arrays(be, g, N)
    hist = be.arr(g2["hist"].astype(F))
    amp_in, amp_out = be.zeros((N, 10, 196)), be.zeros((N, 10, 196))
    W = 358 * (P + 1)
    b = dict(progress=be.arr((g["progress"] - 1).astype(np.int64)), reset=be.zeros(N, np.int64), term=be.zeros(N, np.int64), rew=be.zeros(N),
             raw=be.zeros((N, 5)), obs=be.zeros((N, W + 576)), mids=be.arr(g["env_motion"].astype(np.int64)),
             st=be.arr(g["start_times"].astype(F)), so=be.zeros(N), goff=be.zeros((N, 3)))
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp_in, amp_out, b["mids"], b["st"],
                                b["so"], b["goff"], body_state_hist=hist)
    assert be.im_post_physics(mstruct, lib, prm, sim, buf) == 0
    be.sync()
    obs = be.np(b["obs"])
    np.testing.assert_allclose(obs[:, :W], g2[f"l1u{int(upright)}"], atol=1e-5)
    if upright:
        np.testing.assert_allclose(obs[:, W:], g["task_obs"], atol=1e-5)
    # the history advanced: old slots 1..4 moved to 0..3, slot 4 holds the state this launch saw
    h = be.np(hist)
    np.testing.assert_array_equal(h[:, :P - 1], g2["hist"][:, 1:])
    cur = np.concatenate([g["body_pos"], g["body_rot"], g["body_vel"], g["body_ang_vel"]], axis=-1).astype(F)
    np.testing.assert_array_equal(h[:, P - 1], cur)
    # reset of three envs: their history becomes P copies of the imposed reference state, the observation P + 1 copies of its v1 block
    env_ids = np.array([4, 1, 2], dtype=np.int64)
    ids_d, ph_d = be.arr(env_ids), be.arr(np.array([0.3, 0.6, 0.1], F))
    assert be.im_reset(mstruct, lib, prm, sim, buf, 3, ids_d, ph_d, 0) == 0
    be.sync()
    h, rbs, obs = be.np(hist), be.np(arrs["rbs"]), be.np(b["obs"])
    for k in range(P):
        np.testing.assert_array_equal(h[env_ids, k], rbs[env_ids])
    blocks = obs[env_ids, :W].reshape(3, P + 1, 358)
    for k in range(P):
        np.testing.assert_array_equal(blocks[:, k], blocks[:, P])
    assert np.abs(h[[0, 3, 5], P - 1] - cur[[0, 3, 5]]).max() == 0      # untouched envs


@pytest.mark.parametrize("backend", BACKENDS)
def test_track_body_reward_vs_reference_golden(golden, backend):
    """env.full_body_reward False: the imitation reward over the tracked bodies only == compute_imitation_reward on the `_track_bodies_id`
    subsets (humanoid_im.py:925-936; oracle/gen_golden_task_obs_versions.py), here with the six-body VR-style track list."""
    be = get_backend(backend)
    g, gr, gl = golden("task_fns"), golden("reward_track_bodies"), golden("motion_lib_eval")
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, gl)
    N = g["body_pos"].shape[0]
    track = [model.body_names[i] for i in gr["track_ids"]]
    prm, keepp = make_im_params(be, model, N, power_reward=False, track_bodies=track, track_body_reward=True)
    arrs, sim = _sim_arrays(be, g, N)
    amp_in, amp_out = be.zeros((N, 10, 196)), be.zeros((N, 10, 196))
    b = dict(progress=be.arr((g["progress"] - 1).astype(np.int64)), reset=be.zeros(N, np.int64), term=be.zeros(N, np.int64), rew=be.zeros(N),
             raw=be.zeros((N, 4)), obs=be.zeros((N, 358 + 24 * len(track))), mids=be.arr(g["env_motion"].astype(np.int64)),
             st=be.arr(g["start_times"].astype(F)), so=be.zeros(N), goff=be.zeros((N, 3)))
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp_in, amp_out, b["mids"], b["st"],
                                b["so"], b["goff"])
    assert be.im_post_physics(mstruct, lib, prm, sim, buf) == 0
    be.sync()
    np.testing.assert_allclose(be.np(b["rew"]), gr["reward"], atol=1e-5)
    np.testing.assert_allclose(be.np(b["raw"]), gr["reward_raw"], atol=1e-5)
    assert np.abs(gr["reward"] - g["reward"]).max() > 1e-3      # differs from the full-body reward


@pytest.mark.parametrize("backend", BACKENDS)
def test_amp_obs_v2_vs_reference_golden(golden, backend):
    """R9 `_v2`: env.amp_obs_v=2 -> build_amp_observations_smpl_v2 (humanoid_amp.py:1015-1059): 196 + 12 floats per step (the key bodies'
    heading-local velocities after their positions); the history shift works on the 208-float frames."""
    be = get_backend(backend)
    g, g2, gl = golden("task_fns"), golden("amp_obs_v2"), golden("motion_lib_eval")
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, gl)
    N = g["body_pos"].shape[0]
    prm, keepp = make_im_params(be, model, N, num_amp_obs_per_step=208, amp_obs_v=2)
    arrs, sim = _sim_arrays(be, g, N)
    rng = np.random.default_rng(2)
    amp_in_np = rng.standard_normal((N, 10, 208)).astype(F)
    amp_in, amp_out = be.arr(amp_in_np), be.zeros((N, 10, 208))
    b = dict(progress=be.arr((g["progress"] - 1).astype(np.int64)), reset=be.zeros(N, np.int64), term=be.zeros(N, np.int64), rew=be.zeros(N),
             raw=be.zeros((N, 5)), obs=be.zeros((N, 934)), mids=be.arr(g["env_motion"].astype(np.int64)),
             st=be.arr(g["start_times"].astype(F)), so=be.zeros(N), goff=be.zeros((N, 3)))
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp_in, amp_out, b["mids"], b["st"],
                                b["so"], b["goff"])
    assert be.im_post_physics(mstruct, lib, prm, sim, buf) == 0
    be.sync()
    out = be.np(amp_out)
    np.testing.assert_allclose(out[:, 0], g2["amp_obs_v2"], atol=1e-5)
    np.testing.assert_array_equal(out[:, 1:], amp_in_np[:, :-1])
    np.testing.assert_allclose(be.np(b["obs"])[:, :358], g["self_obs"], atol=1e-5)


@pytest.mark.parametrize("backend", BACKENDS)
def test_post_physics_cycle_motion_zero_out_far_vs_reference_golden(golden, backend):
    """Config-3 branches of post_physics_step (env_im_getup_mcp.yaml: cycle_motion + zero_out_far): point-goal reward,
    in-place clip restart with its new start time / time offset / global offset, cycle-counter gating of reset, and the
    task-obs gating -- against the reference's own functions driven in the reference's order (oracle/gen_golden_cfg3.py)."""
    be = get_backend(backend)
    g = golden("task_fns_cfg3")
    gl = golden("motion_lib_eval")
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, gl)
    N = g["body_pos"].shape[0]
    prm, keepp = make_im_params(be, model, N, power_coefficient=0.00005, cycle_motion=True, zero_out_far=True, close_distance=0.25, far_distance=3.0)
    gg = dict(g)
    gg["dof_pos"] = np.zeros((N, 69), F)
    arrs, sim = _sim_arrays(be, gg, N)
    amp_in, amp_out = be.zeros((N, 10, 196)), be.zeros((N, 10, 196))
    b = dict(progress=be.arr((g["progress"] - 1).astype(np.int64)), reset=be.zeros(N, np.int64), term=be.zeros(N, np.int64), rew=be.zeros(N),
             raw=be.zeros((N, 5)), obs=be.zeros((N, 934)), mids=be.arr(g["env_motion"].astype(np.int64)),
             st=be.arr(g["start_times"].astype(F)), so=be.arr(g["start_off"].astype(F)), goff=be.arr(g["global_offset"].astype(F)),
             cyc=be.arr(g["cycle_counter_in"].astype(np.int32)), pg=be.arr(g["point_goal_prev"].astype(F)), ph=be.arr(g["cycle_phase"].astype(F)))
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp_in, amp_out, b["mids"], b["st"],
                                b["so"], b["goff"], cycle_counter=b["cyc"], point_goal=b["pg"], cycle_phase=b["ph"])
    assert be.im_post_physics(mstruct, lib, prm, sim, buf) == 0
    be.sync()
    o = {k: be.np(v) for k, v in b.items()}
    np.testing.assert_array_equal(o["progress"], g["progress"])
    np.testing.assert_array_equal(o["cyc"], g["cycle_counter_out"])
    np.testing.assert_array_equal(o["st"], g["start_times_out"])          # sample_time_interval of the cycled envs: bit-exact
    np.testing.assert_array_equal(o["so"], g["start_off_out"])
    np.testing.assert_allclose(o["goff"], g["global_offset_out"], atol=1e-6)
    np.testing.assert_allclose(o["raw"], g["reward_raw"], atol=2e-5)
    np.testing.assert_allclose(o["rew"], g["reward"], atol=3e-5)
    np.testing.assert_array_equal(o["reset"], g["reset"])
    np.testing.assert_array_equal(o["term"], g["terminate"])
    np.testing.assert_allclose(o["pg"], g["point_goal_out"], atol=1e-5)
    np.testing.assert_allclose(o["obs"][:, 358:], g["task_obs"], atol=2e-5)
    assert g["pass_len"].sum() > 4 and g["zeros_subset"].sum() > 4 and g["far_subset"].sum() >= 1 and g["reward_far"].sum() > 4
    # the random reference offsets of a clip restart: cycle_motion_xp adds torch.rand(2) (humanoid_im.py:1131-1132), zero_out_far_train a point
    # of the 5 m disk (:1133-1140) to the offset the plain restart computes; envs that do not restart keep theirs
    cycled = g["pass_len"].astype(bool)
    uv = np.random.default_rng(8).random((N, 2)).astype(F)
    for kw, add in ((dict(cycle_motion_xp=True), uv),
                    (dict(zero_out_far_train=True), np.stack([np.cos(uv[:, 1] * F(np.pi) * F(2)) * np.sqrt(uv[:, 0]) * F(5),
                                                              np.sin(uv[:, 1] * F(np.pi) * F(2)) * np.sqrt(uv[:, 0]) * F(5)], axis=1))):
        prm2, keep2 = make_im_params(be, model, N, power_coefficient=0.00005, cycle_motion=True, zero_out_far=True, close_distance=0.25, far_distance=3.0, **kw)
        arrs2, sim2 = _sim_arrays(be, gg, N)
        b2 = dict(b, progress=be.arr((g["progress"] - 1).astype(np.int64)), st=be.arr(g["start_times"].astype(F)), so=be.arr(g["start_off"].astype(F)),
                  goff=be.arr(g["global_offset"].astype(F)), cyc=be.arr(g["cycle_counter_in"].astype(np.int32)), obs=be.zeros((N, 934)),
                  reset=be.zeros(N, np.int64), term=be.zeros(N, np.int64))
        amp_out2, uv_d = be.zeros((N, 10, 196)), be.arr(uv)      # (the struct holds raw addresses: keep the arrays alive)
        buf2 = abi.im_buffers_struct(b2["progress"], b2["reset"], b2["term"], b2["rew"], b2["raw"], b2["obs"], amp_in, amp_out2, b2["mids"],
                                     b2["st"], b2["so"], b2["goff"], cycle_counter=b2["cyc"], point_goal=b2["pg"], cycle_phase=b2["ph"],
                                     offset_rand=uv_d)
        assert be.im_post_physics(mstruct, lib, prm2, sim2, buf2) == 0
        be.sync()
        want = g["global_offset_out"].astype(F).copy()
        want[cycled, :2] += add[cycled]
        np.testing.assert_allclose(be.np(b2["goff"]), want, atol=3e-6)
        np.testing.assert_array_equal(be.np(b2["st"]), g["start_times_out"])


@pytest.mark.parametrize("backend", BACKENDS)
def test_enable_hist_obs_vs_reference_method(golden, backend):
    """env.enableHistObs against the reference's own `HumanoidAMP._compute_humanoid_obs` (humanoid_amp.py:546-557; fixture from
    oracle/gen_golden_hist_obs.py, run on a `__new__`-made task): the AMP history buffer, flattened newest frame first, BEHIND the self observation -- for
    all envs and for an env subset.  The kernels place it through the per-env extra columns of the self observation (`self_obs_extra`), as
    HumanoidIm._refresh_hist_obs fills them; here the same post-physics launch on the fixture's body states."""
    be = get_backend(backend)
    g, gt = golden("hist_obs"), golden("task_fns")
    N = g["obs_all"].shape[0]
    H = g["amp_obs_buf"].shape[1] * g["amp_obs_buf"].shape[2]
    assert g["obs_all"].shape[1] == 358 + H
    np.testing.assert_array_equal(g["obs_all"][:, 358:], g["amp_obs_buf"].reshape(N, H))             # what the reference appends, and where
    np.testing.assert_array_equal(g["obs_subset"], g["obs_all"][g["env_ids"]])
    np.testing.assert_array_equal(g["obs_all"][:, :358], g["obs_without_hist"])
    # the kernels' composition: one post-physics launch on the fixture's body states with the history as the extra columns of the self observation
    sub = {k: (gt[k][:N] if getattr(gt[k], "ndim", 0) >= 1 and gt[k].shape[0] == gt["body_pos"].shape[0] else gt[k]) for k in gt}
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, golden("motion_lib_eval"))
    extra = be.arr(g["amp_obs_buf"].reshape(N, H).astype(F))
    prm, keepp = make_im_params(be, model, N, num_self_obs=358 + H, self_obs_extra=extra)
    arrs, sim = _sim_arrays(be, sub, N)
    amp_in, amp_out = be.zeros((N, 10, 196)), be.zeros((N, 10, 196))
    b = dict(progress=be.arr((sub["progress"] - 1).astype(np.int64)), reset=be.zeros(N, np.int64), term=be.zeros(N, np.int64), rew=be.zeros(N),
             raw=be.zeros((N, 5)), obs=be.zeros((N, 358 + H + 576)), mids=be.arr(sub["env_motion"].astype(np.int64)),
             st=be.arr(sub["start_times"].astype(F)), so=be.zeros(N), goff=be.zeros((N, 3)))
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp_in, amp_out, b["mids"], b["st"], b["so"], b["goff"])
    assert be.im_post_physics(mstruct, lib, prm, sim, buf) == 0
    be.sync()
    obs = be.np(b["obs"])
    np.testing.assert_allclose(obs[:, :358 + H], g["obs_all"], atol=1e-5)                          # == the reference method's output
    np.testing.assert_array_equal(obs[:, 358:358 + H], g["amp_obs_buf"].reshape(N, H))
    np.testing.assert_allclose(obs[:, 358 + H:], sub["task_obs"], atol=1e-5)                       # the task block behind it


@pytest.mark.parametrize("backend", BACKENDS)
def test_amp_ref_table_rows_equal_the_full_builds_at_frame_times(golden, backend):
    """phc_amp_ref_table / phc_im_params_t.amp_ref_table (ABI 33): row f = the AMP observation of the lookup (f, f + 1, blend 0).  At start times on the
    1/30 s grid (sample_time_interval) and history steps of dt = 1/30 s every lookup falls on a frame up to the rounding of its blend factor: exactly 0 for
    most (bit-equal to the full build), <= 1e-4 for the rest that use the table (first-order blend of two rows), full build otherwise.  Against the full
    build of the same backend and against the numpy oracle driven the reference's way (humanoid_amp.py:253-284)."""
    be = get_backend(backend)
    gl = golden("motion_lib_eval")
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, gl)
    N = 6
    prm, keepp = make_im_params(be, model, N)
    track_slot, reset_mask, key_ids, amp_slot, td = keepp
    dof_subset = np.concatenate([np.arange(3 * (j - 1), 3 * j) for j in range(1, 24) if amp_slot[j] >= 0])
    nf, starts = gl["motion_num_frames"].astype(np.int64), gl["length_starts"].astype(np.int64)
    assert np.allclose(gl["motion_dt"], 1 / 30)
    Ftot = int(starts[-1] + nf[-1])
    nxt = np.arange(1, Ftot + 1, dtype=np.int64)
    nxt[starts + nf - 1] = starts + nf - 1
    table = be.zeros((Ftot, 196))
    assert be.amp_ref_table(mstruct, lib, prm, Ftot, be.arr(nxt), table) == 0
    be.sync()
    assert np.isfinite(be.np(table)).all()
    prm_t, keept = make_im_params(be, model, N, amp_ref_table=table)
    rng = np.random.default_rng(5)
    n = 64
    ids = rng.integers(0, N, n).astype(np.int64)
    t0 = np.zeros(n, dtype=F)
    assert be.sample_time_interval(lib, n, be.arr(ids), be.arr(rng.random(n).astype(F)), (t0d := be.zeros(n))) == 0   # multiples of 1/30 s
    be.sync()
    t0 = be.np(t0d).copy()
    t0[:4] = 0.0                                        # history times below zero clamp to frame 0
    out_full, out_tab = be.zeros((n, 10, 196)), be.zeros((n, 10, 196))
    assert be.amp_obs_demo(mstruct, lib, prm, n, be.arr(ids), be.arr(t0), out_full) == 0
    assert be.amp_obs_demo(mstruct, lib, prm_t, n, be.arr(ids), be.arr(t0), out_tab) == 0
    be.sync()
    a, b = be.np(out_tab), be.np(out_full)
    np.testing.assert_allclose(a, b, rtol=0, atol=5e-6)
    assert (a == b).all(axis=-1).mean() > 0.6           # rows of the lookups with blend factor exactly 0 (and of the off-grid ones built in full)
    dt = F(2 * (1 / 60))
    times = (t0[:, None] + (-dt) * np.arange(10, dtype=F)[None]).astype(F)
    ms = po.get_motion_state(gl, np.repeat(ids, 10), times.reshape(-1))
    want = po.build_amp_observations_smpl(ms["root_pos"], ms["root_rot"], ms["root_vel"], ms["root_ang_vel"], ms["dof_pos"], ms["dof_vel"],
                                          ms["rg_pos"][:, key_ids], dof_subset).reshape(n, 10, 196)
    np.testing.assert_allclose(a, want, atol=2e-5)
    # start times OFF the frame grid: every lookup is built in full -- same numbers as without the table
    t1 = (rng.random(n).astype(F) * gl["motion_lengths"][ids]).astype(F)
    o1, o2 = be.zeros((n, 10, 196)), be.zeros((n, 10, 196))
    assert be.amp_obs_demo(mstruct, lib, prm, n, be.arr(ids), be.arr(t1), o1) == 0 and be.amp_obs_demo(mstruct, lib, prm_t, n, be.arr(ids), be.arr(t1), o2) == 0
    be.sync()
    d = np.abs(be.np(o1) - be.np(o2))
    assert d.max() <= 5e-6 and (d == 0).mean() > 0.95


@pytest.mark.parametrize("backend", BACKENDS)
def test_amp_demo_and_reset_vs_oracle(golden, backend):
    """build_amp_obs_demo and the reset composition, against the numpy oracle driven the reference's way."""
    be = get_backend(backend)
    gl = golden("motion_lib_eval")
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, gl)
    N = 6
    prm, keepp = make_im_params(be, model, N)
    track_slot, reset_mask, key_ids, amp_slot, td = keepp
    dof_subset = np.concatenate([np.arange(3 * (j - 1), 3 * j) for j in range(1, 24) if amp_slot[j] >= 0])
    rng = np.random.default_rng(3)
    n = 16
    ids = rng.integers(0, N, n).astype(np.int64)
    t0 = (rng.random(n).astype(F) * gl["motion_lengths"][ids]).astype(F)
    out = be.zeros((n, 10, 196))
    assert be.amp_obs_demo(mstruct, lib, prm, n, be.arr(ids), be.arr(t0), out) == 0
    be.sync()
    dt = F(2 * (1 / 60))
    times = (t0[:, None] + (-dt) * np.arange(10, dtype=F)[None]).astype(F)       # humanoid_amp.py:258-260
    ms = po.get_motion_state(gl, np.repeat(ids, 10), times.reshape(-1))
    want = po.build_amp_observations_smpl(ms["root_pos"], ms["root_rot"], ms["root_vel"], ms["root_ang_vel"], ms["dof_pos"], ms["dof_vel"],
                                          ms["rg_pos"][:, key_ids], dof_subset).reshape(n, 10, 196)
    np.testing.assert_allclose(be.np(out), want, atol=2e-5)
    # the same with has_upright_start False and shape / limb columns: the clip's humanoid is the env of the same index
    extra_np = rng.standard_normal((N, 21)).astype(F)
    extra = be.arr(extra_np)
    prm_s, keep_s = make_im_params(be, model, N, num_amp_obs_per_step=217, remove_base_rot=True, amp_obs_extra=extra)
    out_s = be.zeros((n, 10, 217))
    assert be.amp_obs_demo(mstruct, lib, prm_s, n, be.arr(ids), be.arr(t0), out_s) == 0
    be.sync()
    ex = np.repeat(extra_np[ids], 10, axis=0)
    want_s = po.build_amp_observations_smpl(ms["root_pos"], ms["root_rot"], ms["root_vel"], ms["root_ang_vel"], ms["dof_pos"], ms["dof_vel"],
                                            ms["rg_pos"][:, key_ids], dof_subset, upright=False, shape_params=ex[:, :11],
                                            limb_weight_params=ex[:, 11:]).reshape(n, 10, 217)
    # (upright clips read as a y-up asset: the stripped root frame's x axis can come close to vertical, where the heading atan2 amplifies
    # fp32 rounding -- a handful of elements sit above the 2e-5 of the upright case)
    d = np.abs(be.np(out_s) - want_s)
    assert d.max() < 1e-3 and (d > 2e-5).mean() < 2e-3
    np.testing.assert_array_equal(be.np(out_s)[..., 196:], ex.reshape(n, 10, 21))

    # ---- reset of a subset of envs ----
    nb, nd = 24, 69
    arrs = dict(root=be.zeros((N, 13)), dof=be.zeros((N, nd, 2)), rbs=be.zeros((N, nb, 13)), cf=be.arr(np.ones((N, nb, 3), F)),
                df=be.arr(np.ones((N, nd), F)), pd=be.zeros((N, nd)))
    sim = abi.sim_state_struct(N, arrs["root"], arrs["dof"], arrs["rbs"], arrs["cf"], arrs["df"], arrs["pd"])
    amp = be.zeros((N, 10, 196))
    b = dict(progress=be.arr(np.full(N, 7, np.int64)), reset=be.arr(np.ones(N, np.int64)), term=be.arr(np.ones(N, np.int64)), rew=be.zeros(N),
             raw=be.zeros((N, 5)), obs=be.zeros((N, 934)), mids=be.arr(np.arange(N, dtype=np.int64)), st=be.arr(np.full(N, -1, F)),
             so=be.arr(np.full(N, 3, F)), goff=be.arr(np.ones((N, 3), F)))
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp, amp, b["mids"], b["st"], b["so"], b["goff"])
    env_ids = np.array([4, 1, 2], dtype=np.int64)
    phase = rng.random(3).astype(F)
    assert be.im_reset(mstruct, lib, prm, sim, buf, 3, be.arr(env_ids), be.arr(phase), 0) == 0
    be.sync()
    keep_dev = (b, arrs, amp)      # (`sim` / `buf` hold raw addresses of these device arrays: they must outlive the launches below)
    b = {k: be.np(v) for k, v in b.items()}
    arrs = {k: be.np(v) for k, v in arrs.items()}
    amp = be.np(amp)
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

    # ---- zero_out_far_train (humanoid_im.py:966-980): the reset leaves the humanoid on the clip but moves the REFERENCE to a random point of a
    # 5 m disk (global offset), arms the cycle counter, and the observations of the reset envs see the shifted reference ----
    prm_f, keep_f = make_im_params(be, model, N, zero_out_far=True, zero_out_far_train=True, zero_out_far_steps=90)
    uv = rng.random((N, 2)).astype(F)
    cyc, pg = be.zeros(N, np.int32), be.zeros(N)
    bf = dict(progress=be.arr(np.full(N, 7, np.int64)), reset=be.arr(np.ones(N, np.int64)), term=be.arr(np.ones(N, np.int64)), rew=be.zeros(N),
              raw=be.zeros((N, 5)), obs=be.zeros((N, 934)), mids=be.arr(np.arange(N, dtype=np.int64)), st=be.arr(np.full(N, -1, F)),
              so=be.arr(np.full(N, 3, F)), goff=be.arr(np.ones((N, 3), F)))
    ampf = be.zeros((N, 10, 196))
    uv_d = be.arr(uv)      # (the struct holds raw addresses: keep the array alive)
    buff = abi.im_buffers_struct(bf["progress"], bf["reset"], bf["term"], bf["rew"], bf["raw"], bf["obs"], ampf, ampf, bf["mids"], bf["st"], bf["so"],
                                 bf["goff"], cycle_counter=cyc, point_goal=pg, offset_rand=uv_d)
    ids_d, ph_d = be.arr(env_ids), be.arr(phase)
    assert be.im_reset(mstruct, lib, prm_f, sim, buff, 3, ids_d, ph_d, 0) == 0
    be.sync()
    rd, ang = np.sqrt(uv[env_ids, 0]) * F(5), uv[env_ids, 1] * F(np.pi) * F(2)
    want_off = np.stack([np.cos(ang) * rd, np.sin(ang) * rd, np.zeros(3, F)], axis=1).astype(F)
    np.testing.assert_allclose(be.np(bf["goff"])[env_ids], want_off, atol=2e-6)
    assert (be.np(cyc)[env_ids] == 90).all() and (be.np(cyc)[[0, 3, 5]] == 0).all()
    obs_f = be.np(bf["obs"])
    np.testing.assert_allclose(obs_f[env_ids, :358], so, atol=2e-5)                      # the state itself is the un-shifted reference
    dist = np.linalg.norm(ms["rg_pos"][:, 0] - (ms1["rg_pos"][:, 0] + want_off), axis=-1)
    np.testing.assert_allclose(be.np(pg)[env_ids], dist, atol=2e-5)                       # _point_goal (:792)
    assert (dist > 0.25).any()
    # task observation against the shifted reference with the zero_out_far gating of humanoid_im.py:783-797
    rp, rr, rv, rw = ms1["rg_pos"] + want_off[:, None], ms1["rb_rot"].copy(), ms1["body_vel"].copy(), ms1["body_ang_vel"].copy()
    z = dist > 0.25
    rp[z, 1:], rr[z, 1:], rv[z], rw[z] = ms["rg_pos"][z, 1:], ms["rb_rot"][z, 1:], ms["body_vel"][z], ms["body_ang_vel"][z]
    vz = dist > 3.0
    rp[vz, 0] = (rp[vz, 0] - ms["rg_pos"][vz, 0]) / dist[vz, None] * F(3.0) + ms["rg_pos"][vz, 0]
    to_f = po.compute_imitation_observations_v6(ms["rg_pos"][:, 0], ms["rb_rot"][:, 0], ms["rg_pos"], ms["rb_rot"], ms["body_vel"], ms["body_ang_vel"],
                                                rp, rr, rv, rw)
    np.testing.assert_allclose(obs_f[env_ids, 358:], to_f, atol=3e-5)


@pytest.mark.parametrize("backend", BACKENDS)
def test_reset_from_state_vs_oracle(golden, backend):
    """HumanoidImGetup fall / recovery resets (phc_im_reset_from_state): the env keeps the state it was given; progress /
    reset / terminate / contact cleared, PD target := joint positions, observations recomputed against the reference at the
    env's own motion clock with progress 0, AMP history filled with (fall) or topped by (recovery) the current AMP obs."""
    be = get_backend(backend)
    g = golden("task_fns")
    gl = golden("motion_lib_eval")
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, gl)
    N = g["body_pos"].shape[0]
    prm, keepp = make_im_params(be, model, N)
    track_slot, reset_mask, key_ids, amp_slot, td = keepp
    dof_subset = np.concatenate([np.arange(3 * (j - 1), 3 * j) for j in range(1, 24) if amp_slot[j] >= 0])
    arrs, sim = _sim_arrays(be, g, N)
    rng = np.random.default_rng(5)
    amp0 = rng.standard_normal((N, 10, 196)).astype(F)
    amp = be.arr(amp0)
    st = g["start_times"].astype(F)
    goff = np.zeros((N, 3), F)
    goff[:, :2] = rng.standard_normal((N, 2)).astype(F) * 0.1
    b = dict(progress=be.arr(np.full(N, 9, np.int64)), reset=be.arr(np.ones(N, np.int64)), term=be.arr(np.ones(N, np.int64)), rew=be.zeros(N),
             raw=be.zeros((N, 5)), obs=be.zeros((N, 934)), mids=be.arr(g["env_motion"].astype(np.int64)), st=be.arr(st), so=be.zeros(N),
             goff=be.arr(goff))
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp, amp, b["mids"], b["st"], b["so"], b["goff"])
    fall_ids = np.array([3, 11, 40], dtype=np.int64)
    rec_ids = np.array([7, 20], dtype=np.int64)
    assert be.im_reset_from_state(mstruct, lib, prm, sim, buf, 3, be.arr(fall_ids), 1) == 0
    assert be.im_reset_from_state(mstruct, lib, prm, sim, buf, 2, be.arr(rec_ids), 0) == 0
    be.sync()
    o = {k: be.np(v) for k, v in b.items()}
    a = {k: be.np(v) for k, v in arrs.items()}
    amp = be.np(amp)
    ids = np.concatenate([fall_ids, rec_ids])
    others = np.setdiff1d(np.arange(N), ids)
    assert (o["progress"][ids] == 0).all() and (o["reset"][ids] == 0).all() and (o["term"][ids] == 0).all()
    assert (o["progress"][others] == 9).all() and (o["reset"][others] == 1).all() and (o["obs"][others] == 0).all()
    np.testing.assert_array_equal(a["pd"][ids], g["dof_pos"][ids].astype(F))
    np.testing.assert_array_equal(a["rbs"], np.concatenate([g["body_pos"], g["body_rot"], g["body_vel"], g["body_ang_vel"]], -1).astype(F))  # state kept
    t1 = (F(1) * F(2 * (1 / 60)) + st[ids]).astype(F)
    ms = po.get_motion_state(gl, g["env_motion"][ids], t1, goff[ids])
    bp, br, bv, bw = (g[k][ids].astype(F) for k in ("body_pos", "body_rot", "body_vel", "body_ang_vel"))
    want_self = po.compute_humanoid_observations_smpl_max(bp, br, bv, bw, True, True)
    want_task = po.compute_imitation_observations_v6(bp[:, 0], br[:, 0], bp, br, bv, bw, ms["rg_pos"], ms["rb_rot"], ms["body_vel"], ms["body_ang_vel"])
    np.testing.assert_allclose(o["obs"][ids, :358], want_self, atol=2e-5)
    np.testing.assert_allclose(o["obs"][ids, 358:], want_task, atol=2e-5)
    want_amp = po.build_amp_observations_smpl(bp[:, 0], br[:, 0], bv[:, 0], bw[:, 0], g["dof_pos"][ids].astype(F), g["dof_vel"][ids].astype(F),
                                              bp[:, key_ids], dof_subset)
    np.testing.assert_allclose(amp[fall_ids], np.repeat(want_amp[:3, None], 10, axis=1), atol=2e-5)   # _init_amp_obs_default
    np.testing.assert_allclose(amp[rec_ids, 0], want_amp[3:], atol=2e-5)
    np.testing.assert_array_equal(amp[rec_ids, 1:], amp0[rec_ids, 1:])                                   # recovery keeps its history
    np.testing.assert_array_equal(amp[others], amp0[others])


@pytest.mark.parametrize("backend", BACKENDS)
def test_three_point_tracking_vs_reference_golden(golden, backend):
    """env_vr.yaml: trackBodies = reset_bodies = [Head, L_Hand, R_Hand] -- 72-float task obs in the trackBodies order and a reset
    test on those three bodies only == the reference's functions on the subsets; the reward stays full-body (full_body_reward)."""
    be = get_backend(backend)
    g, gv = golden("task_fns"), golden("task_fns_vr")
    gl = golden("motion_lib_eval")
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, gl)
    N = g["body_pos"].shape[0]
    vr = ["Head", "L_Hand", "R_Hand"]
    prm, keepp = make_im_params(be, model, N, track_bodies=vr, reset_bodies=vr)
    arrs, sim = _sim_arrays(be, g, N)
    amp_in, amp_out = be.zeros((N, 10, 196)), be.zeros((N, 10, 196))
    b = dict(progress=be.arr((g["progress"] - 1).astype(np.int64)), reset=be.zeros(N, np.int64), term=be.zeros(N, np.int64), rew=be.zeros(N),
             raw=be.zeros((N, 5)), obs=be.zeros((N, 358 + 72)), mids=be.arr(g["env_motion"].astype(np.int64)),
             st=be.arr(g["start_times"].astype(F)), so=be.zeros(N), goff=be.zeros((N, 3)))
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp_in, amp_out, b["mids"], b["st"], b["so"], b["goff"])
    assert be.im_post_physics(mstruct, lib, prm, sim, buf) == 0
    be.sync()
    o = {k: be.np(v) for k, v in b.items()}
    np.testing.assert_allclose(o["obs"][:, :358], g["self_obs"], atol=1e-5)
    np.testing.assert_allclose(o["obs"][:, 358:], gv["task_obs"], atol=1e-5)
    np.testing.assert_array_equal(o["reset"], gv["reset"])
    np.testing.assert_array_equal(o["term"], gv["terminate"])
    np.testing.assert_allclose(o["raw"][:, :4], g["reward_raw"], atol=1e-5)     # full-body reward unchanged
    assert gv["terminate"].sum() > 0


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("vr", [False, True])
def test_task_obs_v7_vs_reference_golden(golden, backend, vr):
    """`env.obs_v=7` (the reference's keypoint models): 9 floats per tracked body -- position / velocity differences and reference
    positions in the heading frame -- == compute_imitation_observations_v7, for all bodies and for the three-point subset."""
    be = get_backend(backend)
    g, g7 = golden("task_fns"), golden("task_fns_v7")
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, golden("motion_lib_eval"))
    N = g["body_pos"].shape[0]
    bodies = ["Head", "L_Hand", "R_Hand"] if vr else None
    nt = 3 if vr else 24
    prm, keepp = make_im_params(be, model, N, track_bodies=bodies, obs_v=7)
    prm.num_task_obs = 9 * nt
    arrs, sim = _sim_arrays(be, g, N)
    amp_in, amp_out = be.zeros((N, 10, 196)), be.zeros((N, 10, 196))
    b = dict(progress=be.arr((g["progress"] - 1).astype(np.int64)), reset=be.zeros(N, np.int64), term=be.zeros(N, np.int64), rew=be.zeros(N),
             raw=be.zeros((N, 5)), obs=be.zeros((N, 358 + 9 * nt)), mids=be.arr(g["env_motion"].astype(np.int64)),
             st=be.arr(g["start_times"].astype(F)), so=be.zeros(N), goff=be.zeros((N, 3)))
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp_in, amp_out, b["mids"], b["st"], b["so"], b["goff"])
    assert be.im_post_physics(mstruct, lib, prm, sim, buf) == 0
    be.sync()
    obs = be.np(b["obs"])
    np.testing.assert_allclose(obs[:, :358], g["self_obs"], atol=1e-5)
    np.testing.assert_allclose(obs[:, 358:], g7["task_obs_vr" if vr else "task_obs"], atol=1e-5)
    np.testing.assert_allclose(be.np(b["raw"])[:, :4], g["reward_raw"], atol=1e-5)     # the reward does not depend on the observation version


@pytest.mark.parametrize("backend", BACKENDS)
def test_multi_step_rollout_vs_reference_env(golden, backend):
    """16 consecutive env steps against the thin CPU reference env of oracle/gen_golden_rollout.py (the reference's own jit functions and
    motion library driven in its method order, kinematic stand-in for the physics): per step the reset of the envs the previous step
    flagged (state, observations, re-initialised AMP history), then post-physics on the golden's body state -- progress, reward, flags,
    observations and the ping-ponged AMP history all track the reference across resets, time-outs and terminations."""
    be = get_backend(backend)
    g, gl = golden("rollout_ref_env"), golden("motion_lib_eval")
    model, mstruct, keepm = model_on(be)
    lib, keep = motion_lib_on(be, gl)
    K, N = g["obs"].shape[:2]
    nb, nd = 24, 69
    prm, keepp = make_im_params(be, model, N)
    arrs = dict(root=be.zeros((N, 13)), dof=be.zeros((N, nd, 2)), rbs=be.zeros((N, nb, 13)), cf=be.zeros((N, nb, 3)), df=be.zeros((N, nd)), pd=be.zeros((N, nd)))
    sim = abi.sim_state_struct(N, arrs["root"], arrs["dof"], arrs["rbs"], arrs["cf"], arrs["df"], arrs["pd"])
    amp = [be.zeros((N, 10, 196)), be.zeros((N, 10, 196))]
    cur = 0
    b = dict(progress=be.zeros(N, np.int64), reset=be.arr(np.ones(N, np.int64)), term=be.zeros(N, np.int64), rew=be.zeros(N), raw=be.zeros((N, 5)),
             obs=be.zeros((N, 934)), mids=be.arr(g["motion_ids"].astype(np.int64)), st=be.zeros(N), so=be.zeros(N), goff=be.zeros((N, 3)))

    def bufs(a_in, a_out):
        return abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], a_in, a_out, b["mids"], b["st"], b["so"], b["goff"])

    def put(dst, src):   # host -> backend array, in place
        if isinstance(dst, np.ndarray):
            dst[...] = src
        else:
            dst.copy_(be.arr(np.ascontiguousarray(src)))
    n_resets = n_term = 0
    for k in range(K):
        ids = g["reset_ids"][k]
        ids = ids[ids >= 0].astype(np.int64)
        if len(ids):
            flagged = np.nonzero(be.np(b["reset"]))[0]
            np.testing.assert_array_equal(np.sort(ids), flagged, err_msg=f"step {k}: envs to reset")
            assert be.im_reset(mstruct, lib, prm, sim, bufs(amp[cur], amp[cur]), len(ids), be.arr(ids), be.arr(g["reset_phase"][k][:len(ids)].astype(F)), 0) == 0
            be.sync()
            n_resets += len(ids)
            np.testing.assert_array_equal(be.np(b["st"])[ids], g["start_after_reset"][k][ids], err_msg=f"step {k}: start times")
            np.testing.assert_allclose(be.np(b["obs"])[ids], g["obs_after_reset"][k][ids], atol=2e-5, err_msg=f"step {k}: obs after reset")
            np.testing.assert_allclose(be.np(arrs["root"])[ids], g["root_after_reset"][k][ids], atol=2e-5)
            np.testing.assert_allclose(be.np(arrs["dof"])[ids], g["dof_after_reset"][k][ids], atol=2e-5)
            assert (be.np(b["progress"])[ids] == 0).all() and (be.np(b["reset"]) == 0).all()
        # physics stand-in: the golden's body / joint state of this step
        put(arrs["rbs"], g["state_in"][k].astype(F))
        put(arrs["root"], g["state_in"][k][:, 0].astype(F))
        put(arrs["dof"], g["dof_in"][k].astype(F))
        put(arrs["df"], g["dof_force"][k].astype(F))
        assert be.im_post_physics(mstruct, lib, prm, sim, bufs(amp[cur], amp[1 - cur])) == 0
        be.sync()
        cur = 1 - cur
        np.testing.assert_array_equal(be.np(b["progress"]), g["progress"][k], err_msg=f"step {k}")
        np.testing.assert_array_equal(be.np(b["reset"]), g["reset"][k], err_msg=f"step {k}: reset flags")
        np.testing.assert_array_equal(be.np(b["term"]), g["terminate"][k], err_msg=f"step {k}: terminate flags")
        np.testing.assert_allclose(be.np(b["rew"]), g["rew"][k], atol=1e-5, err_msg=f"step {k}")
        np.testing.assert_allclose(be.np(b["raw"]), g["rew_raw"][k], atol=1e-5, rtol=1e-5, err_msg=f"step {k}")
        np.testing.assert_allclose(be.np(b["obs"]), g["obs"][k], atol=2e-5, err_msg=f"step {k}: observations")
        np.testing.assert_allclose(be.np(amp[cur]), g["amp"][k], atol=2e-5, err_msg=f"step {k}: AMP history")
        n_term += int(g["terminate"][k].sum())
    assert n_resets >= N + 8 and n_term >= 5      # the sequence really exercises resets after time-outs and after terminations
