"""Robot path -- Unitree H1 (BASELINE config 5, 20 bodies) and G1 (env_im_g1_phc, 38 bodies: the 64-lane kernel instantiations) --
against golden vectors produced by the reference's own code (oracle/gen_golden_h1.py [h1|g1]):
model constants vs Humanoid_Batch's reading of the robot's MJCF, clip FK vs `Humanoid_Batch.fk_batch` / `MotionLibReal.load_motions`,
the lookup kernel vs `MotionLibReal.get_motion_state`, and one post-physics step (extended-body reward, per-body observations,
robot AMP observation) vs the reference's jit functions.  Kernel tests run on both backends of tests/backends.py."""
import numpy as np
import pytest

from backends import BACKENDS, get_backend, model_on, motion_lib_on
from phc_amd import abi
from phc_amd.model import load_model
from phc_amd.motion_lib import process_clip_real

F = np.float32
KEY_BODIES = {"h1": ["left_ankle_link", "right_ankle_link", "left_elbow_link", "right_elbow_link"],
              "g1": ["left_ankle_roll_link", "right_ankle_roll_link", "left_zero_link", "right_zero_link"]}
ROBOTS = ["h1", "g1"]


@pytest.mark.parametrize("rb", ROBOTS)
def test_robot_model_matches_reference_skeleton(golden, rb):
    sk = golden(f"skeleton_{rb}")
    m = load_model(f"{rb}_humanoid")
    assert m.body_names == list(sk["node_names"]) and m.all_revolute
    np.testing.assert_array_equal(m.parent, sk["parents"])
    np.testing.assert_allclose(m.local_translation, sk["local_translation"], atol=1e-7)
    # wxyz; g1.xml writes quat="1 0 0 1" for the thumb links: Humanoid_Batch keeps it as read and lets quaternion_to_matrix (2 / |q|^2)
    # absorb the norm, the model compiler normalises
    np.testing.assert_allclose(m.local_rotation, sk["local_rotation"] / np.linalg.norm(sk["local_rotation"], axis=-1, keepdims=True), atol=1e-6)
    np.testing.assert_allclose(m.dof_axis, sk["dof_axis"], atol=1e-12)
    np.testing.assert_allclose(np.stack([m.dof_lower, m.dof_upper], -1), sk["joints_range"], atol=1e-9)
    if rb == "h1":
        assert abs(m.total_mass - 51.436) < 2e-3                                           # env_im_h1_phc.yaml default_humanoid_mass
    else:
        assert m.num_bodies == 38 and m.max_level == 9


def _ext(golden, rb):
    sk = golden(f"skeleton_{rb}")
    e_rot = np.tile(np.array([1.0, 0, 0, 0]), (len(sk["ext_parents"]), 1))
    return sk["ext_parents"], sk["ext_offsets"], e_rot


@pytest.mark.parametrize("rb", ROBOTS)
def test_robot_clip_fk_matches_reference(golden, rb):
    """process_clip_real == Humanoid_Batch.fk_batch(return_full=True) as concatenated by MotionLibReal.load_motions."""
    g = golden(f"motion_lib_{rb}")
    c = golden(f"motion_clips_{rb}")
    m = load_model(f"{rb}_humanoid")
    ep, eo, er = _ext(golden, rb)
    per = [process_clip_real(m.parent, m.local_translation, m.local_rotation, ep, eo, er, c[f"{k}/pose_aa"], c[f"{k}/root_trans_offset"], 30)
           for k in c["keys"]]
    order = [0, 1, 2, 0, 1, 2]   # 6 envs, sequential sampling
    # gavs: the reference differentiates fp32 quaternions with arccos(2 w^2 - 1) (rotation3d.py quat_angle_axis), which for the
    # ~3 mrad frame-to-frame rotations of these clips is conditioned like 1/angle: its own output carries ~1e-3..1e-2 rad/s of
    # rounding noise.  The fp64 restatement is compared at that level (and is the more accurate of the two).
    for k, tol in (("gts", 2e-6), ("gts_t", 2e-6), ("gvs", 2e-4), ("gavs", 1e-2), ("dof_pos", 1e-6), ("dvs", 2e-5)):
        np.testing.assert_allclose(np.concatenate([per[i][k] for i in order]), g[k], atol=tol, err_msg=k)
    for k in ("grs", "grs_t"):   # same rotation AND same sign convention as matrix_to_quaternion
        np.testing.assert_allclose(np.concatenate([per[i][k] for i in order]), g[k], atol=2e-6, err_msg=k)


def _lib_from_golden(golden, rb):
    g = golden(f"motion_lib_{rb}")
    return {k: g[k] for k in ("gts", "grs", "gvs", "gavs", "dvs", "dof_pos", "gts_t", "grs_t", "motion_lengths", "motion_dt", "motion_num_frames",
                              "length_starts")}


@pytest.mark.parametrize("rb", ROBOTS)
@pytest.mark.parametrize("backend", BACKENDS)
def test_robot_motion_state_vs_reference_golden(golden, backend, rb):
    be = get_backend(backend)
    g = golden(f"motion_lib_{rb}")
    lib, keep = motion_lib_on(be, _lib_from_golden(golden, rb))
    n = len(g["ms_ids"])
    NB, NE = g["ms_rg_pos"].shape[1], g["ms_rg_pos_t"].shape[1] - g["ms_rg_pos"].shape[1]
    out = dict(rg_pos=be.zeros((n, NB, 3)), rb_rot=be.zeros((n, NB, 4)), body_vel=be.zeros((n, NB, 3)), body_ang_vel=be.zeros((n, NB, 3)),
               dof_pos=be.zeros((n, NB - 1)), dof_vel=be.zeros((n, NB - 1)), pe=be.zeros((n, NE, 3)), re=be.zeros((n, NE, 4)))
    assert be.motion_state(lib, n, be.arr(g["ms_ids"].astype(np.int64)), be.arr(g["ms_times"].astype(F)), be.arr(g["ms_offset"].astype(F)),
                           out["rg_pos"], out["rb_rot"], out["body_vel"], out["body_ang_vel"], out["dof_pos"], out["dof_vel"], None, None, None,
                           out["pe"], out["re"]) == 0
    be.sync()
    o = {k: be.np(v) for k, v in out.items()}
    for k in ("rg_pos", "body_vel", "body_ang_vel", "dof_pos", "dof_vel"):
        np.testing.assert_allclose(o[k], g["ms_" + k], atol=2e-5, err_msg=k)
    np.testing.assert_allclose(o["rb_rot"], g["ms_rb_rot"], atol=2e-5)       # same slerp on the same (not re-normalised) stored quaternions
    np.testing.assert_allclose(o["pe"], g["ms_rg_pos_t"][:, NB:], atol=2e-5)
    np.testing.assert_allclose(o["re"], g["ms_rg_rot_t"][:, NB:], atol=2e-5)


def robot_im_params(be, model, ext_parent, ext_pos, rb="h1", **extra):
    names = model.body_names
    NB = len(names)
    tabs = abi.task_index_tables(model, names, names, KEY_BODIES[rb], has_dof_subset=False)
    track_slot, reset_mask, key_ids, amp_slot = (be.arr(t) for t in tabs[:4])
    td = be.arr(np.full(64, 0.25, dtype=F))
    ep, eo = be.arr(np.asarray(ext_parent, np.int32)), be.arr(np.asarray(ext_pos, F))
    specs = dict(k_pos=100, k_rot=10, k_vel=0.1, k_ang_vel=0.1, w_pos=0.5, w_rot=0.3, w_vel=0.1, w_ang_vel=0.1)
    prm = abi.im_params_struct(dt=4 * (1 / 200), max_episode_length=300, reward_specs=specs, power_reward=True, power_coefficient=0.0005,
                               enable_early_termination=True, use_mean_termination=False, disable_collision_check=False, local_root_obs=True,
                               root_height_obs=True, num_track_bodies=NB, track_slot=track_slot, reset_mask=reset_mask, num_reset_bodies=NB,
                               first_reset_body=0, termination_distances=td, num_key_bodies=4, key_body_ids=key_ids, num_amp_joints=tabs[4],
                               amp_joint_slot=amp_slot, num_amp_obs_steps=10, num_amp_obs_per_step=13 + 2 * (NB - 1) + 12, num_self_obs=NB * 15 - 2,
                               num_task_obs=NB * 24,
                               dofs_per_joint=1, ext_parent=ep, ext_offset=eo, **extra)
    prm._keepalive = (track_slot, reset_mask, key_ids, amp_slot, td, ep, eo)
    return prm


@pytest.mark.parametrize("rb", ROBOTS)
@pytest.mark.parametrize("backend", BACKENDS)
def test_robot_post_physics_vs_reference_golden(golden, backend, rb):
    """Reward incl. the extended bodies (humanoid_im.py:916-923), power reward over the scalar DoFs, reset flags, self-obs + task-obs
    (H1 298 + 480, G1 568 + 912 floats) and the robot AMP observation (63 / 99) == the reference's functions on the same inputs."""
    be = get_backend(backend)
    g = golden(f"task_fns_{rb}")
    lib, keep = motion_lib_on(be, _lib_from_golden(golden, rb))
    model, mstruct, keepm = model_on(be, f"{rb}_humanoid")
    N, NB = g["body_pos"].shape[:2]
    ND, A, SO = NB - 1, 13 + 2 * (NB - 1) + 12, NB * 15 - 2
    prm = robot_im_params(be, model, g["ext_parent"], g["ext_pos"], rb)
    assert prm.num_amp_joints == ND and prm.num_ext_bodies == {"h1": 3, "g1": 1}[rb]
    rbs = np.concatenate([g["body_pos"], g["body_rot"], g["body_vel"], g["body_ang_vel"]], axis=-1).astype(F)
    arrs = dict(root=be.arr(rbs[:, 0, :]), dof=be.arr(np.stack([g["dof_pos"], g["dof_vel"]], -1).astype(F)), rbs=be.arr(rbs), cf=be.zeros((N, NB, 3)),
                df=be.arr(g["dof_force"].astype(F)), pd=be.zeros((N, ND)))
    sim = abi.sim_state_struct(N, arrs["root"], arrs["dof"], arrs["rbs"], arrs["cf"], arrs["df"], arrs["pd"])
    rng = np.random.default_rng(0)
    amp_in_np = rng.standard_normal((N, 10, A)).astype(F)
    amp_in, amp_out = be.arr(amp_in_np), be.zeros((N, 10, A))
    b = dict(progress=be.arr((g["progress"] - 1).astype(np.int64)), reset=be.zeros(N, np.int64), term=be.zeros(N, np.int64), rew=be.zeros(N),
             raw=be.zeros((N, 5)), obs=be.zeros((N, SO + NB * 24)), mids=be.arr(g["env_motion"].astype(np.int64)), st=be.arr(g["start_times"].astype(F)),
             so=be.zeros(N), goff=be.zeros((N, 3)), rbp=be.zeros((N, NB, 3)), rdp=be.zeros((N, ND)))
    buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp_in, amp_out, b["mids"], b["st"], b["so"],
                                b["goff"], ref_body_pos=b["rbp"], ref_dof_pos=b["rdp"])
    assert be.im_post_physics(mstruct, lib, prm, sim, buf) == 0
    be.sync()
    o = {k: be.np(v) for k, v in b.items()}
    amp_out = be.np(amp_out)
    np.testing.assert_allclose(o["raw"][:, :4], g["reward_raw"], atol=1e-5)
    np.testing.assert_allclose(o["raw"][:, 4], g["power_reward"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(o["rew"], g["reward"] + g["power_reward"], atol=1e-5)
    np.testing.assert_array_equal(o["reset"], g["reset"])
    np.testing.assert_array_equal(o["term"], g["terminate"])
    np.testing.assert_allclose(o["obs"][:, :SO], g["self_obs"], atol=1e-5)
    np.testing.assert_allclose(o["obs"][:, SO:], g["task_obs"], atol=2e-5)
    np.testing.assert_allclose(amp_out[:, 0], g["amp_obs"], atol=1e-5)
    np.testing.assert_array_equal(amp_out[:, 1:], amp_in_np[:, :-1])
    np.testing.assert_allclose(o["rbp"], g["ref1_pos"], atol=2e-5)
    np.testing.assert_allclose(o["rdp"], g["ref1_dof_pos"], atol=2e-5)


@pytest.mark.parametrize("rb", ROBOTS)
@pytest.mark.parametrize("backend", BACKENDS)
def test_robot_multi_step_rollout_vs_reference_env(golden, backend, rb):
    """14 consecutive H1 / G1 env steps against the thin CPU reference env of oracle/gen_golden_rollout_h1.py [h1|g1] (the reference's
    `MotionLibReal`, extended-body reward, robot AMP observation ... in its method order, kinematic stand-in for the physics): resets from
    the clip's joint angles, re-initialised AMP history (63 / 99 x 10), progress / flags / observations across resets, time-outs and
    terminations (G1: the 64-lane instantiations of the reset and post-physics kernels)."""
    be = get_backend(backend)
    g = golden(f"rollout_ref_env_{rb}")
    lib, keep = motion_lib_on(be, _lib_from_golden(golden, rb))
    model, mstruct, keepm = model_on(be, f"{rb}_humanoid")
    K, N = g["obs"].shape[:2]
    NB = g["state_in"].shape[2]
    ND, A, NO = NB - 1, g["amp"].shape[-1], g["obs"].shape[-1]
    prm = robot_im_params(be, model, g["ext_parent"], g["ext_pos"], rb)
    arrs = dict(root=be.zeros((N, 13)), dof=be.zeros((N, ND, 2)), rbs=be.zeros((N, NB, 13)), cf=be.zeros((N, NB, 3)), df=be.zeros((N, ND)), pd=be.zeros((N, ND)))
    sim = abi.sim_state_struct(N, arrs["root"], arrs["dof"], arrs["rbs"], arrs["cf"], arrs["df"], arrs["pd"])
    amp = [be.zeros((N, 10, A)), be.zeros((N, 10, A))]
    cur = 0
    b = dict(progress=be.zeros(N, np.int64), reset=be.arr(np.ones(N, np.int64)), term=be.zeros(N, np.int64), rew=be.zeros(N), raw=be.zeros((N, 5)),
             obs=be.zeros((N, NO)), mids=be.arr(g["motion_ids"].astype(np.int64)), st=be.zeros(N), so=be.zeros(N), goff=be.zeros((N, 3)))
    bufs = lambda a_in, a_out: abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], a_in, a_out, b["mids"], b["st"],
                                                     b["so"], b["goff"])

    def put(dst, src):
        if isinstance(dst, np.ndarray):
            dst[...] = src
        else:
            dst.copy_(be.arr(np.ascontiguousarray(src)))
    n_resets = 0
    for k in range(K):
        ids = g["reset_ids"][k]
        ids = ids[ids >= 0].astype(np.int64)
        if len(ids):
            np.testing.assert_array_equal(np.sort(ids), np.nonzero(be.np(b["reset"]))[0], err_msg=f"step {k}: envs to reset")
            assert be.im_reset(mstruct, lib, prm, sim, bufs(amp[cur], amp[cur]), len(ids), be.arr(ids), be.arr(g["reset_phase"][k][:len(ids)].astype(F)), 0) == 0
            be.sync()
            n_resets += len(ids)
            np.testing.assert_array_equal(be.np(b["st"])[ids], g["start_after_reset"][k][ids], err_msg=f"step {k}: start times")
            np.testing.assert_allclose(be.np(b["obs"])[ids], g["obs_after_reset"][k][ids], atol=3e-5, err_msg=f"step {k}: obs after reset")
            np.testing.assert_allclose(be.np(arrs["root"])[ids], g["root_after_reset"][k][ids], atol=3e-5)
            np.testing.assert_allclose(be.np(arrs["dof"])[ids], g["dof_after_reset"][k][ids], atol=3e-5)
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
        np.testing.assert_allclose(be.np(b["obs"]), g["obs"][k], atol=3e-5, err_msg=f"step {k}: observations")
        np.testing.assert_allclose(be.np(amp[cur]), g["amp"][k], atol=3e-5, err_msg=f"step {k}: AMP history")
    assert n_resets >= N + 6 and g["terminate"].sum() >= 5


# ------------------------------------------------------------------------------------------------------------------ S9: the explicit `pd` torque
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("tag,asset", [("h1_pdv1", "h1_humanoid"), ("h1_pdv2", "h1_humanoid"), ("g1_pdv1", "g1_humanoid")])
def test_pd_torque_held_by_the_stepper_equals_the_reference_compute_torques(golden, backend, tag, asset):
    """S9: `Humanoid._compute_torques` (humanoid.py:1575-1599: clip(p (a scale + q0 - q) - d qd, +-limit), recomputed before every simulate
    call, :1608-1616) run on a `__new__`-made reference task (oracle/gen_golden_torques.py; gains / default pose / limits cut out of the
    reference's `_build_env` and `_process_dof_props`).  The stepper's control_mode 1 computes the torque from the state at the start of
    the simulate call and holds it: after ONE simulate call the published joint force (S5) IS that torque -- through the C ABI, both backends.
    robots.py's tables (what the task installs) equal the reference's."""
    from phc_amd.robots import ROBOTS, apply_robot_gains
    g = {k.split("/", 1)[1]: v for k, v in golden("pd_torques").items() if k.startswith(tag + "/")}
    rb, pd_v = tag.split("_")[0], int(tag[-1])
    np.testing.assert_array_equal(np.asarray(ROBOTS[rb]["p_gains"][pd_v], dtype=F), g["p_gains"])
    np.testing.assert_array_equal(np.asarray(ROBOTS[rb]["d_gains"][pd_v], dtype=F), g["d_gains"])
    np.testing.assert_array_equal(np.asarray(ROBOTS[rb]["default_dof_pos"], dtype=F), g["default_dof_pos"][0])
    np.testing.assert_array_equal(np.broadcast_to(np.asarray(ROBOTS[rb]["torque_limit"], dtype=F), g["torque_limits"].shape), g["torque_limits"])
    be = get_backend(backend)
    m = load_model(asset)
    apply_robot_gains(m, ROBOTS[rb], pd_v)
    ints, floats = m.pack()
    keep = (be.arr(ints), be.arr(floats))
    ms = abi.model_struct(keep[0], keep[1], m.num_bodies, m.num_dof, m.max_level, len(m.contact_body))
    n, nd, nb = g["actions"].shape[0], m.num_dof, m.num_bodies
    root = np.zeros((n, 13), F)
    root[:, 2], root[:, 6] = 3.0, 1.0                                  # in the air: nothing but the drives acts on the joints
    dof = np.stack([g["dof_pos"], g["dof_vel"]], -1).astype(F)
    a = dict(root=be.arr(root), dof=be.arr(dof), rbs=be.zeros((n, nb, 13)), cf=be.zeros((n, nb, 3)), df=be.zeros((n, nd)), pd=be.zeros((n, nd)))
    sim = abi.sim_state_struct(n, a["root"], a["dof"], a["rbs"], a["cf"], a["df"], a["pd"])
    params = abi.sim_params_struct(sim_dt=1 / 200, substeps=1, control_freq_inv=1, control_mode=1)
    # the task's action map in this mode (humanoid_im.py `_torque_target_offset / _scale`): target = default_dof_pos + action_scale * action
    off, scale = be.arr(g["default_dof_pos"][0].astype(F)), be.arr(np.ones(nd, F))
    assert be.sim_step(ms, params, sim, be.arr(g["actions"].astype(F)), off, scale, be.arr(np.zeros(nd, np.int32)), 1) == 0
    be.sync()
    np.testing.assert_allclose(be.np(a["df"]), g["torques"], rtol=2e-5, atol=2e-4)
    sat = np.abs(g["torques"]) >= g["torque_limits"] - 1e-6
    assert 0.1 < sat.mean() < 0.95                                      # both branches of the clip are exercised
