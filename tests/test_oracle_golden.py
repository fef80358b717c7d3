"""Pin the numpy oracle (oracle/phc_oracle.py) against golden vectors produced by the
reference's own code (oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest

import phc_oracle as po

TOL = dict(rtol=0, atol=1e-4)  # north_star: <=1e-4 on FK / quaternion floats
F = np.float32


def test_quat_kat(golden):
    g = golden("quat_kat")
    qa, qb, v, t, em = g["qa"], g["qb"], g["v"], g["t"], g["em"]
    np.testing.assert_allclose(po.quat_mul(qa, qb), g["quat_mul"], atol=1e-6)
    np.testing.assert_array_equal(po.quat_conjugate(qa), g["quat_conjugate"])
    np.testing.assert_allclose(po.my_quat_rotate(qa, v), g["my_quat_rotate"], atol=2e-6)
    ang, ax = po.quat_to_angle_axis(qa)
    np.testing.assert_allclose(ang, g["angle"], atol=1e-5)
    np.testing.assert_allclose(ax, g["axis"], atol=1e-4)
    np.testing.assert_allclose(po.quat_to_exp_map(qa), g["quat_to_exp_map"], atol=1e-4)
    np.testing.assert_allclose(po.quat_to_tan_norm(qa), g["quat_to_tan_norm"], atol=2e-6)
    np.testing.assert_allclose(po.exp_map_to_quat(em), g["exp_map_to_quat"], atol=2e-6)
    np.testing.assert_allclose(po.slerp(qa, qb, t), g["slerp"], atol=1e-5)
    np.testing.assert_allclose(po.calc_heading(qa), g["calc_heading"], atol=1e-5)
    np.testing.assert_allclose(po.calc_heading_quat(qa), g["calc_heading_quat"], atol=1e-6)
    np.testing.assert_allclose(po.calc_heading_quat_inv(qa), g["calc_heading_quat_inv"], atol=1e-6)


def _clips(golden):
    c = golden("motion_clips")
    return [{"pose_quat_global": c[f"{k}/pose_quat_global"], "root_trans_offset": c[f"{k}/root_trans_offset"], "fps": 30}
            for k in c["keys"]]


def test_motion_load_fk_and_velocities(golden):
    """M3-M6: FK + finite-difference velocities == reference MotionLibSMPL.load_motions."""
    sk = golden("skeleton_smpl")
    lib_g = golden("motion_lib_eval")
    clips = _clips(golden)
    lib = po.build_motion_lib(sk["parent_indices"], sk["local_translation"], [clips[i] for i in lib_g["curr_motion_ids"]])
    for k in ("gts", "grs", "lrs", "gvs", "gavs", "dvs"):
        np.testing.assert_allclose(lib[k], lib_g[k], err_msg=k, **TOL)
    for k in ("motion_lengths", "motion_dt"):
        np.testing.assert_array_equal(lib[k], lib_g[k], err_msg=k)
    for k in ("motion_num_frames", "length_starts"):
        np.testing.assert_array_equal(lib[k], lib_g[k], err_msg=k)


def test_frame_blend_indices_bit_exact(golden):
    """M8: frame indices are the bit-exact part of the contract."""
    g = golden("motion_lib_eval")
    ids = g["ms_ids"]
    i0, i1, bl = po.calc_frame_blend(g["ms_times"], g["motion_lengths"][ids], g["motion_num_frames"][ids], g["motion_dt"][ids])
    np.testing.assert_array_equal(i0, g["ms_idx0"])
    np.testing.assert_array_equal(i1, g["ms_idx1"])
    np.testing.assert_array_equal(bl, g["ms_blend"])


def test_sample_time_interval_bit_exact(golden):
    g = golden("motion_lib_eval")
    t = po.sample_time_interval(g["sti_phase"], g["motion_lengths"][g["ms_ids"]])
    np.testing.assert_array_equal(t, g["sti_time"])


def test_get_motion_state(golden):
    """M9 on the reference's own frame tensors."""
    g = golden("motion_lib_eval")
    res = po.get_motion_state(g, g["ms_ids"], g["ms_times"], g["ms_offset"])
    for k in ("root_pos", "root_rot", "dof_pos", "root_vel", "root_ang_vel", "dof_vel", "rg_pos", "rb_rot", "body_vel", "body_ang_vel"):
        np.testing.assert_allclose(res[k], g["ms_" + k], err_msg=k, atol=2e-5, rtol=0)


def test_reward_reset_obs(golden):
    g = golden("task_fns")
    rew, raw = po.compute_imitation_reward(g["body_pos"], g["body_rot"], g["body_vel"], g["body_ang_vel"],
                                           g["ref_pos"], g["ref_rot"], g["ref_vel"], g["ref_ang_vel"])
    np.testing.assert_allclose(rew, g["reward"], atol=1e-5)
    np.testing.assert_allclose(raw, g["reward_raw"], atol=1e-5)
    np.testing.assert_allclose(po.power_reward(g["dof_force"], g["dof_vel"], g["progress"]), g["power_reward"], atol=1e-5, rtol=1e-5)
    rid = g["reset_body_ids"]
    td = np.full((g["body_pos"].shape[0], len(rid)), 0.25, dtype=np.float32)
    reset, term = po.compute_humanoid_im_reset(g["progress"], g["body_pos"][:, rid], g["ref_pos"][:, rid], g["pass_time"], td)
    np.testing.assert_array_equal(reset, g["reset"])
    np.testing.assert_array_equal(term, g["terminate"])
    assert term.sum() > 0 and (1 - term).sum() > 0  # both branches exercised
    reset, term = po.compute_humanoid_im_reset(g["progress"], g["body_pos"][:, rid], g["ref_pos"][:, rid], g["pass_time"], td, use_mean=True)
    np.testing.assert_array_equal(reset, g["reset_mean"])
    np.testing.assert_array_equal(term, g["terminate_mean"])
    so = po.compute_humanoid_observations_smpl_max(g["body_pos"], g["body_rot"], g["body_vel"], g["body_ang_vel"])
    assert so.shape[1] == 358
    np.testing.assert_allclose(so, g["self_obs"], atol=1e-5)
    to = po.compute_imitation_observations_v6(g["body_pos"][:, 0], g["body_rot"][:, 0], g["body_pos"], g["body_rot"], g["body_vel"],
                                              g["body_ang_vel"], g["ref1_pos"], g["ref1_rot"], g["ref1_vel"], g["ref1_ang_vel"])
    assert to.shape[1] == 576
    np.testing.assert_allclose(to, g["task_obs"], atol=1e-5)
    kid = g["key_body_ids"]
    amp = po.build_amp_observations_smpl(g["body_pos"][:, 0], g["body_rot"][:, 0], g["body_vel"][:, 0], g["body_ang_vel"][:, 0],
                                         g["dof_pos"], g["dof_vel"], g["body_pos"][:, kid], g["dof_subset"])
    assert amp.shape[1] == 196
    np.testing.assert_allclose(amp, g["amp_obs"], atol=1e-5)


def test_amass_conversion_matches_reference_poselib(golden):
    """f-4: AMASS -> motion pkl (phc_amd.utils.convert_amass) == the reference's poselib calls driven as its converter script does."""
    from phc_amd.utils.convert_amass import convert
    g = golden("amass_convert")
    out = convert({"clip": {"pose_aa": g["pose_aa_in"], "trans": g["trans"], "betas": np.ones((1, 10)), "gender": "male"}})["clip"]
    np.testing.assert_allclose(out["root_trans_offset"], g["root_trans_offset"], atol=1e-6)   # float32 pelvis offset of the tree
    np.testing.assert_array_equal(out["pose_aa"], g["pose_aa"])
    for k in ("pose_quat_global", "pose_quat"):   # rotations equal up to the quaternion sign (scipy does not canonicalise)
        np.testing.assert_allclose(np.abs((out[k] * g[k]).sum(-1)), 1.0, atol=1e-6, err_msg=k)
    assert (out["beta"] == 0).all() and out["gender"] == "neutral" and out["fps"] == 30.0
    # and the result loads: global rotations -> local -> global round trip of the motion library's clip processing
    sk = golden("skeleton_smpl")
    lib = po.build_motion_lib(sk["parent_indices"], sk["local_translation"], [dict(out, fps=30)])
    np.testing.assert_allclose(np.abs((lib["grs"] * out["pose_quat_global"]).sum(-1)), 1.0, atol=1e-6)


def test_poselib_rotation_kat(golden):
    """The oracle's poselib restatements on the inputs of the reference's own rotation check (core/tests/test_rotation.py): quat_normalize,
    quat_rotate, the rotate / inverse-rotate round trip it asserts, quat_mul_norm and quat_angle_axis."""
    g = golden("poselib_kat")
    np.testing.assert_allclose(po._pl_quat_normalize(g["q"].astype(np.float64)), g["q_normalized"], atol=1e-6)
    np.testing.assert_allclose(po._pl_quat_rotate(g["q_normalized"].astype(np.float64), g["x"].astype(np.float64)), g["rotated"], atol=1e-6)
    rot = np.broadcast_to(g["rot"].astype(np.float64), g["xs"].shape[:-1] + (4,))
    back = po._pl_quat_rotate(po.quat_conjugate(rot), po._pl_quat_rotate(rot, g["xs"]))
    np.testing.assert_allclose(back, g["xs"], atol=1e-6)          # the reference's own assertion (:30), rot is a unit quaternion
    np.testing.assert_allclose(g["roundtrip"], g["xs"], atol=1e-6)
    qa_n = po._pl_quat_normalize(g["qa"])
    np.testing.assert_allclose(qa_n, g["qa_n"], atol=1e-12)
    qb_n = po._pl_quat_normalize(g["qb"])
    np.testing.assert_allclose(po._pl_quat_mul_norm(qa_n, qb_n), g["mul_norm"], atol=1e-12)
    ang, ax = po._pl_quat_angle_axis(po._pl_quat_mul_norm(qa_n, po.quat_conjugate(qb_n)))
    np.testing.assert_allclose(ang, g["diff_angle"], atol=1e-9)
    np.testing.assert_allclose(ax, g["diff_axis"], atol=1e-9)


@pytest.mark.skipif(not os.path.exists("/root/reference/poselib/poselib/skeleton/tests/ant.xml"), reason="reference fixture only exists in the build container")
def test_model_compiler_on_the_reference_ant_fixture(golden):
    """compile_mjcf on the reference's only skeleton fixture (ant.xml: fixed bodies, single hinges, degrees, no free joint) gives the
    tree SkeletonTree.from_mjcf reads from it."""
    from phc_amd.model import compile_mjcf
    g = golden("poselib_kat")
    m = compile_mjcf("/root/reference/poselib/poselib/skeleton/tests/ant.xml")
    assert m["body_names"] == list(g["ant_names"])
    np.testing.assert_array_equal(np.array(m["parent"]), g["ant_parents"])
    np.testing.assert_allclose(np.array(m["local_translation"]), g["ant_local_translation"], atol=1e-7)
    assert len(m["dof_names"]) == 8 and abs(m["dof_upper"][0] - np.deg2rad(40)) < 1e-9


def test_amp_observation_oracle_upright_and_shape_columns(golden):
    """The numpy restatement of build_amp_observations_smpl, `upright=False` and shape / limb columns included, against the reference's
    own outputs (oracle/gen_golden_shape_upright.py)."""
    g, gs = golden("task_fns"), golden("obs_shape_upright")
    kid, sub = g["key_body_ids"], g["dof_subset"]
    for local_root in (True, False):
        for upright in (True, False):
            got = po.build_amp_observations_smpl(g["body_pos"][:, 0], g["body_rot"][:, 0], g["body_vel"][:, 0], g["body_ang_vel"][:, 0], g["dof_pos"],
                                                 g["dof_vel"], g["body_pos"][:, kid], sub, local_root_obs=local_root, upright=upright,
                                                 shape_params=gs["shape"], limb_weight_params=gs["limb"])
            np.testing.assert_allclose(got, gs[f"amp_l{int(local_root)}u{int(upright)}"], atol=2e-6)


@pytest.mark.parametrize("rb", ["h1", "g1"])
def test_robot_oracle_functions_vs_reference_goldens(golden, rb):
    """The numpy restatements of the robot path (MotionLibReal lookup, extended bodies, reward over NB + E bodies, self / task observation
    on robot shapes, robot AMP observation) against the goldens of oracle/gen_golden_h1.py -- outputs of the reference's own
    `MotionLibReal.get_motion_state` and jit functions.  These restatements are what the -m gpu tests at BASELINE configs[4]'s own size
    (4096 envs) compare the HIP path with."""
    g, t = golden(f"motion_lib_{rb}"), golden(f"task_fns_{rb}")
    lib = {k: g[k] for k in ("gts", "grs", "gvs", "gavs", "dvs", "dof_pos", "gts_t", "grs_t", "motion_lengths", "motion_dt", "motion_num_frames",
                             "length_starts")}
    r = po.get_motion_state_robot(lib, g["ms_ids"], g["ms_times"].astype(F), g["ms_offset"].astype(F))
    for k in ("rg_pos", "rb_rot", "body_vel", "body_ang_vel", "dof_pos", "dof_vel", "rg_pos_t", "rg_rot_t", "root_pos", "root_rot"):
        np.testing.assert_allclose(r[k], g["ms_" + k], atol=2e-6, err_msg=k)
    NB = t["body_pos"].shape[1]
    dt = F(4 * (1 / 200))
    mt = (t["progress"].astype(F) * dt + t["start_times"]).astype(F)
    r0 = po.get_motion_state_robot(lib, t["env_motion"], mt, np.zeros((len(mt), 3), F))
    r1 = po.get_motion_state_robot(lib, t["env_motion"], ((t["progress"] + 1).astype(F) * dt + t["start_times"]).astype(F), np.zeros((len(mt), 3), F))
    bpe, bre = po.extend_bodies(t["body_pos"], t["body_rot"], t["ext_parent"], t["ext_pos"])
    rew, raw = po.compute_imitation_reward(bpe, bre, t["body_vel"], t["body_ang_vel"], np.concatenate([r0["rg_pos"], r0["rg_pos_t"][:, NB:]], 1),
                                           np.concatenate([r0["rb_rot"], r0["rg_rot_t"][:, NB:]], 1), r0["body_vel"], r0["body_ang_vel"])
    np.testing.assert_allclose(raw, t["reward_raw"], atol=2e-6)
    np.testing.assert_allclose(rew, t["reward"], atol=2e-6)
    np.testing.assert_allclose(po.power_reward(t["dof_force"], t["dof_vel"], t["progress"]), t["power_reward"], rtol=1e-6, atol=1e-6)
    reset, term = po.compute_humanoid_im_reset(t["progress"], t["body_pos"], r0["rg_pos"], mt >= lib["motion_lengths"][t["env_motion"]],
                                               np.full((len(mt), NB), 0.25, F))
    np.testing.assert_array_equal(reset, t["reset"])
    np.testing.assert_array_equal(term, t["terminate"])
    np.testing.assert_allclose(po.compute_humanoid_observations_smpl_max(t["body_pos"], t["body_rot"], t["body_vel"], t["body_ang_vel"]), t["self_obs"], atol=3e-6)
    to = po.compute_imitation_observations_v6(t["body_pos"][:, 0], t["body_rot"][:, 0], t["body_pos"], t["body_rot"], t["body_vel"], t["body_ang_vel"],
                                              r1["rg_pos"], r1["rb_rot"], r1["body_vel"], r1["body_ang_vel"])
    np.testing.assert_allclose(to, t["task_obs"], atol=5e-6)
    amp = po.build_amp_observations_robot(t["body_pos"][:, 0], t["body_rot"][:, 0], t["body_vel"][:, 0], t["body_ang_vel"][:, 0], t["dof_pos"], t["dof_vel"],
                                          t["body_pos"][:, t["key_body_ids"]])
    np.testing.assert_allclose(amp, t["amp_obs"], atol=2e-6)


@pytest.mark.parametrize("tag", ["h1_pdv1", "h1_pdv2", "g1_pdv1"])
def test_pd_torque_oracle_vs_reference_compute_torques(golden, tag):
    """`compute_torques_pd` == the reference's own `Humanoid._compute_torques` (oracle/gen_golden_torques.py), bit for bit."""
    g = {k.split("/", 1)[1]: v for k, v in golden("pd_torques").items() if k.startswith(tag + "/")}
    tq = po.compute_torques_pd(g["actions"], g["dof_pos"], g["dof_vel"], g["p_gains"], g["d_gains"], g["default_dof_pos"][0], g["torque_limits"])
    np.testing.assert_array_equal(tq, g["torques"])
