"""Stepper (S10) checks, on both backends of tests/backends.py (`hostemu` = g++ build of phc_amd/csrc/phc_aba.h on
the CPU; `hip` = the real k_sim_step kernel on an MI355X, -m gpu): the articulated-body recursion against
  * the fp64 dense mass-matrix oracle (oracle/dyn_oracle.py) -- an independent formulation, and
  * physical invariants (energy / momentum conservation, free fall, standing stability).
There is no reference implementation of the dynamics (closed Isaac Gym): parity is pinned to
the builder's own oracle, see DESIGN.md."""
import numpy as np
import pytest

import dyn_oracle as do
from backends import BACKENDS, get_backend, model_on
from phc_amd import abi

F = np.float32


def random_states(model, n, rng, height=0.95, vel=1.0, pose=0.5):
    nb, nd = model.num_bodies, model.num_dof
    root = np.zeros((n, 13), F)
    root[:, 0:2] = rng.normal(0, 1.0, (n, 2))
    root[:, 2] = height + rng.normal(0, 0.03, n)
    q = rng.normal(0, 1, (n, 4)) * np.array([0.15, 0.15, 0.5, 0]) + np.array([0, 0, 0, 1.0])
    root[:, 3:7] = q / np.linalg.norm(q, axis=-1, keepdims=True)
    root[:, 7:10] = rng.normal(0, vel, (n, 3))
    root[:, 10:13] = rng.normal(0, vel, (n, 3))
    dof = np.zeros((n, nd, 2), F)
    dof[:, :, 0] = rng.normal(0, pose, (n, nd))
    dof[:, :, 1] = rng.normal(0, 2 * vel, (n, nd))
    target = (dof[:, :, 0] + rng.normal(0, 0.3, (n, nd))).astype(F)
    return root, dof, target


def run_step(be, model, mstruct, root, dof, target, params, num_sim_calls=2):
    n = root.shape[0]
    nb, nd = model.num_bodies, model.num_dof
    a = dict(root=be.arr(root), dof=be.arr(dof), rbs=be.zeros((n, nb, 13)), cf=be.zeros((n, nb, 3)), df=be.zeros((n, nd)), pd=be.arr(target))
    sim = abi.sim_state_struct(n, a["root"], a["dof"], a["rbs"], a["cf"], a["df"], a["pd"])
    assert be.sim_step(mstruct, params, sim, None, None, None, None, num_sim_calls) == 0
    be.sync()
    return {k: be.np(v) for k, v in a.items()}


def check_step_against(model, out, r, d, rbs, tau, fc, tag="", scale=1.0):
    """One env's simulator tensors after a step against a reference (root [13], dof [D, 2], rbs [NB, 13], dof force [D], contact force [NB, 3]) at the tolerances the
    fp32 stepper is held to against the fp64 dense oracle (`scale` < 1 tightens all of them)."""
    np.testing.assert_allclose(out["root"], r, atol=3e-4 * scale, rtol=1e-4 * scale, err_msg=f"root {tag}")
    np.testing.assert_allclose(out["dof"][:, 1], d[:, 1], atol=3e-3 * scale, rtol=1e-3 * scale, err_msg=f"dof vel {tag}")
    if model.all_spherical:   # exp-map coordinates may differ by the 2*pi branch: compare as rotations
        for j in range(model.num_bodies - 1):
            qa = do.quat_from_rotvec(np.asarray(out["dof"][3 * j:3 * j + 3, 0], dtype=np.float64))
            qb = do.quat_from_rotvec(np.asarray(d[3 * j:3 * j + 3, 0], dtype=np.float64))
            assert abs(abs(qa @ qb) - 1) < 1e-6 * scale + 1e-12, tag
    else:
        np.testing.assert_allclose(out["dof"][:, 0], d[:, 0], atol=3e-4 * scale, err_msg=f"dof pos {tag}")
    np.testing.assert_allclose(out["rbs"][:, 0:3], rbs[:, 0:3], atol=3e-4 * scale, err_msg=f"body pos {tag}")
    np.testing.assert_allclose(np.abs((out["rbs"][:, 3:7] * rbs[:, 3:7]).sum(-1)), 1, atol=1e-5 * scale + 1e-12)
    np.testing.assert_allclose(out["rbs"][:, 7:13], rbs[:, 7:13], atol=3e-3 * scale, rtol=1e-3 * scale, err_msg=f"body vel {tag}")
    np.testing.assert_allclose(out["df"], tau, atol=0.15 * scale, rtol=2e-3 * scale, err_msg=f"dof force {tag}")
    np.testing.assert_allclose(out["cf"], fc, atol=0.5 * scale, rtol=5e-3 * scale, err_msg=f"contact force {tag}")


def f32_params(prm):
    """The dense oracle's parameter dict holding exactly the numbers the C struct holds (fp32-rounded gravity, damping, dt ...): what the double-precision build of the
    lane code computes with, so that the two can be compared to rounding of fp64."""
    keys = ("gravity_z", "contact_stiffness", "contact_damping", "friction", "friction_viscous", "angular_damping", "max_angular_velocity", "limit_stiffness",
            "limit_damping", "self_stiffness_scale", "self_damping_ratio")
    d = {k: float(getattr(prm, k)) for k in keys}
    d.update(control_mode=int(prm.control_mode), self_collision=int(prm.self_collision))
    return d, float(prm.sim_dt), int(prm.substeps)


@pytest.mark.parametrize("robot,control_mode", [("smpl_humanoid", 0), ("h1_humanoid", 0), ("h1_humanoid", 1), ("g1_humanoid", 2)])
@pytest.mark.parametrize("height,anisotropic,reroot", [(0.95, False, True), (0.80, False, True), (0.80, True, True), (0.80, False, False), (3.0, False, True)])
def test_double_precision_build_of_the_recursion_is_the_dense_scheme(robot, control_mode, height, anisotropic, reroot):
    """oracle/hostemu/hostemu64.cpp -- the kernel's own per-lane recursion (re-rooted solver tree, linearly-implicit drives, penalty contact) compiled with every float a
    double -- against the dense fp64 oracle, BOTH computing with the fp32-rounded parameters of the C struct: equal to 1e-9 (observed ~1e-14).  The recursion and the dense
    M^-1 solve are the same scheme exactly; what separates the fp32 kernel from the oracle elsewhere in this file is rounding alone.  This build is also the
    exact-arithmetic reference of the LAGGED scheme (tests/test_stepper_options.py), which has no dense form."""
    import hostemu_util as hu
    if robot != "smpl_humanoid" and (anisotropic or not reroot):
        pytest.skip("SMPL-only variants")
    model, _, _ = model_on(get_backend("hostemu"), name=robot, anisotropic=anisotropic, reroot=reroot)
    rng = np.random.default_rng(31)
    n = 3
    if model.all_spherical:
        root, dof, target = random_states(model, n, rng, height=height)
    else:
        root, dof, target = random_states(model, n, rng, height=height + (0.1 if robot == "h1_humanoid" else -0.15), vel=0.5, pose=0.15)
        lo, hi = model.dof_limits()
        dof[:, :, 0] = np.clip(dof[:, :, 0], lo + 0.05, hi - 0.05)
        target = np.clip(target, lo, hi).astype(F)
    prm = abi.sim_params_struct(control_mode=control_mode) if not model.all_spherical else abi.sim_params_struct()
    if not model.all_spherical:
        prm.sim_dt = 1.0 / 200.0
    out = hu.sim_step_f64(model, prm, root, dof, target, 2)
    dp, sim_dt, substeps = f32_params(prm)
    worst = 0.0
    for e in range(n):
        r, d, rbs, tau, fc = do.sim_step(model, root[e], dof[e], target[e], params=dp, sim_dt=sim_dt, substeps=substeps, num_sim_calls=2)
        check_step_against(model, {k: v[e] for k, v in out.items()}, r, d, rbs, tau, fc, f"env {e}", scale=1e-5)
        worst = max(worst, float(np.abs(out["rbs"][e] - rbs).max()))
    print(f"{robot} mode {control_mode} height {height}: fp64 recursion vs dense oracle, worst body-state difference {worst:.2e}")
    assert worst < 1e-9


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("num_sim_calls,height,anisotropic", [(1, 0.95, False), (2, 0.95, False), (2, 0.80, False), (1, 3.0, False), (2, 0.95, True), (2, 0.80, True)])
def test_aba_matches_dense_oracle(backend, num_sim_calls, height, anisotropic):
    """Featherstone recursion (fp32, one lane per body) == dense M^-1 solve (fp64), incl. implicit PD + contact.  `anisotropic`: per-axis joint
    gains / armature, the general joint-space inertia D = A + R diag(d) R^T (the SMPL asset's equal gains take the scalar-d expressions)."""
    be = get_backend(backend)
    model, mstruct, keep = model_on(be, anisotropic=anisotropic)
    rng = np.random.default_rng(11)
    n = 6
    root, dof, target = random_states(model, n, rng, height=height)
    params = abi.sim_params_struct()
    out = run_step(be, model, mstruct, root, dof, target, params, num_sim_calls)
    n_contacts = 0
    for e in range(n):
        r, d, rbs, tau, fc = do.sim_step(model, root[e], dof[e], target[e], sim_dt=1 / 60, substeps=2, num_sim_calls=num_sim_calls)
        n_contacts += int((np.abs(fc).sum(-1) > 0).sum())
        check_step_against(model, {k: v[e] for k, v in out.items()}, r, d, rbs, tau, fc, f"env {e}")
    if height < 0.9:
        assert n_contacts > 0, "the low-height case must exercise ground contact"
    if height > 2:
        assert n_contacts == 0


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("self_collision", [0, 1])
def test_rerooted_solve_equals_the_pelvis_rooted_solve(backend, self_collision):
    """The solver tree re-rooted at Spine (7 level-steps per sweep) and the kinematic tree (9) solve the same equations: after two simulate
    calls from random states -- ground contact, PD drives, body-body contact -- every simulator tensor agrees to fp32 rounding."""
    be = get_backend(backend)
    rng = np.random.default_rng(23)
    outs = []
    for reroot in (True, False):
        model, mstruct, keep = model_on(be, reroot=reroot)
        if not outs:
            root, dof, target = random_states(model, 8, rng, height=0.85)
        params = abi.sim_params_struct(self_collision=self_collision, lane_mapping=1)
        outs.append(run_step(be, model, mstruct, root, dof, target, params, 2))
    assert int(model_on(be, reroot=True)[0].solver_tree()["base"]) != 0 and int(model.solver_tree()["base"]) == 0
    a, b = outs
    np.testing.assert_allclose(a["root"], b["root"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(a["dof"][..., 1], b["dof"][..., 1], atol=4e-3, rtol=1e-3)
    np.testing.assert_allclose(a["rbs"][..., 0:3], b["rbs"][..., 0:3], atol=2e-4)
    np.testing.assert_allclose(a["rbs"][..., 7:13], b["rbs"][..., 7:13], atol=4e-3, rtol=1e-3)
    np.testing.assert_allclose(a["df"], b["df"], atol=0.2, rtol=3e-3)
    np.testing.assert_allclose(a["cf"], b["cf"], atol=0.6, rtol=6e-3)
    assert np.abs(a["cf"]).sum() > 0


def _free_params(**kw):
    d = dict(gravity_z=0.0, contact_stiffness=0.0, contact_damping=0.0, friction=0.0, friction_viscous=0.0, angular_damping=0.0)
    d.update(kw)
    return d


def _zero_gain_model(be, zero_armature=False):
    return model_on(be, kp_scale=0.0, kd_scale=0.0, zero_armature=zero_armature)


@pytest.mark.parametrize("backend", BACKENDS)
def test_momentum_and_energy_conservation(backend):
    """No gravity, no contact, no PD, no damping: linear momentum exactly and energy approximately conserved
    (semi-implicit Euler: O(dt) energy drift, bounded)."""
    be = get_backend(backend)
    # armature is a real (reflected) inertia: zero it so that `energy()` is the whole energy
    model, mstruct, keep = _zero_gain_model(be, zero_armature=True)
    rng = np.random.default_rng(5)
    root, dof, target = random_states(model, 3, rng, height=5.0, vel=0.6, pose=0.4)
    params = abi.sim_params_struct(**_free_params())

    def momentum_energy(r, d):
        st = do.State(r, d)
        Q, R, p = do.kinematics(model, st)
        w, v = do.body_velocities(model, st, R, p)
        P_ = sum(model.mass[i] * (v[i] + np.cross(w[i], R[i] @ model.com[i])) for i in range(model.num_bodies))
        return P_, do.energy(model, st, gravity_z=0.0)

    P0 = [momentum_energy(root[e], dof[e]) for e in range(3)]
    drift = {}
    for substeps in (2, 8):
        params = abi.sim_params_struct(substeps=substeps, **_free_params())
        a = dict(root=root, dof=dof)
        for _ in range(15):  # 15 env steps = 0.5 s
            a = run_step(be, model, mstruct, a["root"], a["dof"], target, params, 2)
        dP, dE = 0.0, 0.0
        for e in range(3):
            P1, E1 = momentum_energy(a["root"][e], a["dof"][e])
            dP = max(dP, np.abs(P1 - P0[e][0]).max() / np.abs(P0[e][0]).max())
            dE = max(dE, abs(E1 - P0[e][1]) / abs(P0[e][1]))
        drift[substeps] = (dP, dE)
    # first-order integrator on origin velocities: drift is O(dt) -- small, and shrinking with dt
    assert drift[2][0] < 0.02 and drift[2][1] < 0.08, drift
    assert drift[8][0] < 0.5 * drift[2][0] and drift[8][1] < 0.5 * drift[2][1], drift


@pytest.mark.parametrize("backend", BACKENDS)
def test_free_fall_com_acceleration(backend):
    """In the air the centre of mass accelerates at g.  (Passive joints: with a stiff PD drive the first-order
    integrator exchanges O(dt^2 * qdd * qd) of momentum between the root and the limbs per sub-step -- a documented
    limitation of integrating the root-origin velocity, see DESIGN.md "known limitations".)"""
    be = get_backend(backend)
    model, mstruct, keep = _zero_gain_model(be)
    rng = np.random.default_rng(8)
    root, dof, target = random_states(model, 4, rng, height=6.0, vel=0.5)
    params = abi.sim_params_struct(angular_damping=0.0)

    def com(r, d, vel=False):
        st = do.State(r, d)
        Q, R, p = do.kinematics(model, st)
        w, v = do.body_velocities(model, st, R, p)
        if vel:
            return sum(model.mass[i] * (v[i] + np.cross(w[i], R[i] @ model.com[i])) for i in range(model.num_bodies)) / model.total_mass
        return sum(model.mass[i] * (p[i] + R[i] @ model.com[i]) for i in range(model.num_bodies)) / model.total_mass

    v0 = [com(root[e], dof[e], True) for e in range(4)]
    a = run_step(be, model, mstruct, root, dof, target, params, 2)
    for e in range(4):
        dv = com(a["root"][e], a["dof"][e], True) - v0[e]
        np.testing.assert_allclose(dv, [0, 0, -9.81 * 4 / 120], atol=1e-2)


@pytest.mark.parametrize("backend", BACKENDS)
def test_standing_pose_is_stable_under_pd(backend):
    """Zero pose standing on the ground with PD targets = current pose: stays upright for 2 s, no blow-up
    (kp=800 / kd=80 at dt=1/120 would be unstable with an explicit PD on the light distal links)."""
    be = get_backend(backend)
    model, mstruct, keep = model_on(be)
    n = 2
    nd = model.num_dof
    root = np.zeros((n, 13), F)
    root[:, 2] = 0.93
    root[:, 6] = 1
    dof = np.zeros((n, nd, 2), F)
    dof[1, :, 1] = np.random.default_rng(0).normal(0, 0.5, nd)  # env 1 starts with joint-velocity noise
    target = np.zeros((n, nd), F)
    params = abi.sim_params_struct()
    a = dict(root=root, dof=dof)
    zs = []
    for _ in range(60):
        a = run_step(be, model, mstruct, a["root"], a["dof"], target, params, 2)
        zs.append(a["root"][:, 2].copy())
        assert np.isfinite(a["root"]).all() and np.isfinite(a["dof"]).all()
    zs = np.array(zs)
    assert (zs[-1] > 0.75).all(), f"fell: root heights {zs[-1]}"
    assert np.abs(a["dof"][:, :, 1]).max() < 5.0
    assert np.abs(a["dof"][:, :, 0]).max() < 0.6
    # feet carry the weight: total contact force ~ m g (quasi-static by now)
    fz = a["cf"][:, :, 2].sum(-1)
    np.testing.assert_allclose(fz, model.total_mass * 9.81, rtol=0.25)


# ---------------------------------------------------------------------------------------------------------------
# H1 (config 5): revolute joints with rest rotations, `pd` explicit-torque mode, joint limits
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,control_mode,limits,height", [
    ("h1_humanoid", 0, False, 1.05), ("h1_humanoid", 1, False, 1.05), ("h1_humanoid", 2, False, 1.05), ("h1_humanoid", 1, True, 1.05),
    ("h1_humanoid", 2, True, 0.9), ("h1_humanoid", 0, True, 0.9), ("h1_humanoid", 1, False, 3.0),
    ("g1_humanoid", 0, False, 0.8), ("g1_humanoid", 2, True, 0.7), ("g1_humanoid", 1, False, 3.0)])
def test_robot_aba_matches_dense_oracle(backend, name, control_mode, limits, height):
    """H1: 19 revolute joints (rest rotations on the shoulder links), implicit position drive (`isaac_pd`) and the
    explicit `pd` torque mode (recomputed once per simulate call), joint-limit penalty; 1/200 s x 2 sub-steps x 4 calls.
    G1: 37 revolute joints on 38 bodies (tree depth 10, 14-gram finger links) -- the one-env-per-wavefront instantiation."""
    be = get_backend(backend)
    model, mstruct, keep = model_on(be, name)
    assert model.all_revolute and (model.num_bodies, model.num_dof) == {"h1_humanoid": (20, 19), "g1_humanoid": (38, 37)}[name]
    rng = np.random.default_rng(21)
    n = 5 if name == "h1_humanoid" else 3
    root, dof, target = random_states(model, n, rng, height=height, vel=0.7, pose=0.6 if limits else 0.3)
    lim = dict(limit_stiffness=2000.0, limit_damping=20.0) if limits else {}
    params = abi.sim_params_struct(sim_dt=1 / 200, substeps=2, control_freq_inv=4, control_mode=control_mode, **lim)
    out = run_step(be, model, mstruct, root, dof, target, params, 4)
    n_contacts = n_limits = 0
    lo, hi = model.dof_limits()
    for e in range(n):
        n_limits += int(((dof[e, :, 0] < lo) | (dof[e, :, 0] > hi)).sum())
        r, d, rbs, tau, fc = do.sim_step(model, root[e], dof[e], target[e], params=dict(control_mode=control_mode, **lim), sim_dt=1 / 200,
                                         substeps=2, num_sim_calls=4)
        n_contacts += int((np.abs(fc).sum(-1) > 0).sum())
        np.testing.assert_allclose(out["root"][e], r, atol=3e-4, rtol=1e-4, err_msg=f"root env {e}")
        np.testing.assert_allclose(out["dof"][e, :, 0], d[:, 0], atol=2e-4, err_msg=f"dof pos env {e}")
        np.testing.assert_allclose(out["dof"][e, :, 1], d[:, 1], atol=5e-3, rtol=1e-3, err_msg=f"dof vel env {e}")
        np.testing.assert_allclose(out["rbs"][e][:, 0:3], rbs[:, 0:3], atol=3e-4, err_msg="body pos")
        np.testing.assert_allclose(np.abs((out["rbs"][e][:, 3:7] * rbs[:, 3:7]).sum(-1)), 1, atol=1e-5)
        np.testing.assert_allclose(out["rbs"][e][:, 7:13], rbs[:, 7:13], atol=5e-3, rtol=1e-3, err_msg="body vel")
        np.testing.assert_allclose(out["df"][e], tau, atol=0.05, rtol=2e-3, err_msg="dof force")
        np.testing.assert_allclose(out["cf"][e], fc, atol=0.5, rtol=5e-3, err_msg="contact force")
    if limits:
        assert n_limits > 3, "the wide-pose case must start outside some joint limits"
    if height < 1.0:
        assert n_contacts > 0
    if height > 2:
        assert n_contacts == 0


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("rb,control_mode", [("h1", 0), ("h1", 2), ("g1", 0), ("g1", 2)])
def test_robot_settles_on_its_feet(backend, rb, control_mode):
    """0.4 s of holding the reference's default pose (humanoid.py:1121,1181) after a 5 cm drop: the feet carry the weight, no
    chatter, joints quiet (an un-balanced humanoid tips over later -- that is physics, not tested).  Modes: implicit
    position drive and `pd` with the continuous damper (the held-damper variant, mode 1, chatters on the unloaded foot)."""
    robot_settles_on_its_feet(backend, rb, control_mode)


def robot_settles_on_its_feet(backend, rb, control_mode, **extra):
    be = get_backend(backend)
    from phc_amd.robots import ROBOTS
    model, mstruct, keep = model_on(be, f"{rb}_humanoid")
    n = 4
    nb, nd = model.num_bodies, model.num_dof
    root = np.zeros((n, 13), F)
    root[:, 6] = 1.0
    dof = np.zeros((n, nd, 2), F)
    dof[:, :, 0] = np.asarray(ROBOTS[rb]["default_dof_pos"], F)
    # pelvis height that puts the lowest contact point of the default pose 5 cm above the ground
    st = do.State(root[0], dof[0], model)
    Q, R, p = do.kinematics(model, st)
    low = min(float((p[b] + R[b] @ c)[2] - r) for b, c, r in zip(model.contact_body, model.contact_pos, model.contact_radius))
    h0 = 1.0 if rb == "h1" else 0.05 - low   # H1: the height the test has always used (a ~5 cm drop as well)
    root[:, 2] = h0
    target = dof[:, :, 0].copy()
    params = abi.sim_params_struct(sim_dt=1 / 200, substeps=2, control_freq_inv=4, control_mode=control_mode, limit_stiffness=2000.0, limit_damping=20.0, **extra)
    a = dict(root=be.arr(root), dof=be.arr(dof), rbs=be.zeros((n, nb, 13)), cf=be.zeros((n, nb, 3)), df=be.zeros((n, nd)), pd=be.arr(target))
    sim = abi.sim_state_struct(n, a["root"], a["dof"], a["rbs"], a["cf"], a["df"], a["pd"])
    fz = []
    for k in range(20):   # 20 x 4 simulate calls = 0.4 s
        assert be.sim_step(mstruct, params, sim, None, None, None, None, 4) == 0
        be.sync()
        fz.append(be.np(a["cf"])[:, :, 2].sum(-1))
    r = be.np(a["root"])
    assert np.isfinite(r).all()
    assert (r[:, 2] > h0 - 0.15).all() and (r[:, 2] < h0).all(), (h0, r[:, 2])
    assert (np.abs(r[:, 6]) > 0.98).all(), "pelvis still upright"
    assert np.abs(be.np(a["dof"])[:, :, 1]).max() < 3.0, "joint rates quiet"
    np.testing.assert_allclose(np.mean(fz[8:], axis=0), model.total_mass * 9.81, rtol=0.25)   # feet carry the weight
    cf = be.np(a["cf"])
    feet = {"h1": [5, 10], "g1": [6, 12]}[rb]   # ankle (roll) links
    assert (np.abs(cf[:, feet, 2]).sum(-1) > 0.9 * np.abs(cf[:, :, 2]).sum(-1)).all(), "only the ankle links touch the ground"


# ---------------------------------------------------------------------------------------------------------------
# body-body contact (SURVEY f-1)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", ["smpl_humanoid", "h1_humanoid", "g1_humanoid"])
def test_self_collision_matches_dense_oracle(backend, name):
    """Capsule-capsule penalty contact between non-adjacent bodies: kernel == fp64 dense oracle with the same explicit forces,
    on folded poses that make limbs overlap."""
    be = get_backend(backend)
    model, mstruct, keep = model_on(be, name)
    rng = np.random.default_rng(8)
    n = 6
    root, dof, target = random_states(model, n, rng, height=1.5, vel=0.5, pose=0.9)
    h1 = name != "smpl_humanoid"   # robots: 200 Hz, pd control
    kw = dict(sim_dt=1 / 200, control_freq_inv=4, control_mode=2) if h1 else {}
    params = abi.sim_params_struct(self_collision=1, **kw)
    out = run_step(be, model, mstruct, root, dof, target, params, 1)
    hits = 0
    for e in range(n):
        st = do.State(root[e], dof[e], model)
        Q, R, p = do.kinematics(model, st)
        w, v = do.body_velocities(model, st, R, p)
        Fs, _ = do.self_collision_wrenches(model, R, p, w, v, dict(do.DEFAULT_PARAMS), (1 / 200 if h1 else 1 / 60) / 2)
        hits += int((np.abs(Fs).sum(-1) > 0).sum())
        r, d, rbs, tau, fc = do.sim_step(model, root[e], dof[e], target[e], params=dict(self_collision=1, control_mode=2 if h1 else 0),
                                         sim_dt=1 / 200 if h1 else 1 / 60, substeps=2, num_sim_calls=1)
        np.testing.assert_allclose(out["root"][e], r, atol=5e-4, rtol=2e-4, err_msg=f"root env {e}")
        np.testing.assert_allclose(out["dof"][e, :, 1], d[:, 1], atol=2e-2, rtol=2e-3, err_msg=f"dof vel env {e}")
        np.testing.assert_allclose(out["rbs"][e][:, 0:3], rbs[:, 0:3], atol=5e-4, err_msg="body pos")
        np.testing.assert_allclose(out["cf"][e], fc, atol=1.0, rtol=1e-2, err_msg="contact force (incl. body-body)")
    assert hits >= 6, f"the folded poses must produce overlapping limbs ({hits})"


def _max_penetration(model, root, dof):
    st = do.State(root, dof, model)
    Q, R, p = do.kinematics(model, st)
    cap, masks, worst = model.collision_capsule, model.collision_allow_masks(), 0.0
    for i in range(model.num_bodies):
        for k in range(i + 1, model.num_bodies):
            if (int(masks[i]) >> k) & 1:
                c1, c2 = do.seg_seg_closest(p[i] + R[i] @ cap[i, 0:3], p[i] + R[i] @ cap[i, 3:6], p[k] + R[k] @ cap[k, 0:3], p[k] + R[k] @ cap[k, 3:6])
                worst = max(worst, cap[i, 6] + cap[k, 6] - np.linalg.norm(c1 - c2))
    return worst


@pytest.mark.parametrize("backend", BACKENDS)
def test_self_collision_forces_are_internal_and_separate_the_limbs(backend):
    """Each lane evaluates its half of a colliding pair with the same arithmetic: the published net contact forces of an env
    (no ground in reach) sum to zero; with passive joints and no gravity the overlapping limbs are pushed apart."""
    be = get_backend(backend)
    model, mstruct, keep = model_on(be, "smpl_humanoid", kp_scale=0.0, kd_scale=0.0)
    rng = np.random.default_rng(3)
    n = 8
    root, dof, target = random_states(model, n, rng, height=5.0, vel=0.0, pose=1.0)
    dof[:, :, 1] = 0
    root[:, 7:13] = 0
    params = abi.sim_params_struct(self_collision=1, gravity_z=0.0)
    pen0 = np.array([_max_penetration(model, root[e], dof[e]) for e in range(n)])
    assert (pen0 > 0.02).sum() >= 4, pen0
    out = run_step(be, model, mstruct, root, dof, target, params, 2)
    cf = out["cf"]
    assert np.abs(cf).sum(axis=(1, 2)).max() > 50.0
    np.testing.assert_allclose(cf.sum(axis=1), 0.0, atol=2e-3 * np.abs(cf).sum(axis=1).max())
    r, d = out["root"], out["dof"]
    for _ in range(14):   # 0.5 s in total
        o = run_step(be, model, mstruct, r, d, target, params, 2)
        r, d = o["root"], o["dof"]
    pen1 = np.array([_max_penetration(model, r[e], d[e]) for e in range(n)])
    assert np.isfinite(r).all() and (pen1 < 0.6 * pen0 + 0.005).all(), (pen0, pen1)


@pytest.mark.parametrize("backend", BACKENDS)
def test_sliding_friction_decelerates_at_mu_g(backend):
    """Regularised Coulomb friction: a humanoid lying on its back in the rest pose (stiff PD holds the pose) and sliding at 2 m/s
    decelerates at mu * g (mu = 1) until it sticks; with mu = 0.5 at half that."""
    be = get_backend(backend)
    model, mstruct, keep = model_on(be)
    for mu in (1.0, 0.5):
        n = 2
        root = np.zeros((n, 13), F)
        root[:, 2] = 0.12
        root[:, 3:7] = np.array([0.0, -np.sqrt(0.5), 0.0, np.sqrt(0.5)], F)   # lying on the back: body z axis along world -x... (pitch -90 deg)
        dof = np.zeros((n, 69, 2), F)
        target = np.zeros((n, 69), F)
        params = abi.sim_params_struct(friction=mu)
        out = dict(root=root, dof=dof)
        for _ in range(10):   # settle on the ground (1/3 s)
            out = run_step(be, model, mstruct, out["root"], out["dof"], target, params, 2)
        r = out["root"].copy()
        assert (r[:, 2] < 0.2).all() and np.abs(r[:, 7:10]).max() < 0.5, r
        r[:, 7] = 2.0                     # push: 2 m/s along world x
        r[:, 8:13] = 0
        d = out["dof"].copy()
        d[:, :, 1] = 0
        o = run_step(be, model, mstruct, r, d, target, params, 2)          # one env step = 1/30 s
        o = run_step(be, model, mstruct, o["root"], o["dof"], target, params, 2)
        o = run_step(be, model, mstruct, o["root"], o["dof"], target, params, 2)   # 0.1 s in total
        v = o["root"][:, 7]
        expect = 2.0 - mu * 9.81 * 0.1
        assert np.all(np.abs(v - expect) < 0.25 * mu * 9.81 * 0.1 + 0.05), (mu, v, expect)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("self_collision,inertia_lag", [(0, 0), (1, 0), (1, 1)])
def test_per_env_body_shapes_equal_single_shape_runs(backend, self_collision, inertia_lag):
    """Per-env body shapes (robot.has_shape_variation, humanoid.py:726-766,824-866): ONE launch over K = 3 stacked models (the reference's
    three gender assets: different link offsets, masses, contact-point counts) with an int32 shape id per env gives every env, bit for
    bit, what a single-shape launch of its own model gives -- stepper and the FK-only refresh."""
    from phc_amd.model import load_model, pack_shapes
    from phc_amd.robots import apply_collision_filter
    be = get_backend(backend)
    models = [load_model(f"smpl_{g}_humanoid") for g in range(3)]
    for m in models:
        apply_collision_filter(m, "smpl")
    assert len({len(m.contact_body) for m in models}) > 1 and len({round(m.total_mass, 2) for m in models}) == 3
    ints, floats = pack_shapes(models)
    keep = (be.arr(ints), be.arr(floats))
    m0 = models[0]
    stacked = abi.model_struct(keep[0], keep[1], m0.num_bodies, m0.num_dof, m0.max_level, max(len(m.contact_body) for m in models),
                               num_shapes=3)
    rng = np.random.default_rng(5)
    n = 10
    root, dof, target = random_states(m0, n, rng, height=0.85)
    shape = (np.arange(n) % 3).astype(np.int32)
    params = abi.sim_params_struct(self_collision=self_collision, inertia_lag=inertia_lag)   # (round 6: the lagged instantiation exists for per-env shapes too)

    def run(mstruct, rows, env_shape):
        k = len(rows)
        a = dict(root=be.arr(root[rows]), dof=be.arr(dof[rows]), rbs=be.zeros((k, m0.num_bodies, 13)), cf=be.zeros((k, m0.num_bodies, 3)),
                 df=be.zeros((k, m0.num_dof)), pd=be.arr(target[rows]))
        es = None if env_shape is None else be.arr(env_shape)
        sim = abi.sim_state_struct(k, a["root"], a["dof"], a["rbs"], a["cf"], a["df"], a["pd"], env_shape=es)
        assert be.refresh_body_state(mstruct, sim) == 0
        be.sync()
        fk = be.np(a["rbs"]).copy()
        assert be.sim_step(mstruct, params, sim, None, None, None, None, 2) == 0
        be.sync()
        return fk, {k_: be.np(v) for k_, v in a.items()}

    fk_all, out_all = run(stacked, np.arange(n), shape)
    for g, m in enumerate(models):
        i, f = m.pack()
        kg = (be.arr(i), be.arr(f))
        single = abi.model_struct(kg[0], kg[1], m.num_bodies, m.num_dof, m.max_level, len(m.contact_body))
        rows = np.flatnonzero(shape == g)
        fk_g, out_g = run(single, rows, None)
        assert np.array_equal(fk_all[rows], fk_g)
        for key in ("root", "dof", "rbs", "cf", "df"):
            assert np.array_equal(out_all[key][rows], out_g[key]), (g, key)
    # and the shapes really differ: env 0 (shape 0) and env 1 (shape 1) start from different states anyway, so compare FK of one state
    root[:] = root[0]; dof[:] = dof[0]
    fk_same, _ = run(stacked, np.arange(3), shape[:3])
    assert np.abs(fk_same[0] - fk_same[1]).max() > 1e-3 and np.abs(fk_same[0] - fk_same[2]).max() > 1e-3


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("scene", ["standing_pd", "falling_contact"])
def test_stepper_converges_to_the_continuous_model(backend, scene):
    """pin the MODELLING of the stepper, not only its arithmetic.  `dyn_oracle.ode_solve` integrates the continuous PD +
    penalty-contact ODE explicitly in fp64 at dt = 1/7680 .. 1/30720 s with Richardson extrapolation (no implicit term of the scheme exists in it); the stepper, run over the same 0.025 s
    with sub-steps of 1/120, 1/240, 1/480 and 1/960 s, must approach that solution at first order: the error roughly halves with dt, and at
    the shipped dt = 1/120 s it is small in absolute terms.  Scenes: a humanoid standing on the ground whose PD targets pull it into a bent
    pose (stiff drives + sustained contact + friction), and one dropped from 2 cm with joint velocities (contact onset)."""
    be = get_backend(backend)
    model, mstruct, keep = model_on(be)
    nd = model.num_dof
    rng = np.random.default_rng(3)
    root = np.zeros((1, 13), F)
    root[0, 6] = 1.0
    dof = np.zeros((1, nd, 2), F)
    names = model.body_names
    tgt = np.zeros((1, nd), F)
    st0 = do.State(root[0].astype(np.float64), dof[0].astype(np.float64), model)
    Q, R, p = do.kinematics(model, st0)
    low = min(p[i][2] + (R[i] @ model.contact_pos[k])[2] - model.contact_radius[k] for k, i in enumerate(model.contact_body))
    if scene == "standing_pd":
        root[0, 2] = -low - 0.004                        # resting a few mm into the ground (near its static equilibrium)
        for jn, ax, ang in (("L_Knee", 1, 0.5), ("R_Knee", 1, 0.5), ("L_Hip", 1, -0.3), ("R_Hip", 1, -0.3), ("L_Shoulder", 0, 0.6), ("Torso", 1, 0.2)):
            tgt[0, 3 * (names.index(jn) - 1) + ax] = ang
    else:
        root[0, 2] = -low + 0.02
        dof[0, :, 1] = rng.normal(0, 1.0, nd)
        root[0, 7:10] = (0.3, -0.2, 0.0)
        tgt[0] = rng.normal(0, 0.2, nd)
    T = 0.025
    # explicit integration bounds the stiffness it can take: 8 foot corners x friction_viscous on a ~1 kg foot is a 16 000 1/s mode at the
    # shipped 2000 N s/m (stable only below h = 1/8000 s); both sides of this test run at 500 N s/m, everything else as shipped
    soft = dict(friction_viscous=500.0)
    key = (scene,)
    if key not in _ODE_CACHE:      # (shared by the two backends of one session: ~7 s of fp64 numpy)
        _ODE_CACHE[key] = do.ode_body_positions(model, root[0], dof[0], tgt[0], T, 1 / 7680, params=soft, levels=3)
    pos_ref, ref_err = _ODE_CACHE[key]
    errs = []
    for sub in (1, 2, 4, 8):                             # dt = 1/120 .. 1/960 s; 3 * sub sub-steps cover T
        params = abi.sim_params_struct(sim_dt=sub / (120.0 * sub), substeps=sub, **soft)
        out = run_step(be, model, mstruct, root, dof, tgt, params, num_sim_calls=3)
        errs.append(np.abs(out["rbs"][0][:, 0:3] - pos_ref).max())
    errs = np.array(errs)
    assert ref_err < 0.2 * errs[-1], (ref_err, errs)     # the reference itself is converged well below the finest stepper run
    assert errs[0] < 1e-2, errs                          # the shipped sub-step: millimetres
    ratios = errs[:-1] / errs[1:]
    assert (ratios > 1.5).all() and (ratios < 2.7).all(), (errs, ratios)   # first order: halving dt halves the error


_ODE_CACHE = {}


# ------------------------------------------------------------------------------------------------------------------ rigid ("tgs") ground contact
RIGID = dict(contact_model=1, contact_iterations=4)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("num_sim_calls,height,self_collision", [(1, 0.90, 0), (2, 0.925, 0), (2, 0.93, 1)])
def test_rigid_contact_matches_dense_oracle(backend, num_sim_calls, height, self_collision):
    """contact_model 1 (include/phc_amd.h; solver.contact=tgs): the O(n) recursion with per-point contact impedances and 4 active-set / friction-cone
    passes per sub-step == the fp64 dense solve of the same equations (oracle/dyn_oracle.py accelerations(nud_prev=...)): state, joint torques and
    the published net ground force.  The impedance (dt c = 833 kg per point) is 55 x the penalty model's: the fp32 recursion has to carry it."""
    be = get_backend(backend)
    model, mstruct, keep = model_on(be)
    rng = np.random.default_rng(17)
    n = 6
    root, dof, target = random_states(model, n, rng, height=height, vel=0.5, pose=0.15)
    params = abi.sim_params_struct(self_collision=self_collision, **RIGID)
    out = run_step(be, model, mstruct, root, dof, target, params, num_sim_calls)
    n_contacts = 0
    for e in range(n):
        r, d, rbs, tau, fc = do.sim_step(model, root[e], dof[e], target[e], params=dict(self_collision=self_collision, **RIGID), sim_dt=1 / 60, substeps=2,
                                         num_sim_calls=num_sim_calls)
        n_contacts += int((np.abs(fc).sum(-1) > 0).sum())
        np.testing.assert_allclose(out["root"][e], r, atol=5e-4, rtol=2e-4, err_msg=f"root env {e}")
        np.testing.assert_allclose(out["rbs"][e][:, 0:3], rbs[:, 0:3], atol=5e-4, err_msg="body pos")
        np.testing.assert_allclose(out["rbs"][e][:, 7:13], rbs[:, 7:13], atol=1e-2, rtol=2e-3, err_msg="body vel")
        np.testing.assert_allclose(out["dof"][e, :, 1], d[:, 1], atol=1e-2, rtol=2e-3, err_msg=f"dof vel env {e}")
        np.testing.assert_allclose(out["cf"][e], fc, atol=3.0, rtol=2e-2, err_msg="net ground force per body")
    assert n_contacts > 0


@pytest.mark.parametrize("backend", BACKENDS)
def test_rigid_contact_is_rigid_unilateral_and_carries_the_weight(backend):
    """Physical checks of contact_model 1 on a standing humanoid (PD holds the rest pose): started 3 cm above the ground it lands and comes to rest
    WITHOUT sinking in (lowest contact point within +-0.3 mm of the plane; the penalty model rests ~1 mm inside), the published ground forces sum
    to the body weight, no body is ever pulled down (F_z >= 0: the constraint is unilateral), it does not bounce (restitution 0), and a
    speculative contact (contact_offset) stops the approach AT the plane instead of after a penetration."""
    be = get_backend(backend)
    model, mstruct, keep = model_on(be)
    nd = model.num_dof
    root = np.zeros((1, 13), F)
    root[0, 6] = 1.0
    st0 = do.State(root[0].astype(np.float64), np.zeros((nd, 2)), model)
    Q, R, p = do.kinematics(model, st0)
    low0 = min(p[i][2] + (R[i] @ model.contact_pos[k])[2] - model.contact_radius[k] for k, i in enumerate(model.contact_body))
    root[0, 2] = -low0 + 0.03
    weight = model.mass.sum() * 9.81

    def lowest(rbs):
        q = rbs[:, 3:7].astype(np.float64)
        z = []
        for k, i in enumerate(model.contact_body):
            Rm = do.quat_to_mat(q[i])
            z.append(rbs[i, 2] + (Rm @ model.contact_pos[k])[2] - model.contact_radius[k])
        return min(z)
    for mdl, tol in ((1, 3e-4), (0, 3e-3)):
        params = abi.sim_params_struct(contact_model=mdl, contact_iterations=4)
        out = dict(root=root.copy(), dof=np.zeros((1, nd, 2), F))
        tgt = np.zeros((1, nd), F)
        lows, fzs, vz = [], [], []
        for step in range(45):      # 1.5 s
            out = run_step(be, model, mstruct, out["root"], out["dof"], tgt, params, 2)
            lows.append(lowest(out["rbs"][0])); fzs.append(out["cf"][0][:, 2].copy()); vz.append(out["root"][0, 9])
        lows, fzs, vz = np.array(lows), np.array(fzs), np.array(vz)
        assert (fzs >= -1e-3).all(), "a ground contact never pulls"
        assert lows.min() > -tol * (1 if mdl else 3), (mdl, lows.min())         # never deeper than this
        assert abs(lows[-10:].mean()) < tol and np.abs(vz[-10:]).max() < 0.02, (mdl, lows[-10:], vz[-10:])
        np.testing.assert_allclose(fzs[-10:].sum(-1).mean(), weight, rtol=0.03)
        if mdl == 1:
            touch = int(np.argmax(fzs.sum(-1) > 0))
            assert (vz[touch + 2:touch + 12] < 0.05).all(), "no bounce"


@pytest.mark.parametrize("backend", BACKENDS)
def test_rigid_contact_sliding_friction_decelerates_at_mu_g(backend):
    """The Coulomb cone of contact_model 1 on END-of-step force and slip: a lying humanoid sliding at 2 m/s decelerates at mu g, then sticks."""
    be = get_backend(backend)
    model, mstruct, keep = model_on(be)
    for mu in (1.0, 0.5):
        n = 2
        root = np.zeros((n, 13), F)
        root[:, 2] = 0.12
        root[:, 3:7] = np.array([0.0, -np.sqrt(0.5), 0.0, np.sqrt(0.5)], F)
        dof = np.zeros((n, 69, 2), F)
        target = np.zeros((n, 69), F)
        params = abi.sim_params_struct(friction=mu, **RIGID)
        out = dict(root=root, dof=dof)
        for _ in range(10):
            out = run_step(be, model, mstruct, out["root"], out["dof"], target, params, 2)
        r = out["root"].copy()
        assert (r[:, 2] < 0.2).all() and np.abs(r[:, 7:10]).max() < 0.5, r
        r[:, 7] = 2.0
        r[:, 8:13] = 0
        d = out["dof"].copy()
        d[:, :, 1] = 0
        o = dict(root=r, dof=d)
        for _ in range(3):
            o = run_step(be, model, mstruct, o["root"], o["dof"], target, params, 2)
        v = o["root"][:, 7]
        expect = 2.0 - mu * 9.81 * 0.1
        assert np.all(np.abs(v - expect) < 0.25 * mu * 9.81 * 0.1 + 0.05), (mu, v, expect)
        for _ in range(12):
            o = run_step(be, model, mstruct, o["root"], o["dof"], target, params, 2)
        assert np.abs(o["root"][:, 7]).max() < 0.03, "sticks once the slip is gone"


# ------------------------------------------------------------------------------------------------------------------ several capsules per body (f-1)
def _penetrating_shape_pairs(model, root, dof):
    st = do.State(root, dof, model)
    Q, R, p = do.kinematics(model, st)
    own, cap = model.shape_owner(), model.shape_capsules()
    A = [p[o] + R[o] @ c[0:3] for o, c in zip(own, cap)]
    B = [p[o] + R[o] @ c[3:6] for o, c in zip(own, cap)]
    out = []
    for a, b in model.collision_pairs():
        c1, c2 = do.seg_seg_closest(A[a], B[a], A[b], B[b])
        if cap[a, 6] + cap[b, 6] - np.linalg.norm(c1 - c2) > 1e-3:
            out.append((a, b))
    return out


def test_collision_shapes_of_the_compiled_models():
    """f-1: more than one collision primitive per body.  SMPL: the flat toe boxes (and the hands of the assets whose hand box
    is flat) are two capsules side by side; G1: torso_link carries three shapes and each elbow_roll_link two, with the reference's PER-SHAPE
    filter words (humanoid.py:1205-1226: 40 entries for G1); H1 has one geom per link."""
    from phc_amd.model import load_model
    from phc_amd.robots import _G1_SHAPE_FILTERS, apply_collision_filter
    smpl = apply_collision_filter(load_model("smpl_humanoid"), "smpl")
    assert [smpl.body_names[b] for b in smpl.extra_owner] == ["L_Toe", "R_Toe"] and smpl.num_shapes_collision == 26
    toe = smpl.body_names.index("L_Toe")
    a, b = smpl.collision_capsule[toe], smpl.extra_capsule[0]
    assert abs(a[6] - b[6]) < 1e-12 and a[6] < 0.03 and np.linalg.norm(a[0:3] - b[0:3]) > 0.04      # two thin capsules, one box width apart
    assert len(smpl.collision_pairs()) > 245 and all(smpl.shape_owner()[i] != smpl.shape_owner()[k] for i, k in smpl.collision_pairs())
    g1 = apply_collision_filter(load_model("g1_humanoid"), "g1")
    names = [g1.body_names[b] for b in g1.extra_owner]
    assert names == ["torso_link", "torso_link", "left_elbow_roll_link", "right_elbow_roll_link"] and int(g1.shapes_per_body.sum()) == 40
    # shape order = bodies in order, a body's shapes in geom order: primary shapes take the first word of their body, extras the following ones
    start = np.concatenate([[0], np.cumsum(g1.shapes_per_body)[:-1]])
    for bdy in range(g1.num_bodies):
        if g1.shapes_per_body[bdy]:
            assert g1.shape_filter[bdy] == _G1_SHAPE_FILTERS[start[bdy]]
    for e, (bdy, o) in enumerate(zip(g1.extra_owner, g1.extra_ordinal)):
        assert g1.shape_filter[g1.num_bodies + e] == _G1_SHAPE_FILTERS[start[bdy] + o]
    h1 = load_model("h1_humanoid")
    assert len(h1.extra_owner) == 0


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", ["smpl_humanoid", "g1_humanoid"])
def test_extra_collision_shapes_collide_and_match_the_dense_oracle(backend, name):
    """A pose in which an EXTRA capsule (shape index >= NB: second half of a toe box; G1's head / logo / palm shapes) is what touches: the kernel's
    forces -- accumulated on the shapes' owner bodies -- equal the fp64 dense oracle's, and differ from a model stripped of its extra shapes."""
    be = get_backend(backend)
    model, mstruct, keep = model_on(be, name)
    rng = np.random.default_rng(4)
    robot = name != "smpl_humanoid"
    found = None
    for _ in range(400):
        root, dof, target = random_states(model, 1, rng, height=1.5, vel=0.0, pose=1.0 if not robot else 1.3)
        pairs = _penetrating_shape_pairs(model, root[0], dof[0])
        if any(b >= model.num_bodies or a >= model.num_bodies for a, b in pairs):
            found = (root, dof, target, pairs)
            break
    assert found is not None, "no random pose brought an extra shape into contact"
    root, dof, target, pairs = found
    kw = dict(sim_dt=1 / 200, control_freq_inv=4, control_mode=2) if robot else {}
    out = run_step(be, model, mstruct, root, dof, target, abi.sim_params_struct(self_collision=1, **kw), 1)
    r, d, rbs, tau, fc = do.sim_step(model, root[0], dof[0], target[0], params=dict(self_collision=1, control_mode=2 if robot else 0),
                                     sim_dt=1 / 200 if robot else 1 / 60, substeps=2, num_sim_calls=1)
    np.testing.assert_allclose(out["rbs"][0][:, 0:3], rbs[:, 0:3], atol=5e-4)
    np.testing.assert_allclose(out["cf"][0], fc, atol=1.0, rtol=1e-2)
    # the same pose on a model without the extra shapes gives different body-body forces: the extra shapes really took part
    import copy
    bare = copy.deepcopy(model)
    bare.extra_owner, bare.extra_ordinal, bare.extra_capsule = bare.extra_owner[:0], bare.extra_ordinal[:0], bare.extra_capsule[:0]
    bare.shape_filter = bare.shape_filter[:bare.num_bodies]
    r2, d2, rbs2, tau2, fc2 = do.sim_step(bare, root[0], dof[0], target[0], params=dict(self_collision=1, control_mode=2 if robot else 0),
                                          sim_dt=1 / 200 if robot else 1 / 60, substeps=2, num_sim_calls=1)
    assert np.abs(fc2 - fc).max() > 1.0
