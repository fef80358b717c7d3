"""Round-5 stepper switches (include/phc_amd.h, ABI 35): `inertia_lag` and `force_average`.

`inertia_lag` changes the SCHEME (the articulated inertias of a simulate() call's first sub-step are kept over its other sub-steps), not the model:
it must stay close to the every-sub-step-fresh scheme on one env step, converge to the same continuous model at first order when the step is refined
with the simulate() structure kept, and keep the physical behaviour the fresh scheme is tested for (standing, sliding friction, robots settling).
`force_average` changes only what S4 / S5 publish."""
import numpy as np
import pytest

import dyn_oracle as do
from backends import BACKENDS, get_backend, model_on
from phc_amd import abi
from test_dynamics import random_states, run_step

F = np.float32


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("height", [0.80, 0.95, 3.0])
def test_inertia_lag_stays_close_to_the_fresh_scheme(backend, height):
    """One env step (2 x simulate x 2 sub-steps) from violent random states (joint rates of ~2 rad/s, PD targets 0.3 rad off, 1 m/s root velocities) -- in the
    air, touching down, lying in contact --: the lagged scheme differs from the fresh one by no more than the fresh scheme's own discretisation error
    (its distance to a run with 8 x smaller sub-steps), and is as close to that fine run as the fresh scheme is."""
    be = get_backend(backend)
    model, mstruct, keep = model_on(be)
    rng = np.random.default_rng(5)
    n = 8
    root, dof, target = random_states(model, n, rng, height=height)
    fresh = run_step(be, model, mstruct, root, dof, target, abi.sim_params_struct(), 2)
    lag = run_step(be, model, mstruct, root, dof, target, abi.sim_params_struct(inertia_lag=1), 2)
    fine = run_step(be, model, mstruct, root, dof, target, abi.sim_params_struct(substeps=16), 2)   # the fresh scheme at an 8 x smaller step
    pos = lambda o: o["rbs"][:, :, 0:3]
    vel = lambda o: o["rbs"][:, :, 7:13]
    d_lag, d_fresh = np.abs(pos(lag) - pos(fine)).max(), np.abs(pos(fresh) - pos(fine)).max()
    v_lag, v_fresh = np.abs(vel(lag) - vel(fine)).max(), np.abs(vel(fresh) - vel(fine)).max()
    d_two = np.abs(pos(lag) - pos(fresh)).max()
    print(f"height {height}: distance to the fine run -- fresh {d_fresh:.2e} m / {v_fresh:.2e}, lagged {d_lag:.2e} m / {v_lag:.2e}; lagged vs fresh {d_two:.2e} m")
    assert d_two > 0.0                                   # (the switch does something)
    assert d_two < 1.5 * d_fresh + 1e-4, (d_two, d_fresh)  # the two schemes differ by no more than the fresh scheme's own step error
    assert d_lag < 1.5 * d_fresh + 1e-4 and v_lag < 1.5 * v_fresh + 1e-2, (d_lag, d_fresh, v_lag, v_fresh)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("lag", [0, 1])
@pytest.mark.parametrize("robot,control_mode,height,anisotropic", [("smpl_humanoid", 0, 0.95, False), ("smpl_humanoid", 0, 0.80, False), ("smpl_humanoid", 0, 0.80, True),
                                                                   ("smpl_humanoid", 0, 3.0, False), ("h1_humanoid", 1, 0.85, False), ("g1_humanoid", 2, 0.70, False)])
def test_stepper_equals_the_double_precision_recursion(backend, lag, robot, control_mode, height, anisotropic):
    """Round 6: the LAGGED scheme pinned like the fresh one.  Reference = oracle/hostemu/hostemu64.cpp, the kernel's recursion at double precision -- which IS the dense
    fp64 oracle's scheme to 1e-12 with `inertia_lag` off (tests/test_dynamics.py::test_double_precision_build_of_the_recursion_is_the_dense_scheme) and states the lagged
    scheme in exact arithmetic with it on (no dense form exists: the lagged result depends on the elimination order, oracle/hostemu/hostemu64.cpp).  The fp32 stepper
    (host emulation and the HIP kernel) after one env step (2 x simulate x 2 sub-steps: sub-steps 2 and 4 lagged) from violent random states -- in the air, touching
    down, in contact; SMPL, anisotropic gains, H1 `pd`, G1 -- agrees with it at EXACTLY the tolerances `test_aba_matches_dense_oracle` holds the fresh scheme to."""
    import hostemu_util as hu
    from test_dynamics import check_step_against
    be = get_backend(backend)
    model, mstruct, keep = model_on(be, name=robot, anisotropic=anisotropic)
    rng = np.random.default_rng(17)
    n = 6
    if model.all_spherical:
        root, dof, target = random_states(model, n, rng, height=height)
        prm = abi.sim_params_struct(inertia_lag=lag)
    else:
        root, dof, target = random_states(model, n, rng, height=height, vel=0.5, pose=0.15)
        lo, hi = model.dof_limits()
        dof[:, :, 0] = np.clip(dof[:, :, 0], lo + 0.05, hi - 0.05)
        target = np.clip(target, lo, hi).astype(F)
        prm = abi.sim_params_struct(inertia_lag=lag, control_mode=control_mode, sim_dt=1.0 / 200.0)
    out = run_step(be, model, mstruct, root, dof, target, prm, 2)
    ref = hu.sim_step_f64(model, prm, root, dof, target, 2)
    other = hu.sim_step_f64(model, abi.sim_params_struct(**{**{k: getattr(prm, k) for k in ("control_mode", "sim_dt")}, "inertia_lag": 1 - lag}), root, dof, target, 2)
    for e in range(n):
        check_step_against(model, {k: v[e] for k, v in out.items()}, ref["root"][e], ref["dof"][e], ref["rbs"][e], ref["df"][e], ref["cf"][e], f"{robot} lag {lag} env {e}")
    # the two schemes are different schemes: the fp32 result is closer to its own reference than to the other one's by a wide margin
    d_own, d_other = np.abs(out["rbs"][:, :, 7:13] - ref["rbs"][:, :, 7:13]).max(), np.abs(out["rbs"][:, :, 7:13] - other["rbs"][:, :, 7:13]).max()
    print(f"{robot} lag {lag} height {height}: body velocities vs own fp64 reference {d_own:.2e}, vs the other scheme's {d_other:.2e}")
    assert d_other > 5 * d_own


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("scene", ["standing_pd", "falling_contact"])
def test_inertia_lag_converges_to_the_continuous_model(backend, scene):
    """The gate of the switch: with the simulate() structure kept (2 sub-steps per call, the second one lagged) and the step refined 1/120 -> 1/960 s the
    lagged scheme approaches the explicitly integrated continuous model (dyn_oracle.ode_solve, fp64, no implicit term) at first order, like the fresh one
    (test_dynamics.py::test_stepper_converges_to_the_continuous_model), and its error at the shipped step is of the same size."""
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
        root[0, 2] = -low - 0.004
        for jn, ax, ang in (("L_Knee", 1, 0.5), ("R_Knee", 1, 0.5), ("L_Hip", 1, -0.3), ("R_Hip", 1, -0.3), ("L_Shoulder", 0, 0.6), ("Torso", 1, 0.2)):
            tgt[0, 3 * (names.index(jn) - 1) + ax] = ang
    else:
        root[0, 2] = -low + 0.02
        dof[0, :, 1] = rng.normal(0, 1.0, nd)
        root[0, 7:10] = (0.3, -0.2, 0.0)
        tgt[0] = rng.normal(0, 0.2, nd)
    T = 4 / 120.0                                        # two simulate() calls of two sub-steps at the shipped step
    soft = dict(friction_viscous=500.0)                  # (see test_stepper_converges_to_the_continuous_model: what the explicit reference can take)
    key = (scene,)
    if key not in _ODE_CACHE:
        _ODE_CACHE[key] = do.ode_body_positions(model, root[0], dof[0], tgt[0], T, 1 / 7680, params=soft, levels=3)
    pos_ref, ref_err = _ODE_CACHE[key]
    errs = {0: [], 1: []}
    for lag in (0, 1):
        for k in (1, 2, 4, 8):                           # sub-step 1 / (120 k): 2 k simulate() calls of 2 sub-steps cover T
            params = abi.sim_params_struct(sim_dt=1.0 / (60.0 * k), substeps=2, inertia_lag=lag, **soft)
            out = run_step(be, model, mstruct, root, dof, tgt, params, num_sim_calls=2 * k)
            errs[lag].append(np.abs(out["rbs"][0][:, 0:3] - pos_ref).max())
    e0, e1 = np.array(errs[0]), np.array(errs[1])
    print(f"{scene}: fresh {e0}  ratios {e0[:-1] / e0[1:]}\n{scene}: lag   {e1}  ratios {e1[:-1] / e1[1:]}")
    assert ref_err < 0.2 * e1[-1], (ref_err, e1)
    assert e1[0] < 1e-2 and e1[0] < 1.5 * e0[0] + 1e-4, (e0, e1)   # at the shipped step the lagged scheme is as close to the continuous model as the fresh one
    ratios = e1[:-1] / e1[1:]
    assert (ratios > 1.5).all() and (ratios < 2.7).all(), (e1, ratios)


_ODE_CACHE = {}


@pytest.mark.parametrize("backend", BACKENDS)
def test_inertia_lag_standing_and_touch_down_are_stable(backend):
    """2 s of standing under PD (env 1 with joint-velocity noise) and a drop from 10 cm onto the feet (contacts START in lagged sub-steps too): no blow-up,
    no bounce above the drop height, the feet end up carrying the weight."""
    be = get_backend(backend)
    model, mstruct, keep = model_on(be)
    n, nd = 3, model.num_dof
    root = np.zeros((n, 13), F)
    root[:, 2] = (0.93, 0.93, 1.03)
    root[:, 6] = 1
    dof = np.zeros((n, nd, 2), F)
    dof[1, :, 1] = np.random.default_rng(0).normal(0, 0.5, nd)
    target = np.zeros((n, nd), F)
    params = abi.sim_params_struct(inertia_lag=1)
    a = dict(root=root, dof=dof)
    zmax = 0.0
    for i in range(60):
        a = run_step(be, model, mstruct, a["root"], a["dof"], target, params, 2)
        assert np.isfinite(a["root"]).all() and np.isfinite(a["dof"]).all()
        if i > 15:
            zmax = max(zmax, float(a["root"][2, 2]))
    assert (a["root"][:, 2] > 0.75).all(), f"fell: root heights {a['root'][:, 2]}"
    assert zmax < 0.96, f"the dropped humanoid bounced back to {zmax}"
    assert np.abs(a["dof"][:, :, 1]).max() < 5.0 and np.abs(a["dof"][:, :, 0]).max() < 0.6
    np.testing.assert_allclose(a["cf"][:, :, 2].sum(-1), model.total_mass * 9.81, rtol=0.25)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("rb,control_mode", [("h1", 2), ("g1", 0), ("g1", 2)])
def test_inertia_lag_robots_settle_on_their_feet(backend, rb, control_mode):
    """Revolute models (`pd` mode 2 and the implicit drive, joint limits, hull support points; G1 with up to 40 contact points per body): the settling
    test of test_dynamics.py with the switch on."""
    from test_dynamics import robot_settles_on_its_feet
    robot_settles_on_its_feet(backend, rb, control_mode, inertia_lag=1)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("lag", [0, 1])
def test_force_average_is_the_mean_over_the_sub_steps(backend, lag):
    """`force_average`: contact_force / dof_force of a 2 x 2 sub-step launch == the means of what four one-sub-step launches publish (each of those IS
    its sub-step's value); without the switch the launch publishes the last sub-step's.  The state is the same either way."""
    be = get_backend(backend)
    model, mstruct, keep = model_on(be)
    rng = np.random.default_rng(9)
    n = 6
    root, dof, target = random_states(model, n, rng, height=0.82, vel=0.5)
    last = run_step(be, model, mstruct, root, dof, target, abi.sim_params_struct(self_collision=1, inertia_lag=lag), 2)
    avg = run_step(be, model, mstruct, root, dof, target, abi.sim_params_struct(self_collision=1, inertia_lag=lag, force_average=1), 2)
    for k in ("root", "dof", "rbs"):
        np.testing.assert_array_equal(avg[k], last[k])
    if lag:
        return   # (one-sub-step launches are all fresh: the sub-step values below are those of the fresh scheme)
    one = abi.sim_params_struct(sim_dt=1 / 120, substeps=1, self_collision=1)
    a, cfs, dfs = dict(root=root, dof=dof), [], []
    for _ in range(4):
        a = run_step(be, model, mstruct, a["root"], a["dof"], target, one, 1)
        cfs.append(a["cf"]); dfs.append(a["df"])
    assert np.abs(np.array(cfs)).sum() > 100.0, "the case must exercise contact"
    # (the chained launches round-trip the state through fp32 exp-map coordinates: small differences)
    np.testing.assert_allclose(last["cf"], cfs[-1], atol=1.0, rtol=2e-2)
    np.testing.assert_allclose(avg["cf"], np.mean(cfs, 0), atol=1.0, rtol=2e-2)
    np.testing.assert_allclose(avg["df"], np.mean(dfs, 0), atol=0.3, rtol=2e-2)
    assert np.abs(avg["cf"] - last["cf"]).max() > 1.0   # (and the two publications do differ)



@pytest.mark.parametrize("backend", BACKENDS)
def test_option_combinations_the_solver_masks_cannot_serve_are_refused(backend):
    """no silent fall-backs at the C ABI -- `inertia_lag` with the three-wavefront experiment mapping (no lagged instantiation), the rigid contact model on a
    body with more than 32 contact points (32-bit active / removed masks) and `inertia_lag` above 64 points per body (the touch mask) return PHC_EUNSUPPORTED on both
    backends; the shipped models (SMPL 8, H1 8, G1 40 points on their busiest body) stay inside."""
    be = get_backend(backend)
    model, mstruct, keep = model_on(be)
    root, dof, target = random_states(model, 2, np.random.default_rng(0), height=0.95)
    UNSUPPORTED = -2   # PHC_EUNSUPPORTED (include/phc_amd.h)
    args = lambda p, ms=mstruct: (ms, p) + _sim_args(be, model, root, dof, target)
    assert 0 < mstruct.max_body_contact_pts <= 32
    assert be.sim_step(*args(abi.sim_params_struct(inertia_lag=1))) == 0
    assert be.sim_step(*args(abi.sim_params_struct(inertia_lag=1, lane_mapping=3))) == UNSUPPORTED
    many = type(mstruct).from_buffer_copy(mstruct)
    many.max_body_contact_pts = 40
    assert be.sim_step(*args(abi.sim_params_struct(contact_model=1, contact_iterations=4), many)) == UNSUPPORTED
    assert be.sim_step(*args(abi.sim_params_struct(inertia_lag=1), many)) == 0
    many.max_body_contact_pts = 65
    assert be.sim_step(*args(abi.sim_params_struct(inertia_lag=1), many)) == UNSUPPORTED
    assert be.sim_step(*args(abi.sim_params_struct(), many)) == 0
    be.sync()


def _sim_args(be, model, root, dof, target):
    n, nb, nd = root.shape[0], model.num_bodies, model.num_dof
    a = dict(root=be.arr(root), dof=be.arr(dof), rbs=be.zeros((n, nb, 13)), cf=be.zeros((n, nb, 3)), df=be.zeros((n, nd)), pd=be.arr(target))
    sim = abi.sim_state_struct(n, a["root"], a["dof"], a["rbs"], a["cf"], a["df"], a["pd"])
    _KEEP.append(a)
    return sim, None, None, None, None, 1


_KEEP = []


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("rb,control_mode", [("g1", 2), ("g1", 0), ("h1", 2)])
def test_inertia_lag_is_stable_under_target_noise(backend, rb, control_mode):
    """PD targets re-drawn at every env step (0.1 rad of noise around the default pose: what a rollout's exploration noise does) with the lagged scheme: joint rates and
    the power sum |dof_force * dof_vel| stay at the fresh scheme's.  Round 6 found the case the settling tests miss: G1's 14-gram finger links reach their joint limits
    between two fresh sub-steps; a limit damper that is not in the kept D^-1 is an explicit integrator there (dt c / I = 5e4: 100 rad/s, 55 kW of power term in the env,
    the G1 clips no longer learned).  Such a limit acts as a spring only until the next fresh sub-step (phc_aba.h)."""
    be = get_backend(backend)
    from phc_amd.robots import ROBOTS
    model, mstruct, keep = model_on(be, f"{rb}_humanoid")
    n, nb, nd = 8, model.num_bodies, model.num_dof
    out = {}
    for lag in (0, 1):
        root = np.zeros((n, 13), F)
        root[:, 6] = 1.0
        dof = np.zeros((n, nd, 2), F)
        dof[:, :, 0] = np.asarray(ROBOTS[rb]["default_dof_pos"], F)
        Q, R, p = do.kinematics(model, do.State(root[0], dof[0], model))
        low = min(float((p[b] + R[b] @ c)[2] - r) for b, c, r in zip(model.contact_body, model.contact_pos, model.contact_radius))
        root[:, 2] = 1.0 if rb == "h1" else 0.02 - low
        base = dof[:, :, 0].copy()
        params = abi.sim_params_struct(sim_dt=1 / 200, substeps=2, control_freq_inv=4, control_mode=control_mode, limit_stiffness=2000.0, limit_damping=20.0, inertia_lag=lag)
        a = dict(root=be.arr(root), dof=be.arr(dof), rbs=be.zeros((n, nb, 13)), cf=be.zeros((n, nb, 3)), df=be.zeros((n, nd)), pd=be.arr(base))
        sim = abi.sim_state_struct(n, a["root"], a["dof"], a["rbs"], a["cf"], a["df"], a["pd"])
        rng = np.random.default_rng(1)
        power, rate = [], []
        for k in range(25):
            a["pd"][...] = be.arr((base + 0.1 * rng.standard_normal(base.shape)).astype(F))
            assert be.sim_step(mstruct, params, sim, None, None, None, None, 4) == 0
            be.sync()
            d, f = be.np(a["dof"]), be.np(a["df"])
            power.append(float(np.abs(f * d[:, :, 1]).sum(-1).mean()))
            rate.append(float(np.abs(d[:, :, 1]).max()))
        out[lag] = (np.mean(power[5:]), max(rate))
    assert out[1][1] < 1.5 * out[0][1] + 1.0, f"joint rates with the lag {out[1][1]:.1f} rad/s against {out[0][1]:.1f} fresh"
    assert out[1][0] < 1.3 * out[0][0], f"power with the lag {out[1][0]:.0f} W against {out[0][0]:.0f} W fresh"
