"""TEST INFRASTRUCTURE ONLY -- fp64 oracle for the articulated-body stepper (S10).

There is no reference implementation of the dynamics (Isaac Gym / PhysX is a closed binary,
SURVEY.md section 8c): parity for this part of the path is *unpinned* at the PhysX level.  This
oracle pins the HIP kernel's arithmetic instead, with a formulation that shares nothing with the
kernel's articulated-body recursion:

    generalized coordinates  nu = [w0 (world), v0 (world, root origin), wJ_1 .. wJ_{NB-1} (child frames)]
    M(q) nu_dot + h(q, nu) = tau        with M = sum_i J_i^T M_i J_i built from body Jacobians,
    dense Cholesky-free solve (numpy.linalg.solve) in float64,

plus the same linearly-implicit treatment of the PD drive and of the penalty ground contact
(diag / J^T C J augmentations of M).  Featherstone's ABA must give the same accelerations as the
dense solve -- that is the check (tests/test_dynamics.py), together with physical invariants.

Only tests and bench.py's cpu_baseline leg import this file.
"""
import numpy as np


def cross3(a, b):
    """a x b for 3-vectors (numpy's general `cross` spends 30 us per call on axis bookkeeping; the oracle calls it ~300 times per evaluation)."""
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]])


def skew(a):
    return np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]], dtype=np.float64)


def quat_to_mat(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float64)


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def quat_conj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def quat_from_rotvec(e):
    a = np.linalg.norm(e)
    if a < 1e-12:
        return np.array([0.5 * e[0], 0.5 * e[1], 0.5 * e[2], 1.0])
    return np.concatenate([e / a * np.sin(0.5 * a), [np.cos(0.5 * a)]])


def quat_to_rotvec(q):
    if q[3] < 0:
        q = -q
    s = np.linalg.norm(q[:3])
    if s < 1e-12:
        return 2.0 * q[:3]
    return q[:3] * (2.0 * np.arctan2(s, q[3]) / s)


def quat_wxyz_to_xyzw(q):
    return np.array([q[1], q[2], q[3], q[0]], dtype=np.float64)


class State:
    """Per-env state in the simulator's tensor layout (S1 root_states [13], S2 dof_state [D,2]).  Spherical joints carry
    an exp-map triple per joint, revolute joints one angle; `q[i-1]` is always the child-in-parent rotation of body i
    (for a revolute joint: rest rotation * rot(axis, theta))."""

    def __init__(self, root_states, dof_state, model=None):
        r = np.asarray(root_states, dtype=np.float64)
        d = np.asarray(dof_state, dtype=np.float64)
        self.p0 = r[0:3].copy()
        self.q0 = r[3:7] / np.linalg.norm(r[3:7])
        self.v0 = r[7:10].copy()
        self.w0 = r[10:13].copy()
        self.model = model
        if model is None or model.all_spherical:
            nj = d.shape[0] // 3
            self.nd = [3] * nj
            self.q = np.array([quat_from_rotvec(d[3 * j:3 * j + 3, 0]) for j in range(nj)])
            self.qd = [d[3 * j:3 * j + 3, 1].copy() for j in range(nj)]
            self.theta = None
        else:
            nj = model.num_bodies - 1
            self.nd = [int(model.dof_count[i]) for i in range(1, model.num_bodies)]
            assert all(n == 1 for n in self.nd), "mixed joint types are not built"
            self.theta = d[:, 0].copy()
            self.qd = [d[j:j + 1, 1].copy() for j in range(nj)]
            self.q = np.array([self._rev_quat(j) for j in range(nj)])

    def _rev_quat(self, j):
        m = self.model
        rest = quat_wxyz_to_xyzw(m.local_rotation[j + 1])
        return quat_mul(rest, quat_from_rotvec(m.dof_axis[m.dof_start[j + 1]] * self.theta[j]))

    @property
    def wj(self):  # spherical joints: [NJ, 3] view kept for the existing tests
        return np.array(self.qd)

    def S(self, j):
        """Motion subspace of joint j (body j+1) in the child frame: 3 x nd."""
        if self.nd[j] == 3:
            return np.eye(3)
        return self.model.dof_axis[self.model.dof_start[j + 1]][:, None]

    def root_states(self):
        return np.concatenate([self.p0, self.q0, self.v0, self.w0])

    def dof_state(self):
        if self.theta is None:
            pos = np.concatenate([quat_to_rotvec(q) for q in self.q])
        else:
            pos = self.theta.copy()
        return np.stack([pos, np.concatenate(self.qd)], axis=-1)


DEFAULT_PARAMS = dict(gravity_z=-9.81, contact_stiffness=1.0e5, contact_damping=1.0e3, friction=1.0, friction_viscous=2.0e3,
                      angular_damping=0.01, max_angular_velocity=100.0, control_mode=0, limit_stiffness=0.0, limit_damping=0.0,
                      self_collision=0, self_stiffness_scale=0.25, self_damping_ratio=0.5,
                      # ground-contact model (include/phc_amd.h, ABI 34): 0 penalty, 1 rigid ("tgs")
                      contact_model=0, contact_iterations=4, contact_impedance=1.0e5, max_depenetration_velocity=10.0,
                      bounce_threshold_velocity=0.2, restitution=0.0, contact_offset=0.02)


def kinematics(model, st):
    nb = model.num_bodies
    R = [None] * nb
    Q = [None] * nb
    p = np.zeros((nb, 3))
    for i in range(nb):
        par = model.parent[i]
        if par < 0:
            Q[i] = st.q0
            p[i] = st.p0
        else:
            Q[i] = quat_mul(Q[par], st.q[i - 1])   # st.q already contains a revolute joint's rest rotation
            Q[i] = Q[i] / np.linalg.norm(Q[i])
            p[i] = p[par] + R[par] @ model.local_translation[i].astype(np.float64)
        R[i] = quat_to_mat(Q[i])
    return Q, R, p


def body_velocities(model, st, R, p):
    nb = model.num_bodies
    w = np.zeros((nb, 3))
    v = np.zeros((nb, 3))
    for i in range(nb):
        par = model.parent[i]
        if par < 0:
            w[i], v[i] = st.w0, st.v0
        else:
            w[i] = w[par] + R[i] @ (st.S(i - 1) @ st.qd[i - 1])
            v[i] = v[par] + cross3(w[par], p[i] - p[par])
    return w, v


def seg_seg_closest(p1, q1, p2, q2):
    """Closest points of two segments (Ericson, Real-Time Collision Detection 5.1.9); degenerate segments are points."""
    d1, d2, r = q1 - p1, q2 - p2, p1 - p2
    a, e, f = d1 @ d1, d2 @ d2, d2 @ r
    EPS = 1e-10
    s = t = 0.0
    if a <= EPS and e <= EPS:
        pass
    elif a <= EPS:
        t = min(max(f / e, 0.0), 1.0)
    else:
        c = d1 @ r
        if e <= EPS:
            s = min(max(-c / a, 0.0), 1.0)
        else:
            b = d1 @ d2
            denom = a * e - b * b
            s = min(max((b * f - c * e) / denom, 0.0), 1.0) if denom > 1e-7 * a * e else 0.0
            t = (b * s + f) / e
            if t < 0:
                t, s = 0.0, min(max(-c / a, 0.0), 1.0)
            elif t > 1:
                t, s = 1.0, min(max((b - c) / a, 0.0), 1.0)
    return p1 + d1 * s, p2 + d2 * t


def self_collision_wrenches(model, R, p, w, v, prm, dt):
    """Explicit capsule-capsule penalty forces between the collision SHAPES that may collide (model.collision_pairs(): a body may carry more
    than one capsule since round 4): (F[NB,3], N[NB,3] about each body origin)."""
    nb = model.num_bodies
    F, N = np.zeros((nb, 3)), np.zeros((nb, 3))
    own, cap = model.shape_owner(), model.shape_capsules()
    A = np.array([p[o] + R[o] @ c[0:3] for o, c in zip(own, cap)])
    B = np.array([p[o] + R[o] @ c[3:6] for o, c in zip(own, cap)])
    for s1, s2 in model.collision_pairs():
        for a, b in ((s1, s2), (s2, s1)):      # both directions: each owner receives its own force
            i, k = int(own[a]), int(own[b])
            c1, c2 = seg_seg_closest(A[a], B[a], A[b], B[b])
            n = c1 - c2
            dist = np.linalg.norm(n)
            pen = cap[a, 6] + cap[b, 6] - dist
            if pen <= 0:
                continue
            n = n / dist if dist > 1e-6 else np.array([0.0, 0.0, 1.0])
            cp = c2 + n * (cap[b, 6] - 0.5 * pen)
            vrel = (v[i] + cross3(w[i], cp - p[i])) - (v[k] + cross3(w[k], cp - p[k]))
            mu = model.mass[i] * model.mass[k] / (model.mass[i] + model.mass[k])
            kk = prm["self_stiffness_scale"] * mu / (dt * dt)
            cc = 2.0 * prm["self_damping_ratio"] * np.sqrt(kk * mu)
            fn = kk * pen - cc * (vrel @ n)
            if fn <= 0:
                continue
            F[i] += n * fn
            N[i] += cross3(cp - p[i], n * fn)
    return F, N


def explicit_torque(model, st, target, kp_scale=1.0, kd_scale=1.0):
    """`pd` control mode (humanoid.py:1575-1599): tau = clip(kp (target - q) - kd qd, +-limit), recomputed once per
    gym.simulate call and held over its sub-steps.  Revolute joints only (H1 / G1)."""
    th = st.theta
    qd = np.concatenate(st.qd)
    sp = model.dof_kp * kp_scale * (np.asarray(target, np.float64) - th)
    tau = sp - model.dof_kd * kd_scale * qd
    sat = np.abs(tau) >= model.dof_effort
    # (held torque of mode 1, [spring-or-limit, saturated flag] of mode 2)
    return np.clip(tau, -model.dof_effort, model.dof_effort), np.where(sat, np.sign(tau) * model.dof_effort, sp), sat


def accelerations(model, st, target, params, dt, kp_scale=1.0, kd_scale=1.0, return_parts=False, tau_hold=None, nud_prev=None, cstate=None):
    """Generalized accelerations nu_dot = [alpha0, a0, qdd_1..] of one implicit sub-step.  `nud_prev` (rigid contact model): the solution of
    the previous pass of this sub-step -- active set and friction cone are evaluated on the end-of-step velocities it predicts."""
    prm = dict(DEFAULT_PARAMS, **(params or {}))
    nb = model.num_bodies
    offs = np.concatenate([[6], 6 + np.cumsum(st.nd)]).astype(int)
    nv = int(offs[-1])
    col = lambda i: slice(offs[i - 1], offs[i])   # columns of body i's joint
    Q, R, p = kinematics(model, st)
    w, v = body_velocities(model, st, R, p)
    nu = np.concatenate([st.w0, st.v0] + list(st.qd))
    Jw = np.zeros((nb, 3, nv))
    Jv = np.zeros((nb, 3, nv))
    for i in range(nb):
        Jw[i][:, 0:3] = np.eye(3)
        Jv[i][:, 3:6] = np.eye(3)
        Jv[i][:, 0:3] = -skew(p[i] - p[0])
        k = i
        while k > 0:
            Sw = R[k] @ st.S(k - 1)
            Jw[i][:, col(k)] = Sw
            Jv[i][:, col(k)] = -skew(p[i] - p[k]) @ Sw
            k = model.parent[k]
        assert np.allclose(Jw[i] @ nu, w[i]) and np.allclose(Jv[i] @ nu, v[i])
    # bias accelerations (nu_dot = 0)
    ab_w = np.zeros((nb, 3))
    ab_v = np.zeros((nb, 3))
    for i in range(1, nb):
        par = model.parent[i]
        r = p[i] - p[par]
        ab_w[i] = ab_w[par] + cross3(w[par], R[i] @ (st.S(i - 1) @ st.qd[i - 1]))
        ab_v[i] = ab_v[par] + cross3(ab_w[par], r) + cross3(w[par], cross3(w[par], r))
    M = np.zeros((nv, nv))
    rhs = np.zeros(nv)
    g = np.array([0.0, 0.0, prm["gravity_z"]])
    cn = prm["contact_stiffness"] * dt + prm["contact_damping"]
    fcontact = np.zeros((nb, 3))
    # rigid model: contact points (global index) that pushed in the previous pass / were released for the rest of the sub-step
    cstate = cstate if cstate is not None else {"active": set(), "removed": set()}
    act_now = set()
    for i in range(nb):
        Io = R[i] @ model.inertia_origin[i] @ R[i].T
        mc = R[i] @ (model.mass[i] * model.com[i])
        Mi = np.zeros((6, 6))
        Mi[:3, :3] = Io
        Mi[:3, 3:] = skew(mc)
        Mi[3:, :3] = -skew(mc)
        Mi[3:, 3:] = model.mass[i] * np.eye(3)
        J = np.concatenate([Jw[i], Jv[i]], axis=0)
        bias = np.concatenate([cross3(w[i], Io @ w[i]), cross3(w[i], cross3(w[i], mc))])
        ext = np.concatenate([cross3(mc, g), model.mass[i] * g])
        M += J.T @ Mi @ J
        rhs -= J.T @ (Mi @ np.concatenate([ab_w[i], ab_v[i]]) + bias - ext)
        for k in np.nonzero(model.contact_body == i)[0]:
            arm = R[i] @ model.contact_pos[k]
            rad = model.contact_radius[k]
            depth = rad - (p[i][2] + arm[2])
            if prm["contact_model"] == 1:
                # rigid: unilateral velocity constraint u_n(t + dt) >= v_target through the impedance c; active set and cone from the previous pass
                if depth <= -prm["contact_offset"]:
                    continue
                c = prm["contact_impedance"]
                arm = arm - np.array([0, 0, rad])
                uc = v[i] + cross3(w[i], arm)
                apb = ab_v[i] + cross3(ab_w[i], arm) + cross3(w[i], cross3(w[i], arm))
                Jpt = Jv[i] - skew(arm) @ Jw[i]
                un = uc if nud_prev is None else uc + dt * (Jpt @ nud_prev + apb)
                vt = min(depth / dt, prm["max_depenetration_velocity"]) if depth > 0 else depth / dt
                if prm["restitution"] > 0 and -uc[2] > prm["bounce_threshold_velocity"]:
                    vt = max(vt, -prm["restitution"] * uc[2])
                lam = c * (vt - un[2])
                if lam <= 0:   # idle -- or released: it pushed in the last pass and would have to pull (phc_aba.h aba_ground_contact_rigid)
                    if k in cstate["active"]:
                        cstate["removed"].add(k)
                    continue
                if k in cstate["removed"]:
                    continue
                act_now.add(k)
                ct = rigid_ct(prm, lam, uc, un)
                Cm = np.diag([ct, ct, c])
                F0 = np.array([-ct * uc[0], -ct * uc[1], c * (vt - uc[2])])
                M += dt * Jpt.T @ Cm @ Jpt
                rhs += Jpt.T @ (F0 - dt * Cm @ apb)
                continue
            if depth <= 0:
                continue
            arm = arm - np.array([0, 0, rad])
            uc = v[i] + cross3(w[i], arm)
            fn0 = prm["contact_stiffness"] * depth - cn * uc[2]
            if fn0 <= 0:
                continue
            ut = np.hypot(uc[0], uc[1])
            ct = min(prm["friction_viscous"], prm["friction"] * fn0 / (ut + 1e-6))
            Cm = np.diag([ct, ct, cn])
            F0 = np.array([-ct * uc[0], -ct * uc[1], fn0])
            fcontact[i] += F0 - dt * Cm @ cross3(w[i], cross3(w[i], arm))
            apb = ab_v[i] + cross3(ab_w[i], arm) + cross3(w[i], cross3(w[i], arm))
            Jpt = Jv[i] - skew(arm) @ Jw[i]
            M += dt * Jpt.T @ Cm @ Jpt
            rhs += Jpt.T @ (F0 - dt * Cm @ apb)
    if prm["self_collision"]:
        Fs, Ns = self_collision_wrenches(model, R, p, w, v, prm, dt)
        for i in range(nb):
            rhs += Jw[i].T @ Ns[i] + Jv[i].T @ Fs[i]
            fcontact[i] += Fs[i]
    tau_all = [None] * nb
    dimp_all = [None] * nb
    for i in range(1, nb):
        s, n = model.dof_start[i], st.nd[i - 1]
        kp = model.dof_kp[s:s + n] * kp_scale
        kd = model.dof_kd[s:s + n] * kd_scale
        arm_ = model.dof_armature[s:s + n]
        eff = model.dof_effort[s:s + n]
        qd = st.qd[i - 1]
        if n == 3:
            qt = quat_from_rotvec(np.asarray(target[s:s + 3], dtype=np.float64))
            err = quat_to_rotvec(quat_mul(quat_conj(st.q[i - 1]), qt))
        else:
            err = np.asarray(target[s:s + 1], dtype=np.float64) - st.theta[i - 1]
        if prm["control_mode"] == 1:   # explicit torque held over the simulate call
            tau = np.array(tau_hold[0][s:s + n], dtype=np.float64)
            d = arm_.copy()
        elif prm["control_mode"] == 2:  # spring sampled per simulate call, damper continuous (implicit) unless saturated
            if tau_hold[2][s]:
                tau = np.array(tau_hold[1][s:s + n], dtype=np.float64)
                d = arm_.copy()
            else:
                tau = tau_hold[1][s:s + n] - kd * qd
                d = arm_ + dt * kd
        else:
            tau = np.clip(kp * err, -eff, eff) - (kd + dt * kp) * qd  # spring saturates, damping stays implicit
            d = arm_ + dt * kd + dt * dt * kp
        if n == 1 and prm["limit_stiffness"] > 0:  # joint limit: implicit penalty spring-damper beyond [lo, hi]
            lo, hi = model.dof_limits()
            th = st.theta[i - 1]
            e = (lo[s] - th) if th < lo[s] else ((hi[s] - th) if th > hi[s] else 0.0)
            if e != 0.0:
                kl, dl = prm["limit_stiffness"], prm["limit_damping"]
                tau = tau + kl * e - (dl + dt * kl) * qd
                d = d + dt * dl + dt * dt * kl
        M[col(i), col(i)] += np.diag(d)
        rhs[col(i)] += tau
        tau_all[i], dimp_all[i] = tau, d
    nud = np.linalg.solve(M, rhs)
    cstate["active"] = act_now
    if return_parts:
        return nud, dict(M=M, rhs=rhs, R=R, p=p, w=w, v=v, Q=Q, tau=tau_all, dimp=dimp_all, fcontact=fcontact, offs=offs, Jw=Jw, Jv=Jv, ab_w=ab_w, ab_v=ab_v)
    return nud


def rigid_ct(prm, lam, uc, un):
    """Tangential impedance of a pushing point of the rigid model (phc_aba.h rigid_point_ct): c_t = min(c, mu lambda / max(|u_t|, |u+_t|))."""
    return min(prm["contact_impedance"], prm["friction"] * lam / (max(np.hypot(uc[0], uc[1]), np.hypot(un[0], un[1])) + 1e-9))


def rigid_contact_forces(model, parts, prm, dt, nud, active):
    """Rigid contact model: the contact law evaluated on the end-of-step velocities the FINAL solve predicts -- net ground force per body (S4)."""
    nb = model.num_bodies
    R, p, w, v = parts["R"], parts["p"], parts["w"], parts["v"]
    F = np.zeros((nb, 3))
    c = prm["contact_impedance"]
    for k, i in enumerate(model.contact_body):
        arm = R[i] @ model.contact_pos[k]
        rad = model.contact_radius[k]
        depth = rad - (p[i][2] + arm[2])
        if depth <= -prm["contact_offset"]:
            continue
        arm = arm - np.array([0, 0, rad])
        uc = v[i] + cross3(w[i], arm)
        apb = parts["ab_v"][i] + cross3(parts["ab_w"][i], arm) + cross3(w[i], cross3(w[i], arm))
        Jpt = parts["Jv"][i] - skew(arm) @ parts["Jw"][i]
        un = uc + dt * (Jpt @ nud + apb)
        vt = min(depth / dt, prm["max_depenetration_velocity"]) if depth > 0 else depth / dt
        if prm["restitution"] > 0 and -uc[2] > prm["bounce_threshold_velocity"]:
            vt = max(vt, -prm["restitution"] * uc[2])
        lam = c * (vt - un[2])
        if lam <= 0 or k not in active:   # only the points the final solve let push
            continue
        ct = rigid_ct(prm, lam, uc, un)
        F[i] += np.array([-ct * un[0], -ct * un[1], lam])
    return F


def substep(model, st, target, params, dt, kp_scale=1.0, kd_scale=1.0, tau_hold=None):
    """One linearly-implicit sub-step; returns the applied joint torques (S5)."""
    prm = dict(DEFAULT_PARAMS, **(params or {}))
    nud, parts = accelerations(model, st, target, prm, dt, kp_scale, kd_scale, return_parts=True, tau_hold=tau_hold)
    if prm["contact_model"] == 1:
        cstate = {"active": set(), "removed": set()}
        nud, parts = accelerations(model, st, target, prm, dt, kp_scale, kd_scale, return_parts=True, tau_hold=tau_hold, cstate=cstate)
        for _ in range(1, max(1, int(prm["contact_iterations"]))):   # passes on active set and friction cone
            nud, parts = accelerations(model, st, target, prm, dt, kp_scale, kd_scale, return_parts=True, tau_hold=tau_hold, nud_prev=nud, cstate=cstate)
        parts["fcontact"] = rigid_contact_forces(model, parts, prm, dt, nud, cstate["active"]) + parts["fcontact"]
    damp = 1.0 / (1.0 + dt * prm["angular_damping"])
    st.w0 = (st.w0 + dt * nud[0:3]) * damp
    st.v0 = st.v0 + dt * nud[3:6]
    st.p0 = st.p0 + dt * st.v0
    q = quat_mul(quat_from_rotvec(st.w0 * dt), st.q0)
    st.q0 = q / np.linalg.norm(q)
    nb = model.num_bodies
    offs = parts["offs"]
    tau_applied = []
    for i in range(1, nb):
        qdd = nud[offs[i - 1]:offs[i]]
        s, n = model.dof_start[i], st.nd[i - 1]
        tau_applied.append(parts["tau"][i] - (parts["dimp"][i] - model.dof_armature[s:s + n]) * qdd)
        wj = (st.qd[i - 1] + dt * qdd) * damp
        nrm = np.linalg.norm(wj)
        if nrm > prm["max_angular_velocity"]:
            wj = wj * (prm["max_angular_velocity"] / nrm)
        st.qd[i - 1] = wj
        if n == 3:
            q = quat_mul(st.q[i - 1], quat_from_rotvec(wj * dt))
            st.q[i - 1] = q / np.linalg.norm(q)
        else:
            st.theta[i - 1] += wj[0] * dt
            st.q[i - 1] = st._rev_quat(i - 1)
    return np.concatenate(tau_applied), parts["fcontact"]


def sim_step(model, root_states, dof_state, pd_target, params=None, sim_dt=1 / 60, substeps=2, num_sim_calls=2,
             kp_scale=1.0, kd_scale=1.0):
    """num_sim_calls x gym.simulate, each `substeps` sub-steps of sim_dt/substeps.  Returns new
    (root_states[13], dof_state[D,2], rigid_body_state[NB,13], dof_force[D], contact_force[NB,3])."""
    st = State(root_states, dof_state, model)
    dt = sim_dt / substeps
    tau = np.zeros(model.num_dof)
    fc = np.zeros((model.num_bodies, 3))
    explicit = dict(DEFAULT_PARAMS, **(params or {}))["control_mode"] in (1, 2)
    hold = None
    for k in range(num_sim_calls * substeps):
        if explicit and k % substeps == 0:
            hold = explicit_torque(model, st, pd_target, kp_scale, kd_scale)
        tau, fc = substep(model, st, pd_target, params, dt, kp_scale, kd_scale, tau_hold=hold)
    Q, R, p = kinematics(model, st)
    w, v = body_velocities(model, st, R, p)
    rbs = np.concatenate([p, np.array(Q), v, w], axis=-1)
    return st.root_states(), st.dof_state(), rbs, tau, fc


def energy(model, st, gravity_z=-9.81):
    """Total mechanical energy (kinetic + gravitational), armature excluded -- for the invariants tests."""
    Q, R, p = kinematics(model, st)
    w, v = body_velocities(model, st, R, p)
    E = 0.0
    for i in range(model.num_bodies):
        Io = R[i] @ model.inertia_origin[i] @ R[i].T
        mc = R[i] @ (model.mass[i] * model.com[i])
        E += 0.5 * w[i] @ Io @ w[i] + 0.5 * model.mass[i] * v[i] @ v[i] + v[i] @ cross3(w[i], mc)
        com_z = p[i][2] + (R[i] @ model.com[i])[2]
        E -= model.mass[i] * gravity_z * com_z
    return E


# ------------------------------------------------------------------------------------------------------------------------------------
# The CONTINUOUS model (round 4): what the stepper's linearly-implicit scheme discretises.  `accelerations(..., dt=0)` is exactly the
# right-hand side of the ODE  M(q) nu_dot + h(q, nu) = tau_PD(q, nu) + J^T F_contact(q, nu):  every dt-weighted term of the implicit
# scheme vanishes -- the joint-space matrix is the armature alone, the drive is tau = clamp(kp err, +-effort) - kd qd, a touching point
# pushes with the penalty force F_n = max(0, k depth - d v_n), F_t = -min(c_max, mu F_n / |u_t|) u_t.  `ode_step` integrates it EXPLICITLY
# (symplectic Euler, fp64) at a step far below the stepper's; tests/test_dynamics.py::test_stepper_converges_to_the_continuous_model checks
# that the stepper approaches this solution at FIRST ORDER as its dt -> 0.  That pins the MODELLING of the implicit scheme (which terms
# enter D, the sign and the dt-weights of every implicit correction, the saturation rule), not just its arithmetic -- `sim_step` above
# shares the scheme with the kernel.  (Body-body contact has no continuous limit by construction -- its stiffness is tied to dt -- and
# the rate clamps are excluded: the scenarios stay below them.)
def ode_step(model, st, target, params, h):
    prm = dict(DEFAULT_PARAMS, **(params or {}))
    assert not prm["self_collision"] and prm["control_mode"] == 0
    nud, parts = accelerations(model, st, target, prm, 0.0, return_parts=True)
    c = prm["angular_damping"]
    offs = parts["offs"]
    st.w0 = st.w0 + h * (nud[0:3] - c * st.w0)
    st.v0 = st.v0 + h * nud[3:6]
    st.p0 = st.p0 + h * st.v0
    q = quat_mul(quat_from_rotvec(st.w0 * h), st.q0)
    st.q0 = q / np.linalg.norm(q)
    for i in range(1, model.num_bodies):
        qdd = nud[offs[i - 1]:offs[i]]
        wj = st.qd[i - 1] + h * (qdd - c * st.qd[i - 1])
        st.qd[i - 1] = wj
        if st.nd[i - 1] == 3:
            q = quat_mul(st.q[i - 1], quat_from_rotvec(wj * h))
            st.q[i - 1] = q / np.linalg.norm(q)
        else:
            st.theta[i - 1] += wj[0] * h
            st.q[i - 1] = st._rev_quat(i - 1)


def ode_solve(model, root_states, dof_state, pd_target, T, h, params=None):
    """Integrate the continuous model over [0, T] with explicit steps of (about) h.  Returns (root_states, dof_state, rigid_body_state)."""
    st = State(root_states, dof_state, model)
    n = max(1, int(round(T / h)))
    for _ in range(n):
        ode_step(model, st, pd_target, params, T / n)
    Q, R, p = kinematics(model, st)
    w, v = body_velocities(model, st, R, p)
    return st.root_states(), st.dof_state(), np.concatenate([p, np.array(Q), v, w], axis=-1)


def ode_body_positions(model, root_states, dof_state, pd_target, T, h, params=None, levels=2):
    """Body positions of the continuous model at time T by RICHARDSON extrapolation of the explicit first-order integrator: x(h / 2) * 2 - x(h)
    cancels the O(h) term.  Returns (extrapolated positions [NB, 3], estimate of their error = the difference between the extrapolations from
    (h, h / 2) and (h / 2, h / 4) when `levels` == 3, else the last correction's size)."""
    xs = [ode_solve(model, root_states, dof_state, pd_target, T, h / 2 ** k, params)[2][:, 0:3] for k in range(levels)]
    ex = [2.0 * xs[k + 1] - xs[k] for k in range(levels - 1)]
    err = np.abs(ex[-1] - ex[-2]).max() if levels >= 3 else np.abs(ex[-1] - xs[-1]).max()
    return ex[-1], err
