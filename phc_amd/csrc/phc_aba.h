// phc_aba.h -- articulated-rigid-body stepper (S10 in SURVEY.md section 8a): the replacement for
// the closed `gym.simulate` call (reference phc/env/tasks/humanoid.py:1613,1618).
//
// Algorithm: Featherstone's articulated-body algorithm (Rigid Body Dynamics Algorithms, ch. 7)
// restated for one *lane per body*:
//   * spatial quantities are expressed in WORLD-ALIGNED axes, each body's about its OWN origin
//     (= joint anchor), with classical (not spatial) accelerations.  Parent<->child transforms are
//     then pure translations by r = p_child - p_parent, moment arms stay below one link length
//     (fp32-friendly), and the spherical-joint motion subspace is simply S = [1_3; 0].
//   * the sweeps (articulated inertia leaves->root; accelerations root->leaves) run level-synchronously: at tree level l only
//     the lanes whose body sits at level l work; parent/child hand-off goes through a 28-float LDS slot per body.  A sub-step is
//     TWO sweeps: the accelerations sweep also integrates each joint and produces the next kinematics on its way down.
//     One body per lane (32 lanes / env; 64 above 32 bodies; phc_sim.hip).  Joint families are template instantiations (spherical: SMPL;
//     revolute with rest rotations: H1 / G1).
//   * everything stiff is integrated LINEARLY IMPLICITLY by augmenting the articulated inertia:
//       - PD drive (kp,kd) + armature:  D += R diag(armature + dt*kd + dt^2*kp) R^T,
//         tau_explicit = clamp(kp*err, +-effort) - (kd + dt*kp)*w_joint   (Isaac Gym "isaac_pd" drive, S8)
//       - penalty ground contact with regularised Coulomb friction at sphere / capsule-end / box-corner
//         points: I^A += dt * J^T C J,  p^A -= J^T F0               (C = diag(ct,ct,kn*dt+dn))
//     so kp=800 / kd=80 at dt=1/120 s on light distal links stays stable (explicit PD would not).
// There is no reference implementation of this arithmetic (PhysX is closed): the oracle is
// oracle/dyn_oracle.py (fp64 dense mass-matrix solve, an independent formulation) + physical invariants --
// "parity unpinned" at the PhysX level.
#pragma once
#include "phc_math.h"
#include "../../include/phc_amd.h"

namespace phc {

#define PHC_XCH_STRIDE 28      // floats per exchange slot
#define PHC_BODY_FLOATS 56     // floats per body in phc_model_t.floats (model.py pack())
#define PHC_CAP_STRIDE 20       // 32-bit words per body in the per-env world-capsule exchange area (self-collision)
#define PHC_SC_FSCALE 1024.0f    // fixed-point scales of the body-body force / moment accumulators (1/1024 N, 1/4096 N m)
#define PHC_SC_NSCALE 4096.0f
#define PHC_JT_SPHERICAL 1     // joint types as model.py numbers them
#define PHC_JT_REVOLUTE 2
#define PHC_NTAB 21            // int tables per model (parent level jtype dof_start child0-2 nchild cp_start cp_count order misc collide-mask | solver tree: parent level child0-2 nchild jsrc/bsrc | pointer-jumping anchors)

// set of ground-contact points of one body (bit k = point k of the body's slice of the table): 64 bits -- Unitree G1 has bodies with 40 points
#if defined(PHC_CP_MASK32)
typedef uint32_t CpMask;
#define PHC_CP_CTZ(x) __builtin_ctz(x)
#else
typedef uint64_t CpMask;
#define PHC_CP_CTZ(x) __builtin_ctzll(x)
#endif
#define PHC_CP_BITS ((int)(8 * sizeof(CpMask)))

struct Sym3 { float xx, xy, xz, yy, yz, zz; };

PHC_HD V3 sym_mul(const Sym3& s, V3 v) {
    return v3(s.xx * v.x + s.xy * v.y + s.xz * v.z, s.xy * v.x + s.yy * v.y + s.yz * v.z, s.xz * v.x + s.yz * v.y + s.zz * v.z);
}
PHC_HD Sym3 sym_inv(const Sym3& s) {
    float c00 = s.yy * s.zz - s.yz * s.yz;
    float c01 = s.xz * s.yz - s.xy * s.zz;
    float c02 = s.xy * s.yz - s.xz * s.yy;
    float det = s.xx * c00 + s.xy * c01 + s.xz * c02;
    float id = 1.0f / det;
    Sym3 r;
    r.xx = c00 * id; r.xy = c01 * id; r.xz = c02 * id;
    r.yy = (s.xx * s.zz - s.xz * s.xz) * id;
    r.yz = (s.xy * s.xz - s.xx * s.yz) * id;
    r.zz = (s.xx * s.yy - s.xy * s.xy) * id;
    return r;
}
// R diag(d) R^T
PHC_HD Sym3 rot_diag(const M3& R, V3 d) {
    Sym3 s;
    const float* m = R.m;
    s.xx = m[0] * m[0] * d.x + m[1] * m[1] * d.y + m[2] * m[2] * d.z;
    s.xy = m[0] * m[3] * d.x + m[1] * m[4] * d.y + m[2] * m[5] * d.z;
    s.xz = m[0] * m[6] * d.x + m[1] * m[7] * d.y + m[2] * m[8] * d.z;
    s.yy = m[3] * m[3] * d.x + m[4] * m[4] * d.y + m[5] * m[5] * d.z;
    s.yz = m[3] * m[6] * d.x + m[4] * m[7] * d.y + m[5] * m[8] * d.z;
    s.zz = m[6] * m[6] * d.x + m[7] * m[7] * d.y + m[8] * m[8] * d.z;
    return s;
}
// R S R^T for symmetric S
PHC_HD Sym3 rot_sym(const M3& R, const Sym3& S) {
    // T = R S  (rows of R times S)
    float t[9];
    for (int i = 0; i < 3; ++i) {
        float a = R.m[3 * i], b = R.m[3 * i + 1], c = R.m[3 * i + 2];
        t[3 * i + 0] = a * S.xx + b * S.xy + c * S.xz;
        t[3 * i + 1] = a * S.xy + b * S.yy + c * S.yz;
        t[3 * i + 2] = a * S.xz + b * S.yz + c * S.zz;
    }
    Sym3 r;
    r.xx = t[0] * R.m[0] + t[1] * R.m[1] + t[2] * R.m[2];
    r.xy = t[0] * R.m[3] + t[1] * R.m[4] + t[2] * R.m[5];
    r.xz = t[0] * R.m[6] + t[1] * R.m[7] + t[2] * R.m[8];
    r.yy = t[3] * R.m[3] + t[4] * R.m[4] + t[5] * R.m[5];
    r.yz = t[3] * R.m[6] + t[4] * R.m[7] + t[5] * R.m[8];
    r.zz = t[6] * R.m[6] + t[7] * R.m[7] + t[8] * R.m[8];
    return r;
}

// 6x6 symmetric articulated inertia in blocks: n = A alpha + B a ; f = B^T alpha + C a
struct Inertia6 { Sym3 A; float B[9]; Sym3 C; };
struct Force6 { V3 n, f; };

PHC_HD V3 B_mul(const float* B, V3 v) {
    return v3(B[0] * v.x + B[1] * v.y + B[2] * v.z, B[3] * v.x + B[4] * v.y + B[5] * v.z, B[6] * v.x + B[7] * v.y + B[8] * v.z);
}
PHC_HD V3 Bt_mul(const float* B, V3 v) {
    return v3(B[0] * v.x + B[3] * v.y + B[6] * v.z, B[1] * v.x + B[4] * v.y + B[7] * v.z, B[2] * v.x + B[5] * v.y + B[8] * v.z);
}

// Per-lane registers of the stepper.
struct AbaLane {
    // --- constants (model); everything only body_init needs (mass, inertia, gains, contact points) is re-read
    //     from the L2-resident model there instead of being pinned in registers across the sweeps ---
    int parent, level, dof_start;   // kinematic tree (rooted at body 0, the simulator's root): state, kinematics sweep, integration
    // solver tree (model.py solver_tree(): the same articulation rooted at the body that minimises the depth -- the backward and the
    // acceleration sweeps walk THIS tree; identical to the kinematic tree unless the model re-roots):
    int sparent, slevel, nchild, child[3];   // (children: of the solver tree)
    int anchors;      // kinematics by pointer jumping: the body's anchor in steps 0..3, 8 bits each (0xff: already in the world frame)
    int jsrc;         // body whose joint links this body to its solver parent (itself; its solver parent for a REVERSED body; -1 base)
    int bsrc;         // bodies whose own joint is solved by a reversed body: that body (their kinematic parent), else -1
    V3 r_local;       // offset from parent origin, parent frame
    // --- state ---
    Q4 q;             // joint rotation child-in-parent (root: world rotation)
    V3 wj;            // joint velocity, child frame (root: unused)
                      // (the root's position / velocities ARE its world kinematics p, v, w below: no separate copy in registers)
    V3 target;        // PD target, exp-map
    // --- kinematics (world) ---
    Q4 Q; V3 p, w, v;
    V3 rw, cw, ca;    // solver: reference point minus the solver parent's (world); velocity-product accelerations of the solver joint
    // --- articulated quantities kept between the sweeps ---
    Inertia6 IA; Force6 pA;
    Sym3 Di; V3 u;    // D^-1 and tau_w - p_omega
    // world-frame joint-drive terms, rotated ONCE per sub-step in aba_body_init (the backward level-step is issued once per tree level):
    Sym3 Dw;          // R diag(dimp + arm) R^T (spherical)
    float diso;       // >= 0: the three entries of dimp + arm are equal (Dw = diso * 1, no rotation needed; true of every SMPL joint) -- else -1
    V3 tau_w;         // R tau_local
    V3 aw;            // R axis (revolute)
    V3 tau_local;     // explicit joint torque (child frame), for dof_force publication
    V3 dimp;          // implicit PD diagonal dt kd + dt^2 kp (WITHOUT the armature), child frame
    V3 arm;           // armature
    V3 fcontact;      // net explicit contact force on the body (S4)
    // --- revolute joints only (robots: H1 / G1); for them target.x, dimp.x, arm.x, tau_local = axis * tau carry the scalars ---
    V3 axis;          // joint axis, child (= parent-rest) frame
    Q4 qrest;         // rest rotation child-in-parent (MJCF body quat)
    float th, thd;    // joint angle and rate
    float tau_hold;   // explicit `pd` torque of the current simulate call (control_mode 1)
    // --- body-body contact (self-collision): net explicit force / moment about the origin on this body, set by aba_self_collision ---
    V3 fself, nself;
    // the collision capsule this LANE publishes (a body lane: its body's primary capsule; an idle lane behind the bodies: an extra shape of
    // body cap_owner), in the owner's frame -- loaded once per launch
    V3 cap_a, cap_b; float cap_r, cap_m; int cap_owner;
    // --- rigid ground contact (contact_model 1): the body's world angular acceleration and the classical acceleration of its solver reference
    //     point as the LAST solve of this sub-step left them (aba_accel_level) -- the next pass evaluates active set and friction cone on them ---
    V3 acc_w, acc_v;
    uint32_t c_active, c_removed;   // per contact point of the body (bit k, k < 32): pushed in the previous pass / released for the rest of the sub-step
    // --- lagged articulated inertia (phc_sim_params_t.inertia_lag): the ground-contact points that carried force in the last FRESH sub-step --
    //     the only ones whose impedance the kept I^A contains, hence the only ones a lagged sub-step lets push
    CpMask c_touch;
};

// per-env body shapes (phc_model_t.num_shapes > 1): the env's block of the int / float tables; the scalar header fields stay shared
PHC_HD phc_model_t model_for_env(phc_model_t m, const phc_sim_state_t& s, int64_t env) {
    if (m.num_shapes > 1 && s.env_shape != nullptr && env < s.num_envs) {
        const int sh = s.env_shape[env];
        m.ints += (int64_t)sh * m.int_stride;
        m.floats += (int64_t)sh * m.float_stride;
    }
    return m;
}
PHC_HD const float* model_body(const phc_model_t& m, int j) { return m.floats + j * PHC_BODY_FLOATS; }
PHC_HD int model_tab(const phc_model_t& m, int table, int j) { return m.ints[4 + table * PHC_MAX_BODIES + j]; }

// depth of the solver tree (misc table: [split level, bodies below it, solver depth, solver base])
PHC_HD int model_solver_depth(const phc_model_t& m, bool reroot) { return reroot ? model_tab(m, 11, 2) : m.max_level; }
// a body's solver reference point (body frame), m * (com - it), inertia about it: floats [44..56) -- or the origin's when not re-rooted
struct SolverRef { V3 off, mc; Sym3 Io; };
PHC_HD SolverRef model_solver_ref(const float* f, bool reroot) {
    SolverRef r;
    const int o = reroot ? 47 : 4;
    r.off = reroot ? v3(f[44], f[45], f[46]) : v3(0.f, 0.f, 0.f);
    r.mc = v3(f[o], f[o + 1], f[o + 2]);
    r.Io.xx = f[o + 3]; r.Io.xy = f[o + 4]; r.Io.xz = f[o + 5]; r.Io.yy = f[o + 6]; r.Io.yz = f[o + 7]; r.Io.zz = f[o + 8];
    return r;
}

PHC_HD void aba_load_model(AbaLane& L, const phc_model_t& m, int j, bool reroot = true) {
    L.parent = model_tab(m, 0, j); L.level = model_tab(m, 1, j);
    L.dof_start = model_tab(m, 3, j);
    L.anchors = model_tab(m, 20, j);
    if (reroot) {
        L.sparent = model_tab(m, 13, j); L.slevel = model_tab(m, 14, j);
        L.child[0] = model_tab(m, 15, j); L.child[1] = model_tab(m, 16, j); L.child[2] = model_tab(m, 17, j);
        L.nchild = model_tab(m, 18, j);
        const int js = model_tab(m, 19, j);
        L.jsrc = (js & 0xff) - 1; L.bsrc = ((js >> 8) & 0xff) - 1;
    } else {   // solve on the kinematic tree (the two-slot kernel: its slots are split by kinematic level)
        L.sparent = L.parent; L.slevel = L.level;
        L.child[0] = model_tab(m, 4, j); L.child[1] = model_tab(m, 5, j); L.child[2] = model_tab(m, 6, j);
        L.nchild = model_tab(m, 7, j);
        L.jsrc = j > 0 ? j : -1; L.bsrc = -1;
    }
    L.rw = L.cw = L.ca = v3(0.f, 0.f, 0.f);
    const float* f = model_body(m, j);
    L.r_local = v3(f[0], f[1], f[2]);
    L.arm = v3(f[19], f[20], f[21]);
    L.fself = L.nself = v3(0.f, 0.f, 0.f);
    L.cap_a = v3(f[36], f[37], f[38]); L.cap_b = v3(f[39], f[40], f[41]); L.cap_r = f[42]; L.cap_m = f[3]; L.cap_owner = j;
}
// revolute extras (template JT == PHC_JT_REVOLUTE paths only)
// convenience overload: constants read from the model at every call
template <int JT, bool RIGID = false>
PHC_HD void aba_body_init(AbaLane& L, const phc_model_t& m, const phc_sim_params_t& prm, float dt, int j, bool new_sim_call, bool reroot = true, int pass = 0, bool lag = false);

PHC_HD void aba_load_model_rev(AbaLane& L, const phc_model_t& m, int j) {
    const float* f = model_body(m, j);
    L.axis = v3(f[25], f[26], f[27]);
    L.qrest = q4(f[28], f[29], f[30], f[31]);
    L.tau_hold = 0.f;
}
PHC_HD Q4 rev_joint_quat(const AbaLane& L) { return quat_mul16(L.qrest, quat_from_rotvec(L.axis * L.th)); }

// State load: S1 root_states [N,13], S2 dof_state [N,D,2] (spherical: exp-map triple + joint velocity; revolute: angle +
// rate), S8 pd_target [N,D]
template <int JT>
PHC_HD void aba_load_state(AbaLane& L, const phc_sim_state_t& s, int nd, int64_t env, int j, bool load_target = true) {
    if (j == 0) {
        const float* r = s.root_states + env * 13;
        L.p = v3(r[0], r[1], r[2]); L.q = quat_normalize(q4(r[3], r[4], r[5], r[6]));
        L.v = v3(r[7], r[8], r[9]); L.w = v3(r[10], r[11], r[12]);
        L.wj = v3(0.f, 0.f, 0.f); L.target = v3(0.f, 0.f, 0.f);
        if (JT == PHC_JT_REVOLUTE) { L.th = L.thd = 0.f; }
    } else if (JT == PHC_JT_REVOLUTE) {
        const float* d = s.dof_state + (env * nd + L.dof_start) * 2;
        L.th = d[0]; L.thd = d[1];
        L.q = rev_joint_quat(L);
        L.wj = L.axis * L.thd;
        L.target = load_target ? v3(s.pd_target[env * nd + L.dof_start], 0.f, 0.f) : v3(0.f, 0.f, 0.f);
    } else {
        const float* d = s.dof_state + (env * nd + L.dof_start) * 2;
        L.q = quat_from_rotvec(v3(d[0], d[2], d[4]));
        L.wj = v3(d[1], d[3], d[5]);
        const float* t = s.pd_target + env * nd + L.dof_start;
        L.target = load_target ? v3(t[0], t[1], t[2]) : v3(0.f, 0.f, 0.f);
    }
}

// Exchange buffer: body b's slot is the 28 contiguous floats at base + 28*b (LDS on the device, a stack array in the
// host emulation).  Strides are compile-time constants so slot traffic becomes wide ds_read/ds_write with immediate offsets.
struct Xch {
    float* base;
    static constexpr int bs = PHC_XCH_STRIDE;
    static constexpr int es = 1;
};
PHC_HD float* xslot(const Xch& x, int body) { return x.base + body * Xch::bs; }

// new kinematics of body (level > 0) from its parent's: shared by the initial sweep and the merged forward sweep
PHC_HD void aba_kinematics_from_parent(AbaLane& L, Q4 Qp, V3 pp, V3 wp, V3 vp) {
    const V3 r = quat_rotate(Qp, L.r_local);
    L.p = pp + r;
    L.Q = quat_mul16(Qp, L.q);   // (unit: the root's and every joint's quaternion are normalised where they are integrated / loaded)
    V3 wJw = quat_rotate(L.Q, L.wj);
    L.w = wp + wJw;
    L.v = vp + cross(wp, r);
}
// velocity-product accelerations of the body's joint, c_w = w_p x w_J and c_a = w_p x (w_p x r), from the parent's angular velocity as the
// last kinematics sweep left it in the parent's slot.  Every body at once, before aba_body_init: nothing inside the kinematics
// level-step needs them, and an instruction there is issued once per tree level.  (Root: zero, set by aba_fk_level.)
// With a re-rooted solver tree the "parent" is the SOLVER parent and positions are the bodies' solver reference points (the anchor of
// the joint towards the solver parent: a material point of both bodies, so the rigid-body relations keep their form).
PHC_HD void aba_velocity_products(AbaLane& L, const phc_model_t& m, int j, const Xch& x, bool reroot) {
    if (L.slevel <= 0) { L.rw = L.cw = L.ca = v3(0.f, 0.f, 0.f); return; }
    constexpr int es = Xch::es;
    const float* ps = xslot(x, L.sparent);
    const V3 pp = v3(ps[10 * es], ps[11 * es], ps[12 * es]), wp = v3(ps[13 * es], ps[14 * es], ps[15 * es]);
    L.rw = L.p - pp;
    if (reroot) {   // reference points instead of origins (zero offsets for every body that is not reversed)
        const float* f = model_body(m, j);
        const float* fp = model_body(m, L.sparent);
        L.rw = L.rw + quat_rotate(L.Q, v3(f[44], f[45], f[46]))
                    - quat_rotate(q4(ps[6 * es], ps[7 * es], ps[8 * es], ps[9 * es]), v3(fp[44], fp[45], fp[46]));
    }
    L.cw = cross(wp, L.w - wp);
    L.ca = cross(wp, cross(wp, L.rw));
}
PHC_HD void aba_write_kin(const AbaLane& L, float* s, int es, int o) {
    s[(o + 0) * es] = L.Q.x; s[(o + 1) * es] = L.Q.y; s[(o + 2) * es] = L.Q.z; s[(o + 3) * es] = L.Q.w;
    s[(o + 4) * es] = L.p.x; s[(o + 5) * es] = L.p.y; s[(o + 6) * es] = L.p.z;
    s[(o + 7) * es] = L.w.x; s[(o + 8) * es] = L.w.y; s[(o + 9) * es] = L.w.z;
    s[(o + 10) * es] = L.v.x; s[(o + 11) * es] = L.v.y; s[(o + 12) * es] = L.v.z;
}

// ---- initial kinematics sweep (once per launch), one tree level.  Slot layout: [6..19) = Q(4) p(3) w(3) v(3). ----
PHC_HD void aba_fk_level(AbaLane& L, int level, int j, const Xch& x) {
    if (L.level != level) return;
    constexpr int es = Xch::es;
    if (level == 0) {
        L.Q = L.q;   // (p, w, v: the root's state itself)
    } else {
        const float* ps = xslot(x, L.parent);
        aba_kinematics_from_parent(L, q4(ps[6 * es], ps[7 * es], ps[8 * es], ps[9 * es]), v3(ps[10 * es], ps[11 * es], ps[12 * es]),
                                   v3(ps[13 * es], ps[14 * es], ps[15 * es]), v3(ps[16 * es], ps[17 * es], ps[18 * es]));
    }
    aba_write_kin(L, xslot(x, j), es, 6);
}

// ---- kinematics sweep by POINTER JUMPING (the sweep of every sub-step; the launch's first one stays level by level) ----
// Level by level the sweep is max_level + 1 dependent level-steps of ~70 instructions (9 for SMPL).  Here every body starts from its
// transform relative to its parent -- pose (q, r_local), relative angular velocity q w_J, relative velocity of its origin 0 -- and in
// step k composes it with its anchor's (anchor = parent at first); the anchor's anchor becomes the new anchor, so the distance covered
// doubles per step and ceil(log2(depth + 1)) steps (4 for SMPL) with every lane busy leave the world kinematics in L.Q / p / w / v and
// in slot [6..19), exactly where aba_fk_level leaves them.  Composition of B (relative to frame a) after A (frame a relative to X):
//   Q = Qa Qb,  p = pa + Qa pb,  w = wa + Qa wb,  v = va + wa x (Qa pb) + Qa vb.
PHC_HD int model_jump_steps(const phc_model_t& m) { return model_tab(m, 11, 4); }
PHC_HD void aba_fk_jump_begin(AbaLane& L, int j, const Xch& x) {
    if (L.level < 0) return;
    if (L.level == 0) { L.Q = L.q; }   // (p, w, v: the root's state itself, already world)
    else { L.Q = L.q; L.p = L.r_local; L.w = quat_rotate(L.q, L.wj); L.v = v3(0.f, 0.f, 0.f); }
    aba_write_kin(L, xslot(x, j), Xch::es, 6);
}
// one step: read the anchor's transform as the previous step left it and compose (the caller writes the result back after a barrier)
PHC_HD void aba_fk_jump_step(AbaLane& L, int k, const Xch& x) {
    if (L.level < 0) return;
    const int a = (L.anchors >> (8 * k)) & 0xff;
    if (a == 0xff) return;
    constexpr int es = Xch::es;
    const float* s = xslot(x, a);
    const Q4 Qa = q4(s[6 * es], s[7 * es], s[8 * es], s[9 * es]);
    const V3 pa = v3(s[10 * es], s[11 * es], s[12 * es]), wa = v3(s[13 * es], s[14 * es], s[15 * es]), va = v3(s[16 * es], s[17 * es], s[18 * es]);
    const M3 Ra = quat_to_mat(Qa);
    const V3 r = mat_mul(Ra, L.p);
    L.v = va + cross(wa, r) + mat_mul(Ra, L.v);
    L.w = wa + mat_mul(Ra, L.w);
    L.p = pa + r;
    L.Q = quat_mul16(Qa, L.Q);
}

// one contact point's contribution to the body's articulated quantities: explicit force F0 at `arm` (relative to the solver reference point) and the
// implicit impedance C = diag(c1, c1, c3) / dt of the point:  I^A += J^T (dt C) J,  p^A -= J^T F0  (J = [-[arm]x  1])
PHC_HD void aba_add_point_contact(AbaLane& L, V3 arm, V3 F0, float c1, float c3) {
    L.pA.n -= cross(arm, F0);
    L.pA.f -= F0;
    const float ax = arm.x, ay = arm.y, az = arm.z;
    // A += [a]x C [a]x^T
    L.IA.A.xx += az * az * c1 + ay * ay * c3;
    L.IA.A.yy += az * az * c1 + ax * ax * c3;
    L.IA.A.zz += (ax * ax + ay * ay) * c1;
    L.IA.A.xy -= ax * ay * c3;
    L.IA.A.xz -= ax * az * c1;
    L.IA.A.yz -= ay * az * c1;
    // B += [a]x C   ([a]x = [[0,-az,ay],[az,0,-ax],[-ay,ax,0]], columns scaled by (c1,c1,c3))
    L.IA.B[1] += -az * c1; L.IA.B[2] += ay * c3;
    L.IA.B[3] += az * c1;  L.IA.B[5] += -ax * c3;
    L.IA.B[6] += -ay * c1; L.IA.B[7] += ax * c1;
    // C += C
    L.IA.C.xx += c1; L.IA.C.yy += c1; L.IA.C.zz += c3;
}

// ---- rigid ground contact (phc_sim_params_t.contact_model 1; include/phc_amd.h) ----
// Per contact point a unilateral VELOCITY constraint on the end-of-step normal velocity, u_n(t + dt) >= v_target, carried by the impedance c =
// contact_impedance:  F_n = c (v_target - u_n(t + dt)) >= 0  -- the same linearly-implicit form as the penalty contact (I^A += dt J^T C J), so the
// articulated-body recursion solves ALL contacts of the tree and the joint drives together, exactly, in O(n).  What a single linear solve cannot
// know is the active set (a point only pushes) and where each point sits in its friction cone: they are found by `contact_iterations` passes;
// pass k evaluates them on the velocities the solve of pass k - 1 predicts,
//   u+ = u + dt (a_ref + alpha x r + w x (w x r)),   lambda = c (v_target - u+_n)     (pass 0: u+ = u)
//   a point pushes in pass k iff lambda > 0 -- and it has not been RELEASED: a point that pushed in pass k - 1 and came out with lambda <= 0
//   (the solve needed it to pull) is released for the rest of the sub-step.  Every point therefore changes state at most twice (idle -> pushing ->
//   released): the passes cannot cycle, which plain active-set switching does on statically indeterminate supports (measured: period-2 cycles
//   on a sliding, lying humanoid); a point released wrongly is picked up again by the next sub-step.
//   tangential impedance c_t = min(c, mu lambda / max(|u_t|, |u+_t|)),  F_t = -c_t u_t(t + dt):  a point at rest whose holding force c |u+_t|
//   stays inside the cone keeps the full impedance (it sticks); a slipping point feels mu lambda against its slip (to first order in the change
//   of slip speed over the sub-step) and is caught by the cap when the slip dies -- the friction force can never reverse a slip.
// v_target: depth / dt capped at max_depenetration_velocity for a penetrating point; -gap / dt (the point may close its gap, no more) for a
// speculative point within contact_offset above the plane; -restitution u_n above the bounce threshold.
PHC_HD float rigid_point_velocity_target(const phc_sim_params_t& prm, float dt, float depth, float un_now) {
    float vt = depth > 0.f ? fminf(depth / dt, prm.max_depenetration_velocity) : depth / dt;
    if (prm.restitution > 0.f && -un_now > prm.bounce_threshold_velocity) vt = fmaxf(vt, -prm.restitution * un_now);
    return vt;
}
// one candidate point: geometry and velocities.  false: out of range
struct RigidPoint { V3 arm, uc, un, cc; float depth, lam; };
PHC_HD bool rigid_point(const AbaLane& L, const phc_sim_params_t& prm, float dt, const M3& R, V3 so, const float* cpk, bool predicted, RigidPoint* o) {
    V3 arm = mat_mul(R, v3(cpk[0], cpk[1], cpk[2]));
    const float rad = cpk[3];
    o->depth = rad - (L.p.z + arm.z);
    if (o->depth <= -prm.contact_offset) return false;
    arm.z -= rad;
    o->uc = L.v + cross(L.w, arm);
    o->arm = arm - so;
    o->cc = cross(L.w, cross(L.w, o->arm));
    o->un = predicted ? o->uc + (L.acc_v + cross(L.acc_w, o->arm) + o->cc) * dt : o->uc;
    o->lam = prm.contact_impedance * (rigid_point_velocity_target(prm, dt, o->depth, o->uc.z) - o->un.z);
    return true;
}
// tangential impedance of a pushing point (see above)
PHC_HD float rigid_point_ct(const phc_sim_params_t& prm, const RigidPoint& q) {
    const float us = sqrtf(q.uc.x * q.uc.x + q.uc.y * q.uc.y), up = sqrtf(q.un.x * q.un.x + q.un.y * q.un.y);
    return fminf(prm.contact_impedance, prm.friction * q.lam / (fmaxf(us, up) + 1e-9f));
}
PHC_HD void aba_ground_contact_rigid(AbaLane& L, const phc_sim_params_t& prm, float dt, const M3& R, V3 so, const float* f, const float* cp,
                                     int cp_total, int pass) {
    const float c = prm.contact_impedance;
    if (pass == 0) L.c_active = L.c_removed = 0u;
    uint32_t act = 0u;
    const int cp_count = (L.p.z < f[34] + prm.contact_offset) ? cp_total : 0;
    for (int k = 0; k < cp_count; ++k) {
        RigidPoint q;
        if (!rigid_point(L, prm, dt, R, so, cp + 4 * k, pass > 0, &q)) continue;   // out of range
        const uint32_t bit = k < 32 ? 1u << k : 0u;
        if (q.lam <= 0.f) { L.c_removed |= L.c_active & bit; continue; }           // idle -- or released: it pushed in the last pass and would have to pull
        if (L.c_removed & bit) continue;
        act |= bit;
        const float vt = rigid_point_velocity_target(prm, dt, q.depth, q.uc.z);
        const float ct = rigid_point_ct(prm, q);
        const V3 F0 = v3(-ct * q.uc.x - dt * ct * q.cc.x, -ct * q.uc.y - dt * ct * q.cc.y, c * (vt - q.uc.z) - dt * c * q.cc.z);
        aba_add_point_contact(L, q.arm, F0, dt * ct, dt * c);
    }
    L.c_active = act;
}
// The contact law evaluated on the END-of-sub-step velocities the final solve predicts: the net ground
// force on the body (S4) and, about the body origin, its moment (S6).  `R`: the body's rotation, `so`: origin -> solver reference point (world).
PHC_HD void aba_ground_force_rigid(const AbaLane& L, const phc_sim_params_t& prm, float dt, const M3& R, V3 so, const float* f, const float* cp,
                                   int cp_total, V3* Fout, V3* Nout) {
    V3 F = v3(0.f, 0.f, 0.f), N = v3(0.f, 0.f, 0.f);
    const int cp_count = (L.p.z < f[34] + prm.contact_offset) ? cp_total : 0;
    for (int k = 0; k < cp_count; ++k) {
        RigidPoint q;
        if (!rigid_point(L, prm, dt, R, so, cp + 4 * k, true, &q) || q.lam <= 0.f) continue;
        if (k < 32 && !((L.c_active >> k) & 1u)) continue;   // only the points the final solve let push
        const float ct = rigid_point_ct(prm, q);
        const V3 Fk = v3(-ct * q.un.x, -ct * q.un.y, q.lam);
        F += Fk;
        N += cross(q.arm + so, Fk);
    }
    *Fout = F; *Nout = N;
}

// ---- per-body initialisation of I^A, p^A and of the joint drive (no communication) ----
// `new_sim_call`: first sub-step of a gym.simulate call -- the explicit `pd` torque is recomputed there (humanoid.py:1608-1616).
// `f`: the body's PHC_BODY_FLOATS constants -- straight from the model (L2) or a register copy the caller made once per launch;
// `cp_start / cp_total`: the body's slice of the contact-point table.
// `RIGID` / `pass`: contact_model 1 (include/phc_amd.h) -- the sub-step is solved contact_iterations times; pass 0 decides active set and friction
// cone on the current velocities, pass k > 0 on the end-of-step velocities the previous solve predicts (L.acc_w / L.acc_v); the joint drive is
// formed in pass 0 only (it does not depend on the contact forces).
// `lag` (phc_sim_params_t.inertia_lag, penalty contact only): a sub-step that keeps the articulated inertias I^A and the joint-space inverses D^-1 of the
// previous sub-step (aba_backward_level's bias-only form).  Only p^A and the joint drive are formed here then: bias force and gravity at the current
// state, the explicit part F0 of the contact law for the points whose impedance the kept I^A holds (L.c_touch), body-body forces, drive torque.
template <int JT, bool RIGID>
PHC_HD void aba_body_init(AbaLane& L, const phc_model_t& m, const phc_sim_params_t& prm, float dt, int j, bool new_sim_call,
                          const float* f, int cp_start, int cp_total, bool reroot, int pass, bool lag = false) {
    const float mass = f[3];
    // every spatial quantity of the body is taken about its solver reference point o = p + R off (the origin unless the body is reversed)
    const SolverRef sr = model_solver_ref(f, reroot);
    M3 R = quat_to_mat(L.Q);
    Sym3 Io = rot_sym(R, sr.Io);
    V3 mc = mat_mul(R, sr.mc);
    const V3 so = mat_mul(R, sr.off);
    // rigid-body inertia about the reference point: [[Io, [mc]x], [[mc]x^T, m 1]]
    if (!lag) {
        L.IA.A = Io;
        L.IA.B[0] = 0.f;   L.IA.B[1] = -mc.z; L.IA.B[2] = mc.y;
        L.IA.B[3] = mc.z;  L.IA.B[4] = 0.f;   L.IA.B[5] = -mc.x;
        L.IA.B[6] = -mc.y; L.IA.B[7] = mc.x;  L.IA.B[8] = 0.f;
        L.IA.C.xx = L.IA.C.yy = L.IA.C.zz = mass; L.IA.C.xy = L.IA.C.xz = L.IA.C.yz = 0.f;
    }
    // bias force + gravity (external forces enter p^A with a minus sign)
    V3 g = v3(0.f, 0.f, prm.gravity_z);
    L.pA.n = cross(L.w, sym_mul(Io, L.w)) - cross(mc, g);
    L.pA.f = cross(L.w, cross(L.w, mc)) - g * mass;
    // ground contact: plane z = 0, normal +z
    L.fcontact = v3(0.f, 0.f, 0.f);
    const float* cp = m.floats + PHC_MAX_BODIES * PHC_BODY_FLOATS + cp_start * 4;
    if (RIGID) {
        aba_ground_contact_rigid(L, prm, dt, R, so, f, cp, cp_total, pass);
    } else {
    const float cn = prm.contact_stiffness * dt + prm.contact_damping;
    // broad phase: f[34] bounds |contact point| + radius, so above that height nothing of this body reaches the plane
    const int cp_count = (L.p.z < f[34]) ? cp_total : 0;
    // touching points first (round 3): the heights of up to 32 of the body's points above the plane are formed branch-free with their table
    // loads in flight together (the z row of R only: 4 instructions per point); the loop below then visits the SET BITS -- one iteration per
    // touching point of the busiest lane, each with its point's record already requested -- instead of one dependent table load and one
    // depth test per point (the 8-corner feet made every sub-step walk 8 iterations)
    CpMask touching = 0;
    const int cp_fast = cp_count < PHC_CP_BITS ? cp_count : PHC_CP_BITS;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 8
#endif
    for (int k = 0; k < cp_fast; ++k) {
        const float az = R.m[6] * cp[4 * k] + R.m[7] * cp[4 * k + 1] + R.m[8] * cp[4 * k + 2];
        if (cp[4 * k + 3] - (L.p.z + az) > 0.f) touching |= (CpMask)1 << k;
    }
    if (lag) touching &= L.c_touch;   // (a point that arrives between two fresh sub-steps joins at the next one: its impedance is not in the kept I^A)
    // one iteration per touching point of the busiest lane; the NEXT touching point's record is requested before the current one is worked on
    // (round 4: the table sits in L2 -- ~200 cycles per dependent load -- and the set-bit order is only known at run time)
    struct Cp4 { float x, y, z, r; };
    auto cp_load = [&](int k) { Cp4 c; c.x = cp[4 * k]; c.y = cp[4 * k + 1]; c.z = cp[4 * k + 2]; c.r = cp[4 * k + 3]; return c; };
    auto cp_point = [&](const Cp4& c) -> bool {
        V3 arm = mat_mul(R, v3(c.x, c.y, c.z));
        const float rad = c.r;
        const float depth = rad - (L.p.z + arm.z);
        if (depth <= 0.f) return false;
        arm.z -= rad;  // actual contact location relative to the body origin
        const V3 uc = L.v + cross(L.w, arm);
        arm = arm - so;  // ... relative to the reference point, which the moments and the implicit terms refer to
        const float fn0 = prm.contact_stiffness * depth - cn * uc.z;
        if (fn0 <= 0.f) return false;  // separating: non-adhesive
        const float ut = sqrtf(uc.x * uc.x + uc.y * uc.y);
        const float ct = fminf(prm.friction_viscous, prm.friction * fn0 / (ut + 1e-6f));
        const V3 cc = cross(L.w, cross(L.w, arm));
        const V3 F0 = v3(-ct * uc.x - dt * ct * cc.x, -ct * uc.y - dt * ct * cc.y, fn0 - dt * cn * cc.z);
        L.fcontact += F0;
        if (lag) { L.pA.n -= cross(arm, F0); L.pA.f -= F0; }
        else aba_add_point_contact(L, arm, F0, dt * ct, dt * cn);
        return true;
    };
    CpMask pushed = 0;
    if (touching != 0) {
        CpMask rest = touching;
        int k = PHC_CP_CTZ(rest);
        rest &= rest - 1;
        Cp4 cur = cp_load(k);
        while (true) {
            const bool more = rest != 0;
            const int kn = more ? PHC_CP_CTZ(rest) : k;
            rest &= rest - 1;
            const Cp4 nxt = cp_load(kn);     // (in flight while the current point is worked on)
            if (cp_point(cur)) pushed |= (CpMask)1 << k;
            if (!more) break;
            k = kn; cur = nxt;
        }
    }
    if (!lag) {
        for (int k = PHC_CP_BITS; k < cp_count; ++k) cp_point(cp_load(k));   // (bodies with more points than the mask has bits: the rest one by one; a lagged sub-step leaves them out)
        L.c_touch = pushed;
    }
    }
    // body-body contact forces of this sub-step (explicit; zero unless sim_params.self_collision)
    L.fcontact += L.fself;
    L.pA.n -= L.nself - cross(so, L.fself);   // (nself: about the origin)
    L.pA.f -= L.fself;
    if (RIGID && pass > 0) return;   // the drive terms of pass 0 stand (L.tau_w / Dw / diso / aw / dimp / tau_local)
    if (JT == PHC_JT_REVOLUTE) {
        if (L.level > 0) {
            // joint drive (revolute).  control_mode 0 = Isaac Gym's implicit position drive (`isaac_pd`), linearly implicit as
            // for spherical joints; 1 = `pd`: explicit torque clip(kp (target - q) - kd qd, +-limit) recomputed once per
            // simulate call from the state at that instant and held over its sub-steps (humanoid.py:1575-1599,1608-1616).
            const float kp = f[13], kd = f[16], eff = f[22];
            float tau, d;
            if (prm.control_mode == 1) {
                if (new_sim_call) L.tau_hold = fminf(fmaxf(kp * (L.target.x - L.th) - kd * L.thd, -eff), eff);
                tau = L.tau_hold;
                d = 0.f;
            } else if (prm.control_mode == 2) {
                // `pd` with the damper kept continuous: the spring term is sampled once per simulate call (zero-order hold, as
                // the reference's controller does), the -kd*qd term follows the joint rate implicitly.  A held damper torque
                // on H1's 0.45 kg foot (kd 5 N m s, I 0.005 kg m^2, dt 5 ms: dt kd / I = 4.9 > 2) is an unstable explicit
                // integrator whenever the foot is unloaded; saturated drives are held (constant +-limit), like mode 1.
                if (new_sim_call) {
                    const float sp = kp * (L.target.x - L.th);
                    const float t0 = sp - kd * L.thd;
                    L.tau_hold = (fabsf(t0) >= eff) ? (t0 > 0.f ? eff : -eff) : sp;
                    L.target.y = (fabsf(t0) >= eff) ? 1.f : 0.f;  // saturated flag (target.y is unused by revolute joints)
                }
                if (L.target.y != 0.f) { tau = L.tau_hold; d = 0.f; }
                else { tau = L.tau_hold - kd * L.thd; d = dt * kd; }
            } else {
                tau = fminf(fmaxf(kp * (L.target.x - L.th), -eff), eff) - (kd + dt * kp) * L.thd;
                d = dt * kd + dt * dt * kp;
            }
            // joint limit (URDF <limit lower upper>, enforced by PhysX): implicit penalty spring-damper outside [lo, hi]
            if (prm.limit_stiffness > 0.f) {
                const float lo = f[32], hi = f[33];
                const float e = L.th < lo ? lo - L.th : (L.th > hi ? hi - L.th : 0.f);
                // A lagged sub-step (`inertia_lag`) eliminates with the D^-1 of the last fresh sub-step: a limit that was not engaged THEN has no implicit share in it, and
                // its damper, applied to the current rate without that share, is an explicit integrator -- dt c / I = 5e4 on G1's 14-gram finger links, which reached
                // 100 rad/s under target noise (round 6; `tests/test_stepper_options.py::test_inertia_lag_is_stable_under_target_noise`).  Such a joint gets the spring
                // only until the next fresh sub-step (<= one sub-step later); target.z (unused by revolute joints) remembers what the kept D^-1 contains.
                if (e != 0.f) {
                    if (!lag || L.target.z != 0.f) {
                        tau += prm.limit_stiffness * e - (prm.limit_damping + dt * prm.limit_stiffness) * L.thd;
                        d += dt * prm.limit_damping + dt * dt * prm.limit_stiffness;
                    } else {
                        tau += prm.limit_stiffness * e;
                    }
                }
                if (!lag) L.target.z = e != 0.f ? 1.f : 0.f;
            }
            L.tau_local = L.axis * tau;
            L.dimp = v3(d, 0.f, 0.f);
            L.aw = mat_mul(R, L.axis);
            L.tau_w = L.aw * tau;
        }
        return;
    }
    // joint drive (spherical): geodesic error in the child frame
    if (L.level > 0) {
        Q4 qt = quat_from_rotvec(L.target);
        V3 err = quat_to_rotvec(quat_mul16(quat_conjugate(L.q), qt));
        // effort limit (gear=500): the SPRING term saturates; the damping term stays fully implicit, so a saturated
        // drive can never inject energy (a hard clamp of the total would turn the drive into a constant torque on a
        // 0.02 kg m^2 armature -> 1e4 rad/s^2)
        const V3 kp = v3(f[13], f[14], f[15]), kd = v3(f[16], f[17], f[18]), eff = v3(f[22], f[23], f[24]);
        V3 sp = v3(fminf(fmaxf(kp.x * err.x, -eff.x), eff.x), fminf(fmaxf(kp.y * err.y, -eff.y), eff.y), fminf(fmaxf(kp.z * err.z, -eff.z), eff.z));
        V3 tau = v3(sp.x - (kd.x + dt * kp.x) * L.wj.x, sp.y - (kd.y + dt * kp.y) * L.wj.y, sp.z - (kd.z + dt * kp.z) * L.wj.z);
        V3 d = v3(dt * kd.x + dt * dt * kp.x, dt * kd.y + dt * dt * kp.y, dt * kd.z + dt * dt * kp.z);
        L.tau_local = tau;
        L.dimp = d;
        const V3 dd = d + v3(f[19], f[20], f[21]);   // + armature (read here: not pinned in registers across the sweeps)
        if (dd.x == dd.y && dd.y == dd.z) { L.diso = dd.x; }
        else { L.diso = -1.f; if (!lag) L.Dw = rot_diag(R, dd); }
        L.tau_w = mat_mul(R, tau);
    }
}

template <int JT, bool RIGID>
PHC_HD void aba_body_init(AbaLane& L, const phc_model_t& m, const phc_sim_params_t& prm, float dt, int j, bool new_sim_call, bool reroot, int pass, bool lag) {
    aba_body_init<JT, RIGID>(L, m, prm, dt, j, new_sim_call, model_body(m, j), model_tab(m, 8, j), model_tab(m, 9, j), reroot, pass, lag);
}
// rigid contact model: S4 net ground force of the body from the final solve of the sub-step (+ the body-body forces, as the penalty model
// publishes), and S6: the force sensors read the same wrench (about the body origin, in the body frame)
PHC_HD void aba_publish_contact_rigid(AbaLane& L, const phc_model_t& m, const phc_sim_params_t& prm, const phc_sim_state_t& s, float dt,
                                      int64_t env, int j, bool reroot) {
    const float* f = model_body(m, j);
    const M3 R = quat_to_mat(L.Q);
    const V3 so = mat_mul(R, model_solver_ref(f, reroot).off);
    V3 F, N;
    aba_ground_force_rigid(L, prm, dt, R, so, f, m.floats + PHC_MAX_BODIES * PHC_BODY_FLOATS + model_tab(m, 8, j) * 4, model_tab(m, 9, j), &F, &N);
    L.fcontact = F + L.fself;
    if (s.force_sensor != nullptr)
        for (int k = 0; k < prm.num_force_sensors; ++k)
            if (prm.force_sensor_body[k] == j) {
                const V3 Fl = mat_tmul(R, F), Nl = mat_tmul(R, N);
                float* o = s.force_sensor + (env * prm.num_force_sensors + k) * 6;
                o[0] = Fl.x; o[1] = Fl.y; o[2] = Fl.z; o[3] = Nl.x; o[4] = Nl.y; o[5] = Nl.z;
            }
}

// ---- body-body contact (SURVEY f-1; the reference runs with robot.has_self_collision: True, humanoid.py:1205-1226) ----
// Every body carries one collision capsule (model.py _geom_capsule).  Penalty contact between the capsules of bodies that may
// collide (ArticulationModel.collision_allow_masks: not joint neighbours, no common Isaac Gym filter bit), EXPLICIT in time -- it
// couples bodies across the tree, which the O(n) recursion cannot absorb implicitly -- hence soft and stable by construction:
//     k = alpha * mu / dt^2,  c = 2 zeta sqrt(k mu),  mu = m_i m_j / (m_i + m_j)   (alpha = self_stiffness_scale <= 1: explicit
// Euler on a spring of the pair's reduced mass is stable for k < 4 mu / dt^2).  Each lane evaluates the pairs of its own body;
// lane j evaluates the mirrored pair with the same arithmetic, so forces come out equal and opposite.
PHC_HD void seg_seg_closest(V3 p1, V3 q1, V3 p2, V3 q2, V3* c1, V3* c2) {
    const V3 d1 = q1 - p1, d2 = q2 - p2, r = p1 - p2;
    const float a = dot(d1, d1), e = dot(d2, d2), f = dot(d2, r);
    const float EPS = 1e-10f;
    float s = 0.f, t = 0.f;
    if (a <= EPS && e <= EPS) {
    } else if (a <= EPS) {
        t = fminf(fmaxf(f / e, 0.f), 1.f);
    } else {
        const float c = dot(d1, r);
        if (e <= EPS) {
            s = fminf(fmaxf(-c / a, 0.f), 1.f);
        } else {
            const float b = dot(d1, d2), denom = a * e - b * b;
            s = denom > 1e-7f * a * e ? fminf(fmaxf((b * f - c * e) / denom, 0.f), 1.f) : 0.f;
            t = (b * s + f) / e;
            if (t < 0.f) { t = 0.f; s = fminf(fmaxf(-c / a, 0.f), 1.f); }
            else if (t > 1.f) { t = 1.f; s = fminf(fmaxf((b - c) / a, 0.f), 1.f); }
        }
    }
    *c1 = p1 + d1 * s;
    *c2 = p2 + d2 * t;
}
// Pair-parallel evaluation: the candidate pairs (model ints: count, i | k << 8, ...) are dealt round-robin to the lanes of the
// env's group, so every lane tests different pairs (no redundant work, no per-body imbalance).  A contact adds +-F and the moments
// to the two bodies' accumulators with INTEGER (fixed-point) LDS atomics: integer addition is associative, so the sums do not
// depend on the order lanes arrive in -- bit-reproducible and exactly antisymmetric.
// Per-SHAPE record in the exchange area (PHC_CAP_STRIDE words): [0..4) bounding sphere (centre, radius) | [4..8) a, radius |
// [8..12) b, owner's mass | [12..18) int32 accumulators F(3) N(3) of BODY s (shape s < NB is body s's primary capsule) | [18] owner body.
// Round 4: a body may carry more than one collision capsule (the second half of a flat box -- SMPL toes / hands --, further geoms of a multi-geom
// link -- G1 torso / forearm); the extra capsules are shapes NB .. NB + NX - 1, published by the idle lanes behind the bodies (model.py pack():
// NX in the misc table, records a[3] b[3] radius owner after the contact points), and the candidate list pairs SHAPES.
PHC_HD int model_num_extra_shapes(const phc_model_t& m) { return model_tab(m, 11, 5); }
PHC_HD const float* model_extra_shape(const phc_model_t& m, int e) { return m.floats + PHC_MAX_BODIES * PHC_BODY_FLOATS + m.ints[3] * 4 + e * 8; }
PHC_HD int model_num_pairs(const phc_model_t& m) { return m.ints[4 + PHC_NTAB * PHC_MAX_BODIES]; }
PHC_HD int model_pair(const phc_model_t& m, int q) { return m.ints[4 + PHC_NTAB * PHC_MAX_BODIES + 1 + q]; }
PHC_HD void sc_atomic_add(int32_t* p, int32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(p, v);
#else
    *p += v;   // the host emulation walks the pairs sequentially
#endif
}
PHC_HD void cap_write(float* cap, V3 a, V3 b, float radius, float mass, int owner) {
    const V3 mid = (a + b) * 0.5f;
    cap[0] = mid.x; cap[1] = mid.y; cap[2] = mid.z; cap[3] = 0.5f * norm(b - a) + radius;
    cap[4] = a.x; cap[5] = a.y; cap[6] = a.z; cap[7] = radius;
    cap[8] = b.x; cap[9] = b.y; cap[10] = b.z; cap[11] = mass;
    int32_t* acc = reinterpret_cast<int32_t*>(cap + 12);
    for (int k = 0; k < 6; ++k) acc[k] = 0;
    acc[6] = owner;
}
// extra shape e: the lane behind the bodies that publishes it loads its record once per launch
PHC_HD void aba_load_extra_shape(AbaLane& L, const phc_model_t& m, int e) {
    const float* r = model_extra_shape(m, e);
    L.cap_a = v3(r[0], r[1], r[2]); L.cap_b = v3(r[3], r[4], r[5]); L.cap_r = r[6]; L.cap_owner = (int)r[7];
    L.cap_m = model_body(m, L.cap_owner)[3];
}
// shape s (body lanes: s = body; extra lanes: s = NB + e): world capsule from the OWNER's pose as the last kinematics sweep left it in the owner's
// exchange slot -- one code path for body lanes and extra-shape lanes
PHC_HD void aba_publish_shape(const AbaLane& L, int s, const Xch& x, float* caps) {
    constexpr int es = Xch::es;
    const float* k = xslot(x, L.cap_owner);
    const M3 R = quat_to_mat(q4(k[6 * es], k[7 * es], k[8 * es], k[9 * es]));
    const V3 p = v3(k[10 * es], k[11 * es], k[12 * es]);
    cap_write(caps + PHC_CAP_STRIDE * s, p + mat_mul(R, L.cap_a), p + mat_mul(R, L.cap_b), L.cap_r, L.cap_m, L.cap_owner);
}
// one candidate pair (bodies i < k): bounding spheres, then the capsule-capsule test, then the penalty force into both accumulators.
// Needs both bodies' capsules in `caps` and kinematics (p w v at slot floats [10..19)) in the exchange slots.
// Returns true when the pair is FAR: its surfaces are more than PHC_SC_SKIP_MARGIN apart (see aba_collide_pairs).
#define PHC_SC_SKIP_MARGIN 0.12f
// broad phase of one pair: bounding spheres.  -> 0: the spheres overlap (run the narrow phase), 1: apart but within PHC_SC_SKIP_MARGIN, 2: far
PHC_HD int aba_pair_broad(int i, int k, const float* caps) {
    const float* ci = caps + PHC_CAP_STRIDE * i;
    const float* ck = caps + PHC_CAP_STRIDE * k;
    const V3 dm = v3(ci[0] - ck[0], ci[1] - ck[1], ci[2] - ck[2]);
    const float R = ci[3] + ck[3];
    const float d2 = dot(dm, dm);
    if (d2 <= R * R) return 0;
    return d2 > (R + PHC_SC_SKIP_MARGIN) * (R + PHC_SC_SKIP_MARGIN) ? 2 : 1;
}
// narrow phase: the capsule-capsule test, then the penalty force into both accumulators.  Returns true when the surfaces are more than the margin apart.
PHC_HD bool aba_pair_narrow(const phc_sim_params_t& prm, float dt, int i, int k, const Xch& x, float* caps) {
    const float* ci = caps + PHC_CAP_STRIDE * i;
    const float* ck = caps + PHC_CAP_STRIDE * k;
    const V3 a1 = v3(ci[4], ci[5], ci[6]), b1 = v3(ci[8], ci[9], ci[10]), a2 = v3(ck[4], ck[5], ck[6]), b2 = v3(ck[8], ck[9], ck[10]);
    const float r1 = ci[7], m1 = ci[11], r2 = ck[7], m2 = ck[11];
    V3 c1, c2;
    seg_seg_closest(a1, b1, a2, b2, &c1, &c2);
    V3 n = c1 - c2;
    const float dist = norm(n), pen = r1 + r2 - dist;
    if (pen <= 0.f) return pen < -PHC_SC_SKIP_MARGIN;
    n = dist > 1e-6f ? n * (1.0f / dist) : v3(0.f, 0.f, 1.f);
    const V3 cp = c2 + n * (r2 - 0.5f * pen);                // middle of the overlap
    constexpr int es = Xch::es;
    const int oi = reinterpret_cast<const int32_t*>(ci + 12)[6], ok = reinterpret_cast<const int32_t*>(ck + 12)[6];   // owner bodies of the two shapes (cap_write: acc[6])
    const float* si = xslot(x, oi);
    const float* sk = xslot(x, ok);
    const V3 pi = v3(si[10 * es], si[11 * es], si[12 * es]), wi = v3(si[13 * es], si[14 * es], si[15 * es]), vi = v3(si[16 * es], si[17 * es], si[18 * es]);
    const V3 pk = v3(sk[10 * es], sk[11 * es], sk[12 * es]), wk = v3(sk[13 * es], sk[14 * es], sk[15 * es]), vk = v3(sk[16 * es], sk[17 * es], sk[18 * es]);
    const V3 vrel = (vi + cross(wi, cp - pi)) - (vk + cross(wk, cp - pk));
    const float mu = m1 * m2 / (m1 + m2);
    const float kk = prm.self_stiffness_scale * mu / (dt * dt);
    const float cc = 2.0f * prm.self_damping_ratio * sqrtf(kk * mu);
    const float fn = kk * pen - cc * dot(vrel, n);
    if (fn <= 0.f) return false;                             // non-adhesive
    // force on body i (+) and body k (-), quantised once so that both bodies see exactly opposite values
    const int32_t fx = (int32_t)rintf(n.x * fn * PHC_SC_FSCALE), fy = (int32_t)rintf(n.y * fn * PHC_SC_FSCALE), fz = (int32_t)rintf(n.z * fn * PHC_SC_FSCALE);
    const V3 F = v3((float)fx, (float)fy, (float)fz) * (1.0f / PHC_SC_FSCALE);
    const V3 ni = cross(cp - pi, F), nk = cross(cp - pk, F);
    int32_t* ai = reinterpret_cast<int32_t*>(caps + PHC_CAP_STRIDE * oi + 12);
    int32_t* ak = reinterpret_cast<int32_t*>(caps + PHC_CAP_STRIDE * ok + 12);
    sc_atomic_add(ai + 0, fx); sc_atomic_add(ai + 1, fy); sc_atomic_add(ai + 2, fz);
    sc_atomic_add(ak + 0, -fx); sc_atomic_add(ak + 1, -fy); sc_atomic_add(ak + 2, -fz);
    sc_atomic_add(ai + 3, (int32_t)rintf(ni.x * PHC_SC_NSCALE)); sc_atomic_add(ai + 4, (int32_t)rintf(ni.y * PHC_SC_NSCALE)); sc_atomic_add(ai + 5, (int32_t)rintf(ni.z * PHC_SC_NSCALE));
    sc_atomic_add(ak + 3, -(int32_t)rintf(nk.x * PHC_SC_NSCALE)); sc_atomic_add(ak + 4, -(int32_t)rintf(nk.y * PHC_SC_NSCALE)); sc_atomic_add(ak + 5, -(int32_t)rintf(nk.z * PHC_SC_NSCALE));
    return false;
}
// one candidate pair (bodies i < k), both phases (the host emulation walks the pairs with this).  Returns true when the pair is FAR.
PHC_HD bool aba_collide_pair(const phc_sim_params_t& prm, float dt, int i, int k, const Xch& x, float* caps) {
    const int b = aba_pair_broad(i, k, caps);
    return b == 0 ? aba_pair_narrow(prm, dt, i, k, x, caps) : b == 2;
}
// lane `l` of `nl` lanes of the env's group: its share of the candidate pairs, dealt round-robin, kept in LDS as [pair slot t][thread]
// (round 3; 18 x 32 lanes hold the 245 pairs of the SMPL humanoid, 18 x 64 the 589 of G1)
#define PHC_SC_MAX_PER_LANE 18
template <int NP>
PHC_HD void aba_load_pairs(int* pr /*[NP][stride]*/, int stride, const phc_model_t& m, int l, int nl) {
    const int np = model_num_pairs(m);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int t = 0; t < NP; ++t) {
        const int q = t * nl + l;
        pr[t * stride] = q < np ? model_pair(m, q) : -1;
    }
}
// Temporal coherence over the sub-steps of ONE launch (round 2): the first sub-step tests every candidate pair and remembers in `near`
// (bit t = pair t of this lane) the ones whose surfaces are closer than PHC_SC_SKIP_MARGIN = 0.12 m; the remaining sub-steps (3 x 1/120 s)
// only revisit those.  A pair closing faster than ~5 m/s from beyond the margin is picked up one env step late at the latest (the penalty
// contact is soft by construction, k = mu / (4 dt^2)).  The host emulation / fp64 oracle test every pair in every sub-step; they agree with
// this unless such a pair exists.
// A loop over the SET BITS of the lane's candidate mask (round 3): the first version walked all NP pair slots with a branch around each --
// a wavefront then executes slot t whenever ANY of its 64 lanes has a pair there, i.e. nearly all of them although a lane has 0-2 near pairs
// (one wavefront's timeline: 5.5 k of a sub-step's 38 k cycles, profiles/r03_stepper).  Compacted, the wavefront runs as many iterations as
// its busiest lane has candidates.  The pair list lives in LDS so that the loop can index it.
template <int NP>
PHC_HD void aba_collide_pairs(const int* pr, int stride, const phc_sim_params_t& prm, float dt, const Xch& x, float* caps, uint32_t& near, bool refresh) {
    uint32_t cand = near, nr = 0u;
    if (refresh) {   // every valid slot of the lane
        cand = 0u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int t = 0; t < NP; ++t) cand |= (pr[t * stride] >= 0 ? 1u : 0u) << t;
    }
    while (cand != 0u) {
        const int t = __builtin_ctz(cand);
        cand &= cand - 1u;
        const int pq = pr[t * stride];
        const int i = pq & 0xff, k = pq >> 8;
        const int b = aba_pair_broad(i, k, caps);
        const bool far = b == 0 ? aba_pair_narrow(prm, dt, i, k, x, caps) : b == 2;
        if (!far) nr |= 1u << t;
    }
    if (refresh) near = nr;   // (the later sub-steps keep the first sub-step's set)
}
// net body-body contact force / moment of body j from its accumulators
PHC_HD void aba_collect_self(AbaLane& L, int j, const float* caps) {
    const int32_t* acc = reinterpret_cast<const int32_t*>(caps + PHC_CAP_STRIDE * j + 12);
    L.fself = v3((float)acc[0], (float)acc[1], (float)acc[2]) * (1.0f / PHC_SC_FSCALE);
    L.nself = v3((float)acc[3], (float)acc[4], (float)acc[5]) * (1.0f / PHC_SC_NSCALE);
}

// 6x6 congruence T^T I T and T^T p for a pure translation r (child origin - parent origin)
PHC_HD void shift_to_parent(const Inertia6& I, const Force6& p, V3 r, float* out /*27 floats, stride es*/, int es) {
    // Y = [r]x C ; B' = B + Y
    V3 c0 = v3(I.C.xx, I.C.xy, I.C.xz), c1 = v3(I.C.xy, I.C.yy, I.C.yz), c2 = v3(I.C.xz, I.C.yz, I.C.zz);
    V3 y0 = cross(r, c0), y1 = cross(r, c1), y2 = cross(r, c2);  // columns of Y
    float Bp[9];
    Bp[0] = I.B[0] + y0.x; Bp[1] = I.B[1] + y1.x; Bp[2] = I.B[2] + y2.x;
    Bp[3] = I.B[3] + y0.y; Bp[4] = I.B[4] + y1.y; Bp[5] = I.B[5] + y2.y;
    Bp[6] = I.B[6] + y0.z; Bp[7] = I.B[7] + y1.z; Bp[8] = I.B[8] + y2.z;
    // A' = A + X^T + [r]x B'^T  with X = [r]x B^T ;  columns of B^T are rows of B
    V3 b0 = v3(I.B[0], I.B[1], I.B[2]), b1 = v3(I.B[3], I.B[4], I.B[5]), b2 = v3(I.B[6], I.B[7], I.B[8]);
    V3 x0 = cross(r, b0), x1 = cross(r, b1), x2 = cross(r, b2);  // columns of X
    V3 bp0 = v3(Bp[0], Bp[1], Bp[2]), bp1 = v3(Bp[3], Bp[4], Bp[5]), bp2 = v3(Bp[6], Bp[7], Bp[8]);
    V3 z0 = cross(r, bp0), z1 = cross(r, bp1), z2 = cross(r, bp2);  // columns of Z = [r]x B'^T
    // (X^T)_{ij} = X_{ji} = (x_i)_j ; Z_{ij} = (z_j)_i
    out[0 * es] = I.A.xx + x0.x + z0.x;
    out[1 * es] = I.A.xy + x0.y + z1.x;
    out[2 * es] = I.A.xz + x0.z + z2.x;
    out[3 * es] = I.A.yy + x1.y + z1.y;
    out[4 * es] = I.A.yz + x1.z + z2.y;
    out[5 * es] = I.A.zz + x2.z + z2.z;
    for (int k = 0; k < 9; ++k) out[(6 + k) * es] = Bp[k];
    out[15 * es] = I.C.xx; out[16 * es] = I.C.xy; out[17 * es] = I.C.xz; out[18 * es] = I.C.yy; out[19 * es] = I.C.yz; out[20 * es] = I.C.zz;
    V3 n = p.n + cross(r, p.f);
    out[21 * es] = n.x; out[22 * es] = n.y; out[23 * es] = n.z; out[24 * es] = p.f.x; out[25 * es] = p.f.y; out[26 * es] = p.f.z;
}

PHC_HD void accumulate_child(Inertia6& I, Force6& p, const float* s, int es) {
    I.A.xx += s[0 * es]; I.A.xy += s[1 * es]; I.A.xz += s[2 * es]; I.A.yy += s[3 * es]; I.A.yz += s[4 * es]; I.A.zz += s[5 * es];
    for (int k = 0; k < 9; ++k) I.B[k] += s[(6 + k) * es];
    I.C.xx += s[15 * es]; I.C.xy += s[16 * es]; I.C.xz += s[17 * es]; I.C.yy += s[18 * es]; I.C.yz += s[19 * es]; I.C.zz += s[20 * es];
    p.n.x += s[21 * es]; p.n.y += s[22 * es]; p.n.z += s[23 * es]; p.f.x += s[24 * es]; p.f.y += s[25 * es]; p.f.z += s[26 * es];
}

// ---- backward sweep: articulated inertia, one tree level (leaves -> root) ----
// Lanes at `level` first absorb their children's contributions (written at level+1), then, unless
// they are the root, reduce over their own joint and publish T^T I^a T, T^T p^a for their parent.
// `lag` (phc_sim_params_t.inertia_lag): the BIAS-ONLY level-step of a sub-step that keeps I^A and D^-1 of the previous one.  With U = [A; B^T] the first
// three columns of I^A and I^a = I^A - U D^-1 U^T,
//     p^a = p^A + I^a c + U D^-1 u = p^A + I^A c + U D^-1 (u - U^T c),      U^T c = A c_w + B c_a = (I^A c)_top,
// i.e. two 3x3 products with c, one with D^-1 and two with its result: ~75 multiply-adds and a 6-float hand-over instead of the ~350 and 27 floats of
// the full step -- no 3x3 inverse, no projected inertia, no congruence.
template <int JT>
PHC_HD void aba_backward_level(AbaLane& L, int level, int j, const Xch& x, bool lag = false) {
    if (L.slevel != level) return;
    if (lag) {
        constexpr int es = Xch::es;
        for (int k = 0; k < 3; ++k)
            if (k < L.nchild) {
                const float* s = xslot(x, L.child[k]);
                L.pA.n.x += s[21 * es]; L.pA.n.y += s[22 * es]; L.pA.n.z += s[23 * es]; L.pA.f.x += s[24 * es]; L.pA.f.y += s[25 * es]; L.pA.f.z += s[26 * es];
            }
        if (level == 0) return;
        L.u = L.tau_w - L.pA.n;
        const V3 t = sym_mul(L.IA.A, L.cw) + B_mul(L.IA.B, L.ca);
        const V3 sb = Bt_mul(L.IA.B, L.cw) + sym_mul(L.IA.C, L.ca);
        const V3 z = sym_mul(L.Di, L.u - t);
        const V3 f = L.pA.f + sb + Bt_mul(L.IA.B, z);
        const V3 n = L.pA.n + t + sym_mul(L.IA.A, z) + cross(L.rw, f);
        float* o = xslot(x, j);
        o[21 * es] = n.x; o[22 * es] = n.y; o[23 * es] = n.z; o[24 * es] = f.x; o[25 * es] = f.y; o[26 * es] = f.z;
        return;
    }
    for (int k = 0; k < 3; ++k)
        if (k < L.nchild) accumulate_child(L.IA, L.pA, xslot(x, L.child[k]), Xch::es);
    if (level == 0) return;
    if (JT == PHC_JT_REVOLUTE) {
        // one free axis a (world): D = a^T A a + d is a scalar and D^-1 acts as the rank-1 matrix a a^T / D; with it the
        // spherical-joint expressions below hold unchanged (U D^-1 U^T removes exactly the a-component)
        const V3 a = L.aw;
        const float ids = 1.0f / (dot(a, sym_mul(L.IA.A, a)) + L.dimp.x + L.arm.x);
        L.Di.xx = a.x * a.x * ids; L.Di.xy = a.x * a.y * ids; L.Di.xz = a.x * a.z * ids;
        L.Di.yy = a.y * a.y * ids; L.Di.yz = a.y * a.z * ids; L.Di.zz = a.z * a.z * ids;
    } else if (L.diso < 0.f) {
        Sym3 D = L.Dw;
        D.xx += L.IA.A.xx; D.xy += L.IA.A.xy; D.xz += L.IA.A.xz; D.yy += L.IA.A.yy; D.yz += L.IA.A.yz; D.zz += L.IA.A.zz;
        L.Di = sym_inv(D);
    }
    L.u = L.tau_w - L.pA.n;
    const Sym3& A = L.IA.A;
    const float* B = L.IA.B;
    V3 a0 = v3(A.xx, A.xy, A.xz), a1 = v3(A.xy, A.yy, A.yz), a2 = v3(A.xz, A.yz, A.zz);  // columns (=rows) of A
    V3 bc0 = v3(B[0], B[3], B[6]), bc1 = v3(B[1], B[4], B[7]), bc2 = v3(B[2], B[5], B[8]);  // columns of B
    Inertia6 Ia;
    if (JT != PHC_JT_REVOLUTE && L.diso >= 0.f) {
        // D = A + d 1 with a scalar d: A = D - d 1, so A P = 1 - d P (P = D^-1) and
        //   A_a = A - A P A = d (A P) (symmetric) ;  B_a = B - A P B = d P B ;  C_a = C - B^T P B
        // -- 39 multiply-adds fewer per level-step than the general expressions below
        const float d = L.diso;
        Sym3 D = A;
        D.xx += d; D.yy += d; D.zz += d;
        L.Di = sym_inv(D);
        V3 p0 = v3(L.Di.xx, L.Di.xy, L.Di.xz), p1 = v3(L.Di.xy, L.Di.yy, L.Di.yz), p2 = v3(L.Di.xz, L.Di.yz, L.Di.zz);
        Ia.A.xx = d * dot(a0, p0); Ia.A.xy = d * dot(a0, p1); Ia.A.xz = d * dot(a0, p2);
        Ia.A.yy = d * dot(a1, p1); Ia.A.yz = d * dot(a1, p2); Ia.A.zz = d * dot(a2, p2);
        V3 h0 = sym_mul(L.Di, bc0), h1 = sym_mul(L.Di, bc1), h2 = sym_mul(L.Di, bc2);      // columns of H = P B
        Ia.B[0] = d * h0.x; Ia.B[1] = d * h1.x; Ia.B[2] = d * h2.x;
        Ia.B[3] = d * h0.y; Ia.B[4] = d * h1.y; Ia.B[5] = d * h2.y;
        Ia.B[6] = d * h0.z; Ia.B[7] = d * h1.z; Ia.B[8] = d * h2.z;
        Ia.C.xx = L.IA.C.xx - dot(bc0, h0); Ia.C.xy = L.IA.C.xy - dot(bc0, h1); Ia.C.xz = L.IA.C.xz - dot(bc0, h2);
        Ia.C.yy = L.IA.C.yy - dot(bc1, h1); Ia.C.yz = L.IA.C.yz - dot(bc1, h2); Ia.C.zz = L.IA.C.zz - dot(bc2, h2);
    } else {
        // G = Di A (3x3), H = Di B (3x3)
        V3 g0 = sym_mul(L.Di, a0), g1 = sym_mul(L.Di, a1), g2 = sym_mul(L.Di, a2);         // columns of G
        V3 h0 = sym_mul(L.Di, bc0), h1 = sym_mul(L.Di, bc1), h2 = sym_mul(L.Di, bc2);      // columns of H
        // A_a = A - A G  (symmetric)
        Ia.A.xx = A.xx - dot(a0, g0); Ia.A.xy = A.xy - dot(a0, g1); Ia.A.xz = A.xz - dot(a0, g2);
        Ia.A.yy = A.yy - dot(a1, g1); Ia.A.yz = A.yz - dot(a1, g2); Ia.A.zz = A.zz - dot(a2, g2);
        // B_a = B - A H : (A H)_{ij} = row_i(A) . h_j
        Ia.B[0] = B[0] - dot(a0, h0); Ia.B[1] = B[1] - dot(a0, h1); Ia.B[2] = B[2] - dot(a0, h2);
        Ia.B[3] = B[3] - dot(a1, h0); Ia.B[4] = B[4] - dot(a1, h1); Ia.B[5] = B[5] - dot(a1, h2);
        Ia.B[6] = B[6] - dot(a2, h0); Ia.B[7] = B[7] - dot(a2, h1); Ia.B[8] = B[8] - dot(a2, h2);
        // C_a = C - B^T H : (B^T H)_{ij} = col_i(B) . h_j
        Ia.C.xx = L.IA.C.xx - dot(bc0, h0); Ia.C.xy = L.IA.C.xy - dot(bc0, h1); Ia.C.xz = L.IA.C.xz - dot(bc0, h2);
        Ia.C.yy = L.IA.C.yy - dot(bc1, h1); Ia.C.yz = L.IA.C.yz - dot(bc1, h2); Ia.C.zz = L.IA.C.zz - dot(bc2, h2);
    }
    // p_a = p^A + I^a c + U Di u,  U = [A; B^T]
    V3 du = sym_mul(L.Di, L.u);
    Force6 pa;
    pa.n = L.pA.n + sym_mul(Ia.A, L.cw) + B_mul(Ia.B, L.ca) + sym_mul(A, du);
    pa.f = L.pA.f + Bt_mul(Ia.B, L.cw) + sym_mul(Ia.C, L.ca) + Bt_mul(B, du);
    shift_to_parent(Ia, pa, L.rw, xslot(x, j), Xch::es);
}

// ---- forward sweep, part 1: spatial accelerations, one tree level (root -> leaves).  Slot [0..6): alpha(3) a(3) of the body.
// Only what depends on the parent's acceleration is in the level-step (a level-step is issued once per tree level whatever the
// number of lanes at that level: every instruction here costs max_level + 1 issues per sub-step).  The body's own result stays
// in registers for part 2: the root's alpha -> L.u, a -> L.ca; a joint's world-frame angular acceleration beta -> L.u.
template <int JT>
PHC_HD void aba_accel_level(AbaLane& L, int level, int j, const Xch& x) {
    if (L.slevel != level) return;
    constexpr int es = Xch::es;
    V3 alpha, a;
    if (level == 0) {
        // free root: [A B; B^T C] [alpha; a] = -[n; f]  by block elimination on C
        Sym3 Ci = sym_inv(L.IA.C);
        const float* B = L.IA.B;
        // W = B Ci (3x3), S = A - W B^T
        V3 ci0 = v3(Ci.xx, Ci.xy, Ci.xz), ci1 = v3(Ci.xy, Ci.yy, Ci.yz), ci2 = v3(Ci.xz, Ci.yz, Ci.zz);
        V3 br0 = v3(B[0], B[1], B[2]), br1 = v3(B[3], B[4], B[5]), br2 = v3(B[6], B[7], B[8]);
        V3 w0 = v3(dot(br0, ci0), dot(br0, ci1), dot(br0, ci2));  // rows of W
        V3 w1 = v3(dot(br1, ci0), dot(br1, ci1), dot(br1, ci2));
        V3 w2 = v3(dot(br2, ci0), dot(br2, ci1), dot(br2, ci2));
        Sym3 S;
        S.xx = L.IA.A.xx - dot(w0, br0); S.xy = L.IA.A.xy - dot(w0, br1); S.xz = L.IA.A.xz - dot(w0, br2);
        S.yy = L.IA.A.yy - dot(w1, br1); S.yz = L.IA.A.yz - dot(w1, br2); S.zz = L.IA.A.zz - dot(w2, br2);
        V3 rhs = v3(-L.pA.n.x + dot(w0, L.pA.f), -L.pA.n.y + dot(w1, L.pA.f), -L.pA.n.z + dot(w2, L.pA.f));
        alpha = sym_mul(sym_inv(S), rhs);
        a = -sym_mul(Ci, L.pA.f + Bt_mul(B, alpha));
        L.u = alpha; L.ca = a;
        L.acc_w = alpha; L.acc_v = a;
    } else {
        const float* ps = xslot(x, L.sparent);
        V3 alp = v3(ps[0 * es], ps[1 * es], ps[2 * es]), ap = v3(ps[3 * es], ps[4 * es], ps[5 * es]);
        V3 al1 = alp + L.cw;
        V3 a1 = ap + cross(alp, L.rw) + L.ca;
        V3 beta = sym_mul(L.Di, L.u - sym_mul(L.IA.A, al1) - B_mul(L.IA.B, a1));  // world-frame joint angular acceleration
        alpha = al1 + beta;
        a = a1;
        L.u = beta;
        L.acc_w = alpha; L.acc_v = a;
    }
    float* s = xslot(x, j);
    s[0 * es] = alpha.x; s[1 * es] = alpha.y; s[2 * es] = alpha.z; s[3 * es] = a.x; s[4 * es] = a.y; s[5 * es] = a.z;
    s[19 * es] = L.u.x; s[20 * es] = L.u.y; s[21 * es] = L.u.z;   // (re-rooted trees: aba_accel_finish hands it to the joint's owner)
}

// ---- re-rooted solver tree only (model.py solver_tree()), every body at once ----
// Before the backward sweep: a reversed body's solver joint is its solver parent's joint.  Every body publishes the world-frame drive
// terms of its OWN joint (slot [19..28): tau_w, D_w); a reversed body then takes its solver parent's -- the torque with the opposite
// sign, the joint-space matrix as it is (same physical joint, seen from its other side).
PHC_HD void aba_publish_drive(const AbaLane& L, int j, const Xch& x) {
    constexpr int es = Xch::es;
    float* s = xslot(x, j);
    s[19 * es] = L.tau_w.x; s[20 * es] = L.tau_w.y; s[21 * es] = L.tau_w.z;
    const bool iso = L.diso >= 0.f;
    s[22 * es] = iso ? L.diso : L.Dw.xx; s[23 * es] = iso ? 0.f : L.Dw.xy; s[24 * es] = iso ? 0.f : L.Dw.xz;
    s[25 * es] = iso ? L.diso : L.Dw.yy; s[26 * es] = iso ? 0.f : L.Dw.yz; s[27 * es] = iso ? L.diso : L.Dw.zz;
}
PHC_HD void aba_fetch_drive(AbaLane& L, int j, const Xch& x) {
    if (L.jsrc < 0 || L.jsrc == j) return;
    constexpr int es = Xch::es;
    const float* s = xslot(x, L.jsrc);
    L.tau_w = v3(-s[19 * es], -s[20 * es], -s[21 * es]);
    L.Dw.xx = s[22 * es]; L.Dw.xy = s[23 * es]; L.Dw.xz = s[24 * es]; L.Dw.yy = s[25 * es]; L.Dw.yz = s[26 * es]; L.Dw.zz = s[27 * es];
    const bool iso = L.Dw.xy == 0.f && L.Dw.xz == 0.f && L.Dw.yz == 0.f && L.Dw.xx == L.Dw.yy && L.Dw.yy == L.Dw.zz;
    L.diso = iso ? L.Dw.xx : -1.f;
}
// After the acceleration sweep: what aba_integrate_joint reads from L.u / L.ca.  A joint solved by a reversed body gets that body's
// relative acceleration with the opposite sign (beta_c = alpha_c - alpha_p - w_p x w_c = -(alpha_p - alpha_c - w_c x w_p)); the
// simulator's root, when it is not the solver base, gets its own alpha and the acceleration of its ORIGIN from that of its reference point.
PHC_HD void aba_accel_finish(AbaLane& L, const phc_model_t& m, int j, const Xch& x) {
    if (L.level < 0) return;
    constexpr int es = Xch::es;
    if (L.bsrc >= 0) {
        const float* s = xslot(x, L.bsrc);
        L.u = v3(-s[19 * es], -s[20 * es], -s[21 * es]);
    }
    if (L.level == 0 && L.slevel > 0) {
        const float* s = xslot(x, j);
        const float* f = model_body(m, j);
        const V3 alpha = v3(s[0 * es], s[1 * es], s[2 * es]), a = v3(s[3 * es], s[4 * es], s[5 * es]);
        const V3 d = -quat_rotate(L.Q, v3(f[44], f[45], f[46]));   // origin minus reference point
        L.u = alpha;
        L.ca = a + cross(alpha, d) + cross(L.w, cross(L.w, d));
    }
}

// ---- forward sweep, part 2 (no communication, every body at once): semi-implicit Euler on the body's own joint -- on the floating
// base for the root -- from the accelerations part 1 left in L.u / L.ca.  The new kinematics then follow from one aba_fk_level
// sweep (part 3), which is also the next sub-step's kinematics sweep: a sub-step is three sweeps, two of them short.
template <int JT>
PHC_HD void aba_integrate_joint(AbaLane& L, const phc_sim_params_t& prm, float dt) {
    if (L.level < 0) return;
    const float damp = 1.0f / (1.0f + dt * prm.angular_damping);
    if (L.level == 0) {
        const V3 alpha = L.u, a = L.ca;
        L.v = L.v + a * dt;
        L.w = (L.w + alpha * dt) * damp;
        L.p = L.p + L.v * dt;
        L.q = quat_normalize(quat_mul16(quat_from_rotvec(L.w * dt), L.q));
        return;
    }
    M3 R = quat_to_mat(L.Q);
    V3 qdd = mat_tmul(R, L.u);  // child-frame joint acceleration
    if (JT == PHC_JT_REVOLUTE) {
        const float thdd = dot(L.axis, qdd);
        // torque actually applied over the step (explicit part minus the implicit augmentation), S5 -- kept in tau_local.x
        L.tau_local = v3(dot(L.axis, L.tau_local) - L.dimp.x * thdd, 0.f, 0.f);
        L.thd = (L.thd + thdd * dt) * damp;
        L.thd = fminf(fmaxf(L.thd, -prm.max_angular_velocity), prm.max_angular_velocity);
        L.th += L.thd * dt;
        L.wj = L.axis * L.thd;
        L.q = rev_joint_quat(L);
    } else {
        // torque actually applied over the step (explicit part minus the implicit augmentation), S5
        L.tau_local = v3(L.tau_local.x - L.dimp.x * qdd.x, L.tau_local.y - L.dimp.y * qdd.y, L.tau_local.z - L.dimp.z * qdd.z);
        L.wj = (L.wj + qdd * dt) * damp;
        float wn = norm(L.wj);
        if (wn > prm.max_angular_velocity) L.wj = L.wj * (prm.max_angular_velocity / wn);
        L.q = quat_normalize(quat_mul16(L.q, quat_from_rotvec(L.wj * dt)));
    }
}

// S4 / S5 over the whole control step (phc_sim_params_t.force_average): sub-step s of n adds its net contact force (L.fcontact: ground + body-body)
// and the joint torque it applied (L.tau_local after aba_integrate_joint) to the accumulators; the last one leaves the means where
// aba_store_state / aba_publish_body read them.
PHC_HD void aba_force_accumulate(AbaLane& L, int s, int n, float* acc /*6 floats of this lane: LDS on the device -- not lane registers, the kernel sits at its VGPR limit*/) {
    if (L.level < 0) return;
    V3 f = L.fcontact, t = L.tau_local;
    if (s > 0) { f += v3(acc[0], acc[1], acc[2]); t += v3(acc[3], acc[4], acc[5]); }
    if (s == n - 1) { const float w = 1.0f / (float)n; L.fcontact = f * w; L.tau_local = t * w; return; }
    acc[0] = f.x; acc[1] = f.y; acc[2] = f.z; acc[3] = t.x; acc[4] = t.y; acc[5] = t.z;
}

// ---- state store: S1/S2 (+S5 dof force), and S3/S4 publication from the last kinematics sweep ----
template <int JT>
PHC_HD void aba_store_state(const AbaLane& L, const phc_sim_state_t& s, int nd, int64_t env, int j) {
    if (j == 0) {
        float* r = s.root_states + env * 13;
        r[0] = L.p.x; r[1] = L.p.y; r[2] = L.p.z; r[3] = L.q.x; r[4] = L.q.y; r[5] = L.q.z; r[6] = L.q.w;
        r[7] = L.v.x; r[8] = L.v.y; r[9] = L.v.z; r[10] = L.w.x; r[11] = L.w.y; r[12] = L.w.z;
    } else if (JT == PHC_JT_REVOLUTE) {
        float* d = s.dof_state + (env * nd + L.dof_start) * 2;
        d[0] = L.th; d[1] = L.thd;
        if (s.dof_force) s.dof_force[env * nd + L.dof_start] = L.tau_local.x;
    } else {
        float* d = s.dof_state + (env * nd + L.dof_start) * 2;
        V3 e = quat_to_rotvec(L.q);
        d[0] = e.x; d[1] = L.wj.x; d[2] = e.y; d[3] = L.wj.y; d[4] = e.z; d[5] = L.wj.z;
        if (s.dof_force) {
            float* f = s.dof_force + env * nd + L.dof_start;
            f[0] = L.tau_local.x; f[1] = L.tau_local.y; f[2] = L.tau_local.z;
        }
    }
}
// S6 force sensor on body j (gym.acquire_force_sensor_tensor, humanoid.py:183-190,1031-1040): net ground-contact wrench on the body
// about its origin, in the body's local frame, evaluated on the END-of-step state with the contact law of aba_body_init (explicit
// part: penalty normal force + regularised Coulomb friction).  Runs once per launch on the lanes of the sensor bodies only.
PHC_HD void aba_force_sensor(const AbaLane& L, const phc_model_t& m, const phc_sim_params_t& prm, float dt, int j, float* out6) {
    const float* f = model_body(m, j);
    const M3 R = quat_to_mat(L.Q);
    V3 F = v3(0.f, 0.f, 0.f), N = v3(0.f, 0.f, 0.f);
    const float cn = prm.contact_stiffness * dt + prm.contact_damping;
    const int cp_start = model_tab(m, 8, j), cp_total = model_tab(m, 9, j);
    const float* cp = m.floats + PHC_MAX_BODIES * PHC_BODY_FLOATS + cp_start * 4;
    const int cp_count = (L.p.z < f[34]) ? cp_total : 0;
    for (int k = 0; k < cp_count; ++k) {
        V3 arm = mat_mul(R, v3(cp[4 * k], cp[4 * k + 1], cp[4 * k + 2]));
        const float rad = cp[4 * k + 3];
        const float depth = rad - (L.p.z + arm.z);
        if (depth <= 0.f) continue;
        arm.z -= rad;
        const V3 uc = L.v + cross(L.w, arm);
        const float fn0 = prm.contact_stiffness * depth - cn * uc.z;
        if (fn0 <= 0.f) continue;
        const float ut = sqrtf(uc.x * uc.x + uc.y * uc.y);
        const float ct = fminf(prm.friction_viscous, prm.friction * fn0 / (ut + 1e-6f));
        const V3 F0 = v3(-ct * uc.x, -ct * uc.y, fn0);
        F += F0;
        N += cross(arm, F0);
    }
    const V3 Fl = mat_tmul(R, F), Nl = mat_tmul(R, N);
    out6[0] = Fl.x; out6[1] = Fl.y; out6[2] = Fl.z; out6[3] = Nl.x; out6[4] = Nl.y; out6[5] = Nl.z;
}
PHC_HD void aba_publish_sensors(const AbaLane& L, const phc_model_t& m, const phc_sim_params_t& prm, const phc_sim_state_t& s, float dt,
                                int64_t env, int j) {
    if (s.force_sensor == nullptr) return;
    for (int k = 0; k < prm.num_force_sensors; ++k)
        if (prm.force_sensor_body[k] == j) aba_force_sensor(L, m, prm, dt, j, s.force_sensor + (env * prm.num_force_sensors + k) * 6);
}

PHC_HD void aba_publish_body(const AbaLane& L, const phc_sim_state_t& s, int nb, int64_t env, int j, bool with_contact) {
    float* b = s.rigid_body_state + (env * nb + j) * 13;
    b[0] = L.p.x; b[1] = L.p.y; b[2] = L.p.z; b[3] = L.Q.x; b[4] = L.Q.y; b[5] = L.Q.z; b[6] = L.Q.w;
    b[7] = L.v.x; b[8] = L.v.y; b[9] = L.v.z; b[10] = L.w.x; b[11] = L.w.y; b[12] = L.w.z;
    if (with_contact && s.contact_force) {
        float* c = s.contact_force + (env * nb + j) * 3;
        c[0] = L.fcontact.x; c[1] = L.fcontact.y; c[2] = L.fcontact.z;
    }
}

}  // namespace phc
