// phc_task.h -- per-lane device functions of the imitation task: reference-motion lookup
// (M7-M9), imitation reward (R1/R2), reset test (R5), self / task / AMP observations (R6/R7/R9).
//
// Mapping: one lane per rigid body, 32 lanes per environment (NB <= 32), two environments per
// 64-wide wavefront.  Everything a lane needs from "its" body is private; the few per-env
// quantities (root pose, reward sums, fallen flag) are obtained by broadcast loads and one
// 32-lane reduction.  All functions are PHC_HD so oracle/hostemu can run the same code on CPU.
#pragma once
#include "phc_math.h"
#include "../../include/phc_amd.h"

#if defined(__clang__)
#define PHC_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define PHC_NO_CONTRACT
#endif

namespace phc {

// ---- frame record field offsets (see phc_motion_lib_t) ----
//   pos (NB+E)*3 | rot (NB+E)*4 | vel NB*3 | angvel NB*3 | joints | dof_vel
//   joints: local_rot NB*4 (spherical models: dof_pos = exp-map of the slerped local rotation, motion_lib_base.py:483-484)
//           or dof_pos ND (revolute models: linear blend of the stored angles, motion_lib_real.py:285-291)
// E = extended reference-only bodies (H1 hands / head, motion_lib_real.py:211-215), appended after the NB simulated ones.
PHC_HD int fr_nbe(const phc_motion_lib_t& l) { return l.num_bodies + l.num_ext_bodies; }
PHC_HD int fr_pos(const phc_motion_lib_t& l) { (void)l; return 0; }
PHC_HD int fr_rot(const phc_motion_lib_t& l) { return fr_nbe(l) * 3; }
PHC_HD int fr_vel(const phc_motion_lib_t& l) { return fr_nbe(l) * 7; }
PHC_HD int fr_angvel(const phc_motion_lib_t& l) { return fr_nbe(l) * 7 + l.num_bodies * 3; }
PHC_HD int fr_lrot(const phc_motion_lib_t& l) { return fr_nbe(l) * 7 + l.num_bodies * 6; }
PHC_HD int fr_dvel(const phc_motion_lib_t& l) {
    return fr_lrot(l) + (l.dofs_per_joint == 1 ? (l.num_bodies - 1) : l.num_bodies * 4);
}

struct FrameRef { int64_t f0, f1; float blend; int64_t idx0, idx1; };

// M8 _calc_frame_blend (motion_lib_base.py:549-559) + the length_starts offset (:447-448).
// Index arithmetic is the bit-exact part of the contract: no fma contraction in here.
// (the clip's table entries are loaded separately from the arithmetic, so that a kernel can request them at its very top)
struct FrameTab { float len, dt; int nf; int64_t start; };
PHC_HD FrameTab frame_tab(const phc_motion_lib_t& lib, int64_t mid) {
    FrameTab t;
    t.len = lib.motion_lengths[mid];
    t.dt = lib.motion_dt[mid];
    // frame counts are far below 2^24, so 32-bit integers and their float conversions give the same values as torch's int64
    // arithmetic (the 64-bit conversions cost ~10 instructions each on the device)
    t.nf = (int)lib.motion_num_frames[mid];
    t.start = lib.length_starts[mid];
    return t;
}
PHC_HD FrameRef frame_ref(const FrameTab& tab, float time) {
    PHC_NO_CONTRACT
    const float len = tab.len, dt = tab.dt;
    const int nf = tab.nf;
    float phase = time / len;
    phase = fminf(fmaxf(phase, 0.0f), 1.0f);  // torch.clip
    if (time < 0.f) time = 0.f;
    float nfm1 = (float)(nf - 1);
    float prod = phase * nfm1;
    FrameRef r;
    const int i0 = (int)prod;                 // .long(): truncation, prod >= 0
    const int i1 = (i0 + 1 < nf - 1) ? i0 + 1 : nf - 1;
    r.idx0 = i0;
    r.idx1 = i1;
    float sub = (float)i0 * dt;
    float bl = (time - sub) / dt;
    r.blend = fminf(fmaxf(bl, 0.0f), 1.0f);
    r.f0 = r.idx0 + tab.start;
    r.f1 = r.idx1 + tab.start;
    return r;
}
PHC_HD FrameRef frame_ref(const phc_motion_lib_t& lib, int64_t mid, float time) { return frame_ref(frame_tab(lib, mid), time); }

// M7 sample_time_interval (motion_lib_base.py:414-423)
PHC_HD float sample_time_interval(const phc_motion_lib_t& lib, int64_t mid, float phase) {
    PHC_NO_CONTRACT
    const float curr_fps = (float)(1.0 / 30.0);
    float t = (phase * lib.motion_lengths[mid]) / curr_fps;
    return (float)((int)t) * curr_fps;   // .long(): t < 2^24 frames
}

// env time (humanoid_im.py:879,752,1118): progress * dt + start + offset, each op rounded to fp32
PHC_HD float motion_time(int64_t progress, float dt, float start, float start_off) {
    PHC_NO_CONTRACT
    float a = (float)progress * dt;
    float b = a + start;
    return b + start_off;
}
// fut_tracks sample k (humanoid_im.py:744-745): (progress + 1) * dt + k * traj_sample_timestep + start + offset, left to right in fp32
PHC_HD float motion_time_future(int64_t progress1, float dt, float k_ts, float start, float start_off) {
    PHC_NO_CONTRACT
    float a = (float)progress1 * dt;
    float b = a + k_ts;
    float c = b + start;
    return c + start_off;
}
// humanoid_im.py:1126: -progress_buf * dt
PHC_HD float neg_progress_time(int64_t progress, float dt) {
    PHC_NO_CONTRACT
    return (float)(-progress) * dt;
}
// humanoid_amp.py:575-603 / 253-284: t0 + (-dt * k)
PHC_HD float history_time(float t0, float dt, int k) {
    PHC_NO_CONTRACT
    float ts = (-dt) * (float)k;
    return t0 + ts;
}

PHC_HD V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }
PHC_HD Q4 ld4(const float* p) { return q4(p[0], p[1], p[2], p[3]); }
PHC_HD void st3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
PHC_HD void st4(float* p, Q4 q) { p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w; }
PHC_HD V3 lerp3(V3 a, V3 b, float t) {  // (1-t)*a + t*b, the reference's form (motion_lib_base.py:474-480)
    float s = 1.0f - t;
    return v3(s * a.x + t * b.x, s * a.y + t * b.y, s * a.z + t * b.z);
}

struct BodyState { V3 pos; Q4 rot; V3 vel; V3 angvel; };

// M9 get_motion_state, the part of it that belongs to body j (motion_lib_base.py:450-488)
PHC_HD BodyState ref_body(const phc_motion_lib_t& lib, const FrameRef& fr, int j) {
    const float* a = lib.frames + fr.f0 * (int64_t)lib.frame_stride;
    const float* b = lib.frames + fr.f1 * (int64_t)lib.frame_stride;
    BodyState s;
    s.pos = lerp3(ld3(a + fr_pos(lib) + 3 * j), ld3(b + fr_pos(lib) + 3 * j), fr.blend);
    s.vel = lerp3(ld3(a + fr_vel(lib) + 3 * j), ld3(b + fr_vel(lib) + 3 * j), fr.blend);
    s.angvel = lerp3(ld3(a + fr_angvel(lib) + 3 * j), ld3(b + fr_angvel(lib) + 3 * j), fr.blend);
    s.rot = slerp(ld4(a + fr_rot(lib) + 4 * j), ld4(b + fr_rot(lib) + 4 * j), fr.blend);
    return s;
}
// The same in two halves: the LOADS of body j's two frame records, and the blend.  A caller that has independent work puts it between the two, so that
// the records' memory latency is covered (a wavefront issues in order: a load's value is waited for at its first use, not at the load).
struct BodyRaw { V3 pa, pb, va, vb, wa, wb; Q4 ra, rb; };
PHC_HD BodyRaw ref_body_raw(const phc_motion_lib_t& lib, const FrameRef& fr, int j) {
    const float* a = lib.frames + fr.f0 * (int64_t)lib.frame_stride;
    const float* b = lib.frames + fr.f1 * (int64_t)lib.frame_stride;
    BodyRaw q;
    q.pa = ld3(a + fr_pos(lib) + 3 * j); q.pb = ld3(b + fr_pos(lib) + 3 * j);
    q.va = ld3(a + fr_vel(lib) + 3 * j); q.vb = ld3(b + fr_vel(lib) + 3 * j);
    q.wa = ld3(a + fr_angvel(lib) + 3 * j); q.wb = ld3(b + fr_angvel(lib) + 3 * j);
    q.ra = ld4(a + fr_rot(lib) + 4 * j); q.rb = ld4(b + fr_rot(lib) + 4 * j);
    return q;
}
PHC_HD BodyState ref_body_blend(const BodyRaw& q, float blend) {
    BodyState s;
    s.pos = lerp3(q.pa, q.pb, blend);
    s.vel = lerp3(q.va, q.vb, blend);
    s.angvel = lerp3(q.wa, q.wb, blend);
    s.rot = slerp(q.ra, q.rb, blend);
    return s;
}
// extended reference body e (record slot NB+e): position and rotation only (rg_pos_t / rg_rot_t, motion_lib_real.py:300-312)
PHC_HD void ref_body_ext(const phc_motion_lib_t& lib, const FrameRef& fr, int e, V3* pos, Q4* rot) {
    const int j = lib.num_bodies + e;
    const float* a = lib.frames + fr.f0 * (int64_t)lib.frame_stride;
    const float* b = lib.frames + fr.f1 * (int64_t)lib.frame_stride;
    *pos = lerp3(ld3(a + fr_pos(lib) + 3 * j), ld3(b + fr_pos(lib) + 3 * j), fr.blend);
    *rot = slerp(ld4(a + fr_rot(lib) + 4 * j), ld4(b + fr_rot(lib) + 4 * j), fr.blend);
}
// get_root_pos_smpl (motion_lib_base.py:522-547): position-only lookup of the root
PHC_HD V3 ref_root_pos_lerp(const phc_motion_lib_t& lib, const FrameRef& fr) {
    const float* a = lib.frames + fr.f0 * (int64_t)lib.frame_stride;
    const float* b = lib.frames + fr.f1 * (int64_t)lib.frame_stride;
    return lerp3(ld3(a + fr_pos(lib)), ld3(b + fr_pos(lib)), fr.blend);
}
// joint of body j >= 1.  Spherical: dof_pos = quat_to_exp_map(slerp(local_rot)) (motion_lib_base.py:483-484,564-567), dof_vel
// lerp.  Revolute (dofs_per_joint 1): both are linear blends of the stored scalars (motion_lib_real.py:285-291), carried in .x.
PHC_HD void ref_joint(const phc_motion_lib_t& lib, const FrameRef& fr, int j, V3* dof_pos, V3* dof_vel) {
    const float* a = lib.frames + fr.f0 * (int64_t)lib.frame_stride;
    const float* b = lib.frames + fr.f1 * (int64_t)lib.frame_stride;
    if (lib.dofs_per_joint == 1) {
        const float s = 1.0f - fr.blend;
        *dof_pos = v3(s * a[fr_lrot(lib) + j - 1] + fr.blend * b[fr_lrot(lib) + j - 1], 0.f, 0.f);
        *dof_vel = v3(s * a[fr_dvel(lib) + j - 1] + fr.blend * b[fr_dvel(lib) + j - 1], 0.f, 0.f);
        return;
    }
    Q4 lr = slerp(ld4(a + fr_lrot(lib) + 4 * j), ld4(b + fr_lrot(lib) + 4 * j), fr.blend);
    *dof_pos = quat_to_exp_map(lr);
    *dof_vel = lerp3(ld3(a + fr_dvel(lib) + 3 * (j - 1)), ld3(b + fr_dvel(lib) + 3 * (j - 1)), fr.blend);
}
// dof_pos alone, in the same two halves as ref_body_raw / ref_body_blend
struct JointPosRaw { Q4 la, lb; };   // spherical: the two local rotations; revolute: the two angles in la.x / lb.x
PHC_HD JointPosRaw ref_joint_pos_raw(const phc_motion_lib_t& lib, const FrameRef& fr, int j) {
    const float* a = lib.frames + fr.f0 * (int64_t)lib.frame_stride;
    const float* b = lib.frames + fr.f1 * (int64_t)lib.frame_stride;
    JointPosRaw q;
    if (lib.dofs_per_joint == 1) { q.la = q4(a[fr_lrot(lib) + j - 1], 0.f, 0.f, 1.f); q.lb = q4(b[fr_lrot(lib) + j - 1], 0.f, 0.f, 1.f); }
    else { q.la = ld4(a + fr_lrot(lib) + 4 * j); q.lb = ld4(b + fr_lrot(lib) + 4 * j); }
    return q;
}
PHC_HD V3 ref_joint_pos_blend(const phc_motion_lib_t& lib, const JointPosRaw& q, float blend) {
    if (lib.dofs_per_joint == 1) return v3((1.0f - blend) * q.la.x + blend * q.lb.x, 0.f, 0.f);
    return quat_to_exp_map(slerp(q.la, q.lb, blend));
}
// Spherical joint of body j >= 1 for the AMP observation of a REFERENCE frame (humanoid_amp.py:575-603,253-284): the reference goes
// local rotation -> slerp -> quat_to_exp_map (dof_pos, motion_lib_base.py:483-484) -> exp_map_to_quat -> tan_norm (dof_to_obs_smpl,
// humanoid.py:1756-1765).  Composed in closed form here: with q = (v, w), a = normalize_angle(2 acos w), s = sqrt(1 - w^2), the exp-map is
// (v / s) a, so exp_map_to_quat sees the angle |a| |v| / s about sign(a) v / |v| -- for a unit q that is +-q, for the not-quite-unit
// rotations a motion file holds the factor |v| / s is what the reference computes, and it is kept (it moves small joint angles by up to
// 1e-4 on the synthetic clips).  Both 1e-5 thresholds of the chain are kept.  One acos, one sin / cos pair, two square roots instead of
// acos + 2 atan2 + 3 sin / cos pairs + ~14 divisions (round 3: this chain was half of the reset kernel's instruction stream).
PHC_HD Q4 exp_map_round_trip(Q4 q) {
    const float sin_half = t_sqrt(1.0f - q.w * q.w);
    const bool mask1 = fabsf(sin_half) > 1e-5f;                   // quat_to_angle_axis (NaN for |w| > 1 -> false, as torch.abs(nan) > x)
    const float angle = normalize_angle(2.0f * acosf(q.w));
    const float vn = t_sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
    const float ang2 = normalize_angle(fabsf(angle) * vn * t_rcp(sin_half));   // exp_map_to_quat: |exp-map|, then normalize_angle
    const bool mask2 = fabsf(ang2) > 1e-5f;
    const float k = (angle < 0.f ? -1.0f : 1.0f) * t_rcp(vn);
    float sn, cs;
    t_sincos(0.5f * ang2, &sn, &cs);
    return (mask1 && mask2) ? q4(q.x * k * sn, q.y * k * sn, q.z * k * sn, cs) : q4(0.f, 0.f, 0.f, 1.f);
}
PHC_HD void ref_joint_rot(const phc_motion_lib_t& lib, const FrameRef& fr, int j, Q4* rot, V3* dof_vel) {
    const float* a = lib.frames + fr.f0 * (int64_t)lib.frame_stride;
    const float* b = lib.frames + fr.f1 * (int64_t)lib.frame_stride;
    *rot = exp_map_round_trip(slerp(ld4(a + fr_lrot(lib) + 4 * j), ld4(b + fr_lrot(lib) + 4 * j), fr.blend));
    *dof_vel = lerp3(ld3(a + fr_dvel(lib) + 3 * (j - 1)), ld3(b + fr_dvel(lib) + 3 * (j - 1)), fr.blend);
}
// joint coordinates in the simulator tensors: 3 (exp-map) or 1 (angle) consecutive DoFs starting at ds
PHC_HD void ld_joint_state(const phc_sim_state_t& sim, int nd, int64_t env, int ds, int dpj, V3* pos, V3* vel) {
    const float* d = sim.dof_state + (env * nd + ds) * 2;
    if (dpj == 1) { *pos = v3(d[0], 0.f, 0.f); *vel = v3(d[1], 0.f, 0.f); }
    else { *pos = v3(d[0], d[2], d[4]); *vel = v3(d[1], d[3], d[5]); }
}
PHC_HD void st_joint(float* p, int dpj, V3 v) {
    p[0] = v.x;
    if (dpj != 1) { p[1] = v.y; p[2] = v.z; }
}

PHC_HD BodyState load_body(const float* rigid_body_state, int64_t env, int nb, int j) {
    const float* p = rigid_body_state + (env * nb + j) * 13;
    BodyState s;
    s.pos = ld3(p); s.rot = ld4(p + 3); s.vel = ld3(p + 7); s.angvel = ld3(p + 10);
    return s;
}
PHC_HD void store_body(float* rigid_body_state, int64_t env, int nb, int j, const BodyState& s) {
    float* p = rigid_body_state + (env * nb + j) * 13;
    st3(p, s.pos); st4(p + 3, s.rot); st3(p + 7, s.vel); st3(p + 10, s.angvel);
}

// `if not upright: root_rot = remove_base_rot(root_rot)` (humanoid.py:1936-1939): quat_mul(q, conj(0.5, 0.5, 0.5, 0.5)); the heading and
// the root-rotation observation are taken from the result
PHC_HD Q4 obs_root_rot(const phc_im_params_t& prm, Q4 q) {
    return prm.remove_base_rot ? quat_mul(q, q4(-0.5f, -0.5f, -0.5f, 0.5f)) : q;
}
// random reference offset of zero_out_far_train (humanoid_im.py:971-977,1135-1140): a point of the 5 m disk from two uniform draws
PHC_HD void disk_offset(float u, float v, float* ox, float* oy) {
    PHC_NO_CONTRACT
    const float rd = sqrtf(u) * 5.0f;                  // torch.sqrt(torch.rand) * max_distance
    const float ang = (v * 3.14159265358979323846f) * 2.0f;   // torch.rand * np.pi * 2
    *ox = cosf(ang) * rd;
    *oy = sinf(ang) * rd;
}
// per-env constant columns (shape parameters, limb weights) appended to an observation: the `nl` lanes of the env share the copy
PHC_HD void obs_extra_lane(const float* src, int n, int lane, int nl, float* dst) {
    if (src == nullptr) return;
    for (int k = lane; k < n; k += nl) dst[k] = src[k];
}

// ---- R6: compute_humanoid_observations_smpl_max (humanoid.py:1995-2050), lane j's slices ----
// `sensors`: the env's S6 force-sensor readings [S*6] (self_obs_v 3: compute_humanoid_observations_smpl_max_v3, humanoid.py:2113-2169,
// appends them after the angular-velocity block) or nullptr.
PHC_HD void self_obs_lane(const phc_im_params_t& prm, int nb, int j, const BodyState& body, const BodyState& root,
                          Q4 hinv, float* obs, const float* sensors = nullptr, int64_t env = -1, bool hist_step = false) {
    // `hist_step` (self_obs_v 2, humanoid.py:2054-2108): `body` is this body's state at one of the time steps, `root` / `hinv` stay the CURRENT
    // root; the height column is the root's z at that time step (for j == 0 `body` IS the root then) and the raw root rotation is that step's
    if (env >= 0 && j < nb && prm.num_self_obs_extra > 0 && prm.self_obs_extra)
        obs_extra_lane(prm.self_obs_extra + env * prm.num_self_obs_extra, prm.num_self_obs_extra, j, nb, obs + prm.num_self_obs - prm.num_self_obs_extra);
    int off = 0;
    if (prm.root_height_obs) { if (j == 0) obs[0] = body.pos.z; off = 1; }
    if (j >= 1) st3(obs + off + (j - 1) * 3, quat_rotate(hinv, body.pos - root.pos));
    float tn[6];
    if (j == 0 && !prm.local_root_obs) quat_to_tan_norm(hist_step ? body.rot : obs_root_rot(prm, root.rot), tn);  // :2026-2028 / :2085-2087
    else quat_to_tan_norm(quat_mul(hinv, body.rot), tn);
    float* pr = obs + off + (nb - 1) * 3 + j * 6;
    for (int k = 0; k < 6; ++k) pr[k] = tn[k];
    st3(obs + off + (nb - 1) * 3 + nb * 6 + j * 3, quat_rotate(hinv, body.vel));
    st3(obs + off + (nb - 1) * 3 + nb * 9 + j * 3, quat_rotate(hinv, body.angvel));
    if (prm.self_obs_v == 3 && sensors != nullptr && j < prm.num_force_sensors) {   // lane s copies sensor s
        float* o = obs + off + (nb - 1) * 3 + nb * 12 + j * 6;
        for (int k = 0; k < 6; ++k) o[k] = sensors[j * 6 + k];
    }
}

// self_obs_v 2: the P history states of body j then the current one, each a v1 block of S1 = num_self_obs / (P + 1) floats (:2091-2107);
// `shift`: afterwards the history advances by one step (`_update_tensor_history`, humanoid.py:1627-1631 -- called at the top of the NEXT
// post_physics_step with the state this one saw); `fill`: every history slot := the current state (`_init_tensor_history`, :1621-1625)
PHC_HD void self_obs_v2_lane(const phc_im_params_t& prm, float* hist_all, int nb, int64_t env, int j, const BodyState& body, const BodyState& root,
                             Q4 hinv, float* obs, bool shift, bool fill) {
    const int P = prm.num_self_obs_hist, S1 = prm.num_self_obs / (P + 1);
    float* h = hist_all + ((env * P) * nb + j) * 13;       // slot k of body j: h + k * nb * 13
    if (fill)
        for (int k = 0; k < P; ++k) { float* p = h + (int64_t)k * nb * 13; st3(p, body.pos); st4(p + 3, body.rot); st3(p + 7, body.vel); st3(p + 10, body.angvel); }
    for (int k = 0; k < P; ++k) {
        const float* p = h + (int64_t)k * nb * 13;
        BodyState s;
        s.pos = ld3(p); s.rot = ld4(p + 3); s.vel = ld3(p + 7); s.angvel = ld3(p + 10);
        self_obs_lane(prm, nb, j, s, root, hinv, obs + k * S1, nullptr, -1, true);
        if (shift && k >= 1) { float* q = h + (int64_t)(k - 1) * nb * 13; st3(q, s.pos); st4(q + 3, s.rot); st3(q + 7, s.vel); st3(q + 10, s.angvel); }
    }
    self_obs_lane(prm, nb, j, body, root, hinv, obs + P * S1, nullptr, -1, true);
    if (shift && P >= 1) { float* q = h + (int64_t)(P - 1) * nb * 13; st3(q, body.pos); st4(q + 3, body.rot); st3(q + 7, body.vel); st3(q + 10, body.angvel); }
}

// ---- R7: compute_imitation_observations_v6 (humanoid_im.py:1309-1358), time_steps = 1 ----
// Other versions (env.obs_v), all at time_steps = 1 (fut_tracks off), J = tracked bodies, the tracked body in slot s:
//   1 `compute_imitation_observations`   (:1203-1236)  [dpos 3J | drot 6J | dvel 3J | dangvel 3J]
//   2 `_v2` (:1240-1278)  v1 + [ref_dof_pos - dof_pos of the joints of tracked bodies 1.. , 3 (J-1)]   (`dof` / `ref_dof`: this body's joint)
//   3 `_v3` (:1281-1306)  [dpos 3J | drot 6J]
//   8 `_v8` (:1396-1458)  v1 + [ref pos - root pos 3J | ref rot 6J | ref vel 3J | ref angvel 3J], all heading-local
//   9 `_v9` (:1462-1515)  [dpos 3J | drot 6J | root dvel 3 | root dangvel 3 | ref pos - root pos 3J | ref rot 6J]
PHC_HD void task_obs_lane(const phc_im_params_t& prm, int slot, const BodyState& body, const BodyState& root,
                          const BodyState& ref, Q4 hinv, Q4 h, float* tobs, const V3* dof = nullptr, const V3* ref_dof = nullptr) {
    const int jt = prm.num_track_bodies;
    if (prm.obs_v == 1 || prm.obs_v == 2 || prm.obs_v == 3 || prm.obs_v == 8 || prm.obs_v == 9) {
        float tn[6];
        st3(tobs + slot * 3, quat_rotate(hinv, ref.pos - body.pos));
        quat_to_tan_norm(quat_mul(quat_mul(hinv, quat_mul(ref.rot, quat_conjugate(body.rot))), h), tn);
        for (int k = 0; k < 6; ++k) tobs[jt * 3 + slot * 6 + k] = tn[k];
        if (prm.obs_v == 3) return;
        if (prm.obs_v == 9) {
            if (slot == 0) {   // root velocity differences only (:1488-1494): the root is the first tracked body
                st3(tobs + jt * 9, quat_rotate(hinv, ref.vel - body.vel));
                st3(tobs + jt * 9 + 3, quat_rotate(hinv, ref.angvel - body.angvel));
            }
            st3(tobs + jt * 9 + 6 + slot * 3, quat_rotate(hinv, ref.pos - root.pos));
            quat_to_tan_norm(quat_mul(hinv, ref.rot), tn);
            for (int k = 0; k < 6; ++k) tobs[jt * 12 + 6 + slot * 6 + k] = tn[k];
            return;
        }
        st3(tobs + jt * 9 + slot * 3, quat_rotate(hinv, ref.vel - body.vel));
        st3(tobs + jt * 12 + slot * 3, quat_rotate(hinv, ref.angvel - body.angvel));
        if (prm.obs_v == 2 && slot >= 1 && dof != nullptr && ref_dof != nullptr)
            st3(tobs + jt * 15 + (slot - 1) * 3, *ref_dof - *dof);
        if (prm.obs_v == 8) {
            st3(tobs + jt * 15 + slot * 3, quat_rotate(hinv, ref.pos - root.pos));
            quat_to_tan_norm(quat_mul(hinv, ref.rot), tn);
            for (int k = 0; k < 6; ++k) tobs[jt * 18 + slot * 6 + k] = tn[k];
            st3(tobs + jt * 24 + slot * 3, quat_rotate(hinv, ref.vel));
            st3(tobs + jt * 27 + slot * 3, quat_rotate(hinv, ref.angvel));
        }
        return;
    }
    if (prm.obs_v == 7) {
        // compute_imitation_observations_v7 (humanoid_im.py:1362-1393, the keypoint models): no rotation terms
        st3(tobs + slot * 3, quat_rotate(hinv, ref.pos - body.pos));
        st3(tobs + jt * 3 + slot * 3, quat_rotate(hinv, ref.vel - body.vel));
        st3(tobs + jt * 6 + slot * 3, quat_rotate(hinv, ref.pos - root.pos));
        return;
    }
    st3(tobs + slot * 3, quat_rotate(hinv, ref.pos - body.pos));
    Q4 drot = quat_mul(ref.rot, quat_conjugate(body.rot));
    Q4 dl = quat_mul(quat_mul(hinv, drot), h);
    float tn[6];
    quat_to_tan_norm(dl, tn);
    for (int k = 0; k < 6; ++k) tobs[jt * 3 + slot * 6 + k] = tn[k];
    st3(tobs + jt * 9 + slot * 3, quat_rotate(hinv, ref.vel - body.vel));
    st3(tobs + jt * 12 + slot * 3, quat_rotate(hinv, ref.angvel - body.angvel));
    st3(tobs + jt * 15 + slot * 3, quat_rotate(hinv, ref.pos - root.pos));
    quat_to_tan_norm(quat_mul(hinv, ref.rot), tn);
    for (int k = 0; k < 6; ++k) tobs[jt * 18 + slot * 6 + k] = tn[k];
}

// ---- R9: build_amp_observations_smpl (humanoid_amp.py:967-1011), lane j's slices of one 196-float step ----
// layout: [root_h? | root_rot 6 | root_vel 3 | root_ang_vel 3 | dof_obs 6*NJ | dof_vel 3*NJ | key_pos 3*K]
PHC_HD void amp_obs_root(const phc_im_params_t& prm, V3 root_pos, Q4 root_rot, V3 root_vel, V3 root_angvel, Q4 hinv, float* a) {
    int off = 0;
    if (prm.root_height_obs) { a[0] = root_pos.z; off = 1; }
    float tn[6];
    root_rot = obs_root_rot(prm, root_rot);
    quat_to_tan_norm(prm.local_root_obs ? quat_mul(hinv, root_rot) : root_rot, tn);
    for (int k = 0; k < 6; ++k) a[off + k] = tn[k];
    st3(a + off + 6, quat_rotate(hinv, root_vel));
    st3(a + off + 9, quat_rotate(hinv, root_angvel));
}
PHC_HD void amp_obs_joint(const phc_im_params_t& prm, int slot, V3 dof_pos, V3 dof_vel, float* a) {
    const int off = (prm.root_height_obs ? 1 : 0) + 12;
    if (prm.dofs_per_joint == 1) {  // build_amp_observations_robot (humanoid_amp.py:1063-1104): dof_obs = dof_pos
        a[off + slot] = dof_pos.x;
        a[off + prm.num_amp_joints + slot] = dof_vel.x;
        return;
    }
    float tn[6];
    quat_to_tan_norm(exp_map_to_quat(dof_pos), tn);  // dof_to_obs_smpl humanoid.py:1756-1765
    for (int k = 0; k < 6; ++k) a[off + slot * 6 + k] = tn[k];
    st3(a + off + prm.num_amp_joints * 6 + slot * 3, dof_vel);
}
PHC_HD void amp_obs_joint_rot(const phc_im_params_t& prm, int slot, Q4 rot, V3 dof_vel, float* a) {   // spherical joints, see ref_joint_rot
    const int off = (prm.root_height_obs ? 1 : 0) + 12;
    float tn[6];
    quat_to_tan_norm(rot, tn);
    for (int k = 0; k < 6; ++k) a[off + slot * 6 + k] = tn[k];
    st3(a + off + prm.num_amp_joints * 6 + slot * 3, dof_vel);
}
// amp_obs_v 2 (build_amp_observations_smpl_v2, humanoid_amp.py:1015-1059): the key bodies' velocities, heading-local, follow their positions
PHC_HD void amp_obs_key(const phc_im_params_t& prm, int k, V3 key_pos, V3 key_vel, V3 root_pos, Q4 hinv, float* a) {
    const int off = (prm.root_height_obs ? 1 : 0) + 12 + prm.num_amp_joints * (prm.dofs_per_joint == 1 ? 2 : 9);
    st3(a + off + k * 3, quat_rotate(hinv, key_pos - root_pos));
    if (prm.amp_obs_v == 2) st3(a + off + prm.num_key_bodies * 3 + k * 3, quat_rotate(hinv, key_vel));
}

// ---- R1/R5 per-lane partials, reduced over the 32-lane group by the caller ----
struct RewardPartial { float pos, rot, vel, angvel, power, dist, root_dist; int fallen; };

PHC_HD RewardPartial reward_partial(const phc_im_params_t& prm, int64_t env, int nb, int j, const BodyState& body, const BodyState& ref) {
    RewardPartial p;
    V3 d = ref.pos - body.pos;
    p.pos = (d.x * d.x + d.y * d.y + d.z * d.z) / 3.0f;  // (diff**2).mean(-1)   humanoid_im.py:1530-1531
    Q4 dq = quat_mul(ref.rot, quat_conjugate(body.rot));
    float ang = quat_to_angle_axis(dq, nullptr);          // :1535-1537
    p.rot = ang * ang;
    d = ref.vel - body.vel;
    p.vel = (d.x * d.x + d.y * d.y + d.z * d.z) / 3.0f;
    d = ref.angvel - body.angvel;
    p.angvel = (d.x * d.x + d.y * d.y + d.z * d.z) / 3.0f;
    p.power = 0.f; p.root_dist = 0.f;
    // compute_humanoid_im_reset :1586-1588
    float dist = norm(body.pos - ref.pos);
    int in_reset = prm.reset_mask[j];
    p.dist = in_reset ? dist : 0.f;
    p.fallen = (in_reset && dist > prm.termination_distances[j]) ? 1 : 0;
    return p;
}

}  // namespace phc
