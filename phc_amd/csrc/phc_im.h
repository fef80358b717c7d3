// phc_im.h -- per-lane bodies of the imitation-task kernels (post-physics step, reset, AMP demo).
// One lane per rigid body, 32 (or, above 32 bodies, 64) lanes per environment.  PHC_HD so oracle/hostemu can drive the same
// code lane-by-lane on the CPU.  Reference call sites are cited at each step.
#pragma once
#include "phc_task.h"

namespace phc {

#if defined(PHC_SIM_PROFILE) && defined(__HIPCC__)
// one env's timeline through the post-physics kernel (scripts/probes/post_timeline.py; -DPHC_SIM_PROFILE library only).  Every stamp drains the
// memory counters first: the deltas are the SERIALISED cost of each section.
static __device__ unsigned long long g_phc_ptl[32];
static __device__ long long g_phc_ptl_env = -1;
#if defined(__HIP_DEVICE_COMPILE__)
#define PHC_PTL(i, ENV, LANE) if ((long long)(ENV) == g_phc_ptl_env) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); \
        const unsigned long long t_ = __builtin_readcyclecounter(); if ((LANE) == 0) g_phc_ptl[i] = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define PHC_PTL(i, ENV, LANE)
#endif
#else
#define PHC_PTL(i, ENV, LANE)
#endif

// the env's clip: sampled_motion_ids[env], or the env's own index when the caller passes no table (one clip per env, humanoid_im.py:121
// `_sampled_motion_ids = arange(num_envs)`: the table load is then one dependent memory round trip the lookup chain does not need)
PHC_HD int64_t motion_id_of(const phc_im_buffers_t& buf, int64_t env) { return buf.sampled_motion_ids ? buf.sampled_motion_ids[env] : env; }

// fetch-and-increment (device: one atomic per finished env; host emulation: OpenMP atomic capture)
PHC_HD int phc_atomic_inc(int32_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return atomicAdd(p, 1);
#else
    int v;
#pragma omp atomic capture
    v = (*p)++;
    return v;
#endif
}

// AMP observation of one time step computed from the *reference motion* (no offset):
// HumanoidAMP._init_amp_obs_ref / build_amp_obs_demo (humanoid_amp.py:575-603,253-284).
// Lane j writes its slices of a[0..A).
// (the part that depends on the frame pair only -- also what phc_amp_ref_table tabulates per frame, with f0 = f1 = f and blend 0)
PHC_HD void amp_obs_from_frames_lane(const phc_motion_lib_t& lib, const phc_im_params_t& prm, int nb, int j, const FrameRef& fr, float* a);
PHC_HD void amp_obs_from_ref_lane(const phc_motion_lib_t& lib, const phc_im_params_t& prm, int nb, int j,
                                  int64_t mid, float t, float* a) {
    if (j < nb && prm.num_amp_obs_extra > 0 && prm.amp_obs_extra)   // the clip's humanoid (motion id == env id)
        obs_extra_lane(prm.amp_obs_extra + mid * prm.num_amp_obs_extra, prm.num_amp_obs_extra, j, nb, a + prm.num_amp_obs_per_step - prm.num_amp_obs_extra);
    amp_obs_from_frames_lane(lib, prm, nb, j, frame_ref(lib, mid, t), a);
}
PHC_HD void amp_obs_from_frames_lane(const phc_motion_lib_t& lib, const phc_im_params_t& prm, int nb, int j, const FrameRef& fr, float* a) {
    BodyState root = ref_body(lib, fr, 0);
    Q4 hinv = calc_heading_quat_inv(obs_root_rot(prm, root.rot));
    if (j == 0) amp_obs_root(prm, root.pos, root.rot, root.vel, root.angvel, hinv, a);
    if (j >= 1 && j < nb) {
        int slot = prm.amp_joint_slot[j];
        if (slot >= 0) {
            V3 dp, dv;
            if (prm.dofs_per_joint == 1) {
                ref_joint(lib, fr, j, &dp, &dv);
                amp_obs_joint(prm, slot, dp, dv, a);
            } else {
                Q4 lr;
                ref_joint_rot(lib, fr, j, &lr, &dv);
                amp_obs_joint_rot(prm, slot, lr, dv, a);
            }
        }
    }
    if (j < prm.num_key_bodies) {
        BodyState kb = ref_body(lib, fr, prm.key_body_ids[j]);
        amp_obs_key(prm, j, kb.pos, kb.vel, root.pos, hinv, a);
    }
}

// AMP observation of the current step from simulator state:
// HumanoidAMP._compute_amp_observations (humanoid_amp.py:672-707)
PHC_HD void amp_obs_from_sim_lane(const phc_im_params_t& prm, const phc_sim_state_t& sim, int nb, int nd, int64_t env, int j,
                                  const BodyState& root, Q4 hinv, const int* dof_start_tab, float* a) {
    if (j < nb && prm.num_amp_obs_extra > 0 && prm.amp_obs_extra)
        obs_extra_lane(prm.amp_obs_extra + env * prm.num_amp_obs_extra, prm.num_amp_obs_extra, j, nb, a + prm.num_amp_obs_per_step - prm.num_amp_obs_extra);
    if (j == 0) amp_obs_root(prm, root.pos, root.rot, root.vel, root.angvel, hinv, a);
    if (j >= 1 && j < nb) {
        int slot = prm.amp_joint_slot[j];
        if (slot >= 0) {
            V3 dp, dv;
            ld_joint_state(sim, nd, env, dof_start_tab[j], prm.dofs_per_joint, &dp, &dv);
            amp_obs_joint(prm, slot, dp, dv, a);
        }
    }
    if (j < prm.num_key_bodies) {
        BodyState kb = load_body(sim.rigid_body_state, env, nb, prm.key_body_ids[j]);
        amp_obs_key(prm, j, kb.pos, kb.vel, root.pos, hinv, a);
    }
}

// History shift of HumanoidAMP._update_hist_amp_obs (humanoid_amp.py:662-670), ping-pong:
// out[env][1..S) = in[env][0..S-1).  Cooperative over the `nl` lanes of the env, float4 wide.
// Floats between two envs' history windows: S * A, or phc_im_buffers_t.amp_env_stride when the windows live in longer per-env strips.
PHC_HD int64_t amp_env_stride(const phc_im_params_t& prm, const phc_im_buffers_t& buf) {
    return buf.amp_env_stride > 0 ? buf.amp_env_stride : (int64_t)prm.num_amp_obs_steps * prm.num_amp_obs_per_step;
}
PHC_HD void amp_shift_lane(const phc_im_params_t& prm, const phc_im_buffers_t& buf, int64_t env, int lane, int nl) {
    const int A = prm.num_amp_obs_per_step, S = prm.num_amp_obs_steps;
    // the caller's new window starts one frame BEFORE the old one in the same strip: the old frames already sit where the shift would put
    // them (phc_im_buffers_t.amp_env_stride: the strip is 2 S frames long, the window walks down it and is moved back up every S steps)
    if (buf.amp_obs_out + A == buf.amp_obs_in) return;
    const float* src = buf.amp_obs_in + env * amp_env_stride(prm, buf);
    float* dst = buf.amp_obs_out + env * amp_env_stride(prm, buf) + A;
    const int n = (S - 1) * A;
    if ((A & 3) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        int i = lane;
        for (; i + 3 * nl < n / 4; i += 4 * nl) {   // four independent 16-byte loads in flight per lane
            const float4 a = s4[i], b = s4[i + nl], c = s4[i + 2 * nl], e = s4[i + 3 * nl];
            d4[i] = a; d4[i + nl] = b; d4[i + 2 * nl] = c; d4[i + 3 * nl] = e;
        }
        for (; i < n / 4; i += nl) d4[i] = s4[i];
    } else {
        for (int i = lane; i < n; i += nl) dst[i] = src[i];
    }
}

// Per-env scalar state of one post-physics step, computed redundantly by every lane of the env (a handful of loads)
// BEFORE any lane work, because the config-3 options change what the lanes look up:
//   cycle_motion  (humanoid_im.py:1120-1150): an env whose clip ran out restarts the clip in place -- new start time,
//                 time offset cancelling progress, global xy offset pinning the clip's root to the humanoid's root;
//                 the reward still uses the old time/offset (it is computed before _compute_reset, humanoid.py:1642-1645);
//   getup         (humanoid_im_getup.py:203-216): while recovery_counter > 0 the env is neither reset nor advanced.
struct ImStepCtx {
    int64_t progress;     // progress_buf value written back (getup recovery: not advanced)
    float t_rew; V3 goff_rew;    // reward lookup
    float t0; V3 goff;           // reset lookup and everything after it
    float t1;                    // observation lookup ("next frame")
    float start, start_off;      // motion_start_times / _offset after cycling
    int pass_time;               // the pass_time handed to compute_humanoid_im_reset
    int cycled, cycle_cnt, recovery_cnt;
};

PHC_HD ImStepCtx im_post_prologue(const phc_motion_lib_t& lib, const phc_im_params_t& prm, const phc_sim_state_t& sim,
                                  const phc_im_buffers_t& buf, int64_t env, int64_t progress) {
    ImStepCtx c;
    const int64_t mid = motion_id_of(buf, env);
    c.start = buf.motion_start_times[env];
    c.start_off = buf.motion_start_times_offset[env];
    c.goff_rew = ld3(buf.global_offset + env * 3);
    c.goff = c.goff_rew;
    c.t_rew = motion_time(progress, prm.dt, c.start, c.start_off);  // humanoid_im.py:879
    c.t0 = c.t_rew;
    // pre_physics_step: _update_cycle_count / _update_recovery_count (humanoid_im.py:1076-1079,1110; humanoid_im_getup.py:198-201)
    c.cycle_cnt = buf.cycle_counter ? (buf.cycle_counter[env] > 1 ? buf.cycle_counter[env] - 1 : 0) : 0;
    c.recovery_cnt = buf.recovery_counter ? (buf.recovery_counter[env] > 1 ? buf.recovery_counter[env] - 1 : 0) : 0;
    c.cycled = 0;
    const bool pass_len = c.t_rew >= lib.motion_lengths[mid];
    c.pass_time = pass_len ? 1 : 0;
    if (prm.cycle_motion) {
        c.pass_time = (progress >= (int64_t)prm.max_episode_length - 1) ? 1 : 0;  // :1120,1124
        if (pass_len) {
            c.cycled = 1;
            c.start_off = neg_progress_time(progress, prm.dt);                       // :1126
            c.start = sample_time_interval(lib, mid, buf.cycle_phase[env]);         // :1127
            c.cycle_cnt = 60;                                                       // :1128
            const FrameRef fr = frame_ref(lib, mid, c.start);
            const V3 rp = ref_root_pos_lerp(lib, fr);                               // get_root_pos_smpl motion_lib_base.py:522-547
            const float* rs = sim.root_states + env * 13;
            c.goff.x = rs[0] - rp.x; c.goff.y = rs[1] - rp.y;                       // :1146 (z keeps its value)
            if (buf.offset_rand && prm.cycle_motion_xp) {                           // :1131-1132 (+ torch.rand(2): up to one metre)
                c.goff.x += buf.offset_rand[env * 2]; c.goff.y += buf.offset_rand[env * 2 + 1];
            } else if (buf.offset_rand && prm.zero_out_far && prm.zero_out_far_train) {   // :1133-1140
                float ox, oy;
                disk_offset(buf.offset_rand[env * 2], buf.offset_rand[env * 2 + 1], &ox, &oy);
                c.goff.x += ox; c.goff.y += oy;
            }
            c.t0 = motion_time(progress, prm.dt, c.start, c.start_off);             // :1148
        }
    }
    c.progress = progress;
    if (c.recovery_cnt > 0) c.progress = progress - 1;  // humanoid_im_getup.py:215
    c.t1 = motion_time(c.progress + 1, prm.dt, c.start, c.start_off);  // humanoid_im.py:752
    return c;
}

// zero_out_far gating of the task-obs reference (humanoid_im.py:783-797): beyond close_distance only the root target
// survives (the other bodies' targets collapse onto the current pose -> zero differences), beyond far_distance the root
// target becomes a far_distance-long direction.  Returns the distance (-> _point_goal).
PHC_HD float zero_out_far_ref(const phc_im_params_t& prm, int slot, const BodyState& body, const BodyState& root,
                              const BodyState& ref_root, BodyState* ref) {
    const float distance = norm(root.pos - ref_root.pos);
    if (distance > prm.close_distance) {
        if (slot >= 1) { ref->pos = body.pos; ref->rot = body.rot; }
        ref->vel = body.vel; ref->angvel = body.angvel;
    }
    if (distance > prm.far_distance && slot == 0) {
        V3 d = ref->pos - body.pos;
        ref->pos = v3(d.x / distance * prm.far_distance, d.y / distance * prm.far_distance, d.z / distance * prm.far_distance) + body.pos;
    }
    return distance;
}

// env.occl_training (humanoid_im.py:796-804,845-851): the reference state of an occluded tracked body is replaced by the simulated one before the
// task observation is formed -- all four fields for obs_v 4 / 5 / 6 / 8 / 9, the position (and the unused rotation) for obs_v 7
PHC_HD bool occluded(const phc_im_buffers_t& buf, const phc_im_params_t& prm, int64_t env, int slot) {
    return buf.occl_mask != nullptr && slot >= 0 && buf.occl_mask[env * prm.num_track_bodies + slot] != 0;
}
PHC_HD void occlude_ref(const phc_im_buffers_t& buf, const phc_im_params_t& prm, int64_t env, int slot, const BodyState& body, BodyState* rt) {
    if (!occluded(buf, prm, env, slot) || (prm.obs_v >= 1 && prm.obs_v <= 3)) return;
    if (prm.obs_v == 7) { rt->pos = body.pos; rt->rot = body.rot; }
    else *rt = body;
}

// env.fut_tracks (humanoid_im.py:741-747): the blocks of the T - 1 further reference samples behind the standard one (obs_v 6 / 7 / 9 lay the
// samples out time-major: one standard block each)
// Called at the very END of the lane functions with everything re-derived from memory (`body` / `root`: the simulated -- or, in a reset,
// the imposed -- state): inlined where the first block is formed, its registers stacked on top of the callers' live reference frames and
// cost the post-physics and reset kernels a wavefront of occupancy each (189 / 149 VGPRs vs 160 / 118) for a path no shipped config takes.
PHC_HD void task_obs_future_lane(const phc_motion_lib_t& lib, const phc_im_params_t& prm, int64_t mid, int64_t progress1, float start, float start_off,
                                 V3 goff, int slot, int j, const BodyState& body, const BodyState& root, float* tobs) {
    const int T = prm.num_traj_samples;
    if (T <= 1 || slot < 0 || !(prm.obs_v == 6 || prm.obs_v == 7 || prm.obs_v == 9)) return;
    const Q4 hroot = obs_root_rot(prm, root.rot);
    const Q4 hinv = calc_heading_quat_inv(hroot), h = calc_heading_quat(hroot);
    const int block = prm.num_task_obs / T;
    for (int k = 1; k < T; ++k) {
        PHC_NO_CONTRACT
        const float kts = (float)k * prm.traj_sample_timestep;
        BodyState r = ref_body(lib, frame_ref(lib, mid, motion_time_future(progress1, prm.dt, kts, start, start_off)), j);
        r.pos += goff;
        task_obs_lane(prm, slot, body, root, r, hinv, h, tobs + k * block);
    }
}

// post_physics_step for lane (env, j); `progress` is the already incremented progress_buf value
// (humanoid.py:1637).  Writes obs / AMP slices, returns the partials the caller reduces over the env.
PHC_HD RewardPartial im_post_lane(const phc_model_t& model, const phc_motion_lib_t& lib, const phc_im_params_t& prm,
                                  const phc_sim_state_t& sim, const phc_im_buffers_t& buf, int64_t env, int j, const ImStepCtx& c,
                                  const FrameTab& tab, const BodyState& body, const BodyState& root) {
    const int nb = model.num_bodies, nd = model.num_dof;
    RewardPartial rp;
    rp.pos = rp.rot = rp.vel = rp.angvel = rp.power = rp.dist = rp.root_dist = 0.f; rp.fallen = 0;
    const int64_t mid = motion_id_of(buf, env);
    if (j >= nb) {
        // R3: extended bodies of the full-body reward (humanoid_im.py:916-923) -- lane NB+e carries extended body e: current
        // pose = parent pose composed with a fixed offset, reference from the extended record slots; only the position
        // and rotation terms include them (body_vel / body_ang_vel stay NB wide).
        const int e = j - nb;
        if (e >= prm.num_ext_bodies || prm.track_body_reward) return rp;
        const BodyState par = load_body(sim.rigid_body_state, env, nb, prm.ext_parent[e]);
        BodyState cur;
        cur.pos = quat_rotate(par.rot, ld3(prm.ext_offset + 3 * e)) + par.pos;
        cur.rot = par.rot;
        BodyState ref;
        ref_body_ext(lib, frame_ref(lib, mid, c.cycled ? c.t_rew : c.t0), e, &ref.pos, &ref.rot);
        ref.pos += c.cycled ? c.goff_rew : c.goff;
        const V3 d = ref.pos - cur.pos;
        rp.pos = (d.x * d.x + d.y * d.y + d.z * d.z) / 3.0f;
        const float ang = quat_to_angle_axis(quat_mul(ref.rot, quat_conjugate(cur.rot)), nullptr);
        rp.rot = ang * ang;
        return rp;
    }
    const FrameRef fr0 = frame_ref(tab, c.t0), fr1 = frame_ref(tab, c.t1);
    PHC_PTL(2, env, j)
    // Order of work (round 3, profiles/r03_task/post_physics_timeline.txt): the two frame-record pairs are REQUESTED first, then everything that
    // needs the simulator state only -- self observation, AMP frame -- is computed and stored while they are on their way, then the blends,
    // the reward partials and the task observation.
    const BodyRaw q0 = ref_body_raw(lib, fr0, j), q1 = ref_body_raw(lib, fr1, j);
    JointPosRaw qj;
    const bool want_dof = buf.ref_dof_pos != nullptr && j >= 1;
    if (want_dof) qj = ref_joint_pos_raw(lib, fr1, j);
    PHC_PTL(3, env, j)
    // observations for the next policy step (humanoid_im.py:694-726)
    const Q4 hroot = obs_root_rot(prm, root.rot);
    Q4 hinv = calc_heading_quat_inv(hroot), h = calc_heading_quat(hroot);
    float* obs = buf.obs_buf + env * (int64_t)(prm.num_self_obs + prm.num_task_obs);
    if (prm.self_obs_v == 2 && buf.body_state_hist) self_obs_v2_lane(prm, buf.body_state_hist, nb, env, j, body, root, hinv, obs, true, false);
    else self_obs_lane(prm, nb, j, body, root, hinv, obs, (sim.force_sensor && prm.self_obs_v == 3) ? sim.force_sensor + env * (int64_t)(prm.num_force_sensors * 6) : nullptr, env);
    PHC_PTL(4, env, j)
    // AMP observation of this step -> slot 0 of the new history (humanoid_amp.py:204-209)
    {
        float* amp = buf.amp_obs_out + env * amp_env_stride(prm, buf);
        amp_obs_from_sim_lane(prm, sim, nb, nd, env, j, root, hinv, model.ints + 4 + 3 * PHC_MAX_BODIES, amp);
    }
    PHC_PTL(5, env, j)
    BodyState r0 = ref_body_blend(q0, fr0.blend), r1 = ref_body_blend(q1, fr1.blend);
    r0.pos += c.goff; r1.pos += c.goff;  // motion_lib_base.py:476
    // R1 / R5 partials
    rp = reward_partial(prm, env, nb, j, body, r0);
    if (c.cycled) {  // the reward was computed before the clip restarted: old time, old offset (humanoid.py:1644-1645)
        BodyState rr = ref_body(lib, frame_ref(lib, mid, c.t_rew), j);
        rr.pos += c.goff_rew;
        RewardPartial q = reward_partial(prm, env, nb, j, body, rr);
        rp.pos = q.pos; rp.rot = q.rot; rp.vel = q.vel; rp.angvel = q.angvel;
        if (j == 0) rp.root_dist = norm(body.pos - rr.pos);
    } else if (j == 0) {
        rp.root_dist = norm(body.pos - r0.pos);  // humanoid_im.py:892
    }
    if (prm.track_body_reward && prm.track_slot[j] < 0) rp.pos = rp.rot = rp.vel = rp.angvel = 0.f;   // full_body_reward False (:925-936)
    if (occluded(buf, prm, env, prm.track_slot[j])) { rp.dist = 0.f; rp.fallen = 0; }   // _compute_reset: ref = body for occluded bodies (:1180-1181)
    // R2 power partial: sum |tau * qdot| over this body's joint (humanoid_im.py:939-946)
    if (prm.power_reward && j >= 1) {
        int ds = model.ints[4 + 3 * PHC_MAX_BODIES + j];
        const float* d = sim.dof_state + (env * nd + ds) * 2;
        const float* f = sim.dof_force + env * nd + ds;
        rp.power = fabsf(f[0] * d[1]);
        if (prm.dofs_per_joint != 1) rp.power += fabsf(f[1] * d[3]) + fabsf(f[2] * d[5]);
    }
    PHC_PTL(6, env, j)
    int slot = prm.track_slot[j];
    if (slot >= 0) {
        BodyState rt = r1;
        if (prm.zero_out_far && !(prm.obs_v >= 1 && prm.obs_v <= 3)) {   // (:783: versions 4 / 5 / 6 / 8 / 9, and 7 at :829)
            BodyState rroot = (j == 0) ? r1 : ref_body(lib, fr1, 0);
            if (j != 0) rroot.pos += c.goff;
            const float dist = zero_out_far_ref(prm, slot, body, root, rroot, &rt);
            if (j == 0 && buf.point_goal) buf.point_goal[env] = dist;  // :792
        }
        occlude_ref(buf, prm, env, slot, body, &rt);
        V3 jd = v3(0.f, 0.f, 0.f), jv, rjd = jd, rjv;
        if (prm.obs_v == 2 && j >= 1) {   // humanoid_im.py:775-778: the joint of a tracked body, simulator vs reference at t + dt
            ld_joint_state(sim, nd, env, model.ints[4 + 3 * PHC_MAX_BODIES + j], prm.dofs_per_joint, &jd, &jv);
            ref_joint(lib, fr1, j, &rjd, &rjv);
        }
        task_obs_lane(prm, slot, body, root, rt, hinv, h, obs + prm.num_self_obs, &jd, &rjd);
    }
    PHC_PTL(7, env, j)
    // side-effect buffers of _compute_task_obs (humanoid_im.py:855-868)
    if (buf.ref_body_pos) st3(buf.ref_body_pos + (env * nb + j) * 3, r1.pos);
    if (buf.ref_body_rot) st4(buf.ref_body_rot + (env * nb + j) * 4, r1.rot);
    if (buf.ref_body_vel) st3(buf.ref_body_vel + (env * nb + j) * 3, r1.vel);
    if (want_dof)
        st_joint(buf.ref_dof_pos + env * nd + model.ints[4 + 3 * PHC_MAX_BODIES + j], prm.dofs_per_joint, ref_joint_pos_blend(lib, qj, fr1.blend));
    PHC_PTL(8, env, j)
    if (prm.num_traj_samples > 1)   // env.fut_tracks: the further reference samples' blocks (see task_obs_future_lane)
        task_obs_future_lane(lib, prm, mid, c.progress + 1, c.start, c.start_off, c.goff, prm.track_slot[j], j, load_body(sim.rigid_body_state, env, nb, j),
                             load_body(sim.rigid_body_state, env, nb, 0), buf.obs_buf + env * (int64_t)(prm.num_self_obs + prm.num_task_obs) + prm.num_self_obs);
    return rp;
}

// Per-env epilogue (lane 0) once the partials are reduced: reward (humanoid_im.py:1524-1554, 939-946),
// reset / terminate (:1117-1190, 1581-1608), progress write-back.
PHC_HD void im_post_finalize(const phc_motion_lib_t& lib, const phc_im_params_t& prm, const phc_im_buffers_t& buf, int nb,
                             int64_t env, const ImStepCtx& c, int64_t progress, float s_pos, float s_rot, float s_vel, float s_angvel,
                             float s_power, float s_dist, float root_dist, float prev_point_goal, int any_fallen, int n_reset_bodies) {
    float J = (float)nb, JE = (float)(nb + prm.num_ext_bodies);  // position / rotation means include the extended bodies
    if (prm.track_body_reward) J = JE = (float)prm.num_track_bodies;
    float r_pos = expf(-prm.k_pos * (s_pos / JE));
    float r_rot = expf(-prm.k_rot * (s_rot / JE));
    float r_vel = expf(-prm.k_vel * (s_vel / J));
    float r_ang = expf(-prm.k_ang_vel * (s_angvel / J));
    float rew = prm.w_pos * r_pos + prm.w_rot * r_rot + prm.w_vel * r_vel + prm.w_ang_vel * r_ang;
    const int nraw = prm.power_reward ? 5 : 4;
    float* raw = buf.reward_raw + env * nraw;
    if (prm.zero_out_far) {
        // compute_point_goal_reward + half-weight imitation reward inside the transition radius (humanoid_im.py:890-905,1558-1562)
        const float pg = fminf(prev_point_goal - root_dist, 1.0f / 3.0f) * 9.0f;
        const bool far = root_dist > 0.25f;  // transition_distance :891
        raw[0] = pg + (far ? 0.f : r_pos * 0.5f);
        raw[1] = far ? 0.f : r_rot * 0.5f; raw[2] = far ? 0.f : r_vel * 0.5f; raw[3] = far ? 0.f : r_ang * 0.5f;
        rew = pg + (far ? 0.f : rew * 0.5f);
    } else {
        raw[0] = r_pos; raw[1] = r_rot; raw[2] = r_vel; raw[3] = r_ang;
    }
    if (prm.power_reward) {
        float pr = -prm.power_coefficient * s_power;
        if (progress <= 3) pr = 0.f;
        rew += pr;
        raw[4] = pr;
    }
    buf.rew_buf[env] = rew;
    // _compute_reset
    int fallen = any_fallen;
    if (prm.use_mean_termination) {
        // torch.norm(...).mean(-1, keepdim=True) > termination_distance[0]  (humanoid_im.py:1586)
        float mean = s_dist / (float)n_reset_bodies;
        fallen = mean > prm.termination_distances[prm.first_reset_body] ? 1 : 0;
    }
    int64_t terminated = 0;
    if (prm.enable_early_termination) {
        if (!(progress > 1)) fallen = 0;
        if (prm.disable_collision_check) fallen = 0;
        terminated = fallen ? 1 : 0;
    }
    int64_t reset = c.pass_time ? 1 : terminated;
    if (!c.pass_time && c.cycle_cnt > 0) { reset = 0; terminated = 0; }  // humanoid_im.py:1186-1188
    if (c.recovery_cnt > 0) { reset = 0; terminated = 0; }               // humanoid_im_getup.py:212-214
    buf.terminate_buf[env] = terminated;
    buf.reset_buf[env] = reset;
    if (reset && buf.reset_list) {
        // PHC_RESET_SUBLISTS sub-lists (envs of workgroup b go to sub-list b % 16) keep the atomics off a single hot address
        const int sub = (int)((env >> 3) & (PHC_RESET_SUBLISTS - 1));
        const int i = phc_atomic_inc(buf.reset_count + (buf.reset_slot * PHC_RESET_SUBLISTS + sub) * PHC_RESET_COUNT_STRIDE);
        buf.reset_list[sub * buf.reset_sublist_cap + i] = (int32_t)env;
    }
    buf.progress_buf[env] = c.progress;
    if (buf.cycle_counter) buf.cycle_counter[env] = c.cycle_cnt;
    if (buf.recovery_counter) buf.recovery_counter[env] = c.recovery_cnt;
    if (c.cycled) {
        buf.motion_start_times[env] = c.start;
        buf.motion_start_times_offset[env] = c.start_off;
        st3(buf.global_offset + env * 3, c.goff);
    }
}

// Reset of one env, lane j: HumanoidIm._reset_envs (humanoid.py:585-621; humanoid_amp.py:378-398,508-528,
// 559-637; humanoid_im.py:955-1023).  `t` = sampled start time.
// `parts`: bit 0 = the state, the self observation and the per-env scalars; bit 1 = the task observation and the ref_* side buffers.  The two
// halves share nothing but the reference frame at t (the imposed state IS that frame), so the kernel gives each its own lane group: the reset
// launch lasts as long as its longest dependent chain (profiles/r03_task/reset_group_timeline.txt).
PHC_HD void im_reset_lane(const phc_model_t& model, const phc_motion_lib_t& lib, const phc_im_params_t& prm,
                          const phc_sim_state_t& sim, const phc_im_buffers_t& buf, int64_t env, int j, float t, bool clear_reset_flag,
                          int parts = 3) {
    const int nb = model.num_bodies, nd = model.num_dof;
    const int64_t mid = motion_id_of(buf, env);
    const bool p_state = (parts & 1) != 0, p_task = (parts & 2) != 0;
    if (j < nb) {
        const FrameRef fr = frame_ref(lib, mid, t);
        BodyState rs = ref_body(lib, fr, j);     // global offset was just zeroed (humanoid_im.py:956-957)
        BodyState root = (j == 0) ? rs : ref_body(lib, fr, 0);
        // _set_env_state (humanoid_amp.py:605-637)
        if (p_state) {
        store_body(sim.rigid_body_state, env, nb, j, rs);
        if (sim.contact_force) st3(sim.contact_force + (env * nb + j) * 3, v3(0.f, 0.f, 0.f));  // humanoid.py:619
        if (j == 0) {
            float* r = sim.root_states + env * 13;
            st3(r, rs.pos); st4(r + 3, rs.rot); st3(r + 7, rs.vel); st3(r + 10, rs.angvel);
        } else {
            V3 dp, dv;
            ref_joint(lib, fr, j, &dp, &dv);
            const int ds = model.ints[4 + 3 * PHC_MAX_BODIES + j];
            float* d = sim.dof_state + (env * nd + ds) * 2;
            d[0] = dp.x; d[1] = dv.x;
            if (prm.dofs_per_joint != 1) { d[2] = dp.y; d[3] = dv.y; d[4] = dp.z; d[5] = dv.z; }
            st_joint(sim.pd_target + env * nd + ds, prm.dofs_per_joint, dp);  // set_dof_position_target_tensor_indexed(dof_pos) humanoid.py:605
            if (sim.dof_force) st_joint(sim.dof_force + env * nd + ds, prm.dofs_per_joint, v3(0.f, 0.f, 0.f));
        }
        }
        // observations of the reset envs (humanoid.py:595 -> humanoid_im.py:694-726), progress_buf == 0
        const Q4 hroot = obs_root_rot(prm, root.rot);
        Q4 hinv = calc_heading_quat_inv(hroot), h = calc_heading_quat(hroot);
        float* obs = buf.obs_buf + env * (int64_t)(prm.num_self_obs + prm.num_task_obs);
        if (p_state) {
        // (reset envs: the sensor tensor keeps its last reading until the next simulate call, as gym's does -- humanoid.py:1463)
        if (prm.self_obs_v == 2 && buf.body_state_hist) self_obs_v2_lane(prm, buf.body_state_hist, nb, env, j, rs, root, hinv, obs, false, true);   // humanoid.py:592-595
        else self_obs_lane(prm, nb, j, rs, root, hinv, obs, (sim.force_sensor && prm.self_obs_v == 3) ? sim.force_sensor + env * (int64_t)(prm.num_force_sensors * 6) : nullptr, env);
        }
        if (p_task) {
        const float t1 = motion_time(1, prm.dt, t, 0.f);
        const FrameRef fr1 = frame_ref(lib, mid, t1);
        BodyState r1 = ref_body(lib, fr1, j);
        V3 goff = v3(0.f, 0.f, 0.f);
        if (buf.offset_rand && prm.zero_out_far && prm.zero_out_far_train)   // the far-away start (humanoid_im.py:966-980): set AFTER the state
            disk_offset(buf.offset_rand[env * 2], buf.offset_rand[env * 2 + 1], &goff.x, &goff.y);   // was imposed, seen by the observations
        r1.pos += goff;
        int slot = prm.track_slot[j];
        if (slot >= 0) {
            BodyState rt = r1;
            if (prm.zero_out_far && !(prm.obs_v >= 1 && prm.obs_v <= 3)) {
                BodyState rroot = (j == 0) ? r1 : ref_body(lib, fr1, 0);
                if (j != 0) rroot.pos += goff;
                const float dist = zero_out_far_ref(prm, slot, rs, root, rroot, &rt);
                if (j == 0 && buf.point_goal) buf.point_goal[env] = dist;
            }
            occlude_ref(buf, prm, env, slot, rs, &rt);
            V3 jd = v3(0.f, 0.f, 0.f), jv, rjd = jd, rjv;
            if (prm.obs_v == 2 && j >= 1) { ref_joint(lib, fr, j, &jd, &jv); ref_joint(lib, fr1, j, &rjd, &rjv); }   // (the imposed state is the reference at t)
            task_obs_lane(prm, slot, rs, root, rt, hinv, h, obs + prm.num_self_obs, &jd, &rjd);
        }
        if (buf.ref_body_pos) st3(buf.ref_body_pos + (env * nb + j) * 3, r1.pos);
        if (buf.ref_body_rot) st4(buf.ref_body_rot + (env * nb + j) * 4, r1.rot);
        if (buf.ref_body_vel) st3(buf.ref_body_vel + (env * nb + j) * 3, r1.vel);
        if (buf.ref_dof_pos && j >= 1) {
            V3 dp, dv;
            ref_joint(lib, fr1, j, &dp, &dv);
            st_joint(buf.ref_dof_pos + env * nd + model.ints[4 + 3 * PHC_MAX_BODIES + j], prm.dofs_per_joint, dp);
        }
        if (prm.num_traj_samples > 1) {   // env.fut_tracks (see task_obs_future_lane): the imposed state is the reference at t
            const FrameRef frt = frame_ref(lib, mid, t);
            V3 go = v3(0.f, 0.f, 0.f);
            if (buf.offset_rand && prm.zero_out_far && prm.zero_out_far_train) disk_offset(buf.offset_rand[env * 2], buf.offset_rand[env * 2 + 1], &go.x, &go.y);
            task_obs_future_lane(lib, prm, mid, 1, t, 0.f, go, prm.track_slot[j], j, ref_body(lib, frt, j), ref_body(lib, frt, 0),
                                 buf.obs_buf + env * (int64_t)(prm.num_self_obs + prm.num_task_obs) + prm.num_self_obs);
        }
        }
    }
    if (j == 0 && p_state) {
        buf.motion_start_times[env] = t;           // humanoid_amp.py:524
        buf.motion_start_times_offset[env] = 0.f;  // humanoid_im.py:956
        V3 goff = v3(0.f, 0.f, 0.f);
        const bool far_start = buf.offset_rand && prm.zero_out_far && prm.zero_out_far_train;
        if (far_start) disk_offset(buf.offset_rand[env * 2], buf.offset_rand[env * 2 + 1], &goff.x, &goff.y);
        st3(buf.global_offset + env * 3, goff);
        buf.progress_buf[env] = 0; buf.terminate_buf[env] = 0;  // humanoid.py:616-618
        if (clear_reset_flag) buf.reset_buf[env] = 0;
        if (buf.cycle_counter) buf.cycle_counter[env] = far_start ? prm.zero_out_far_steps : 0;        // humanoid_im.py:960,980
        if (buf.recovery_counter) buf.recovery_counter[env] = 0;  // humanoid_im_getup.py:155
    }
}

// Reset of one env that KEEPS the simulator state the caller put there (HumanoidImGetup fall / recovery episodes,
// humanoid_im_getup.py:136-196): the shared tail of _reset_envs (humanoid.py:585-621) -- PD target := joint positions,
// progress / reset / terminate / contact cleared -- then _compute_observations(env_ids) against the reference at the
// env's unchanged motion clock with progress 0, and (fill_history) _init_amp_obs_default (humanoid_amp.py:569-573).
// rigid_body_state of the env must already be current (phc_refresh_body_state_indexed).
PHC_HD void im_reset_from_state_lane(const phc_model_t& model, const phc_motion_lib_t& lib, const phc_im_params_t& prm,
                                     const phc_sim_state_t& sim, const phc_im_buffers_t& buf, int64_t env, int j, int fill_history) {
    const int nb = model.num_bodies, nd = model.num_dof;
    const int64_t mid = motion_id_of(buf, env);
    const int S = prm.num_amp_obs_steps, A = prm.num_amp_obs_per_step;
    if (j < nb) {
        BodyState body = load_body(sim.rigid_body_state, env, nb, j);
        BodyState root = load_body(sim.rigid_body_state, env, nb, 0);
        if (sim.contact_force) st3(sim.contact_force + (env * nb + j) * 3, v3(0.f, 0.f, 0.f));
        if (j >= 1) {
            const int ds = model.ints[4 + 3 * PHC_MAX_BODIES + j];
            V3 dp, dv;
            ld_joint_state(sim, nd, env, ds, prm.dofs_per_joint, &dp, &dv);
            st_joint(sim.pd_target + env * nd + ds, prm.dofs_per_joint, dp);  // humanoid.py:605
        }
        const V3 goff = ld3(buf.global_offset + env * 3);
        const float t1 = motion_time(1, prm.dt, buf.motion_start_times[env], buf.motion_start_times_offset[env]);
        const FrameRef fr1 = frame_ref(lib, mid, t1);
        BodyState r1 = ref_body(lib, fr1, j);
        r1.pos += goff;
        const Q4 hroot = obs_root_rot(prm, root.rot);
    Q4 hinv = calc_heading_quat_inv(hroot), h = calc_heading_quat(hroot);
        float* obs = buf.obs_buf + env * (int64_t)(prm.num_self_obs + prm.num_task_obs);
        if (prm.self_obs_v == 2 && buf.body_state_hist) self_obs_v2_lane(prm, buf.body_state_hist, nb, env, j, body, root, hinv, obs, false, true);
        else self_obs_lane(prm, nb, j, body, root, hinv, obs, (sim.force_sensor && prm.self_obs_v == 3) ? sim.force_sensor + env * (int64_t)(prm.num_force_sensors * 6) : nullptr, env);
        int slot = prm.track_slot[j];
        if (slot >= 0) {
            BodyState rt = r1;
            if (prm.zero_out_far && !(prm.obs_v >= 1 && prm.obs_v <= 3)) {
                BodyState rroot = (j == 0) ? r1 : ref_body(lib, fr1, 0);
                if (j != 0) rroot.pos += goff;
                const float dist = zero_out_far_ref(prm, slot, body, root, rroot, &rt);
                if (j == 0 && buf.point_goal) buf.point_goal[env] = dist;
            }
            occlude_ref(buf, prm, env, slot, body, &rt);
            V3 jd = v3(0.f, 0.f, 0.f), jv, rjd = jd, rjv;
            if (prm.obs_v == 2 && j >= 1) {
                ld_joint_state(sim, nd, env, model.ints[4 + 3 * PHC_MAX_BODIES + j], prm.dofs_per_joint, &jd, &jv);
                ref_joint(lib, fr1, j, &rjd, &rjv);
            }
            task_obs_lane(prm, slot, body, root, rt, hinv, h, obs + prm.num_self_obs, &jd, &rjd);
        }
        if (buf.ref_body_pos) st3(buf.ref_body_pos + (env * nb + j) * 3, r1.pos);
        if (buf.ref_body_rot) st4(buf.ref_body_rot + (env * nb + j) * 4, r1.rot);
        if (buf.ref_body_vel) st3(buf.ref_body_vel + (env * nb + j) * 3, r1.vel);
        if (buf.ref_dof_pos && j >= 1) {
            V3 dp, dv;
            ref_joint(lib, fr1, j, &dp, &dv);
            st_joint(buf.ref_dof_pos + env * nd + model.ints[4 + 3 * PHC_MAX_BODIES + j], prm.dofs_per_joint, dp);
        }
        // _compute_amp_observations(env_ids) -> slot 0; _init_amp_obs_default copies it into every history slot
        float* amp = buf.amp_obs_out + env * amp_env_stride(prm, buf);
        const int nfill = fill_history ? S : 1;
        for (int k = 0; k < nfill; ++k)
            amp_obs_from_sim_lane(prm, sim, nb, nd, env, j, root, hinv, model.ints + 4 + 3 * PHC_MAX_BODIES, amp + k * A);
        if (prm.num_traj_samples > 1)   // env.fut_tracks (see task_obs_future_lane)
            task_obs_future_lane(lib, prm, mid, 1, buf.motion_start_times[env], buf.motion_start_times_offset[env], ld3(buf.global_offset + env * 3),
                                 prm.track_slot[j], j, load_body(sim.rigid_body_state, env, nb, j), load_body(sim.rigid_body_state, env, nb, 0),
                                 buf.obs_buf + env * (int64_t)(prm.num_self_obs + prm.num_task_obs) + prm.num_self_obs);
    }
    if (j == 0) {
        buf.progress_buf[env] = 0; buf.terminate_buf[env] = 0; buf.reset_buf[env] = 0;  // humanoid.py:616-618
    }
}

// AMP history frame k of a reset env: slot 0 from the (just imposed) state == reference at t; slots 1..S-1 from
// the reference motion at t - k dt (humanoid_amp.py:559-603).  Independent of im_reset_lane, so the kernel gives
// every (env, k) pair its own 32-lane group.
PHC_HD void im_reset_amp_lane(const phc_motion_lib_t& lib, const phc_im_params_t& prm, const phc_im_buffers_t& buf, int nb,
                              int64_t env, int j, float t, int k) {
    const int A = prm.num_amp_obs_per_step;
    float* amp = buf.amp_obs_out + env * amp_env_stride(prm, buf);
    amp_obs_from_ref_lane(lib, prm, nb, j, motion_id_of(buf, env), history_time(t, prm.dt, k), amp + k * A);
}
// The same from the per-frame table (phc_im_params_t.amp_ref_table, row f = the build for the pair (f, f + 1) at blend 0): the history times of a reset
// fall on frames of the clip up to the rounding of the blend factor b -- exactly 0 for ~5 lookups in 6: the row is then the full build bit for bit --
// so for b <= PHC_AMP_TABLE_BLEND_TOL frame k's observation is (1 - b) T[f0] + b T[f1]; any other lookup (a few % land just BELOW the next frame,
// b ~ 1: the reference's slerp then still averages the previous pair, which row f1 does not hold) is built in full.
// Error of the first-order form: T[f1] is the build for the pair (f1, f1 + 1), not the pair (f0, f1) at blend 1, so the deviation from the full
// build is bounded by b * |T[f1] - build(f0, f1, 1)| + O(b^2) <= 1e-4 x (how far a column moves between consecutive frames): <= 1e-4 even for a
// velocity column that changes by a full unit per frame, i.e. inside the north-star bar by construction; measured maximum 1.2e-6 (blend factors
// above ~4e-5 need t / dt > 300, tests/test_env_gpu.py::test_reset_amp_history_from_the_per_frame_table_equals_the_lookups compares at 2e-5).
// `nl` lanes.  All loads of a chunk are requested before its stores.
#define PHC_AMP_TABLE_BLEND_TOL 1e-4f
PHC_HD void amp_obs_from_table_lane(const phc_motion_lib_t& lib, const phc_im_params_t& prm, int nb, int j, int nl, int64_t mid, float t, float* a) {
    const int A = prm.num_amp_obs_per_step, E = prm.num_amp_obs_extra, W = A - E;
    const FrameRef fr = frame_ref(lib, mid, t);
    const float b = fr.blend;
    if (j < nb && E > 0 && prm.amp_obs_extra) obs_extra_lane(prm.amp_obs_extra + mid * E, E, j, nb, a + W);
    if (!(b <= PHC_AMP_TABLE_BLEND_TOL)) {
        amp_obs_from_frames_lane(lib, prm, nb, j, fr, a);
        return;
    }
    const float* r0 = prm.amp_ref_table + fr.f0 * (int64_t)W;
    const float* r1 = prm.amp_ref_table + fr.f1 * (int64_t)W;
    const float s0 = 1.0f - b;
    for (int c0 = j; c0 < W; c0 += 8 * nl) {
        float u[8], v[8];
        for (int i = 0; i < 8; ++i) { const int c = c0 + i * nl; if (c < W) { u[i] = r0[c]; v[i] = r1[c]; } }
        for (int i = 0; i < 8; ++i) { const int c = c0 + i * nl; if (c < W) a[c] = s0 * u[i] + b * v[i]; }
    }
}
PHC_HD void im_reset_amp_table_lane(const phc_motion_lib_t& lib, const phc_im_params_t& prm, const phc_im_buffers_t& buf, int nb,
                                    int64_t env, int j, int nl, float t, int k) {
    amp_obs_from_table_lane(lib, prm, nb, j, nl, motion_id_of(buf, env), history_time(t, prm.dt, k),
                            buf.amp_obs_out + env * amp_env_stride(prm, buf) + k * prm.num_amp_obs_per_step);
}

}  // namespace phc
