/* phc_amd.h -- C ABI of libphc_amd.so: the MI355X-native humanoid env-step hot path.
 *
 * The reference (ZhengyiLuo/PHC) is pure Python; its boundary for this path is the
 * Isaac Gym tensor API seen by the task plus the task's own
 * @torch.jit.script functions.  Each entry point below names the reference interface it
 * replaces (paths relative to the reference root).  All pointers are DEVICE pointers
 * (HBM) unless a field says "host"; `stream` is a hipStream_t passed as void*.  No torch
 * types cross this boundary: the host side (phc_amd/*.py, ctypes) hands over
 * tensor.data_ptr() values and the current stream.
 *
 * Return value: 0 on success, a negative PHC_E* code on argument errors, or the positive
 * hipError_t of a failed launch.
 *
 * Layouts are the reference's (Isaac Gym) layouts so that torch views with the same
 * shapes as `humanoid.py:201-235` alias these buffers:
 *   root_states      f32 [N,13]      pos3 quat4(xyzw) linvel3 angvel3        (S1)
 *   dof_state        f32 [N,D,2]     (pos, vel) pairs; spherical joints = exp-map (S2)
 *   rigid_body_state f32 [N,NB,13]   pos3 quat4 linvel3 angvel3 per body      (S3)
 *   contact_force    f32 [N,NB,3]                                             (S4)
 *   dof_force        f32 [N,D]                                                (S5)
 */
#ifndef PHC_AMD_H
#define PHC_AMD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHC_ABI_VERSION 37
#define PHC_MAX_BODIES 64   /* bodies (incl. extended reference bodies) per articulation; also the slot count of the model tables */
#define PHC_EINVAL (-1)
#define PHC_EUNSUPPORTED (-2)
#define PHC_RESET_SUBLISTS 16
#define PHC_RESET_COUNT_STRIDE 32   /* ints between two sub-list counters: one 128-byte line each */

/* Compiled articulation (phc_amd/model.py: ArticulationModel.pack()).  Replaces
 * gym.load_asset / get_actor_dof_properties / get_actor_rigid_body_properties
 * (phc/env/tasks/humanoid.py:768-990,1093-1106). */
typedef struct {
    int32_t num_bodies;       /* NB <= PHC_MAX_BODIES; up to 32 bodies an env takes 32 lanes (two envs per wavefront), above one wavefront */
    int32_t num_dof;          /* D */
    int32_t max_level;        /* tree depth */
    int32_t num_contact_pts;
    const int32_t* ints;      /* packed tables, see model.py pack() */
    const float* floats;
    int32_t num_collision_pairs; /* body pairs that may collide (listed after the int tables); capacity 576 (NB <= 32) / 1152, see phc_sim_step */
    /* Per-env body shapes (robot.has_shape_variation, humanoid.py:726-766,824-866): K compiled articulations with ONE topology (names,
     * parents, joint types) but their own link offsets, masses, inertias, gains, collision geometry.  ints / floats then hold K blocks
     * int_stride / float_stride elements apart, phc_sim_state_t.env_shape picks an env's block, and the scalar fields above are the
     * maxima over the shapes.  num_shapes <= 1: one model for all envs. */
    int32_t num_shapes;
    int32_t int_stride;
    int32_t float_stride;
    int32_t max_body_contact_pts; /* ABI 36: the largest number of ground-contact points any one body carries (int table 9; the maximum over the shapes).  The per-point
                                   * masks of the solver are finite: phc_sim_step returns PHC_EUNSUPPORTED for contact_model 1 above 32 (c_active / c_removed) and for
                                   * inertia_lag above 64 (c_touch: the tail points of a lagged sub-step would lose their force while their impedance stays in I^A) */
} phc_model_t;

/* Flat reference-motion buffer.  Replaces MotionLibBase's gts/grs/lrs/gvs/gavs/dvs tensors
 * (phc/utils/motion_lib_base.py:300-318; H1/G1: motion_lib_real.py:196-215).  One record per frame, fields contiguous:
 *   [pos (NB+E)*3 | rot (NB+E)*4 | vel NB*3 | angvel NB*3 | joints | dof_vel | pad]
 *   joints / dof_vel = local_rot NB*4 / (NB-1)*3 for spherical models (dofs_per_joint 3: dof_pos is the exp-map of the
 *   slerped local rotation), dof_pos ND / ND for revolute models (dofs_per_joint 1: linear blend, motion_lib_real.py:285-291);
 *   E = num_ext_bodies reference-only bodies (H1 hands / head) appended to the position and rotation blocks.
 * One lookup touches one contiguous run instead of six (ten) separate tensors. */
typedef struct {
    const float* frames;              /* [F, frame_stride] */
    int64_t num_frames_total;         /* F */
    int32_t frame_stride;             /* floats per frame record */
    int32_t num_bodies;
    int32_t num_motions;              /* one clip per env (motion_lib_base.py:205-216) */
    const float* motion_lengths;      /* [M] seconds */
    const float* motion_dt;           /* [M] */
    const int64_t* motion_num_frames; /* [M] */
    const int64_t* length_starts;     /* [M] first frame of each clip */
    int32_t num_ext_bodies;           /* E (0 for SMPL) */
    int32_t dofs_per_joint;           /* 3 spherical (0 is read as 3), 1 revolute */
} phc_motion_lib_t;

/* Simulator-owned state tensors (S1-S5, S8). */
typedef struct {
    int32_t num_envs;
    float* root_states;
    float* dof_state;
    float* rigid_body_state;
    float* contact_force;
    float* dof_force;
    float* pd_target;                 /* [N,D] set_dof_position_target_tensor (humanoid.py:1559-1560) */
    float* force_sensor;              /* [N,S,6] S6: gym.acquire_force_sensor_tensor (humanoid.py:183-190), force then torque of each sensor in
                                         the sensor body's LOCAL frame (create_humanoid_force_sensors :1031-1040: identity pose on the
                                         body, use_world_frame False); nullable.  Written by phc_sim_step at the end of the step: the net
                                         ground-contact wrench on the body about its origin (PhysX's reading is closed: parity unpinned) */
    const int32_t* env_shape;         /* [N] shape block of each env (phc_model_t.num_shapes > 1); nullable = block 0 */
    const float* pd_ref;              /* [N,D] env.res_action (humanoid_im.py:1094-1099): non-null = residual actions, pd_tar = pd_ref +
                                         scale * action limited to the current joint position +- pi / 2; the caller passes the task's
                                         ref_dof_pos (phc_im_buffers_t.ref_dof_pos, the reference pose of the next frame); nullable */
} phc_sim_state_t;

/* Solver parameters.  Replaces gymapi.SimParams as filled by parse_sim_params
 * (phc/run_hydra.py:76-111, phc/data/cfg/sim/default_sim.yaml). */
typedef struct {
    float sim_dt;                     /* physx.step_dt, 1/60 */
    int32_t substeps;                 /* 2 */
    int32_t control_freq_inv;         /* simulate() calls per env step, 2 */
    float gravity_z;                  /* -9.81 */
    float contact_stiffness;          /* penalty ground contact, N/m */
    float contact_damping;            /* N s/m */
    float friction;                   /* mu (plane static=dynamic=1, env_im.yaml) */
    float friction_viscous;           /* regularised Coulomb: tangential damping N s/m before the cone clamp */
    float angular_damping;            /* asset_options.angular_damping 0.01 (humanoid.py:819-822) */
    float max_angular_velocity;       /* 100 */
    float contact_offset;             /* 0.02: contact activates this far above the plane with zero force */
    int32_t control_mode;             /* 0: implicit position drive (`isaac_pd`, S8); 1: `pd` -- explicit torque
                                         clip(kp (target - q) - kd qd, +-limit) recomputed every simulate call (S9,
                                         humanoid.py:1575-1599,1608-1616; robot_control.yaml); 2: as 1 but only the spring
                                         term is held, the damper follows the joint rate implicitly (stable on light
                                         links).  1 and 2: revolute models only. */
    float limit_stiffness;            /* joint-limit penalty spring N m/rad outside [lower, upper] (revolute models; 0 = off) */
    float limit_damping;              /* N m s/rad */
    int32_t self_collision;           /* 1: penalty contact between the collision capsules of non-adjacent bodies (robot.has_self_collision,
                                         humanoid.py:1205-1226), explicit in time with k = self_stiffness_scale * mu / dt^2 per pair
                                         (mu = reduced mass) and damping ratio self_damping_ratio */
    float self_stiffness_scale;       /* <= 1 (explicit stability bound is 4); default 0.25 */
    float self_damping_ratio;         /* default 0.5 */
    int32_t lane_mapping;             /* stepper thread mapping: one body per lane (32 lanes per env, 2 envs per wavefront; NB > 32: 64 lanes, 1 env).
                                         0 / 1 = the product kernel (two wavefronts per SIMD); 3 = the same kernel compiled for three wavefronts per
                                         SIMD (SMPL-family penalty kernel only; measured slower at every env count, profiles/r04_stepper/occupancy_2_vs_3_waves_per_simd.txt --
                                         an experiment knob); other values PHC_EUNSUPPORTED (2 was the two-bodies-per-lane kernel, removed in ABI 31) */
    int32_t num_force_sensors;        /* S <= 4: force sensors (env.force_sensor_joints, default L_Ankle / R_Ankle, humanoid.py:268) */
    int32_t force_sensor_body[4];     /* body id of each sensor */
    /* ---- ABI 34: ground-contact model (solver.contact) ----
     * 0 = penalty spring-damper with regularised Coulomb friction (contact_stiffness / contact_damping / friction_viscous above).
     * 1 = rigid ("tgs"): what parse_sim_params asks PhysX for (phc/run_hydra.py:88-91 solver_type 1 = TGS, num_position_iterations 4,
     *     phc/data/cfg/sim/default_sim.yaml contact_offset / bounce_threshold_velocity / max_depenetration_velocity) restated for the
     *     articulated-body recursion: a velocity-level unilateral constraint per contact point, u_n(t + dt) >= v_target, enforced through
     *     the impedance contact_impedance (N s/m; its compliance 1 / c is the constraint's CFM), v_target = min(depth / dt,
     *     max_depenetration_velocity) for a penetrating point, -gap / dt for a speculative one inside contact_offset, and
     *     -restitution * u_n(t) when the approach speed exceeds bounce_threshold_velocity; Coulomb cone |F_t| <= mu F_n evaluated on the
     *     END-of-step normal force and slip velocity; active set and cone are found by contact_iterations fixed-point passes, each an exact
     *     O(n) articulated-body solve of the whole tree with the contact impedances of the previous pass. */
    int32_t contact_model;
    int32_t contact_iterations;       /* passes per sub-step in model 1 (PhysX num_position_iterations: 4); >= 2 (a single pass never re-evaluates the active set
                                         or the friction cone on predicted velocities: PHC_EINVAL) */
    float contact_impedance;          /* model 1: N s/m per contact point (default 1e5: dt * c = 833 kg of apparent mass per point) */
    float max_depenetration_velocity; /* m/s (default_sim.yaml: 10) */
    float bounce_threshold_velocity;  /* m/s (default_sim.yaml: 0.2) */
    float restitution;                /* plane / shape restitution (env_im.yaml plane.restitution: 0) */
    /* ---- ABI 35 ---- */
    int32_t inertia_lag;              /* 1 (solver.inertia_lag; contact_model 0 only): the articulated inertias I^A and the joint-space inverses are formed in the FIRST
                                         sub-step of every simulate() call and kept over its other sub-steps, which only redo the bias-force recursion
                                         (forces, drives and velocity products at the current state; a third of the backward sweep's arithmetic).  The kept
                                         I^A is O(dt) old, the velocities it produces differ by O(dt^2) per sub-step: the scheme stays first-order consistent
                                         (tests/test_dynamics.py: dt-convergence with the switch on).  A ground-contact point that starts to touch in a
                                         lagged sub-step joins at the next fresh one (its impedance is not in the kept I^A).  0 = every sub-step fresh.
                                         The host side (HumanoidIm) sets 1 by default for contact_model 0 since round 6: pinned against the double-precision build
                                         of this recursion (oracle/hostemu/hostemu64.cpp; tests/test_stepper_options.py). */
    int32_t force_average;            /* 1 (solver.force_average): contact_force (S4) and dof_force (S5) are the MEANS over all sub-steps of the env step
                                         (what a power / contact reward integrates over the control interval); 0 = the last sub-step's values -- what Isaac
                                         Gym's tensors hold after simulate() with substeps > 1, as far as its documentation says (humanoid.py:185,193-194) */
} phc_sim_params_t;

/* Imitation-task parameters (phc/env/tasks/humanoid_im.py:37-123, env_im.yaml). */
typedef struct {
    float dt;                         /* control dt = control_freq_inv * sim_dt, as fp32 */
    int32_t max_episode_length;       /* episode_length 300 */
    float k_pos, k_rot, k_vel, k_ang_vel, w_pos, w_rot, w_vel, w_ang_vel; /* reward_specs :57 */
    int32_t power_reward;             /* env.power_reward */
    float power_coefficient;          /* :107 */
    int32_t enable_early_termination;
    int32_t use_mean_termination;     /* flags.im_eval and not strict_eval (:1174) */
    int32_t disable_collision_check;  /* flags.no_collision_check */
    int32_t local_root_obs, root_height_obs;
    int32_t num_track_bodies;         /* len(trackBodies) */
    const int32_t* track_slot;        /* [NB] slot of body in trackBodies or -1 */
    const int32_t* reset_mask;        /* [NB] 1 if body in reset_bodies */
    int32_t num_reset_bodies;         /* len(reset_bodies) */
    int32_t first_reset_body;         /* body id of reset_bodies[0]: `termination_distance[0]` of the mean variant (:1586) */
    const float* termination_distances; /* [NB] per body (humanoid_im.py:539-543), live-editable by the learner (im_amp.py:174) */
    int32_t num_key_bodies;
    const int32_t* key_body_ids;      /* [K] */
    int32_t num_amp_joints;           /* joints in dof_subset (19 for SMPL) */
    const int32_t* amp_joint_slot;    /* [NB] slot of joint (body j) in dof_subset order or -1 */
    int32_t num_amp_obs_steps;        /* numAMPObsSteps 10 */
    int32_t num_amp_obs_per_step;     /* 196 */
    int32_t num_self_obs, num_task_obs; /* 358, 576 */
    /* config 3 (getup / MCP composer envs, env_im_getup_mcp.yaml) */
    int32_t cycle_motion;             /* env.cycle_motion: restart the clip in place instead of ending the episode (:1120-1150) */
    int32_t zero_out_far;             /* env.zero_out_far: point-goal reward + task-obs gating when far from the reference (:783-797,890-905) */
    float close_distance;             /* humanoid.py:328 (0.25) */
    float far_distance;               /* humanoid.py:329 (3) */
    /* config 5 (H1 / G1 robots) */
    int32_t dofs_per_joint;           /* 3 spherical (0 is read as 3), 1 revolute: DoF layout of dof_state / AMP obs (humanoid_amp.py:1063-1104) */
    int32_t num_ext_bodies;           /* robot.extend_config entries used by the full-body reward (humanoid_im.py:74-82,916-923) */
    const int32_t* ext_parent;        /* [E] body id of each extended body's parent */
    const float* ext_offset;          /* [E,3] position in the parent frame */
    int32_t obs_v;                    /* task-observation version: 6 (0 is read as 6; `compute_imitation_observations_v6`, humanoid_im.py:1300-1360: 24 floats per
                                         tracked body), 7 (`_v7`, :1362-1393, the keypoint models: position / velocity differences + reference positions, 9),
                                         or 1 / 2 / 3 / 8 / 9 (`compute_imitation_observations`, `_v2`, `_v3`, `_v8`, `_v9`, :1203-1306,1395-1515: 15 J,
                                         15 J + 3 (J - 1), 9 J, 30 J, 18 J + 6 floats; 2 and 9 need the root as the first tracked body) */
    int32_t self_obs_v;               /* 1 (0 is read as 1): compute_humanoid_observations_smpl_max; 3: `_v3` (humanoid.py:2113-2169) = the same
                                         followed by the force-sensor readings [S*6] (num_self_obs grows by 6 S, humanoid.py:683); 2: `_v2`
                                         (:2054-2108): the v1 block for each of the past_track_steps previous body states and the current one,
                                         all relative to the CURRENT root position / heading (num_self_obs = (P + 1) x the v1 size, :513-514) */
    int32_t num_force_sensors;        /* S of phc_sim_state_t.force_sensor (self_obs_v 3) */
    int32_t amp_obs_v;                /* 1 (0 is read as 1): build_amp_observations_smpl; 2: `_v2` (humanoid_amp.py:1015-1059) = the same followed by the
                                         key bodies' heading-local velocities [3 K] (num_amp_obs_per_step grows by 3 K, :303) */
    int32_t remove_base_rot;          /* robot.has_upright_start False: the observation functions strip the asset's base rotation (0.5,0.5,0.5,0.5)
                                         from the root rotation before taking the heading (`remove_base_rot`, humanoid.py:1936-1939 and every
                                         `if not upright:` of humanoid.py / humanoid_amp.py / humanoid_im.py) */
    int32_t num_self_obs_extra;       /* per-env constant columns at the end of the self observation: body-shape parameters
                                         (robot.has_shape_obs, humanoid.py:669-673,2043-2044) then limb weights (has_weight_obs, :676,2046-2047);
                                         counted in num_self_obs */
    int32_t num_amp_obs_extra;        /* the same at the end of every AMP step (has_shape_obs_disc / has_weight_obs_disc, humanoid_amp.py:1005-1008);
                                         counted in num_amp_obs_per_step */
    int32_t num_traj_samples;         /* env.fut_tracks: T = numTrajSamples reference frames in the task observation -- the next one and T - 1 more,
                                         traj_sample_timestep apart (humanoid_im.py:39-47,741-747); obs_v 6 / 7 / 9 (time-major blocks); 0 / 1 = off;
                                         num_task_obs counts all T blocks */
    float traj_sample_timestep;       /* 1 / env.trajSampleTimestepInv (30) */
    int32_t track_body_reward;        /* env.full_body_reward False: the imitation reward runs over the tracked bodies only (humanoid_im.py:925-936:
                                         the `_track_bodies_id` subsets; means over len(trackBodies), no extended bodies) */
    int32_t num_self_obs_hist;        /* P = env.past_track_steps (5) of self_obs_v 2; 0 otherwise */
    int32_t zero_out_far_train;       /* env.zero_out_far_train (with zero_out_far): a reset / a clip restart moves the reference to a random spot of a
                                         5 m disk around the humanoid and arms the cycle counter with zero_out_far_steps (humanoid_im.py:966-980,1133-1140) */
    int32_t zero_out_far_steps;       /* env.zero_out_far_steps (90) */
    int32_t cycle_motion_xp;          /* env.cycle_motion_xp: a clip restart shifts the reference by up to one metre in x and y (:1131-1132) */
    const float* self_obs_extra;      /* [N, num_self_obs_extra] row of the env */
    const float* amp_obs_extra;       /* [N, num_amp_obs_extra] row of the env (observations from simulator state) or of the MOTION (observations
                                         built from the reference clip: the clip carries its humanoid's shape, motion_lib_base.py:244,
                                         humanoid_amp.py:253-284,575-603) -- the same row, clip i belongs to env i */
    const float* amp_ref_table;       /* [F, num_amp_obs_per_step - num_amp_obs_extra] (nullable; ABI 33): the AMP observation of every frame of the motion
                                         library (build_amp_observations of the frame itself), written by phc_amp_ref_table.  With it a reset fills an
                                         env's AMP history from S consecutive ROWS instead of S lookups + observation builds: start times are multiples
                                         of 1/30 s (sample_time_interval) and the history steps back by dt, so on 30 fps clips stepped at dt = k/30 every
                                         history time falls on a frame up to fp32 rounding of the blend factor: exactly 0 for ~5 lookups in 6 (the row IS
                                         the full build then, bit for bit), below 1e-4 for most others (first-order blend with the next row, error
                                         ~1e-6); lookups with any other blend factor (a few % land just below the next frame) are built in full.
                                         NULL: every history frame is looked up and built (robots at 50 Hz). */
} phc_im_params_t;

/* Task-owned per-env buffers (phc/env/tasks/base_task.py:99-105, humanoid_amp.py:109-116,
 * humanoid_im.py:71-121).  int64 where the reference uses torch.long. */
typedef struct {
    int64_t* progress_buf;            /* [N] */
    int64_t* reset_buf;               /* [N] */
    int64_t* terminate_buf;           /* [N] */
    float* rew_buf;                   /* [N] */
    float* reward_raw;                /* [N,4 or 5] */
    float* obs_buf;                   /* [N, num_self_obs+num_task_obs] */
    float* amp_obs_in;                /* [N,S,A] history before this step */
    float* amp_obs_out;               /* [N,S,A] history after this step (ping-pong; may equal amp_obs_in only for reset) */
    const int64_t* sampled_motion_ids;/* [N]; NULL = the identity (env i follows clip i: humanoid_im.py:121) -- saves a dependent load per lookup chain */
    float* motion_start_times;        /* [N] */
    float* motion_start_times_offset; /* [N] */
    float* global_offset;             /* [N,3] */
    float* ref_body_pos;              /* [N,NB,3] side-effect buffers of _compute_task_obs (:855-868), nullable */
    float* ref_body_rot;              /* [N,NB,4] nullable */
    float* ref_body_vel;              /* [N,NB,3] nullable */
    float* ref_dof_pos;               /* [N,D] nullable */
    int32_t* cycle_counter;           /* [N] humanoid_im.py:72,1077-1078,1128,1186; nullable unless cycle_motion */
    int32_t* recovery_counter;        /* [N] humanoid_im_getup.py:62,203-216; nullable (plain HumanoidIm) */
    float* point_goal;                /* [N] humanoid_im.py:95,792,898; nullable unless zero_out_far */
    const float* cycle_phase;         /* [N] the caller's torch.rand draw for _sample_time of cycled envs (:1127); nullable unless cycle_motion */
    /* device-side lists of the envs that finished in the last post-physics launch (nullable): phc_im_post_physics appends every
     * env whose reset flag it sets to one of PHC_RESET_SUBLISTS sub-lists (sub-list = (env / 8) % 16, capacity reset_sublist_cap
     * each) and counts them in reset_count[reset_slot][sub]; phc_im_reset_done then works on exactly those envs (dense wavefronts
     * instead of a masked sweep over all envs) and zeroes reset_count[(reset_slot + 1) % 3][*] for the next step -- the caller
     * rotates reset_slot 0,1,2 per step. */
    int32_t* reset_list;              /* [PHC_RESET_SUBLISTS * reset_sublist_cap] */
    int32_t* reset_count;             /* [3, PHC_RESET_SUBLISTS, PHC_RESET_COUNT_STRIDE], counter in element 0 */
    int32_t reset_slot;
    int32_t reset_sublist_cap;        /* >= 8 * ceil(ceil(N / 8) / PHC_RESET_SUBLISTS) */
    float* body_state_hist;           /* [N, P, NB, 13] self_obs_v 2: the P previous rigid-body states, oldest first (`_rigid_body_*_hist`,
                                         humanoid.py:229-232,1621-1631); shifted by the post-physics launch, filled by the resets; nullable otherwise */
    const float* offset_rand;         /* [N,2] the caller's torch.rand draw for the random reference offsets of zero_out_far_train / cycle_motion_xp
                                         (row of the env; refreshed by the caller before every launch that may use it); nullable unless one is set */
    const uint8_t* occl_mask;         /* [N, J] env.occl_training (humanoid_im.py:96-97,796-804,845-851,1081-1092,1180-1181): non-zero = the tracked body
                                         (slot order) is occluded: its reference state in the task observation and its reference position in
                                         the early-termination distance are the simulated ones; the caller keeps the mask; nullable */
    int64_t amp_env_stride;           /* floats between two envs' AMP history windows; 0 = S * P (plain [N,S,P] buffers).  Larger: every env owns a strip of
                                         more than S frames and amp_obs_in / amp_obs_out point at window positions inside env 0's strip.  When
                                         amp_obs_out + P == amp_obs_in (the new window starts one frame before the old one) phc_im_post_physics
                                         writes the new frame only -- the history shift of humanoid_amp.py:662-670 (14 KB of traffic per env and
                                         step) is then implicit; the caller moves the window back to the end of the strip every (strip - S) steps
                                         with one ordinary shifting call (HumanoidIm: strips of 2 S frames) */
    uint64_t* reset_rng_counter;      /* [1] device-side call counter of phc_im_reset_done (nullable).  Non-null: the start-time draws of a call are keyed by
                                         (seed, *reset_rng_counter) -- the counter argument is ignored -- and phc_im_post_physics advances it by one -- a launch captured in
                                         a hipGraph then draws fresh phases on every replay (the host-side counter argument is frozen at capture) */
} phc_im_buffers_t;

int32_t phc_abi_version(void);

/* M9: MotionLibBase.get_motion_state (phc/utils/motion_lib_base.py:437-520; robots: MotionLibReal.get_motion_state,
 * motion_lib_real.py:236-361) incl. M8 _calc_frame_blend (:549-559).  Outputs nullable.  n lookups; offset nullable [n,3].
 * dof_pos / dof_vel are [n,ND]; rg_pos_ext / rb_rot_ext [n,E,3/4] are the extended bodies' part of rg_pos_t / rg_rot_t. */
int32_t phc_motion_state(const phc_motion_lib_t* lib, int32_t n, const int64_t* motion_ids, const float* motion_times,
                         const float* offset, float* rg_pos /*[n,NB,3]*/, float* rb_rot /*[n,NB,4]*/,
                         float* body_vel /*[n,NB,3]*/, float* body_ang_vel /*[n,NB,3]*/, float* dof_pos /*[n,ND]*/,
                         float* dof_vel /*[n,ND]*/, int64_t* frame_idx0 /*[n]*/, int64_t* frame_idx1 /*[n]*/,
                         float* blend /*[n]*/, float* rg_pos_ext /*[n,E,3]*/, float* rb_rot_ext /*[n,E,4]*/, void* stream);

/* M7: MotionLibBase.sample_time_interval (motion_lib_base.py:414-423); `phase` is the
 * caller's torch.rand draw so the reference RNG stream is preserved. */
int32_t phc_sample_time_interval(const phc_motion_lib_t* lib, int32_t n, const int64_t* motion_ids, const float* phase,
                                 float* motion_times, void* stream);

/* A2 + S8 + S10 + S7: pre_physics_step's action->PD target (humanoid.py:1522-1572,1711-1713),
 * then control_freq_inv x gym.simulate (humanoid.py:1602-1619) with `substeps` sub-steps each,
 * then publication of body state / dof force / contact force (humanoid_amp.py:639-660).
 * actions nullable (then pd_target is used as is).  freeze_mask [D] nullable: 1 -> target forced to 0
 * (humanoid.py:1549-1554). */
int32_t phc_sim_step(const phc_model_t* model, const phc_sim_params_t* params, const phc_sim_state_t* sim,
                     const float* actions /*[N,D]*/, const float* pd_action_offset /*[D]*/, const float* pd_action_scale /*[D]*/,
                     const int32_t* freeze_mask /*[D]*/, int32_t num_sim_calls, void* stream);

/* S7 alone: forward kinematics from (root_states, dof_state) to rigid_body_state. */
int32_t phc_refresh_body_state(const phc_model_t* model, const phc_sim_state_t* sim, void* stream);

/* S7 for a list of envs (after a teleport of root_states / dof_state: gym.set_*_indexed + refresh). */
int32_t phc_refresh_body_state_indexed(const phc_model_t* model, const phc_sim_state_t* sim, int32_t num, const int64_t* env_ids,
                                       void* stream);

/* post_physics_step of HumanoidIm in one launch (humanoid.py:1634-1650, humanoid_amp.py:194-210,
 * humanoid_im.py:694-948,1117-1190): progress_buf += 1, reference lookup at t and t+dt,
 * imitation + power reward, reset / terminate, self obs, task obs v6, AMP obs + history shift. */
int32_t phc_im_post_physics(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm,
                            const phc_sim_state_t* sim, const phc_im_buffers_t* buf, void* stream);

/* phc_im_params_t.amp_ref_table: row f = AMP observation (without the per-env extra columns) of the lookup (f0 = f, f1 = next_frame[f], blend 0), i.e.
 * of a time that falls on frame f of its clip; next_frame[f] = f + 1, or f at the last frame of a clip (the pair matters even at blend 0: the
 * reference's slerp returns the MEAN of two nearly equal rotations whatever the blend factor, isaacgym torch_utils slerp).
 * Rebuild after every (re)load of the motion library and whenever prm's AMP options change. */
int32_t phc_amp_ref_table(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm, int64_t num_frames,
                          const int64_t* next_frame /*[num_frames]*/, float* table /*[num_frames, A - extra]*/, void* stream);

/* Humanoid.reset(env_ids) -> _reset_envs (humanoid.py:585-621, humanoid_amp.py:378-398,508-528,559-637,
 * humanoid_im.py:955-1023): per listed env sample a start time from `phase`, impose the reference
 * state on root/dof/body tensors and PD targets, zero progress/reset/terminate/contact, recompute
 * obs for those envs, and rebuild their AMP history from the reference motion.
 * env_ids == NULL selects the MASKED mode: num_reset must be num_envs, phase is [num_envs], and exactly the envs
 * with reset_buf != 0 are reset (lets a rollout loop reset done envs without a device->host sync); in this mode
 * reset_buf itself is left untouched (the caller zeroes it after the launch). */
int32_t phc_im_reset(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm,
                     const phc_sim_state_t* sim, const phc_im_buffers_t* buf, int32_t num_reset,
                     const int64_t* env_ids, const float* phase /*[num_reset]*/, int32_t start_at_zero, void* stream);

/* `reset_done()`: phc_im_reset's masked mode with the start-time phase drawn INSIDE the kernel, so that the rollout idiom
 * "reset the envs that are done" is one launch with no host round trip and no auxiliary torch kernels:
 * phase(env) = 24-bit uniform from a 32-bit avalanche hash of env under the stream key splitmix64(seed, counter) (like
 * torch.rand: 24 mantissa bits).  The caller advances `counter` per call. */
int32_t phc_im_reset_done(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm,
                          const phc_sim_state_t* sim, const phc_im_buffers_t* buf, uint64_t seed, uint64_t counter,
                          int32_t start_at_zero, void* stream);

/* HumanoidImGetup._reset_fall_episode + the shared tail of _reset_envs (humanoid_im_getup.py:166-196, humanoid.py:585-621,
 * humanoid_amp.py:559-573): the caller has written root_states / dof_state of the listed envs (a stored fall state);
 * after phc_refresh_body_state_indexed made their rigid_body_state current, this call zeroes
 * progress/reset/terminate/contact, sets the PD target to the joint positions, recomputes their observations against
 * the reference at the env's (unchanged) motion clock and recomputes the current AMP observation; fill_history != 0
 * copies it into every history slot (_init_amp_obs_default) -- 0 is the "recovery episode" case (:160-164) where the
 * env keeps its state and its AMP history. */
int32_t phc_im_reset_from_state(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm,
                                const phc_sim_state_t* sim, const phc_im_buffers_t* buf, int32_t num_reset,
                                const int64_t* env_ids, int32_t fill_history, void* stream);

/* HumanoidAMP.build_amp_obs_demo (humanoid_amp.py:253-284): n samples x S steps back in time. */
int32_t phc_amp_obs_demo(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm, int32_t n,
                         const int64_t* motion_ids, const float* motion_times0, float* amp_obs_demo /*[n,S,A]*/, void* stream);

/* P5: CommonAgent.discount_values (phc/learning/common_agent.py:493-505), tensors [T,N]. */
int32_t phc_gae(int32_t horizon, int32_t n, const float* fdones, const float* values, const float* rewards,
                const float* next_values, float gamma, float tau, float* advs, void* stream);

/* poselib FK for motion loading (M5): SkeletonState.global_transformation
 * (poselib/poselib/skeleton/skeleton3d.py:390-426) in fp32 on device.  local_rot [T,NB,4],
 * root_trans [T,3] -> global_rot [T,NB,4], global_pos [T,NB,3]. */
int32_t phc_fk(const phc_model_t* model, int64_t num_frames, const float* local_rot, const float* root_trans,
               float* global_rot, float* global_pos, void* stream);

/* P1: RunningMeanStd.forward in train mode (phc/utils/running_mean_std.py:69-111) on a device batch x [rows, cols] fp32 -- or, with
 * row_index [rows] (int64), on the minibatch x[row_index] of a larger tensor without materialising it (the dataset gather of
 * common_agent.py:357-398 folded into the pass):
 *   out = clamp((x - float(norm_mean)) / sqrt(float(norm_var) + epsilon), -clamp, clamp)      (:95-96; fp32 or bf16 [rows, cols], may be NULL)
 *   run_mean / run_var / run_count (fp64, in place; NULL = no update, eval mode or frozen) <- parallel-variance update with the batch
 *   mean and unbiased variance (:56-67,100-104).  norm_* may alias run_* (output from the statistics BEFORE the update, as the
 *   reference computes it) or be a frozen copy (amp_agent.py:527-532 `running_mean_std_temp`).
 * out_stride (ABI 23): elements between two output rows (0 = cols): the output may be the left columns of a wider buffer whose K is
 *   padded to a GEMM-friendly multiple (934 -> 1024: the first-layer GEMMs of the update run 20-30 % faster, profiles/r02_notes.md).
 * workspace: phc_running_norm_workspace(rows, cols) bytes of device memory (only used when updating); its LAST 8 bytes (a ticket
 * counter of the finishing kernel) must be zero before the first call and are left at zero. */
int64_t phc_running_norm_workspace(int64_t rows, int32_t cols);
int32_t phc_running_norm(const float* x, const int64_t* row_index, int64_t rows, int32_t cols, const double* norm_mean, const double* norm_var, float epsilon,
                         float clamp, void* out, int32_t out_bf16, int32_t out_stride, double* run_mean, double* run_var, double* run_count,
                         double* workspace, void* stream);

/* Column sums of a bf16 matrix x [rows, cols] -> out fp32 [cols]: the bias gradient of a linear layer (autograd's `sum(0)` of
 * the output gradient, AddmmBackward).  workspace: phc_colsum_workspace(rows, cols) bytes. */
int64_t phc_colsum_workspace(int64_t rows, int32_t cols);
int32_t phc_colsum_bf16(const void* x, int64_t rows, int32_t cols, float* out, float* workspace, void* stream);
/* The same behind a ReLU whose output `y` was saved (round 2: layers whose ReLU rides in the GEMM epilogue): gm = gy where y > 0 else 0
 * (torch's ThresholdBackward, phc/learning/network_builder.py:126-137 `activation: relu`) written as bf16 [rows, cols], and its column sums
 * (the bias gradient) -- one pass instead of two.  Same workspace size. */
int32_t phc_colsum_relu_bf16(const void* gy, const void* y, int64_t rows, int32_t cols, void* gm, float* out, float* workspace, void* stream);
/* ABI 35: `out` == NULL in phc_colsum_bf16 / phc_colsum_relu_bf16 / `gw_gb` == NULL in phc_linear1_backward runs the FIRST stage only -- the per-chunk partial
 * sums stay in `workspace`, [phc_colsum_chunks(rows)] (resp. [phc_linear1_chunks(rows)], `cols + 1` wide) rows -- and phc_colsum_finish_batch later finishes
 * up to any number of such sums in one launch per PHC_COLSUM_MAX_JOBS of them: the bias gradients of a backward pass (autograd's AccumulateGrad of
 * `AddmmBackward`'s `sum(0)`, read by nobody before clip_grad_norm_ + Adam, phc/learning/amp_agent.py:669-676) cost one dependent ~5 us launch at the end
 * of the pass instead of one per layer.  `accumulate` != 0 adds to `out` instead of storing. */
#define PHC_COLSUM_MAX_JOBS 16
typedef struct {
    const float* partial;             /* [nchunks, cols] fp32: a first stage's workspace (must stay untouched until the batch has run) */
    float* out;                       /* [cols] fp32 */
    int32_t nchunks, cols, accumulate;
} phc_colsum_job_t;
int32_t phc_colsum_chunks(int64_t rows);
int32_t phc_linear1_chunks(int64_t rows);
int32_t phc_colsum_finish_batch(int32_t count, const phc_colsum_job_t* jobs /* host array */, void* stream);
/* Weight gradient of a linear layer as a split-K batched GEMM (autograd's `grad_output.t() @ input`, AddmmBackward): `part` [slabs, n] bf16 are the
 * slab products; out [n] fp32 receives their sum (accumulate 0) or has it added (accumulate != 0: a parameter's second and later gradient
 * contributions of a step, torch's AccumulateGrad).  part and out 16-byte aligned. */
int32_t phc_sum_slabs_bf16(const void* part, int32_t slabs, int64_t n, float* out, int32_t accumulate, void* stream);
/* ABI 37.  Operand of a split-precision linear layer (`+learning.params.config.actor_precision=split_bf16`; no reference counterpart: the reference's layers are fp32
 * `torch.nn.Linear`s, phc/learning/network_builder.py, and this is the precision option between bf16 and fp32 GEMMs).  x fp32 [rows, cols], rows `ld_in` floats apart,
 * optionally gated (gate fp32, rows `ld_gate` apart: x where gate > 0, else 0 -- the ReLU mask of a backward pass; NULL = no gate), is cut into bf16 head h = bf16(x) and
 * tail l = bf16(x - h) and written as the three chunks ONE bf16 GEMM with a three times longer reduction reads:
 *   out[row * row_stride + c * chunk_stride + col] (bf16), c = 0, 1, 2 = (h, h, l) for order 0, (h, l, h) for order 1,
 * with zeros in columns cols .. cols_pad - 1 and rows rows .. rows_pad - 1.  (x h)(w h) + (x l)(w h) + (x h)(w l) = an order-1 operand against an order-0 one.
 * extra_mode 1 writes the constant 1 into column `cols` of the valid rows, extra_mode 2 writes extra[row] (fp32 [rows]) there (cols_pad > cols then): an activation
 * operand with the ones column against a weight operand carrying the bias in that column makes the bias part of the product, and row `cols` of the weight-gradient
 * product (gradient operand^T x activation operand) is the bias gradient.  extra_mode 0: `extra` unused.
 * cols_pad, row_stride, chunk_stride multiples of 4; out 8-byte aligned. */
int32_t phc_split3_bf16(const float* x, int64_t ld_in, const float* gate, int64_t ld_gate, int64_t rows, int32_t cols, int64_t rows_pad, int32_t cols_pad,
                        const float* extra, int32_t extra_mode, void* out, int64_t row_stride, int64_t chunk_stride, int32_t order, void* stream);

/* Discriminator loss pieces (phc/learning/amp_agent.py:732-808 `_disc_loss`).
 * phc_disc_bce: logits [n_agent + n_demo] (agent and replay rows first, demo rows last; bf16 or fp32):
 *   stats[0] = scale * 0.5 (BCEWithLogits(agent, 0) + BCEWithLogits(demo, 1)), stats[1] = mean(agent < 0), stats[2] = mean(demo > 0),
 *   stats[3] = mean(agent logits), stats[4] = mean(demo logits) (ABI 36: `stats` holds 5 floats; the reference's `disc/agent_logit`, `disc/demo_logit`
 *   scalars, amp_agent.py:911-912);
 *   grad [same shape / type] = d stats[0] / d logits.
 * phc_weighted_sumsq: out[0] = sum_i coefs[i] * |tensors[i]|^2 over count <= 4 device tensors of sizes[i] elements (all fp32 or all
 *   bf16): the logit regulariser + weight decay in one pass, or -- one bf16 tensor, coef = c / rows -- the gradient penalty
 *   c * mean_rows(sum_cols g^2).  out[1 + i] = |tensors[i]|^2 (ABI 20: `out` holds 1 + count floats; the last one of the weight call
 *   is the reference's `disc_logit_loss`, amp_agent.py:757-758).  workspace: phc_sumsq_workspace() bytes. 
 * (Round 6: one block per 1024 logits, the last block to finish adds the per-block sums; they live in a static device buffer, so launches of phc_disc_bce on DIFFERENT streams
 * of one process must not overlap.) */
int32_t phc_disc_bce(const void* logits, int32_t is_bf16, int32_t n_agent, int32_t n_demo, float scale, void* grad, float* stats,
                     void* stream);
int64_t phc_sumsq_workspace(void);
int32_t phc_weighted_sumsq(int32_t count, const void* const* tensors, const int64_t* sizes, const float* coefs, int32_t is_bf16, float* out,
                           double* workspace, void* stream);

/* Rollout bookkeeping of one env step (phc/learning/amp_agent.py:321-341 inside play_steps; rl_games' current_rewards / current_lengths):
 *   exp_rewards[i] = reward_scale * rewards[i];  exp_dones[i] = dones[i] != 0;  terminated_mask[i] = terminate[i] != 0 (the mask of the
 *   next-value write, :327-329);  terminated_flags[i] += terminated_mask[i];  reward_raw_acc[k] += mean_i reward_raw[i, k];
 *   current_rewards[i] = (current_rewards[i] + rewards[i]) * (1 - done);  current_lengths[i] = (current_lengths[i] + 1) * (1 - done). */
int32_t phc_rollout_bookkeeping(const float* rewards, float reward_scale, const int64_t* dones, const int64_t* terminate, const float* reward_raw,
                                int32_t num_reward_terms, int64_t num_envs, float* exp_rewards, uint8_t* exp_dones, float* terminated_flags,
                                float* terminated_mask, float* reward_raw_acc, float* current_rewards, float* current_lengths, void* stream);

/* Rollout policy step (phc/learning/amp_agent.py:309-341 with rl_games' ModelA2CContinuousLogStd in eval mode), per env r:
 *   actions = mu + exp(logstd) * noise, mus = mu, sigmas = exp(logstd), neglogp = neglogp(actions | mu, sigma)            (mu != NULL)
 *   values[r] = unnorm(value[r]) = sqrt(float(value_var) + epsilon) * clamp(value[r], -5, 5) + float(value_mean)             (value != NULL;
 *               value_mean == NULL: no un-normalisation), times (1 - mask[r]) when mask is given (next_values of terminated envs).
 * mu [N, D] / value [N] are the network heads (bf16 when is_bf16, else fp32); outputs fp32 (rows of the experience buffer). */
int32_t phc_policy_sample(const void* mu, const void* value, int32_t is_bf16, const float* logstd, const float* noise, const double* value_mean,
                          const double* value_var, float epsilon, const float* mask, int64_t num_envs, int32_t num_actions, float* actions, float* mus,
                          float* sigmas, float* neglogp, float* values, void* stream);

/* Linear layer with one output (the value head `a2c_network.value`, 512 -> 1), bf16: y [rows] = x [rows, cols] w [cols] + b[0];
 * backward: gx [rows, cols] = gy w^T (optional), gw_gb fp32 [cols + 1] = (gy^T x, sum gy).  workspace: phc_linear1_workspace(). */
int64_t phc_linear1_workspace(int64_t rows, int32_t cols);
int32_t phc_linear1_forward(const void* x, const void* w, const void* b, int64_t rows, int32_t cols, void* y, void* stream);
int32_t phc_linear1_backward(const void* x, const void* w, const void* gy, int64_t rows, int32_t cols, void* gx, float* gw_gb,
                             float* workspace, void* stream);

/* P9: gradient clipping + optimizer step on the flat fp32 parameter (phc/learning/amp_agent.py:669-676: `clip_grad_norm_(grad_norm)`
 * then `optimizer.step()` with torch.optim.Adam): grad *= min(1, max_norm / (|grad| + 1e-6)) in place (max_norm <= 0: no clipping),
 * then Adam with L2 weight decay; `step` is the 1-based step count (bias corrections computed on the host in fp64).
 * grad_norm_out (optional, device) receives |grad| before clipping; param_bf16 (optional, [n] bf16) receives the updated parameter
 * rounded to bf16 -- the copy the next step's GEMMs read.  step_device (optional, device int64): incremented by one and used for the
 * bias corrections instead of `step` -- for launches replayed from a captured hipGraph, whose by-value arguments are frozen.
 * workspace: phc_adam_workspace() bytes. */
int64_t phc_adam_workspace(void);
int32_t phc_adam_clip_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                           float eps, float weight_decay, int64_t step, float max_norm, double* workspace, float* grad_norm_out,
                           void* param_bf16, int64_t* step_device, void* stream);

/* P8: actor + critic part of the PPO loss and its gradient (phc/learning/amp_agent.py:598-640 with rl_games' neglogp, bound_loss
 * common_agent.py:512-520 and torch_ext.policy_kl), tensors on the device, value_size 1:
 *   loss = mean(max(-adv r, -adv clamp(r, 1 -+ e_clip))) + critic_coef mean(c_loss) - entropy_coef entropy + bounds_loss_coef mean(b_loss),
 *   r = exp(old_neglogp - neglogp(actions | mu, exp(logstd)))
 * mu [B, D] and value [B] are the network heads (bf16 when is_bf16, else fp32); grad_mu / grad_value receive d loss / d mu, d loss / d value
 * in the same type; stats[7] = loss, mean a_loss, mean c_loss, mean b_loss, entropy, mean kl(policy || old policy), mean(|r - 1| > e_clip)
 * (ABI 36: the clip fraction, `actor_clipped` of common_agent.py:570-571 -> `loss/clip_frac`).
 * old_values is only read when clip_value.  row_index (optional, [B] int64): minibatch row r takes actions / old_* / advantages /
 * returns from row row_index[r] of the rollout tensors (mu, value and the gradients stay minibatch-ordered).
 * workspace: phc_ppo_loss_workspace() bytes. */
typedef struct {
    float e_clip, critic_coef, entropy_coef, bounds_loss_coef;
    int32_t clip_value;
} phc_ppo_params_t;
int64_t phc_ppo_loss_workspace(void);
int32_t phc_ppo_loss(const void* mu, const void* value, int32_t is_bf16, const float* logstd, const float* actions, const float* old_neglogp,
                     const float* advantages, const float* returns, const float* old_values, const float* old_mu, const float* old_sigma,
                     const int64_t* row_index, int64_t batch, int32_t num_actions, const phc_ppo_params_t* prm, void* grad_mu, void* grad_value, float* stats,
                     double* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PHC_AMD_H */
