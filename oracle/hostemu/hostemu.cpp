// hostemu.cpp -- TEST INFRASTRUCTURE ONLY (lives under oracle/: only tests and bench.py's cpu_baseline leg use it).
// Compiles the *same* per-lane functions the HIP kernels are made of (phc_amd/csrc/*.h are
// host+device) with g++ and drives them lane by lane on the CPU, so that the kernel math can be
// checked against the oracle on a machine without a GPU (`pytest -m "not gpu"`).
// The product never loads this library: phc_amd/_lib.py only ever opens libphc_amd.so.
#include <cstring>
#include <vector>
#include <omp.h>
#include "../../phc_amd/csrc/phc_aba.h"
#include "../../phc_amd/csrc/phc_im.h"

using namespace phc;

// Stepper: same phase sequence as k_sim_step, lanes looped inside each phase.
template <int JT>
static int emu_sim_step_t(const phc_model_t* model_all, const phc_sim_params_t* prm, const phc_sim_state_t* sim, const float* actions,
                          const float* pd_off, const float* pd_scale, const int32_t* freeze, int num_sim_calls, int do_step) {
    const int nb = model_all->num_bodies, nd = model_all->num_dof;
    const int ndj = JT == PHC_JT_REVOLUTE ? 1 : 3;
#pragma omp parallel for schedule(static)
    for (int64_t env = 0; env < sim->num_envs; ++env) {
        std::vector<float> xch(PHC_MAX_BODIES * PHC_XCH_STRIDE);
        const phc_model_t model_env = model_for_env(*model_all, *sim, env);
        const phc_model_t* model = &model_env;
        AbaLane L[PHC_MAX_BODIES];
        for (int j = 0; j < PHC_MAX_BODIES; ++j) L[j].level = L[j].slevel = -1;
        for (int j = 0; j < nb; ++j) {
            aba_load_model(L[j], *model, j);
            if (JT == PHC_JT_REVOLUTE) aba_load_model_rev(L[j], *model, j);
            if (do_step && actions && j >= 1) {
                for (int k = 0; k < ndj; ++k) {
                    const int d = L[j].dof_start + k;
                    volatile float prod = pd_scale[d] * actions[env * nd + d];
                    float t = pd_off[d] + prod;
                    if (freeze && freeze[d]) t = 0.f;
                    sim->pd_target[env * nd + d] = t;
                }
            }
            aba_load_state<JT>(L[j], *sim, nd, env, j);
        }
        Xch x;
        x.base = xch.data();
        for (int j = 0; j < nb; ++j) aba_fk_jump_begin(L[j], j, x);   // initial kinematics by pointer jumping, as the kernel does
        for (int k = 0, ks = model_jump_steps(*model); k < ks; ++k) {
            for (int j = 0; j < nb; ++j) aba_fk_jump_step(L[j], k, x);
            for (int j = 0; j < nb; ++j) aba_write_kin(L[j], xslot(x, j), Xch::es, 6);
        }
        if (do_step) {
            const float dt = prm->sim_dt / (float)prm->substeps;
            const int nsub = num_sim_calls * prm->substeps;
            std::vector<float> caps(PHC_MAX_BODIES * PHC_CAP_STRIDE);
            float favg[PHC_MAX_BODIES * 6];
            const int sd = model_solver_depth(*model, true);
            const bool rerooted = model_tab(*model, 11, 3) != 0;
            for (int s = 0; s < nsub; ++s) {
                if (prm->self_collision) {
                    for (int j = 0; j < nb; ++j) aba_publish_shape(L[j], j, x, caps.data());
                    for (int e = 0, nx = model_num_extra_shapes(*model); e < nx; ++e) {
                        AbaLane X;
                        aba_load_extra_shape(X, *model, e);
                        aba_publish_shape(X, nb + e, x, caps.data());
                    }
                    for (int q = 0, np = model_num_pairs(*model); q < np; ++q)
                        aba_collide_pair(*prm, dt, model_pair(*model, q) & 0xff, model_pair(*model, q) >> 8, x, caps.data());
                    for (int j = 0; j < nb; ++j) aba_collect_self(L[j], j, caps.data());
                }
                for (int j = 0; j < nb; ++j) aba_velocity_products(L[j], *model, j, x, true);
                const bool rigid = prm->contact_model == 1;
                const int passes = rigid ? (prm->contact_iterations < 1 ? 1 : prm->contact_iterations) : 1;
                const bool lag = !rigid && prm->inertia_lag != 0 && (s % prm->substeps) != 0;
                for (int pass = 0; pass < passes; ++pass) {
                    for (int j = 0; j < nb; ++j) {
                        if (rigid) aba_body_init<JT, true>(L[j], *model, *prm, dt, j, s % prm->substeps == 0, true, pass);
                        else aba_body_init<JT, false>(L[j], *model, *prm, dt, j, s % prm->substeps == 0, true, 0, lag);
                    }
                    if (JT == PHC_JT_SPHERICAL && rerooted && pass == 0) {
                        for (int j = 0; j < nb; ++j) aba_publish_drive(L[j], j, x);
                        for (int j = 0; j < nb; ++j) aba_fetch_drive(L[j], j, x);
                    }
                    for (int l = sd; l >= 0; --l) for (int j = 0; j < nb; ++j) aba_backward_level<JT>(L[j], l, j, x, lag);
                    for (int l = 0; l <= sd; ++l) for (int j = 0; j < nb; ++j) aba_accel_level<JT>(L[j], l, j, x);
                }
                if (rigid && (s == nsub - 1 || prm->force_average)) for (int j = 0; j < nb; ++j) aba_publish_contact_rigid(L[j], *model, *prm, *sim, dt, env, j, true);
                if (JT == PHC_JT_SPHERICAL && rerooted) for (int j = 0; j < nb; ++j) aba_accel_finish(L[j], *model, j, x);
                for (int j = 0; j < nb; ++j) aba_integrate_joint<JT>(L[j], *prm, dt);
                if (prm->force_average) for (int j = 0; j < nb; ++j) aba_force_accumulate(L[j], s, nsub, favg + 6 * j);
                for (int j = 0; j < nb; ++j) aba_fk_jump_begin(L[j], j, x);
                for (int k = 0, ks = model_jump_steps(*model); k < ks; ++k) {   // all bodies read the previous step's slots, then all write
                    for (int j = 0; j < nb; ++j) aba_fk_jump_step(L[j], k, x);
                    for (int j = 0; j < nb; ++j) aba_write_kin(L[j], xslot(x, j), Xch::es, 6);
                }
            }
        }
        for (int j = 0; j < nb; ++j) {
            if (do_step) aba_store_state<JT>(L[j], *sim, nd, env, j);
            aba_publish_body(L[j], *sim, nb, env, j, do_step != 0);
            if (do_step && prm->contact_model != 1) aba_publish_sensors(L[j], *model, *prm, *sim, prm->sim_dt / (float)prm->substeps, env, j);
        }
    }
    return 0;
}

extern "C" {

void emu_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int emu_max_threads(void) { return omp_get_max_threads(); }

int emu_motion_state(const phc_motion_lib_t* lib, int n, const int64_t* ids, const float* times, const float* offset,
                     float* rg_pos, float* rb_rot, float* body_vel, float* body_ang_vel, float* dof_pos, float* dof_vel,
                     int64_t* idx0, int64_t* idx1, float* blend, float* rg_pos_ext, float* rb_rot_ext) {
    const int nb = lib->num_bodies, ne = lib->num_ext_bodies;
    for (int64_t i = 0; i < n; ++i) {
        FrameRef fr = frame_ref(*lib, ids[i], times[i]);
        if (idx0) idx0[i] = fr.idx0;
        if (idx1) idx1[i] = fr.idx1;
        if (blend) blend[i] = fr.blend;
        for (int e = 0; e < ne; ++e) {
            V3 p; Q4 q;
            ref_body_ext(*lib, fr, e, &p, &q);
            if (offset) p += ld3(offset + i * 3);
            if (rg_pos_ext) st3(rg_pos_ext + (i * ne + e) * 3, p);
            if (rb_rot_ext) st4(rb_rot_ext + (i * ne + e) * 4, q);
        }
        for (int j = 0; j < nb; ++j) {
            BodyState s = ref_body(*lib, fr, j);
            if (offset) s.pos += ld3(offset + i * 3);
            if (rg_pos) st3(rg_pos + (i * nb + j) * 3, s.pos);
            if (rb_rot) st4(rb_rot + (i * nb + j) * 4, s.rot);
            if (body_vel) st3(body_vel + (i * nb + j) * 3, s.vel);
            if (body_ang_vel) st3(body_ang_vel + (i * nb + j) * 3, s.angvel);
            if (j >= 1 && (dof_pos || dof_vel)) {
                V3 dp, dv;
                ref_joint(*lib, fr, j, &dp, &dv);
                const int dpj = lib->dofs_per_joint == 1 ? 1 : 3;
                if (dof_pos) st_joint(dof_pos + (i * (nb - 1) + (j - 1)) * dpj, dpj, dp);
                if (dof_vel) st_joint(dof_vel + (i * (nb - 1) + (j - 1)) * dpj, dpj, dv);
            }
        }
    }
    return 0;
}

int emu_sample_time_interval(const phc_motion_lib_t* lib, int n, const int64_t* ids, const float* phase, float* out) {
    for (int i = 0; i < n; ++i) out[i] = sample_time_interval(*lib, ids[i], phase[i]);
    return 0;
}

int emu_im_post_physics(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm,
                        const phc_sim_state_t* sim, const phc_im_buffers_t* buf) {
#pragma omp parallel for schedule(static)
    for (int64_t env = 0; env < sim->num_envs; ++env) {
        const int64_t progress = buf->progress_buf[env] + 1;
        const ImStepCtx c = im_post_prologue(*lib, *prm, *sim, *buf, env, progress);
        const float prev_goal = (prm->zero_out_far && buf->point_goal) ? buf->point_goal[env] : 0.f;
        float s[6] = {0, 0, 0, 0, 0, 0};
        float root_dist = 0.f;
        int fallen = 0;
        const FrameTab tab = frame_tab(*lib, motion_id_of(*buf, env));
        const BodyState root = load_body(sim->rigid_body_state, env, model->num_bodies, 0);
        for (int lane = 0; lane < PHC_MAX_BODIES; ++lane) {
            const BodyState body = load_body(sim->rigid_body_state, env, model->num_bodies, lane < model->num_bodies ? lane : 0);
            RewardPartial rp = im_post_lane(*model, *lib, *prm, *sim, *buf, env, lane, c, tab, body, root);
            s[0] += rp.pos; s[1] += rp.rot; s[2] += rp.vel; s[3] += rp.angvel; s[4] += rp.power; s[5] += rp.dist;
            if (lane == 0) root_dist = rp.root_dist;
            fallen |= rp.fallen;
        }
        for (int lane = 0; lane < PHC_MAX_BODIES; ++lane) amp_shift_lane(*prm, *buf, env, lane, PHC_MAX_BODIES);   // (after the new frame, as the kernel)
        im_post_finalize(*lib, *prm, *buf, model->num_bodies, env, c, progress, s[0], s[1], s[2], s[3], s[4], s[5], root_dist, prev_goal,
                         fallen, prm->num_reset_bodies > 0 ? prm->num_reset_bodies : 1);
    }
    return 0;
}

int emu_im_reset_from_state(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm, const phc_sim_state_t* sim,
                            const phc_im_buffers_t* buf, int num_reset, const int64_t* env_ids, int fill_history) {
    for (int r = 0; r < num_reset; ++r)
        for (int lane = PHC_MAX_BODIES - 1; lane >= 0; --lane) im_reset_from_state_lane(*model, *lib, *prm, *sim, *buf, env_ids[r], lane, fill_history);
    return 0;
}

int emu_im_reset(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm, const phc_sim_state_t* sim,
                 const phc_im_buffers_t* buf, int num_reset, const int64_t* env_ids, const float* phase, int start_at_zero) {
#pragma omp parallel for schedule(dynamic, 8)
    for (int r = 0; r < num_reset; ++r) {
        const int64_t env = env_ids ? env_ids[r] : r;
        if (!env_ids && buf->reset_buf[env] == 0) continue;
        const int64_t mid = motion_id_of(*buf, env);   // (NULL table = identity, as on the device: ABI 33)
        const float t = start_at_zero ? 0.f : sample_time_interval(*lib, mid, phase[r]);
        for (int k = 0; k < prm->num_amp_obs_steps; ++k)
            for (int lane = 0; lane < PHC_MAX_BODIES; ++lane) {
                if (prm->amp_ref_table) im_reset_amp_table_lane(*lib, *prm, *buf, model->num_bodies, env, lane, PHC_MAX_BODIES, t, k);
                else im_reset_amp_lane(*lib, *prm, *buf, model->num_bodies, env, lane, t, k);
            }
        for (int lane = PHC_MAX_BODIES - 1; lane >= 0; --lane) im_reset_lane(*model, *lib, *prm, *sim, *buf, env, lane, t, true);
    }
    return 0;
}

int emu_amp_obs_demo(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm, int n,
                     const int64_t* ids, const float* times0, float* out) {
    const int S = prm->num_amp_obs_steps, A = prm->num_amp_obs_per_step;
    for (int64_t g = 0; g < (int64_t)n * S; ++g) {
        const int64_t i = g / S;
        const int k = (int)(g - i * S);
        for (int lane = 0; lane < PHC_MAX_BODIES; ++lane) {
            if (prm->amp_ref_table) amp_obs_from_table_lane(*lib, *prm, model->num_bodies, lane, PHC_MAX_BODIES, ids[i], history_time(times0[i], prm->dt, k), out + g * A);
            else amp_obs_from_ref_lane(*lib, *prm, model->num_bodies, lane, ids[i], history_time(times0[i], prm->dt, k), out + g * A);
        }
    }
    return 0;
}

// phc_amp_ref_table: row f = the AMP observation of the lookup (f, next_frame[f], blend 0)
int emu_amp_ref_table(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm, int64_t num_frames,
                      const int64_t* next_frame, float* table) {
    const int W = prm->num_amp_obs_per_step - prm->num_amp_obs_extra;
#pragma omp parallel for schedule(static)
    for (int64_t f = 0; f < num_frames; ++f) {
        FrameRef fr;
        fr.f0 = f; fr.f1 = next_frame[f]; fr.idx0 = fr.idx1 = 0; fr.blend = 0.f;
        for (int lane = 0; lane < PHC_MAX_BODIES; ++lane) amp_obs_from_frames_lane(*lib, *prm, model->num_bodies, lane, fr, table + f * (int64_t)W);
    }
    return 0;
}

int emu_sim_step(const phc_model_t* model, const phc_sim_params_t* prm, const phc_sim_state_t* sim, const float* actions,
                 const float* pd_off, const float* pd_scale, const int32_t* freeze, int num_sim_calls, int do_step) {
    if (do_step) {   // the option checks of phc_sim_step (phc_sim.hip), mirrored: same refusals on both backends
        if (prm->contact_model == 1 && prm->inertia_lag) return PHC_EUNSUPPORTED;
        if (prm->inertia_lag && prm->lane_mapping == 3) return PHC_EUNSUPPORTED;
        if (prm->contact_model == 1 && model->max_body_contact_pts > 32) return PHC_EUNSUPPORTED;
        if (prm->inertia_lag && model->max_body_contact_pts > PHC_CP_BITS) return PHC_EUNSUPPORTED;
    }
    if (model->num_dof == model->num_bodies - 1 && model->num_bodies > 2)
        return emu_sim_step_t<PHC_JT_REVOLUTE>(model, prm, sim, actions, pd_off, pd_scale, freeze, num_sim_calls, do_step);
    return emu_sim_step_t<PHC_JT_SPHERICAL>(model, prm, sim, actions, pd_off, pd_scale, freeze, num_sim_calls, do_step);
}

}  // extern "C"
