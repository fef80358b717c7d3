// phc_kernels.hip -- gfx950 kernels + the extern "C" entry points declared in include/phc_amd.h.
//
// Thread mapping used by every env kernel: ONE LANE PER RIGID BODY, G = 32 lanes per environment (two environments per
// 64-wide wavefront) for articulations of up to 32 bodies incl. the extended reference bodies, G = 64 (one environment per
// wavefront) above.  Task kernels: 256-thread workgroups (8 or 4 envs), per-env sums by G-lane butterfly shuffles.
// Stepper: one wavefront per workgroup (phc_sim.hip).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "phc_aba.h"  // model table accessors (k_fk)
#include "phc_im.h"

using namespace phc;

// Sum / or over the G lanes of an env's group, result in every lane.  Inside a row of 16 lanes the butterfly runs on DPP operands (quad_perm xor 1,
// xor 2, row_half_mirror, row_mirror: folded into the add, ~4 cycles each); only the steps across rows are ds_bpermute round trips (~64 cycles
// each, round 3: seven sums x five dependent permutes were 1.8 k cycles of the post-physics wavefront's 38 k).
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int G>
__device__ __forceinline__ float group_sum(float v) {
    v += __int_as_float(dpp_i<0xB1>(__float_as_int(v)));    // quad_perm:[1,0,3,2]
    v += __int_as_float(dpp_i<0x4E>(__float_as_int(v)));    // quad_perm:[2,3,0,1]
    v += __int_as_float(dpp_i<0x141>(__float_as_int(v)));   // row_half_mirror
    v += __int_as_float(dpp_i<0x140>(__float_as_int(v)));   // row_mirror
#pragma unroll
    for (int m = 16; m < G; m <<= 1) v += __shfl_xor(v, m, G);
    return v;
}
template <int G>
__device__ __forceinline__ int group_or(int v) {
    v |= dpp_i<0xB1>(v); v |= dpp_i<0x4E>(v); v |= dpp_i<0x141>(v); v |= dpp_i<0x140>(v);
#pragma unroll
    for (int m = 16; m < G; m <<= 1) v |= __shfl_xor(v, m, G);
    return v;
}

// ------------------------------------------------------------------------------------------
// post_physics_step of the imitation task.  blockDim = 256 (8 envs).
// ------------------------------------------------------------------------------------------
// The task kernels are instantiated per joint family (DPJ = DoFs per joint: 3 spherical / 1 revolute) with the struct fields
// that select the family pinned to compile-time constants, so the SMPL instantiation carries none of the robot branches.
template <int DPJ>
__device__ __forceinline__ void pin_family(phc_motion_lib_t& lib, phc_im_params_t& prm) {
    lib.dofs_per_joint = DPJ; prm.dofs_per_joint = DPJ;
    if (DPJ == 3) { lib.num_ext_bodies = 0; prm.num_ext_bodies = 0; }
}

template <int DPJ, int G>
__global__ __launch_bounds__(256) void k_im_post_physics(phc_model_t model, phc_motion_lib_t lib, phc_im_params_t prm,
                                                        phc_sim_state_t sim, phc_im_buffers_t buf, int n_reset_bodies) {
    pin_family<DPJ>(lib, prm);
    const int lane = threadIdx.x & (G - 1);
    const int64_t env = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (env >= sim.num_envs) return;  // whole lane group exits together
    PHC_PTL(0, env, lane)
    if (blockIdx.x == 0 && threadIdx.x == 0 && buf.reset_rng_counter) *buf.reset_rng_counter += 1;   // (one writer; no reset launch runs concurrently)
    // Everything the launch reads that does not depend on something else it reads is REQUESTED here, before the first wait: the clip's table
    // entries, the body's and the root's state, the per-env scalars of the prologue (a wavefront waits for a load at its first use; this
    // kernel's wavefronts spend three quarters of their life waiting, profiles/r03_task/post_physics_timeline.txt)
    const int64_t progress = buf.progress_buf[env] + 1;  // humanoid.py:1637
    const FrameTab tab = frame_tab(lib, motion_id_of(buf, env));
    const int jb = lane < model.num_bodies ? lane : 0;
    const BodyState body = load_body(sim.rigid_body_state, env, model.num_bodies, jb);
    const BodyState root = load_body(sim.rigid_body_state, env, model.num_bodies, 0);
    const ImStepCtx c = im_post_prologue(lib, prm, sim, buf, env, progress);
    const float prev_goal = (prm.zero_out_far && buf.point_goal) ? buf.point_goal[env] : 0.f;  // read before lane 0 overwrites it
    PHC_PTL(1, env, lane)
    RewardPartial rp = im_post_lane(model, lib, prm, sim, buf, env, lane, c, tab, body, root);
    PHC_PTL(9, env, lane)
    amp_shift_lane(prm, buf, env, lane, G);   // (every S-th step; reads the old window, writes rows 1.. of the new one: after the frame in row 0)
    float s_pos = group_sum<G>(rp.pos), s_rot = group_sum<G>(rp.rot), s_vel = group_sum<G>(rp.vel), s_ang = group_sum<G>(rp.angvel);
    float s_pow = group_sum<G>(rp.power), s_dist = group_sum<G>(rp.dist);
    int fallen = group_or<G>(rp.fallen);
    PHC_PTL(10, env, lane)
    if (lane == 0)
        im_post_finalize(lib, prm, buf, model.num_bodies, env, c, progress, s_pos, s_rot, s_vel, s_ang, s_pow, s_dist, rp.root_dist,
                         prev_goal, fallen, n_reset_bodies);
    PHC_PTL(11, env, lane)
}
#ifdef PHC_SIM_PROFILE
extern "C" int32_t phc_debug_post_timeline(unsigned long long* out32, long long env) {
    if (out32 && hipMemcpyFromSymbol(out32, HIP_SYMBOL(phc::g_phc_ptl), sizeof(unsigned long long) * 32) != hipSuccess) return -1;
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(phc::g_phc_ptl), z, sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(phc::g_phc_ptl_env), &env, sizeof(env)) == hipSuccess ? 0 : -1;
}
#endif

// phc_amp_ref_table: one lane group per frame of the library.
template <int DPJ, int G>
__global__ __launch_bounds__(256) void k_amp_ref_table(phc_model_t model, phc_motion_lib_t lib, phc_im_params_t prm, int64_t num_frames,
                                                       const int64_t* __restrict__ next_frame, float* __restrict__ table) {
    pin_family<DPJ>(lib, prm);
    const int lane = threadIdx.x & (G - 1);
    const int64_t f = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (f >= num_frames) return;
    FrameRef fr;
    fr.f0 = f; fr.f1 = next_frame[f]; fr.idx0 = fr.idx1 = 0; fr.blend = 0.f;
    amp_obs_from_frames_lane(lib, prm, model.num_bodies, lane, fr, table + f * (int64_t)(prm.num_amp_obs_per_step - prm.num_amp_obs_extra));
}

// HumanoidImGetup fall / recovery resets: one lane group per listed env (state kept, see im_reset_from_state_lane).
template <int G>
__global__ __launch_bounds__(256) void k_im_reset_from_state(phc_model_t model, phc_motion_lib_t lib, phc_im_params_t prm, phc_sim_state_t sim,
                                                            phc_im_buffers_t buf, int num_reset, const int64_t* __restrict__ env_ids,
                                                            int fill_history) {
    const int lane = threadIdx.x & (G - 1);
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (r >= num_reset) return;
    im_reset_from_state_lane(model, lib, prm, sim, buf, env_ids[r], lane, fill_history);
}

// Reset of a list of envs.  One lane group per (env, AMP history frame k): group k == 0 also imposes the state
// and recomputes the observations.  blockDim = 256.
// counter-based uniform in [0,1): the host folds (seed, counter) into one 64-bit stream key (splitmix64); per env a 32-bit
// avalanche hash (murmur3 finaliser rounds) of the env id under that key, top 24 bits -> float like torch.rand
__device__ __forceinline__ float hash_u01(uint64_t key, uint32_t env) {
    uint32_t x = env * 0x9E3779B1u ^ (uint32_t)key;
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    x += (uint32_t)(key >> 32);
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}
__host__ __device__ static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

#ifdef PHC_SIM_PROFILE   // one lane group's timeline through the reset kernel (scripts/probes/reset_timeline.py; the product library has none of this)
__device__ unsigned long long g_phc_rtl[64];
__device__ int g_phc_rtl_group = -1;   // r * 16 + k of the group that stamps
extern "C" int32_t phc_debug_reset_timeline(unsigned long long* out64, int32_t group) {
    if (out64 && hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_phc_rtl), sizeof(g_phc_rtl)) != hipSuccess) return -1;
    unsigned long long z[64] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phc_rtl), z, sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phc_rtl_group), &group, sizeof(group)) == hipSuccess ? 0 : -1;
}
#define PHC_RTL(i) if (rtl_on) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_readcyclecounter(); \
        if (lane == 0) g_phc_rtl[i] = t_; __builtin_amdgcn_sched_barrier(0); }
#else
#define PHC_RTL(i)
#endif

// Measured alternative (round 2, profiles/r02_notes.md): ONE workgroup per listed env with its S history-frame groups side by side
// (blockDim = G * S), so that the S lookups of an env -- S + 1 consecutive clip frames -- share a CU's L1: 46.6 us vs 34 us for this
// geometry (the ten groups of an env then hit one clip region, i.e. the same HBM channels, at the same instant).  Not kept.
template <int DPJ, bool RNG, int G>
__global__ __launch_bounds__(256) void k_im_reset(phc_model_t model, phc_motion_lib_t lib, phc_im_params_t prm, phc_sim_state_t sim,
                                                 phc_im_buffers_t buf, int num_reset, const int64_t* __restrict__ env_ids,
                                                 const float* __restrict__ phase, int start_at_zero, uint64_t rng_key) {
    pin_family<DPJ>(lib, prm);
    const int lane = threadIdx.x & (G - 1);
    // listed env (grid.x); grid.y: 0 = state + self observation, 1 = task observation, 2 + k = AMP history frame k -- the heaviest groups
    // are dispatched first and no group carries more than one lookup chain
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int k = (int)blockIdx.y;
#ifdef PHC_SIM_PROFILE
    const bool rtl_on = (int)(r * 16 + k) == g_phc_rtl_group;
#endif
    PHC_RTL(0)
    int64_t env;
    if (RNG && buf.reset_list) {
        // reset_done() on the device-built list of finished envs: dense wavefronts, blocks beyond the count leave at once
        const int cap = buf.reset_sublist_cap, r32 = (int)r;
        // group r works on entry r / 16 of sub-list r % 16: concurrently running wavefronts draw from all sub-lists (a sub-list holds
        // envs of every 16th workgroup, whose clips sit at a fixed stride in HBM -- walking one sub-list at a time camps on channels)
        const int sub = r32 & (PHC_RESET_SUBLISTS - 1), i = r32 >> 4;
        if (r32 < PHC_RESET_SUBLISTS && k == 0 && lane == 0)   // next step's counters
            buf.reset_count[(((buf.reset_slot + 1) % 3) * PHC_RESET_SUBLISTS + r32) * PHC_RESET_COUNT_STRIDE] = 0;
        if (i >= cap || i >= buf.reset_count[(buf.reset_slot * PHC_RESET_SUBLISTS + sub) * PHC_RESET_COUNT_STRIDE]) return;
        env = buf.reset_list[sub * cap + i];
    } else {
        if (r >= num_reset) return;
        // env_ids == NULL: masked mode over all envs (reset every env whose reset_buf is set) -- no host sync needed.
        // The flag is NOT cleared here (other groups of the same env still read it).
        env = env_ids ? env_ids[r] : r;
        if (!env_ids && buf.reset_buf[env] == 0) return;
    }
    PHC_RTL(1)
    const int64_t mid = motion_id_of(buf, env);
    // _sample_ref_state (humanoid_im.py:1000-1023): StateInit.Random -> sample_time_interval; Start / flags.test -> 0
    // (start_at_zero with a null phase array is only legal in the RNG-free instantiation's list mode)
    // (device-side call counter, phc_im_buffers_t.reset_rng_counter: folded into the key so that a captured launch draws anew on every replay)
    const uint64_t key = (RNG && buf.reset_rng_counter) ? splitmix64(rng_key ^ (*buf.reset_rng_counter * 0x9E6C63D0876A9A47ull)) : rng_key;
    const float t = start_at_zero ? 0.f : sample_time_interval(lib, mid, RNG ? hash_u01(key, (uint32_t)env) : phase[r]);
    PHC_RTL(2)
    if (k < 2) im_reset_lane(model, lib, prm, sim, buf, env, lane, t, env_ids != nullptr, 1 << k);
    PHC_RTL(3)
    if (k >= 2) {
        if (prm.amp_ref_table != nullptr) im_reset_amp_table_lane(lib, prm, buf, model.num_bodies, env, lane, G, t, k - 2);
        else im_reset_amp_lane(lib, prm, buf, model.num_bodies, env, lane, t, k - 2);
    }
    PHC_RTL(4)
}

// build_amp_obs_demo: n samples x S history steps.  One lane group per (sample, step).
template <int G>
__global__ __launch_bounds__(256) void k_amp_obs_demo(phc_model_t model, phc_motion_lib_t lib, phc_im_params_t prm, int n,
                                                     const int64_t* __restrict__ motion_ids, const float* __restrict__ times0,
                                                     float* __restrict__ out) {
    const int lane = threadIdx.x & (G - 1);
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;   // sample (grid.x), history step k (grid.y)
    const int k = (int)blockIdx.y;
    const int S = prm.num_amp_obs_steps, A = prm.num_amp_obs_per_step;
    if (i >= n) return;
    // (demo start times are multiples of 1/30 s too -- sample_time_interval, humanoid_amp.py:262 -- : the per-frame table serves them as it serves resets)
    if (prm.amp_ref_table != nullptr) amp_obs_from_table_lane(lib, prm, model.num_bodies, lane, G, motion_ids[i], history_time(times0[i], prm.dt, k), out + (i * S + k) * A);
    else amp_obs_from_ref_lane(lib, prm, model.num_bodies, lane, motion_ids[i], history_time(times0[i], prm.dt, k), out + (i * S + k) * A);
}

// M9 standalone: get_motion_state for n (id, time) pairs.  One lane group per lookup.
template <int G>
__global__ __launch_bounds__(256) void k_motion_state(phc_motion_lib_t lib, int n, const int64_t* __restrict__ ids,
                                                     const float* __restrict__ times, const float* __restrict__ offset,
                                                     float* rg_pos, float* rb_rot, float* body_vel, float* body_ang_vel,
                                                     float* dof_pos, float* dof_vel, int64_t* idx0, int64_t* idx1, float* blend,
                                                     float* rg_pos_ext, float* rb_rot_ext) {
    const int lane = threadIdx.x & (G - 1);
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    if (i >= n) return;
    const int nb = lib.num_bodies;
    const FrameRef fr = frame_ref(lib, ids[i], times[i]);
    if (lane == 0) {
        if (idx0) idx0[i] = fr.idx0;
        if (idx1) idx1[i] = fr.idx1;
        if (blend) blend[i] = fr.blend;
    }
    if (lane >= nb) {
        const int e = lane - nb, ne = lib.num_ext_bodies;
        if (e < ne && (rg_pos_ext || rb_rot_ext)) {
            V3 p; Q4 q;
            ref_body_ext(lib, fr, e, &p, &q);
            if (offset) p += ld3(offset + i * 3);
            if (rg_pos_ext) st3(rg_pos_ext + (i * ne + e) * 3, p);
            if (rb_rot_ext) st4(rb_rot_ext + (i * ne + e) * 4, q);
        }
        return;
    }
    BodyState s = ref_body(lib, fr, lane);
    if (offset) s.pos += ld3(offset + i * 3);
    if (rg_pos) st3(rg_pos + (i * nb + lane) * 3, s.pos);
    if (rb_rot) st4(rb_rot + (i * nb + lane) * 4, s.rot);
    if (body_vel) st3(body_vel + (i * nb + lane) * 3, s.vel);
    if (body_ang_vel) st3(body_ang_vel + (i * nb + lane) * 3, s.angvel);
    if (lane >= 1 && (dof_pos || dof_vel)) {
        V3 dp, dv;
        ref_joint(lib, fr, lane, &dp, &dv);
        const int dpj = lib.dofs_per_joint == 1 ? 1 : 3;
        if (dof_pos) st_joint(dof_pos + (i * (nb - 1) + (lane - 1)) * dpj, dpj, dp);
        if (dof_vel) st_joint(dof_vel + (i * (nb - 1) + (lane - 1)) * dpj, dpj, dv);
    }
}

__global__ void k_sample_time_interval(phc_motion_lib_t lib, int n, const int64_t* __restrict__ ids,
                                       const float* __restrict__ phase, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sample_time_interval(lib, ids[i], phase[i]);
}

// P5 GAE: one thread per env, reversed scan over the horizon (common_agent.py:493-505)
__global__ void k_gae(int T, int n, const float* __restrict__ fdones, const float* __restrict__ values,
                      const float* __restrict__ rewards, const float* __restrict__ next_values, float gamma, float tau,
                      float* __restrict__ advs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float last = 0.f;
    for (int t = T - 1; t >= 0; --t) {
        const int64_t k = (int64_t)t * n + i;
        const float not_done = 1.0f - fdones[k];
        const float delta = rewards[k] + gamma * next_values[k] - values[k];
        last = delta + gamma * tau * not_done * last;
        advs[k] = last;
    }
}

// M5 poselib FK (skeleton3d.py:390-426): one thread per frame, bodies in tree order; quat_mul_norm semantics
// (positive real part, unit norm; rotation3d.py:31-98,195-201).
__device__ __forceinline__ Q4 pl_quat_mul_norm(Q4 a, Q4 b) {
    Q4 q = quat_mul16(a, b);
    if (q.w < 0.f) q = q4(-q.x, -q.y, -q.z, -q.w);
    float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-9f);
    return q4(q.x / n, q.y / n, q.z / n, q.w / n);
}
__global__ void k_fk(phc_model_t model, int64_t T, const float* __restrict__ local_rot, const float* __restrict__ root_trans,
                     float* __restrict__ grot, float* __restrict__ gpos) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= T) return;
    const int nb = model.num_bodies;
    for (int j = 0; j < nb; ++j) {
        const int p = model_tab(model, 0, j);
        Q4 lq = ld4(local_rot + (f * nb + j) * 4);
        if (p < 0) {
            st4(grot + (f * nb + j) * 4, lq);
            st3(gpos + (f * nb + j) * 3, ld3(root_trans + f * 3));
        } else {
            Q4 pq = ld4(grot + (f * nb + p) * 4);
            const float* mb = model_body(model, j);
            // poselib quat_rotate = Im(q * (v,0) * conj(q))
            V3 off = quat_rotate(pq, v3(mb[0], mb[1], mb[2]));
            st4(grot + (f * nb + j) * 4, pl_quat_mul_norm(pq, lq));
            st3(gpos + (f * nb + j) * 3, off + ld3(gpos + (f * nb + p) * 3));
        }
    }
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
static inline int32_t launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}
static inline int env_blocks(int64_t groups, int lanes) { return (int)((groups * lanes + 255) / 256); }
// lanes per env: 32 while the articulation (with its extended reference bodies) fits, else 64
static inline int group_lanes(int num_bodies, int num_ext) { return num_bodies + num_ext > 32 ? 64 : 32; }

extern "C" {

int32_t phc_abi_version(void) { return PHC_ABI_VERSION; }

int32_t phc_motion_state(const phc_motion_lib_t* lib, int32_t n, const int64_t* motion_ids, const float* motion_times,
                         const float* offset, float* rg_pos, float* rb_rot, float* body_vel, float* body_ang_vel,
                         float* dof_pos, float* dof_vel, int64_t* frame_idx0, int64_t* frame_idx1, float* blend, float* rg_pos_ext,
                         float* rb_rot_ext, void* stream) {
    if (!lib || n < 0 || lib->num_bodies + lib->num_ext_bodies > PHC_MAX_BODIES) return PHC_EINVAL;
    if (n == 0) return 0;
    if (group_lanes(lib->num_bodies, lib->num_ext_bodies) == 64)
        hipLaunchKernelGGL(k_motion_state<64>, dim3(env_blocks(n, 64)), dim3(256), 0, (hipStream_t)stream, *lib, n, motion_ids, motion_times, offset,
                           rg_pos, rb_rot, body_vel, body_ang_vel, dof_pos, dof_vel, frame_idx0, frame_idx1, blend, rg_pos_ext, rb_rot_ext);
    else
        hipLaunchKernelGGL(k_motion_state<32>, dim3(env_blocks(n, 32)), dim3(256), 0, (hipStream_t)stream, *lib, n, motion_ids, motion_times, offset,
                           rg_pos, rb_rot, body_vel, body_ang_vel, dof_pos, dof_vel, frame_idx0, frame_idx1, blend, rg_pos_ext, rb_rot_ext);
    return launch_status();
}

int32_t phc_sample_time_interval(const phc_motion_lib_t* lib, int32_t n, const int64_t* motion_ids, const float* phase,
                                 float* motion_times, void* stream) {
    if (!lib || n < 0) return PHC_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_sample_time_interval, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *lib, n, motion_ids, phase, motion_times);
    return launch_status();
}

static int32_t check_model(const phc_model_t* m) {
    if (!m || m->num_bodies < 1 || m->num_bodies > PHC_MAX_BODIES || !m->ints || !m->floats) return PHC_EINVAL;
    // all-spherical (SMPL family) or all-revolute (H1 / G1) articulations
    if (m->num_dof != 3 * (m->num_bodies - 1) && m->num_dof != m->num_bodies - 1) return PHC_EUNSUPPORTED;
    return 0;
}

static int32_t check_im(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm) {
    int32_t rc = check_model(model);
    if (rc) return rc;
    if (!lib || !prm || lib->num_bodies != model->num_bodies) return PHC_EINVAL;
    const int dpj = model->num_dof == model->num_bodies - 1 && model->num_bodies > 2 ? 1 : 3;
    if ((lib->dofs_per_joint == 1 ? 1 : 3) != dpj || (prm->dofs_per_joint == 1 ? 1 : 3) != dpj) return PHC_EINVAL;
    if (prm->num_ext_bodies < 0 || prm->num_ext_bodies != lib->num_ext_bodies || model->num_bodies + prm->num_ext_bodies > PHC_MAX_BODIES) return PHC_EINVAL;
    if (prm->num_ext_bodies > 0 && (!prm->ext_parent || !prm->ext_offset)) return PHC_EINVAL;
    if (!prm->track_slot || !prm->reset_mask || !prm->termination_distances || !prm->key_body_ids || !prm->amp_joint_slot) return PHC_EINVAL;
    if (prm->num_key_bodies > 32) return PHC_EUNSUPPORTED;
    return 0;
}

int32_t phc_im_post_physics(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm,
                            const phc_sim_state_t* sim, const phc_im_buffers_t* buf, void* stream) {
    int32_t rc = check_im(model, lib, prm);
    if (rc) return rc;
    if (!sim || !buf || buf->amp_obs_in == buf->amp_obs_out) return PHC_EINVAL;
    if (prm->cycle_motion && (!buf->cycle_counter || !buf->cycle_phase)) return PHC_EINVAL;
    if (prm->zero_out_far && (!buf->point_goal || prm->track_slot == nullptr)) return PHC_EINVAL;
    if (sim->num_envs == 0) return 0;
    const int n_reset_bodies = prm->num_reset_bodies > 0 ? prm->num_reset_bodies : 1;
    const int g = group_lanes(model->num_bodies, prm->num_ext_bodies);
    const dim3 grid(env_blocks(sim->num_envs, g));
#define PHC_POST(DPJ, G) hipLaunchKernelGGL((k_im_post_physics<DPJ, G>), grid, dim3(256), 0, (hipStream_t)stream, *model, *lib, *prm, *sim, *buf, n_reset_bodies)
    if (prm->dofs_per_joint == 1) { if (g == 64) PHC_POST(1, 64); else PHC_POST(1, 32); }
    else { if (g == 64) PHC_POST(3, 64); else PHC_POST(3, 32); }
#undef PHC_POST
    return launch_status();
}

int32_t phc_amp_ref_table(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm, int64_t num_frames,
                          const int64_t* next_frame, float* table, void* stream) {
    int32_t rc = check_im(model, lib, prm);
    if (rc) return rc;
    if (!table || !next_frame || num_frames < 0 || num_frames > lib->num_frames_total) return PHC_EINVAL;
    if (num_frames == 0) return 0;
    const int g = group_lanes(model->num_bodies, prm->num_ext_bodies);
    const int64_t blocks = (num_frames * g + 255) / 256;
    if (blocks > 0x7fffffffLL) return PHC_EUNSUPPORTED;
#define PHC_TAB(DPJ, G) hipLaunchKernelGGL((k_amp_ref_table<DPJ, G>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *model, *lib, *prm, num_frames, next_frame, table)
    if (prm->dofs_per_joint == 1) { if (g == 64) PHC_TAB(1, 64); else PHC_TAB(1, 32); }
    else { if (g == 64) PHC_TAB(3, 64); else PHC_TAB(3, 32); }
#undef PHC_TAB
    return launch_status();
}

int32_t phc_im_reset(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm, const phc_sim_state_t* sim,
                     const phc_im_buffers_t* buf, int32_t num_reset, const int64_t* env_ids, const float* phase,
                     int32_t start_at_zero, void* stream) {
    int32_t rc = check_im(model, lib, prm);
    if (rc) return rc;
    if (!sim || !buf || num_reset < 0 || (!start_at_zero && !phase)) return PHC_EINVAL;
    if (num_reset == 0) return 0;
    const int g = group_lanes(model->num_bodies, prm->num_ext_bodies);
    const dim3 grid(env_blocks(num_reset, g), prm->num_amp_obs_steps + 2);
#define PHC_RESET(DPJ, G) hipLaunchKernelGGL((k_im_reset<DPJ, false, G>), grid, dim3(256), 0, (hipStream_t)stream, *model, *lib, *prm, *sim, *buf, num_reset, env_ids, phase, start_at_zero, 0ull)
    if (prm->dofs_per_joint == 1) { if (g == 64) PHC_RESET(1, 64); else PHC_RESET(1, 32); }
    else { if (g == 64) PHC_RESET(3, 64); else PHC_RESET(3, 32); }
#undef PHC_RESET
    return launch_status();
}

int32_t phc_im_reset_done(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm, const phc_sim_state_t* sim,
                          const phc_im_buffers_t* buf, uint64_t seed, uint64_t counter, int32_t start_at_zero, void* stream) {
    int32_t rc = check_im(model, lib, prm);
    if (rc) return rc;
    if (!sim || !buf) return PHC_EINVAL;
    if (sim->num_envs == 0) return 0;
    if (buf->reset_list && (!buf->reset_count || buf->reset_sublist_cap * PHC_RESET_SUBLISTS < sim->num_envs)) return PHC_EINVAL;
    const int n = buf->reset_list ? buf->reset_sublist_cap * PHC_RESET_SUBLISTS : sim->num_envs;   // groups to launch
    // (with a device-side call counter the host one stays out of the key: a captured launch and an eager one then draw the same numbers)
    const uint64_t key = splitmix64(splitmix64(seed) ^ ((buf->reset_rng_counter ? 0ull : counter) * 0xD1342543DE82EF95ull));
    const int g = group_lanes(model->num_bodies, prm->num_ext_bodies);
    const dim3 grid(env_blocks(n, g), prm->num_amp_obs_steps + 2);
#define PHC_RESET(DPJ, G) hipLaunchKernelGGL((k_im_reset<DPJ, true, G>), grid, dim3(256), 0, (hipStream_t)stream, *model, *lib, *prm, *sim, *buf, n, nullptr, nullptr, start_at_zero, key)
    if (prm->dofs_per_joint == 1) { if (g == 64) PHC_RESET(1, 64); else PHC_RESET(1, 32); }
    else { if (g == 64) PHC_RESET(3, 64); else PHC_RESET(3, 32); }
#undef PHC_RESET
    return launch_status();
}

int32_t phc_im_reset_from_state(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm,
                                const phc_sim_state_t* sim, const phc_im_buffers_t* buf, int32_t num_reset, const int64_t* env_ids,
                                int32_t fill_history, void* stream) {
    int32_t rc = check_im(model, lib, prm);
    if (rc) return rc;
    if (!sim || !buf || num_reset < 0 || (num_reset > 0 && !env_ids)) return PHC_EINVAL;
    if (num_reset == 0) return 0;
    if (group_lanes(model->num_bodies, prm->num_ext_bodies) == 64)
        hipLaunchKernelGGL(k_im_reset_from_state<64>, dim3(env_blocks(num_reset, 64)), dim3(256), 0, (hipStream_t)stream, *model, *lib, *prm, *sim,
                           *buf, num_reset, env_ids, fill_history);
    else
        hipLaunchKernelGGL(k_im_reset_from_state<32>, dim3(env_blocks(num_reset, 32)), dim3(256), 0, (hipStream_t)stream, *model, *lib, *prm, *sim,
                           *buf, num_reset, env_ids, fill_history);
    return launch_status();
}

int32_t phc_amp_obs_demo(const phc_model_t* model, const phc_motion_lib_t* lib, const phc_im_params_t* prm, int32_t n,
                         const int64_t* motion_ids, const float* motion_times0, float* amp_obs_demo, void* stream) {
    int32_t rc = check_im(model, lib, prm);
    if (rc) return rc;
    if (n < 0) return PHC_EINVAL;
    if (n == 0) return 0;
    if (group_lanes(model->num_bodies, prm->num_ext_bodies) == 64)
        hipLaunchKernelGGL(k_amp_obs_demo<64>, dim3(env_blocks(n, 64), prm->num_amp_obs_steps), dim3(256), 0, (hipStream_t)stream, *model, *lib,
                           *prm, n, motion_ids, motion_times0, amp_obs_demo);
    else
        hipLaunchKernelGGL(k_amp_obs_demo<32>, dim3(env_blocks(n, 32), prm->num_amp_obs_steps), dim3(256), 0, (hipStream_t)stream, *model, *lib,
                           *prm, n, motion_ids, motion_times0, amp_obs_demo);
    return launch_status();
}

int32_t phc_gae(int32_t horizon, int32_t n, const float* fdones, const float* values, const float* rewards,
                const float* next_values, float gamma, float tau, float* advs, void* stream) {
    if (horizon < 0 || n < 0) return PHC_EINVAL;
    if (horizon == 0 || n == 0) return 0;
    hipLaunchKernelGGL(k_gae, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, horizon, n, fdones, values, rewards,
                       next_values, gamma, tau, advs);
    return launch_status();
}

int32_t phc_fk(const phc_model_t* model, int64_t num_frames, const float* local_rot, const float* root_trans, float* global_rot,
               float* global_pos, void* stream) {
    if (!model || model->num_bodies < 1 || model->num_bodies > PHC_MAX_BODIES || num_frames < 0) return PHC_EINVAL;
    if (num_frames == 0) return 0;
    hipLaunchKernelGGL(k_fk, dim3((unsigned)((num_frames + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *model, num_frames,
                       local_rot, root_trans, global_rot, global_pos);
    return launch_status();
}

}  // extern "C"
