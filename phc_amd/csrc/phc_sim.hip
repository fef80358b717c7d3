// phc_sim.hip -- the articulated-body stepper kernel (S10) and its two C-ABI entry points.
//
// Own translation unit because it is compiled with different code-generation flags than the task kernels
// (phc_amd/build.py): `-ffast-math -fno-slp-vectorize`.
//   * -fno-slp-vectorize: the SLP vectoriser packs the 3-vector algebra into v_pk_fma_f32 and then needs ~840 v_mov to
//     marshal register pairs (3975 static instructions, 16 scratch ops); without it 3466 instructions and no scratch;
//   * -ffast-math: v_rcp / v_sqrt / hardware sin-cos instead of the IEEE division and libm expansions: 1949 static
//     instructions.  The stepper has no reference arithmetic to match bit-for-bit (Isaac Gym is closed); its oracle
//     tolerances (tests/test_dynamics.py) hold with these approximations.  The task kernels, which ARE pinned to the
//     reference at 1e-5 and rely on IEEE NaN/division semantics, are NOT compiled this way.
#include <hip/hip_runtime.h>
#include "phc_aba.h"

using namespace phc;

// Phase profile of the stepper (scripts/sim_phase_profile.py builds a SEPARATE library with -DPHC_SIM_PROFILE; the product
// library never contains this): per-wavefront s_memtime deltas accumulated per phase, summed over wavefronts into a device array.
#ifdef PHC_SIM_PROFILE
__device__ unsigned long long g_phc_prof[16];
__device__ unsigned long long g_phc_prof_wg[8192][10];   // the same per workgroup (= wavefront), of the LAST launch: the launch lasts as long as its slowest wavefront
__device__ unsigned long long g_phc_prof_where[8192][2];   // round 5: [start cycle of the wavefront, XCC_ID << 32 | HW_ID]: WHERE and WHEN the slow wavefronts ran
extern "C" int32_t phc_debug_profile_wg(unsigned long long* out, int32_t nwg) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phc_prof_wg), sizeof(unsigned long long) * 10 * (nwg < 8192 ? nwg : 8192)) == hipSuccess ? 0 : -1;
}
extern "C" int32_t phc_debug_profile_where(unsigned long long* out, int32_t nwg) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phc_prof_where), sizeof(unsigned long long) * 2 * (nwg < 8192 ? nwg : 8192)) == hipSuccess ? 0 : -1;
}
extern "C" int32_t phc_debug_profile(unsigned long long* out16, int32_t reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phc_prof), sizeof(g_phc_prof)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_phc_prof), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
// phase ablation (scripts/probes/sim_ablation.py): bit b set = phase b of the list there is skipped (timing only; the results are then meaningless)
__device__ int g_phc_skip;
extern "C" int32_t phc_debug_set_skip(int32_t mask) { return hipMemcpyToSymbol(HIP_SYMBOL(g_phc_skip), &mask, sizeof(mask)) == hipSuccess ? 0 : -1; }
// single-wave timeline (scripts/probes/sim_timeline.py): workgroup g_phc_tl_block stamps s_memtime at PHC_TL(id) points of sub-step 1
__device__ unsigned long long g_phc_tl[512];
__device__ int g_phc_tl_block = -1;
extern "C" int32_t phc_debug_timeline(unsigned long long* out512, int32_t block) {
    if (out512 && hipMemcpyFromSymbol(out512, HIP_SYMBOL(g_phc_tl), sizeof(g_phc_tl)) != hipSuccess) return -1;
    unsigned long long z[512] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phc_tl), z, sizeof(z)) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phc_tl_block), &block, sizeof(block)) == hipSuccess ? 0 : -1;
}
#define PHC_TL_DECL int tl_n = 0; const bool tl_on = (int)blockIdx.x == g_phc_tl_block;
#define PHC_TL(id) if (tl_on && tl_sub == 1 && tl_n < 255) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); \
        if (threadIdx.x == 0) { g_phc_tl[2 * tl_n] = (unsigned long long)(id); g_phc_tl[2 * tl_n + 1] = t_; } ++tl_n; __builtin_amdgcn_sched_barrier(0); }
#define PHC_SKIP_DECL const int skip_mask = g_phc_skip;
#define PHC_SKIP(b) ((skip_mask >> (b)) & 1)
#define PHC_PROF_DECL unsigned long long prof_acc[10] = {0}; unsigned long long prof_t = __builtin_readcyclecounter(); const unsigned long long prof_t0 = prof_t; \
        const unsigned long long prof_hw = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#define PHC_PROF(i) if (!PHC_SKIP(15)) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_ = __builtin_readcyclecounter(); prof_acc[i] += t_ - prof_t; prof_t = t_; __builtin_amdgcn_sched_barrier(0); }
#define PHC_PROF_FLUSH if (threadIdx.x == 0 && !PHC_SKIP(15)) { for (int i_ = 0; i_ < 10; ++i_) { atomicAdd(&g_phc_prof[i_], prof_acc[i_]); if (blockIdx.x < 8192) g_phc_prof_wg[blockIdx.x][i_] = prof_acc[i_]; } atomicAdd(&g_phc_prof[15], 1ull); \
        if (blockIdx.x < 8192) { g_phc_prof_where[blockIdx.x][0] = prof_t0; g_phc_prof_where[blockIdx.x][1] = prof_hw; } }
#else
#define PHC_PROF_DECL
#define PHC_PROF(i)
#define PHC_PROF_FLUSH
#define PHC_SKIP_DECL
#define PHC_SKIP(b) false
#define PHC_TL_DECL
#define PHC_TL(id)
#endif

// A2: pd_tar = offset + scale * action (humanoid.py:1711-1713); env.res_action (sim.pd_ref set): reference joint position + scale * action,
// kept within pi / 2 of the current joint position (humanoid_im.py:1094-1099); frozen DoFs -> 0 (humanoid.py:1549-1554)
__device__ __forceinline__ float pd_target_of(const phc_sim_state_t& sim, const float* __restrict__ actions, const float* __restrict__ pd_off,
                                              const float* __restrict__ pd_scale, const int32_t* __restrict__ freeze, int64_t env, int nd, int d) {
    const float sa = __fmul_rn(pd_scale[d], actions[env * nd + d]);
    float t;
    if (sim.pd_ref != nullptr) {
        const float q = sim.dof_state[(env * nd + d) * 2];
        const float half_pi = 1.57079637f;   // float32(np.pi / 2)
        t = fmaxf(fminf(__fadd_rn(sim.pd_ref[env * nd + d], sa), __fadd_rn(q, half_pi)), __fsub_rn(q, half_pi));
    } else {
        t = __fadd_rn(pd_off[d], sa);
    }
    if (freeze != nullptr && freeze[d]) t = 0.f;
    return t;
}


// ------------------------------------------------------------------------------------------
// Staged epilogue.  A lane would otherwise issue ~50 scattered 4-byte global stores, all wavefronts at the same instant.  The output slices of the E consecutive envs of ONE wavefront are contiguous and 16-byte aligned in every
// simulator tensor (E * 13, E * NB * 13, E * ND * 2, ... floats), so the lanes first write their values into the (now idle) LDS
// exchange area in exactly that layout and the wavefront then streams each slice out with coalesced dwordx4 / dwordx2 stores.
// ------------------------------------------------------------------------------------------
struct StageLayout { int root, dof, force, contact, rbs, total; };
__device__ __forceinline__ StageLayout stage_layout(int E, int nb, int nd, bool with_force, bool with_contact) {
    StageLayout o;
    o.root = 0;
    o.dof = o.root + ((E * 13 + 3) & ~3);
    o.force = o.dof + ((E * nd * 2 + 3) & ~3);
    o.contact = o.force + (with_force ? ((E * nd + 3) & ~3) : 0);
    o.rbs = o.contact + (with_contact ? ((E * nb * 3 + 3) & ~3) : 0);
    o.total = o.rbs + ((E * nb * 13 + 3) & ~3);
    return o;
}
// a phc_sim_state_t whose tensors are the staging slices, indexed by the env's position in the wavefront
__device__ __forceinline__ phc_sim_state_t stage_state(const phc_sim_state_t& sim, float* stage, const StageLayout& o) {
    phc_sim_state_t st = sim;
    st.root_states = stage + o.root;
    st.dof_state = stage + o.dof;
    st.dof_force = sim.dof_force ? stage + o.force : nullptr;
    st.contact_force = sim.contact_force ? stage + o.contact : nullptr;
    st.rigid_body_state = stage + o.rbs;
    return st;
}
template <int W>
__device__ __forceinline__ void stage_copy_out(float* __restrict__ dst, const float* __restrict__ src, int n, int tid, int nthreads) {
    if (W == 4) {
        for (int i = tid; i < n / 4; i += nthreads) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    } else {
        for (int i = tid; i < n / 2; i += nthreads) reinterpret_cast<float2*>(dst)[i] = reinterpret_cast<const float2*>(src)[i];
    }
}
// all lanes of the wavefront: stream the staged slices of envs [env0, env0 + E) to the simulator tensors
template <int E>
__device__ __forceinline__ void stage_flush(const phc_sim_state_t& sim, const float* stage, const StageLayout& o, int64_t env0, int nb, int nd) {
    constexpr int W = (E % 4 == 0) ? 4 : 2;
    const int tid = threadIdx.x;
    stage_copy_out<W>(sim.root_states + env0 * 13, stage + o.root, E * 13, tid, 64);
    stage_copy_out<W>(sim.dof_state + env0 * nd * 2, stage + o.dof, E * nd * 2, tid, 64);
    if (sim.dof_force) stage_copy_out<W>(sim.dof_force + env0 * nd, stage + o.force, E * nd, tid, 64);
    if (sim.contact_force) stage_copy_out<W>(sim.contact_force + env0 * nb * 3, stage + o.contact, E * nb * 3, tid, 64);
    stage_copy_out<W>(sim.rigid_body_state + env0 * nb * 13, stage + o.rbs, E * nb * 13, tid, 64);
}
__device__ __forceinline__ bool stage_aligned(const phc_sim_state_t& sim) {
    const uintptr_t a = (uintptr_t)sim.root_states | (uintptr_t)sim.dof_state | (uintptr_t)sim.rigid_body_state |
                        (uintptr_t)sim.dof_force | (uintptr_t)sim.contact_force;
    return (a & 15) == 0;
}

// ------------------------------------------------------------------------------------------
// S10: the stepper.  One lane per body, GRP = 32 lanes per env (articulations of up to 32 bodies: two envs per wavefront) or
// GRP = 64 (up to 64 bodies -- Unitree G1 has 38: one env per wavefront), blockDim = 64: ONE wavefront per workgroup, so
// the level-synchronous tree sweeps synchronise with a single-wave barrier and every SIMD of the chip carries
// two independent dependency chains (2048 wavefronts at N = 4096).
// Measured alternative (round 1, profiles/r01_notes.md): a level-major mapping (workgroup = 16 envs, wavefront = same
// body of many envs, multi-wave barriers) runs ~100 % of its lanes but leaves 3 of 4 SIMDs idle at N = 4096 and needs
// >256 VGPRs: 214-252 us vs 158 us for this mapping.  __launch_bounds__(64, 2): two wavefronts per SIMD (<= 256 VGPRs,
// 68 B/lane of scratch) beats one (272 registers, no scratch: 195 us) and three (168 VGPRs, 412 B scratch: 280 us).
// ------------------------------------------------------------------------------------------
// OCC: wavefronts per SIMD the register allocation aims at.  2 (<= 256 VGPRs, no scratch) is what every launch uses.  OCC = 3 (168 VGPRs, 116 B / lane
// of scratch) keeps 3072 wavefronts resident instead of 2048 -- and was measured SLOWER at every size (round 4, profiles/r04_stepper/occupancy_2_vs_3_waves_per_simd.txt:
// 96.6 vs 78.0 us at 4096 envs, 168.9 vs 145.3 at 8192, 241.0 vs 211.9 at 12288): the spilled wavefront's longer stream costs more than the
// third resident wavefront hides.  Kept behind lane_mapping = 3 so that the measurement can be repeated; never chosen automatically.
// (Round 5: the per-lane force accumulators of `force_average` took the workgroup's LDS from 16.9 to 18.4 KB; eight workgroups per CU = two per SIMD still fit
// the 160 KB, a third per SIMD would not -- the compiler says so when it builds this instantiation; the knob now measures the 168-VGPR code at occupancy 2.)
// LAG: the instantiation whose sub-steps behind the first one of a simulate() call keep its articulated inertias (phc_sim_params_t.inertia_lag); a
// template parameter, not a run-time branch: with the switch compiled into the one kernel it took 256 VGPRs + 12 spilled SGPRs instead of 224 and the
// every-sub-step-fresh launch went from 77 to 82 us (round 5, same box).
template <bool STEP, int JT, int GRP, bool SHAPES = false, bool RIGID = false, int OCC = 2, bool LAG = false>
__global__ __launch_bounds__(64, OCC) void k_sim_step(phc_model_t model_all, phc_sim_params_t prm, phc_sim_state_t sim,
                                                const float* __restrict__ actions, const float* __restrict__ pd_off,
                                                const float* __restrict__ pd_scale, const int32_t* __restrict__ freeze,
                                                int num_sim_calls, const int64_t* __restrict__ env_ids, int num_listed) {
    __shared__ float xch_all[64 * PHC_XCH_STRIDE];   // one exchange slot per lane == body
    __shared__ float cap_all[64 * PHC_CAP_STRIDE];
    __shared__ int pair_all[PHC_SC_MAX_PER_LANE * 64];   // candidate pairs of body-body contact: [pair slot][thread]
    __shared__ float favg_all[STEP ? 64 * 6 : 1];        // force_average: per-lane sums of S4 / S5 over the sub-steps
    const int lane = threadIdx.x & (GRP - 1);
    const int grp = threadIdx.x / GRP;
    const int64_t slot = (int64_t)blockIdx.x * (64 / GRP) + grp;
    // env_ids (refresh of a teleported subset only): slot -> listed env
    const int64_t env = (!STEP && env_ids != nullptr) ? (slot < num_listed ? env_ids[slot] : sim.num_envs) : slot;
    // SHAPES (per-env body shapes): the env's block of the model tables -- a per-lane pointer pair; the single-shape instantiation keeps the
    // tables behind scalar registers (with the select compiled in unconditionally the kernel spilled 188 B / lane: 35 MB of scratch traffic)
    const phc_model_t model = SHAPES ? model_for_env(model_all, sim, env) : model_all;
    const int nb = model.num_bodies, nd = model.num_dof;
    const bool active = env < sim.num_envs && lane < nb;
    const int body = lane;   // one lane per body, in the model's body order
    Xch x;
    x.base = xch_all + grp * GRP * PHC_XCH_STRIDE;

    AbaLane L;
    L.level = L.slevel = -1;
    PHC_SKIP_DECL
    PHC_PROF_DECL
    PHC_TL_DECL
    if (active) {
        aba_load_model(L, model, body);
        if (JT == PHC_JT_REVOLUTE) aba_load_model_rev(L, model, body);
        // (the state is requested BEFORE the new PD targets are stored: no load of this prologue has to wait behind a store, and the targets go
        //  into the lane's registers directly instead of through memory)
        const bool new_targets = STEP && actions != nullptr && body >= 1;
        aba_load_state<JT>(L, sim, nd, env, body, !new_targets);
        if (new_targets) {
            float tg[3] = {0.f, 0.f, 0.f};
            for (int k = 0; k < (JT == PHC_JT_REVOLUTE ? 1 : 3); ++k) {
                const int d = L.dof_start + k;
                tg[k] = pd_target_of(sim, actions, pd_off, pd_scale, freeze, env, nd, d);
                sim.pd_target[env * nd + d] = tg[k];
            }
            L.target = v3(tg[0], tg[1], tg[2]);
        }
    }
    // the idle lanes behind the bodies publish the extra collision shapes (phc_aba.h): their capsule records are loaded once, here
    const bool shape_lane = active || (STEP && prm.self_collision && env < sim.num_envs && lane < nb + model_num_extra_shapes(model));
    if (shape_lane && !active) aba_load_extra_shape(L, model, lane - nb);
    // initial kinematics by pointer jumping too (round 4: 4 composition steps instead of max_level + 1 = 9 level-steps for the SMPL tree)
    if (!PHC_SKIP(8)) {
        const int jsteps = model_jump_steps(model);
        aba_fk_jump_begin(L, body, x);
        __syncthreads();
        for (int k = 0; k < jsteps; ++k) {
            aba_fk_jump_step(L, k, x);
            __syncthreads();
            if (active) aba_write_kin(L, xslot(x, body), Xch::es, 6);
            __syncthreads();
        }
    }
    PHC_PROF(0)
    if (STEP) {
        const float dt = prm.sim_dt / (float)prm.substeps;
        const int nsub = num_sim_calls * prm.substeps;
        float* caps = cap_all + grp * GRP * PHC_CAP_STRIDE;
        uint32_t near_pairs = 0;
        if (prm.self_collision) aba_load_pairs<PHC_SC_MAX_PER_LANE>(pair_all + threadIdx.x, 64, model, lane, GRP);
        // the backward / acceleration sweeps walk the solver tree (model.py solver_tree(): re-rooted where that makes it shallower)
        const int solver_depth = model_solver_depth(model, true);
        const int jump_steps = model_jump_steps(model);
        const bool rerooted = model_tab(model, 11, 3) != 0;
        for (int s = 0; s < nsub; ++s) {
            const int tl_sub = s; (void)tl_sub;
            PHC_TL(1)
            if (prm.self_collision && !PHC_SKIP(0)) {   // body-body contact from the kinematics the last sweep left in the exchange slots
                if (shape_lane) aba_publish_shape(L, lane, x, caps);   // body lanes: the primary capsules; the idle lanes behind them: the extra shapes
                __syncthreads();
                if (env < sim.num_envs) aba_collide_pairs<PHC_SC_MAX_PER_LANE>(pair_all + threadIdx.x, 64, prm, dt, x, caps, near_pairs, s == 0);
                __syncthreads();
                if (active) aba_collect_self(L, body, caps);
            }
            PHC_PROF(1)
            PHC_TL(2)
            if (active && !PHC_SKIP(1)) aba_velocity_products(L, model, body, x, true);
            // contact_model 1 (rigid): the sub-step's solve is repeated contact_iterations times, each pass with the active set and friction cone the
            // previous one implies (phc_aba.h aba_ground_contact_rigid); the penalty model is the single pass it always was
            const int passes = RIGID ? (prm.contact_iterations < 1 ? 1 : prm.contact_iterations) : 1;
            // inertia_lag (round 5; penalty contact): the sub-steps behind the first one of a simulate() call keep its articulated inertias and joint-space
            // inverses and only redo the bias-force recursion (aba_body_init / aba_backward_level, `lag`)
            const bool lag = LAG && !RIGID && (s % prm.substeps) != 0;
            for (int pass = 0; pass < passes; ++pass) {
            PHC_TL(3)
            if (active && !PHC_SKIP(1)) aba_body_init<JT, RIGID>(L, model, prm, dt, body, s % prm.substeps == 0, true, pass, lag);
            if (active && PHC_SKIP(9)) aba_body_init<JT, RIGID>(L, model, prm, dt, body, s % prm.substeps == 0, true, pass, lag);   // (profiling builds: the phase a second time, loads warm -- its pure instruction cost)
            PHC_PROF(2)
            PHC_TL(4)
            if (JT == PHC_JT_SPHERICAL && rerooted && pass == 0 && !PHC_SKIP(2)) {   // reversed bodies take the drive terms of their solver parent's joint
                if (active) aba_publish_drive(L, body, x);
                __syncthreads();
                if (active) aba_fetch_drive(L, body, x);
                __syncthreads();
            }
            PHC_PROF(3)
            PHC_TL(5)
            if (!PHC_SKIP(3)) for (int l = solver_depth; l >= 0; --l) { aba_backward_level<JT>(L, l, body, x, lag); __syncthreads(); PHC_TL(120 + l) }
            PHC_PROF(4)
            if (!PHC_SKIP(4)) {
                for (int l = 0; l <= solver_depth; ++l) { aba_accel_level<JT>(L, l, body, x); __syncthreads(); PHC_TL(140 + l) }
            }
            }
            if (RIGID && active && (s == nsub - 1 || prm.force_average)) aba_publish_contact_rigid(L, model, prm, sim, dt, env, body, true);   // S4 / S6 from the final solve
            if (JT == PHC_JT_SPHERICAL && rerooted && !PHC_SKIP(4)) aba_accel_finish(L, model, body, x);
            PHC_PROF(5)
            PHC_TL(6)
            if (!PHC_SKIP(5)) aba_integrate_joint<JT>(L, prm, dt);
            if (prm.force_average) aba_force_accumulate(L, s, nsub, favg_all + threadIdx.x * 6);   // S4 / S5 as means over the sub-steps of the env step instead of the last one's values
            PHC_PROF(6)
            PHC_TL(7)
            if (!PHC_SKIP(6)) aba_fk_jump_begin(L, body, x);   // kinematics by pointer jumping: jump_steps composition steps instead of max_level + 1 level-steps
            __syncthreads();
            for (int k = 0; k < (PHC_SKIP(6) ? 0 : jump_steps); ++k) {
                aba_fk_jump_step(L, k, x);
                __syncthreads();
                if (active) aba_write_kin(L, xslot(x, body), Xch::es, 6);
                __syncthreads();
                PHC_TL(160 + k)
            }
            PHC_PROF(7)
            PHC_TL(8)
        }
    }
    // S7: the last forward sweep already produced the end-of-step kinematics
    constexpr int E = 64 / GRP;
    const int64_t env0 = (int64_t)blockIdx.x * E;
    const StageLayout so = stage_layout(E, nb, nd, sim.dof_force != nullptr, sim.contact_force != nullptr);
    if (PHC_SKIP(7)) {
    } else if (STEP && E >= 2 && env0 + E <= sim.num_envs && stage_aligned(sim) && so.total <= 64 * PHC_XCH_STRIDE) {   // staged epilogue, see above
        const phc_sim_state_t st = stage_state(sim, xch_all, so);
        if (active) {
            aba_store_state<JT>(L, st, nd, grp, body);
            aba_publish_body(L, st, nb, grp, body, true);
        }
        __syncthreads();
        stage_flush<E>(sim, xch_all, so, env0, nb, nd);
    } else if (active) {
        if (STEP) aba_store_state<JT>(L, sim, nd, env, body);
        aba_publish_body(L, sim, nb, env, body, STEP);
    }
    if (STEP && !RIGID && active && sim.force_sensor != nullptr) aba_publish_sensors(L, model, prm, sim, prm.sim_dt / (float)prm.substeps, env, body);   // S6
    PHC_PROF(8)
    if (STEP) { PHC_PROF_FLUSH }
}

template <bool STEP, int JT, bool SHAPES, bool RIGID>
static void sim_launch_cm(const phc_model_t* model, const phc_sim_params_t& prm, const phc_sim_state_t* sim, const float* actions,
                          const float* off, const float* scale, const int32_t* freeze, int num_sim_calls, hipStream_t stream,
                          const int64_t* env_ids, int num_listed) {
    const int64_t groups = env_ids ? num_listed : sim->num_envs;
    const bool wide = model->num_bodies > 32;   // more bodies than a 32-lane group holds: one env per wavefront
    const bool occ3 = STEP && !RIGID && !SHAPES && JT == PHC_JT_SPHERICAL && !wide && prm.lane_mapping == 3;   // (experiment knob, see k_sim_step)
    const bool lag = STEP && !RIGID && prm.inertia_lag != 0;
    if (occ3)
        hipLaunchKernelGGL((k_sim_step<STEP, JT, 32, SHAPES, RIGID, (STEP && !RIGID && !SHAPES && JT == PHC_JT_SPHERICAL) ? 3 : 2>), dim3((groups + 1) / 2), dim3(64), 0, stream,
                           *model, prm, *sim, actions, off, scale, freeze, num_sim_calls, env_ids, num_listed);
    else if (lag && wide)
        hipLaunchKernelGGL((k_sim_step<STEP, JT, 64, SHAPES, RIGID, 2, STEP && !RIGID>), dim3(groups), dim3(64), 0, stream, *model, prm, *sim, actions, off, scale, freeze,
                           num_sim_calls, env_ids, num_listed);
    else if (lag)
        hipLaunchKernelGGL((k_sim_step<STEP, JT, 32, SHAPES, RIGID, 2, STEP && !RIGID>), dim3((groups + 1) / 2), dim3(64), 0, stream, *model, prm, *sim, actions, off, scale, freeze,
                           num_sim_calls, env_ids, num_listed);
    else if (wide)
        hipLaunchKernelGGL((k_sim_step<STEP, JT, 64, SHAPES, RIGID>), dim3(groups), dim3(64), 0, stream, *model, prm, *sim, actions, off, scale, freeze,
                           num_sim_calls, env_ids, num_listed);
    else
        hipLaunchKernelGGL((k_sim_step<STEP, JT, 32, SHAPES, RIGID>), dim3((groups + 1) / 2), dim3(64), 0, stream, *model, prm, *sim, actions, off, scale, freeze,
                           num_sim_calls, env_ids, num_listed);
}
template <bool STEP, int JT, bool SHAPES>
static void sim_launch_jt(const phc_model_t* model, const phc_sim_params_t& prm, const phc_sim_state_t* sim, const float* actions,
                          const float* off, const float* scale, const int32_t* freeze, int num_sim_calls, hipStream_t stream,
                          const int64_t* env_ids, int num_listed) {
    if (STEP && prm.contact_model == 1)   // rigid ground contact: its own instantiation, the penalty kernel is untouched by it
        sim_launch_cm<STEP, JT, SHAPES, STEP>(model, prm, sim, actions, off, scale, freeze, num_sim_calls, stream, env_ids, num_listed);
    else
        sim_launch_cm<STEP, JT, SHAPES, false>(model, prm, sim, actions, off, scale, freeze, num_sim_calls, stream, env_ids, num_listed);
}

template <bool STEP>
static void sim_launch(const phc_model_t* model, const phc_sim_params_t& prm, const phc_sim_state_t* sim, const float* actions,
                       const float* off, const float* scale, const int32_t* freeze, int num_sim_calls, hipStream_t stream,
                       const int64_t* env_ids = nullptr, int num_listed = 0) {
    if (model->num_dof == model->num_bodies - 1 && model->num_bodies > 2)  // one revolute joint per body (robots; one shape)
        sim_launch_jt<STEP, PHC_JT_REVOLUTE, false>(model, prm, sim, actions, off, scale, freeze, num_sim_calls, stream, env_ids, num_listed);
    else if (model->num_shapes > 1 && sim->env_shape != nullptr)   // per-env body shapes (SMPL family)
        sim_launch_jt<STEP, PHC_JT_SPHERICAL, true>(model, prm, sim, actions, off, scale, freeze, num_sim_calls, stream, env_ids, num_listed);
    else
        sim_launch_jt<STEP, PHC_JT_SPHERICAL, false>(model, prm, sim, actions, off, scale, freeze, num_sim_calls, stream, env_ids, num_listed);
}

static inline int32_t launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int32_t)e;
}

extern "C" {

static int32_t check_model(const phc_model_t* m) {
    if (!m || m->num_bodies < 1 || m->num_bodies > PHC_MAX_BODIES || !m->ints || !m->floats) return PHC_EINVAL;
    // all-spherical (SMPL family) or all-revolute (H1 / G1) articulations
    if (m->num_dof != 3 * (m->num_bodies - 1) && m->num_dof != m->num_bodies - 1) return PHC_EUNSUPPORTED;
    if (m->num_shapes > 1 && m->num_dof != 3 * (m->num_bodies - 1)) return PHC_EUNSUPPORTED;   // per-env shapes: SMPL family only
    if (m->num_shapes > 1 && (m->int_stride <= 0 || m->float_stride <= 0)) return PHC_EINVAL;
    return 0;
}

int32_t phc_sim_step(const phc_model_t* model, const phc_sim_params_t* params, const phc_sim_state_t* sim, const float* actions,
                     const float* pd_action_offset, const float* pd_action_scale, const int32_t* freeze_mask,
                     int32_t num_sim_calls, void* stream) {
    int32_t rc = check_model(model);
    if (rc) return rc;
    if (!params || !sim || sim->num_envs < 0 || params->substeps < 1 || num_sim_calls < 0) return PHC_EINVAL;
    if (actions && (!pd_action_offset || !pd_action_scale)) return PHC_EINVAL;
    if (sim->num_envs == 0) return 0;
    // pairs are dealt round-robin to the lanes of an env's group: PHC_SC_MAX_PER_LANE each
    if (params->self_collision && model->num_collision_pairs > PHC_SC_MAX_PER_LANE * (model->num_bodies > 32 ? 64 : 32)) return PHC_EUNSUPPORTED;
    if (params->lane_mapping != 0 && params->lane_mapping != 1 && params->lane_mapping != 3) return PHC_EUNSUPPORTED;   // (2 was the two-bodies-per-lane kernel of rounds 1-2: removed)
    if (params->contact_model != 0 && params->contact_model != 1) return PHC_EUNSUPPORTED;
    if (params->contact_model == 1 && (params->contact_iterations < 2 || !(params->contact_impedance > 0.f))) return PHC_EINVAL;
    if (params->contact_model == 1 && params->inertia_lag) return PHC_EUNSUPPORTED;   // (the rigid model re-solves every sub-step contact_iterations times with fresh impedances)
    if (params->inertia_lag && params->lane_mapping == 3) return PHC_EUNSUPPORTED;   // (the three-wavefront experiment build has no lagged instantiation: it would silently run fresh)
    if (params->contact_model == 1 && model->max_body_contact_pts > 32) return PHC_EUNSUPPORTED;   // c_active / c_removed are 32-bit masks: a point beyond them could never be released
    if (params->inertia_lag && model->max_body_contact_pts > PHC_CP_BITS) return PHC_EUNSUPPORTED;  // c_touch: tail points would alternate between full and no force
    sim_launch<true>(model, *params, sim, actions, pd_action_offset, pd_action_scale, freeze_mask, num_sim_calls, (hipStream_t)stream);
    return launch_status();
}

int32_t phc_refresh_body_state(const phc_model_t* model, const phc_sim_state_t* sim, void* stream) {
    int32_t rc = check_model(model);
    if (rc) return rc;
    if (!sim || sim->num_envs < 0) return PHC_EINVAL;
    if (sim->num_envs == 0) return 0;
    phc_sim_params_t prm = {};
    prm.substeps = 1;
    sim_launch<false>(model, prm, sim, nullptr, nullptr, nullptr, nullptr, 0, (hipStream_t)stream);
    return launch_status();
}

int32_t phc_refresh_body_state_indexed(const phc_model_t* model, const phc_sim_state_t* sim, int32_t num, const int64_t* env_ids,
                                       void* stream) {
    int32_t rc = check_model(model);
    if (rc) return rc;
    if (!sim || num < 0 || (num > 0 && !env_ids)) return PHC_EINVAL;
    if (num == 0) return 0;
    phc_sim_params_t prm = {};
    prm.substeps = 1;
    sim_launch<false>(model, prm, sim, nullptr, nullptr, nullptr, nullptr, 0, (hipStream_t)stream, env_ids, num);
    return launch_status();
}

}  // extern "C"
