"""TEST INFRASTRUCTURE -- one env step of the HIP task checked against the numpy oracle (oracle/phc_oracle.py, pinned to the
reference's goldens by tests/test_oracle_golden.py) and the fp64 dense dynamics oracle (oracle/dyn_oracle.py), driven with the task's
own tensors.  Shared by tests/test_env_gpu.py (N = 64, BASELINE configs[1] at 4096) and tests/test_config_sizes_gpu.py (configs[2]
at 8192 envs with a multi-clip library, configs[4] H1 at 4096, G1 at 4096).

Reference order of a step (phc/env/tasks/humanoid.py:1634-1650, humanoid_im.py:694-948,1117-1190): progress += 1; reward and reset at
t = progress dt + start + offset; observations against t + dt; AMP history shifted by one with the new frame in front."""
import numpy as np
import torch

import phc_oracle as po

F = np.float32


def lib_dict(task):
    ml = task._motion_lib
    robot = getattr(ml, "dofs_per_joint", 3) == 1
    keys = ("gts", "grs", "gvs", "gavs", "dvs", "dof_pos", "gts_t", "grs_t") if robot else ("gts", "grs", "gvs", "gavs", "lrs", "dvs")
    d = {k: getattr(ml, k).cpu().numpy() for k in keys}
    d.update(motion_lengths=ml._motion_lengths.cpu().numpy(), motion_dt=ml._motion_dt.cpu().numpy(),
             motion_num_frames=ml._motion_num_frames.cpu().numpy(), length_starts=ml.length_starts.cpu().numpy())
    return d


class StepChecker:
    """`before()` snapshots what a step overwrites, `after()` recomputes every post-physics output with the oracle."""

    def __init__(self, task, lib=None):
        self.task = task
        self.robot = bool(task._is_robot)
        self.lib = lib if lib is not None else lib_dict(task)   # (the library only changes on resample_motions(): cache it across steps)
        self.nb, self.nd = task.num_bodies, task.num_dof

    def before(self):
        t = self.task
        self.amp_before = t._amp_obs_buf.clone().cpu().numpy()
        self.prog_before = t.progress_buf.cpu().numpy().copy()
        self.root0 = t._root_states.cpu().numpy().copy()
        self.dof0 = t._dof_state.view(t.num_envs, self.nd, 2).cpu().numpy().copy()
        self.st_before = t._motion_start_times.cpu().numpy().copy()
        self.so_before = t._motion_start_times_offset.cpu().numpy().copy()
        self.cycle_before = t._cycle_counter.cpu().numpy().copy()
        self.goff_before = t._global_offset.cpu().numpy().copy()

    def dynamics(self, actions, envs, pos_atol=1e-3, root_atol=2e-3):
        """The stepper's result on `envs` against the fp64 dense oracle stepped from the pre-step state (same model, same PD targets)."""
        import dyn_oracle as do
        t = self.task
        if t.control_mode == "pd":
            tgt = (t._torque_target_offset + t._torque_target_scale * torch.clip(actions, -10, 10)).cpu().numpy()
        else:
            tgt = (t._pd_action_offset + t._pd_action_scale * actions).cpu().numpy()
        tgt[:, t._freeze_mask.cpu().numpy() != 0] = 0
        sp = t._sim_params
        params = dict(self_collision=int(sp.self_collision), control_mode=int(sp.control_mode), limit_stiffness=float(sp.limit_stiffness),
                      limit_damping=float(sp.limit_damping), contact_stiffness=float(sp.contact_stiffness), contact_damping=float(sp.contact_damping),
                      friction=float(sp.friction), friction_viscous=float(sp.friction_viscous), contact_model=int(sp.contact_model),
                      contact_iterations=int(sp.contact_iterations), contact_impedance=float(sp.contact_impedance),
                      max_depenetration_velocity=float(sp.max_depenetration_velocity), bounce_threshold_velocity=float(sp.bounce_threshold_velocity),
                      restitution=float(sp.restitution), contact_offset=float(sp.contact_offset))
        if int(sp.inertia_lag):
            # the lagged scheme (the task's default since round 6) has no dense form: its exact-arithmetic reference is the double-precision build of the kernel's own
            # recursion (oracle/hostemu/hostemu64.cpp), which equals the dense oracle to 1e-12 with the lag off (tests/test_dynamics.py)
            import hostemu_util as hu
            envs = list(envs)
            ref = hu.sim_step_f64(t.model, sp, self.root0[envs], self.dof0[envs], tgt[envs], t.control_freq_inv, float(t._kp_scale), float(t._kd_scale))
            for k, e in enumerate(envs):
                np.testing.assert_allclose(t._rigid_body_pos[e].cpu().numpy(), ref["rbs"][k][:, 0:3], atol=pos_atol, err_msg=f"env {e}: body positions vs the fp64 recursion (lagged scheme)")
                np.testing.assert_allclose(t._root_states[e].cpu().numpy(), ref["root"][k], atol=root_atol, rtol=1e-3, err_msg=f"env {e}: root state vs the fp64 recursion (lagged scheme)")
            return
        for e in envs:
            r, d, rbs, tau, fc = do.sim_step(t.model, self.root0[e], self.dof0[e], tgt[e], params=params, sim_dt=t.sim_dt, substeps=int(sp.substeps),
                                             num_sim_calls=t.control_freq_inv)
            np.testing.assert_allclose(t._rigid_body_pos[e].cpu().numpy(), rbs[:, 0:3], atol=pos_atol, err_msg=f"env {e}: body positions vs fp64 dense oracle")
            np.testing.assert_allclose(t._root_states[e].cpu().numpy(), r, atol=root_atol, rtol=1e-3, err_msg=f"env {e}: root state vs fp64 dense oracle")

    def after(self, obs, rew, done, info, atol=1e-4):
        t, lib, robot = self.task, self.lib, self.robot
        torch.cuda.synchronize()
        n, nb = t.num_envs, self.nb
        dt = F(t.dt)
        prog = t.progress_buf.cpu().numpy()
        np.testing.assert_array_equal(prog, self.prog_before + 1)
        st, so = t._motion_start_times.cpu().numpy(), t._motion_start_times_offset.cpu().numpy()
        mids = t._sampled_motion_ids.cpu().numpy()
        goff = t._global_offset.cpu().numpy()
        lookup = po.get_motion_state_robot if robot else po.get_motion_state
        cyc = t._cycle_counter.cpu().numpy()
        wrapped = np.zeros(n, bool)
        if t.cycle_motion:
            # humanoid_im.py:1117-1145: a clip that ran out restarts at a freshly sampled time, the clock offset cancels the progress, the reference is
            # moved under the humanoid, and 60 steps of grace follow
            t_old = (prog.astype(F) * dt + self.st_before + self.so_before).astype(F)
            wrapped = t_old >= lib["motion_lengths"][mids]
            if wrapped.any():
                w = np.flatnonzero(wrapped)
                np.testing.assert_array_equal(so[w], (-(prog[w].astype(F)) * dt).astype(F))
                np.testing.assert_array_equal(st[w], po.sample_time_interval(t._cycle_phase.cpu().numpy()[w], lib["motion_lengths"][mids[w]]))
                np.testing.assert_array_equal(cyc[w], 60)
                rr = lookup(lib, mids[w], st[w], None)
                np.testing.assert_allclose(goff[w, :2], t._rigid_body_pos[:, 0, :2].cpu().numpy()[w] - rr["root_pos"][:, :2], atol=2e-5)
            np.testing.assert_array_equal(st[~wrapped], self.st_before[~wrapped])
            np.testing.assert_array_equal(cyc[~wrapped], np.maximum(self.cycle_before[~wrapped] - 1, 0))
        t0 = (prog.astype(F) * dt + st + so).astype(F)
        t1 = ((prog + 1).astype(F) * dt + st + so).astype(F)
        r0, r1 = lookup(lib, mids, t0, goff), lookup(lib, mids, t1, goff)
        # the reward is computed BEFORE `_compute_reset` restarts a clip that ran out (humanoid.py:1634-1650: reward, reset, observations): for those
        # envs its reference is the OLD clock (past the end: clamped to the last frame) under the old offset
        rr = r0 if not wrapped.any() else lookup(lib, mids, (prog.astype(F) * dt + self.st_before + self.so_before).astype(F), self.goff_before)
        bp, br = t._rigid_body_pos.cpu().numpy(), t._rigid_body_rot.cpu().numpy()
        bv, bav = t._rigid_body_vel.cpu().numpy(), t._rigid_body_ang_vel.cpu().numpy()
        assert np.isfinite(bp).all() and np.isfinite(bv).all()
        # ---- reward (R1, R2) ----
        if robot:
            ep, eo = t.extend_body_parent_ids.cpu().numpy(), t.extend_body_pos_in_parent[0].cpu().numpy()
            bpe, bre = po.extend_bodies(bp, br, ep, eo)
            rw, raw = po.compute_imitation_reward(bpe, bre, bv, bav, np.concatenate([rr["rg_pos"], rr["rg_pos_t"][:, nb:]], 1),
                                                  np.concatenate([rr["rb_rot"], rr["rg_rot_t"][:, nb:]], 1), rr["body_vel"], rr["body_ang_vel"], t.reward_specs)
        else:
            rw, raw = po.compute_imitation_reward(bp, br, bv, bav, rr["rg_pos"], rr["rb_rot"], rr["body_vel"], rr["body_ang_vel"], t.reward_specs)
        pr = po.power_reward(t.dof_force_tensor.cpu().numpy(), t._dof_vel.cpu().numpy(), prog, coef=t.power_coefficient)
        np.testing.assert_allclose(info["reward_raw"].cpu().numpy()[:, :4], raw, atol=atol)
        np.testing.assert_allclose(rew.cpu().numpy(), rw + pr, atol=atol, rtol=1e-4)
        # ---- reset / terminate (R5), bit-exact ----
        rid = t._reset_bodies_id.cpu().numpy()
        td = np.broadcast_to(t._termination_distances.cpu().numpy()[rid], (n, len(rid)))
        pass_len = t0 >= lib["motion_lengths"][mids]
        pass_max = prog >= t.max_episode_length - 1
        pass_time = pass_max if t.cycle_motion else pass_len
        reset, term = po.compute_humanoid_im_reset(prog, bp[:, rid], r0["rg_pos"][:, rid], pass_time, td)
        if t.cycle_motion:
            rec = (~pass_time) & (cyc > 0)       # humanoid_im.py:1186-1188
            reset[rec], term[rec] = 0, 0
        np.testing.assert_array_equal(done.cpu().numpy(), reset)
        np.testing.assert_array_equal(info["terminate"].cpu().numpy(), term)
        # ---- observations (R6-R8) ----
        so_ = po.compute_humanoid_observations_smpl_max(bp, br, bv, bav)
        tid = t._track_bodies_id.cpu().numpy()
        to = po.compute_imitation_observations_v6(bp[:, 0], br[:, 0], bp[:, tid], br[:, tid], bv[:, tid], bav[:, tid], r1["rg_pos"][:, tid], r1["rb_rot"][:, tid],
                                                  r1["body_vel"][:, tid], r1["body_ang_vel"][:, tid])
        np.testing.assert_allclose(obs.cpu().numpy(), np.concatenate([so_, to], -1), atol=atol)
        # ---- AMP observation and history (R9, R10) ----
        kid = t._key_body_ids.cpu().numpy()
        if robot:
            amp = po.build_amp_observations_robot(bp[:, 0], br[:, 0], bv[:, 0], bav[:, 0], t._dof_pos.cpu().numpy(), t._dof_vel.cpu().numpy(), bp[:, kid])
        else:
            amp = po.build_amp_observations_smpl(bp[:, 0], br[:, 0], bv[:, 0], bav[:, 0], t._dof_pos.cpu().numpy(), t._dof_vel.cpu().numpy(), bp[:, kid],
                                                 t.dof_subset.numpy())
        S, A = t._num_amp_obs_steps, amp.shape[-1]
        a = info["amp_obs"].cpu().numpy().reshape(n, S, A)
        np.testing.assert_allclose(a[:, 0], amp, atol=atol)
        np.testing.assert_array_equal(a[:, 1:], self.amp_before[:, :-1])
        return dict(wrapped=wrapped, reset=reset, terminate=term)
