"""TEST INFRASTRUCTURE ONLY -- a thin CPU "reference env": K consecutive env steps of HumanoidIm with the reference's OWN jit functions
and motion library called in the order its methods call them, and a kinematic stand-in for the physics (body state after a step :=
reference state of that time + noise, growing on some envs so that they terminate) -- pins the orchestration across steps that no
single-step golden sees: progress counting, reset / terminate flags feeding the next step's reset, re-initialisation of the reset
envs (state, observations, AMP history) and the AMP history shift (SURVEY.md 8c "Task classes").  Order restated from

    BaseTask.step                       base_task.py:216-234
    Humanoid.post_physics_step          humanoid.py:1634-1650   (progress += 1, refresh, reward, reset, observations)
    HumanoidIm._compute_reward/_reset   humanoid_im.py:876-948,1117-1190
    HumanoidIm._compute_task_obs        humanoid_im.py:728-871
    HumanoidAMP.post_physics_step       humanoid_amp.py:194-210 (_update_hist_amp_obs, _compute_amp_observations)
    Humanoid._reset_envs / HumanoidAMP  humanoid.py:585-621, humanoid_amp.py:378-398,508-528,559-603, humanoid_im.py:955-1023

Run in the build container:  python oracle/gen_golden_rollout.py   -> tests/golden/rollout_ref_env.npz
(same synthetic clips and library as tests/golden/motion_lib_eval.npz, so the test builds its library from that fixture)"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

ref_shim.install()
import joblib  # noqa: E402
import torch  # noqa: E402

from gen_golden import KEY_BODIES, MJCF, OUT, RESET_BODIES, t2n  # noqa: E402
from phc_amd.utils.synthetic_motion import make_motion_dict  # noqa: E402


def main():
    torch.set_num_threads(1)
    itu = ref_shim.ref_module("phc.utils.isaacgym_torch_utils")
    him = ref_shim.ref_module("phc.env.tasks.humanoid_im")
    hum = ref_shim.ref_module("phc.env.tasks.humanoid")
    hamp = ref_shim.ref_module("phc.env.tasks.humanoid_amp")
    from phc.utils.flags import flags
    from phc.utils.motion_lib_base import FixHeightMode
    from phc.utils.motion_lib_smpl import MotionLibSMPL
    from poselib.poselib.skeleton.skeleton3d import SkeletonTree
    from easydict import EasyDict

    tree = SkeletonTree.from_mjcf(MJCF)
    names = list(tree.node_names)
    parents = t2n(tree.parent_indices).astype(np.int32)
    clips = make_motion_dict(parents, 3, seed=7, body_names=names, lengths=[31, 45, 38])     # == gen_golden.py
    for k in clips:
        clips[k]["root_trans_offset"] = torch.from_numpy(clips[k]["root_trans_offset"])
    tmp = tempfile.mkdtemp()
    pkl = os.path.join(tmp, "synthetic.pkl")
    joblib.dump(clips, pkl)
    NM = 6
    cwd = os.getcwd()
    os.chdir(tmp)
    cfg = EasyDict({"motion_file": pkl, "device": torch.device("cpu"), "fix_height": FixHeightMode.full_fix, "min_length": -1, "max_length": -1,
                    "im_eval": False, "multi_thread": False, "smpl_type": "smpl", "randomrize_heading": True, "step_dt": 1 / 30})
    flags.test, flags.im_eval = True, False
    lib = MotionLibSMPL(cfg)
    lib.load_motions(skeleton_trees=[tree] * NM, gender_betas=torch.zeros(NM, 17), limb_weights=np.zeros((NM, 10)), random_sample=False,
                     start_idx=0, max_len=-1)
    os.chdir(cwd)

    E, K, S, dt = 8, 16, 10, 1 / 30
    NB, ND = 24, 69
    gr = torch.Generator().manual_seed(77)
    rid = torch.tensor([names.index(b) for b in RESET_BODIES])
    kid = torch.tensor([names.index(b) for b in KEY_BODIES])
    dof_subset = torch.from_numpy(np.concatenate([np.arange(i * 3, i * 3 + 3) for i, nm in enumerate(names[1:]) if nm not in ("L_Hand", "R_Hand", "L_Toe", "R_Toe")]))
    specs = {"k_pos": 100., "k_rot": 10., "k_vel": 0.1, "k_ang_vel": 0.1, "w_pos": 0.5, "w_rot": 0.3, "w_vel": 0.1, "w_ang_vel": 0.1}
    term_dist = torch.full((E, NB), 0.25)
    motion_ids = torch.arange(E) % NM

    def amp_from(state_root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel, key_pos):
        n = state_root_pos.shape[0]
        return hamp.build_amp_observations_smpl(state_root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel, key_pos, torch.zeros(n, 11),
                                                torch.zeros(n, 10), dof_subset, True, True, True, False, False, True)

    def observations(ids, st, progress, start):
        """_compute_observations(env_ids): self obs of the current state + task obs against the reference one step ahead (humanoid_im.py:694-726,752)"""
        self_obs = hum.compute_humanoid_observations_smpl_max(st["bp"][ids], st["br"][ids], st["bv"][ids], st["bw"][ids], torch.zeros(len(ids), 11),
                                                              torch.zeros(len(ids), 10), True, True, True, False, False)
        t1 = (progress[ids] + 1) * dt + start[ids] + torch.zeros(len(ids))
        r1 = lib.get_motion_state(motion_ids[ids], t1, offset=torch.zeros(len(ids), 3))
        task_obs = him.compute_imitation_observations_v6(st["bp"][ids, 0], st["br"][ids, 0], st["bp"][ids], st["br"][ids], st["bv"][ids], st["bw"][ids],
                                                         r1["rg_pos"], r1["rb_rot"], r1["body_vel"], r1["body_ang_vel"], 1, True)
        return torch.cat([self_obs, task_obs], dim=-1)

    st = {k: torch.zeros(E, NB, n) for k, n in (("bp", 3), ("br", 4), ("bv", 3), ("bw", 3))}
    st.update(dp=torch.zeros(E, ND), dv=torch.zeros(E, ND))
    progress = torch.zeros(E, dtype=torch.long)
    start = torch.zeros(E)
    amp = torch.zeros(E, S, 196)
    obs = torch.zeros(E, 934)
    reset_buf = torch.ones(E, dtype=torch.long)      # everything resets before the first step (Humanoid.reset())
    log = {k: [] for k in ("reset_ids", "reset_phase", "start_after_reset", "obs_after_reset", "root_after_reset", "dof_after_reset",
                           "state_in", "dof_in", "dof_force", "progress", "rew", "rew_raw", "reset", "terminate", "obs", "amp")}
    for k in range(K):
        # ---- top of step: reset of the envs the previous step flagged (base_task.py:216-220 -> _reset_envs) ----
        ids = torch.nonzero(reset_buf)[:, 0]
        torch.manual_seed(1000 + k)
        phase = torch.rand(ids.shape)                                # the draw sample_time_interval makes (reproduced by reseeding)
        if len(ids):
            torch.manual_seed(1000 + k)
            t = lib.sample_time_interval(motion_ids[ids])            # motion_lib_base.py:413-422 (quantised to 1/30 s)
            r = lib.get_motion_state(motion_ids[ids], t, offset=torch.zeros(len(ids), 3))
            st["bp"][ids], st["br"][ids], st["bv"][ids], st["bw"][ids] = r["rg_pos"], r["rb_rot"], r["body_vel"], r["body_ang_vel"]
            st["dp"][ids], st["dv"][ids] = r["dof_pos"], r["dof_vel"]
            progress[ids] = 0
            start[ids] = t
            reset_buf[ids] = 0
            # _init_amp_obs: current frame from the imposed state, history from the reference at t - j dt (humanoid_amp.py:559-603)
            amp[ids, 0] = amp_from(st["bp"][ids, 0], st["br"][ids, 0], st["bv"][ids, 0], st["bw"][ids, 0], st["dp"][ids], st["dv"][ids], st["bp"][ids][:, kid])
            th = (t.unsqueeze(-1) + (-dt * (torch.arange(0, S - 1) + 1))).view(-1)
            rh = lib.get_motion_state(torch.tile(motion_ids[ids].unsqueeze(-1), [1, S - 1]).view(-1), th)
            amp[ids, 1:] = amp_from(rh["root_pos"], rh["root_rot"], rh["root_vel"], rh["root_ang_vel"], rh["dof_pos"], rh["dof_vel"],
                                    rh["rg_pos"][:, kid]).view(len(ids), S - 1, 196)
            obs[ids] = observations(ids, st, progress, start)
        pad = lambda x, fill=0: torch.cat([x, torch.full((E - len(ids),) + tuple(x.shape[1:]), fill, dtype=x.dtype)])
        log["reset_ids"].append(pad(ids, -1)); log["reset_phase"].append(pad(phase)); log["start_after_reset"].append(start.clone())
        log["obs_after_reset"].append(obs.clone())   # (the re-initialised AMP history shows up, shifted by one, in the next `amp`)
        log["root_after_reset"].append(torch.cat([st["bp"][:, 0], st["br"][:, 0], st["bv"][:, 0], st["bw"][:, 0]], -1).clone())
        log["dof_after_reset"].append(torch.stack([st["dp"], st["dv"]], -1).clone())

        # ---- physics stand-in: the state after this step = reference one step ahead + noise (envs 3, 7 drift away and terminate) ----
        tn = (progress + 1) * dt + start
        rn = lib.get_motion_state(motion_ids, tn, offset=torch.zeros(E, 3))
        noise = lambda shape, s: torch.randn(*shape, generator=gr) * s
        scale = torch.full((E, 1, 1), 0.02)
        scale[3] = 0.02 + 0.05 * k
        scale[7] = 0.3 if k % 5 == 4 else 0.02
        st["bp"] = rn["rg_pos"] + noise((E, NB, 3), 1.0) * scale
        st["br"] = itu.quat_mul(itu.exp_map_to_quat(noise((E * NB, 3), 0.1)).view(E, NB, 4), rn["rb_rot"])
        st["bv"] = rn["body_vel"] + noise((E, NB, 3), 0.2)
        st["bw"] = rn["body_ang_vel"] + noise((E, NB, 3), 0.3)
        st["dp"] = rn["dof_pos"] + noise((E, ND), 0.05)
        st["dv"] = rn["dof_vel"] + noise((E, ND), 0.3)
        dof_force = noise((E, ND), 30.0)

        # ---- post_physics_step ----
        progress = progress + 1
        mt = progress * dt + start
        r0 = lib.get_motion_state(motion_ids, mt, offset=torch.zeros(E, 3))
        rew, rew_raw = him.compute_imitation_reward(st["bp"][:, 0], st["br"][:, 0], st["bp"], st["br"], st["bv"], st["bw"], r0["rg_pos"], r0["rb_rot"],
                                                    r0["body_vel"], r0["body_ang_vel"], specs)
        power_reward = -0.0005 * torch.abs(dof_force * st["dv"]).sum(dim=-1)
        power_reward[progress <= 3] = 0
        rew = rew + power_reward
        rew_raw = torch.cat([rew_raw, power_reward[:, None]], dim=-1)
        pass_time = mt >= lib._motion_lengths[motion_ids]
        reset_buf, term = him.compute_humanoid_im_reset(reset_buf, progress, torch.zeros(E, NB, 3), torch.zeros(4, dtype=torch.long), st["bp"][:, rid],
                                                        r0["rg_pos"][:, rid], pass_time, True, term_dist[:, rid], False, False)
        obs = observations(torch.arange(E), st, progress, start)
        amp = torch.cat([torch.zeros(E, 1, 196), amp[:, :S - 1]], dim=1)           # _update_hist_amp_obs
        amp[:, 0] = amp_from(st["bp"][:, 0], st["br"][:, 0], st["bv"][:, 0], st["bw"][:, 0], st["dp"], st["dv"], st["bp"][:, kid])
        log["state_in"].append(torch.cat([st["bp"], st["br"], st["bv"], st["bw"]], -1).clone()); log["dof_in"].append(torch.stack([st["dp"], st["dv"]], -1).clone())
        log["dof_force"].append(dof_force); log["progress"].append(progress.clone()); log["rew"].append(rew); log["rew_raw"].append(rew_raw)
        log["reset"].append(reset_buf.clone()); log["terminate"].append(term.clone()); log["obs"].append(obs.clone()); log["amp"].append(amp.clone())
    out = {k: t2n(torch.stack(v)) for k, v in log.items()}
    out["motion_ids"] = t2n(motion_ids)
    np.savez_compressed(os.path.join(OUT, "rollout_ref_env.npz"), **out)
    r = out["reset"]
    print("rollout golden:", {k: v.shape for k, v in out.items() if k in ("obs", "amp", "reset_ids")}, "resets per step", r.sum(-1).tolist(),
          "terminations", int(out["terminate"].sum()), os.path.getsize(os.path.join(OUT, "rollout_ref_env.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
