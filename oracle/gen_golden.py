"""TEST INFRASTRUCTURE ONLY -- generate golden vectors from the *reference's own code*.

Run in the build container (needs /root/reference):

    python oracle/gen_golden.py

Imports the unmodified reference through ``oracle/ref_shim.py`` and writes small
``.npz`` fixtures into ``tests/golden/``.  The reference ships no golden vectors
of its own, so these are the pins for
``oracle/phc_oracle.py`` (checked on CPU) and, through it, for the HIP kernels
(checked on the GPU box, where the reference does not exist).
"""
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

from phc_amd.utils.synthetic_motion import make_motion_dict  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
MJCF = os.path.join(ref_shim.REFERENCE_ROOT, "phc/data/assets/mjcf/smpl_0_humanoid.xml")

KEY_BODIES = ["R_Ankle", "L_Ankle", "R_Wrist", "L_Wrist"]
RESET_BODIES = ['Pelvis', 'L_Hip', 'L_Knee', 'R_Hip', 'R_Knee', 'Torso', 'Spine', 'Chest', 'Neck', 'Head', 'L_Thorax',
                'L_Shoulder', 'L_Elbow', 'L_Wrist', 'L_Hand', 'R_Thorax', 'R_Shoulder', 'R_Elbow', 'R_Wrist', 'R_Hand']


def t2n(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def main():
    os.makedirs(OUT, exist_ok=True)
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

    # ---------------- skeleton (model-compiler pin) ----------------
    tree = SkeletonTree.from_mjcf(MJCF)
    names = list(tree.node_names)
    parents = t2n(tree.parent_indices).astype(np.int32)
    np.savez(os.path.join(OUT, "skeleton_smpl.npz"), node_names=np.array(names), parent_indices=parents,
             local_translation=t2n(tree.local_translation))

    # ---------------- quaternion KATs (R12) ----------------
    g = torch.Generator().manual_seed(1234)
    n = 257
    qa = itu.normalize(torch.randn(n, 4, generator=g))
    qb = itu.normalize(torch.randn(n, 4, generator=g))
    # edge cases: identity, antipodal, nearly-equal, w=+-1
    qa[0] = torch.tensor([0, 0, 0, 1.0]); qb[0] = torch.tensor([0, 0, 0, 1.0])
    qa[1] = torch.tensor([0, 0, 0, -1.0]); qb[1] = -qb[1]
    qb[2] = qa[2]
    qb[3] = itu.normalize(qa[3] + 1e-4 * torch.randn(4, generator=g))
    qb[4] = -qa[4]
    qa[5] = torch.tensor([1.0, 0, 0, 0]); qa[6] = torch.tensor([0, 1.0, 0, 0]); qa[7] = torch.tensor([0, 0, 1.0, 0])
    v = torch.randn(n, 3, generator=g)
    tt = torch.rand(n, 1, generator=g)
    tt[8] = 0.0; tt[9] = 1.0
    em = torch.randn(n, 3, generator=g) * 1.5
    em[10] = 0.0; em[11] = torch.tensor([1e-7, 0, 0]); em[12] = torch.tensor([0, 0, 3.5])
    ang, ax = itu.quat_to_angle_axis(qa)
    np.savez(os.path.join(OUT, "quat_kat.npz"), qa=t2n(qa), qb=t2n(qb), v=t2n(v), t=t2n(tt), em=t2n(em),
             quat_mul=t2n(itu.quat_mul(qa, qb)), quat_conjugate=t2n(itu.quat_conjugate(qa)),
             my_quat_rotate=t2n(itu.my_quat_rotate(qa, v)), angle=t2n(ang), axis=t2n(ax),
             quat_to_exp_map=t2n(itu.quat_to_exp_map(qa)), quat_to_tan_norm=t2n(itu.quat_to_tan_norm(qa)),
             exp_map_to_quat=t2n(itu.exp_map_to_quat(em)), slerp=t2n(itu.slerp(qa, qb, tt)),
             calc_heading=t2n(itu.calc_heading(qa)), calc_heading_quat=t2n(itu.calc_heading_quat(qa)),
             calc_heading_quat_inv=t2n(itu.calc_heading_quat_inv(qa)))

    # ---------------- synthetic AMASS-shaped clips + reference MotionLibSMPL (M2-M9) ----------------
    lengths = [31, 45, 38]
    clips = make_motion_dict(parents, 3, seed=7, body_names=names, lengths=lengths)
    for k in clips:
        clips[k]["root_trans_offset"] = torch.from_numpy(clips[k]["root_trans_offset"])  # must be f64 torch (motion_lib_smpl.py:130,145)
    tmp = tempfile.mkdtemp()
    pkl = os.path.join(tmp, "synthetic.pkl")
    joblib.dump(clips, pkl)
    np.savez_compressed(os.path.join(OUT, "motion_clips.npz"),
                        **{f"{k}/pose_quat_global": v["pose_quat_global"] for k, v in clips.items()},
                        **{f"{k}/root_trans_offset": t2n(v["root_trans_offset"]) for k, v in clips.items()},
                        **{f"{k}/pose_aa": v["pose_aa"] for k, v in clips.items()},
                        keys=np.array(list(clips.keys())), fps=np.array([30] * 3))

    N = 6
    cwd = os.getcwd()
    os.chdir(tmp)  # "data/smpl" must not exist relative to cwd -> mesh_parsers None
    cfg = EasyDict({"motion_file": pkl, "device": torch.device("cpu"), "fix_height": FixHeightMode.full_fix,
                    "min_length": -1, "max_length": -1, "im_eval": False, "multi_thread": False,
                    "smpl_type": "smpl", "randomrize_heading": True, "step_dt": 1 / 30})

    def dump_lib(lib, tag, extra=None):
        d = {k: t2n(getattr(lib, k)) for k in ("gts", "grs", "lrs", "gvs", "gavs", "grvs", "gravs", "dvs")}
        d.update(motion_lengths=t2n(lib._motion_lengths), motion_fps=t2n(lib._motion_fps), motion_dt=t2n(lib._motion_dt),
                 motion_num_frames=t2n(lib._motion_num_frames), length_starts=t2n(lib.length_starts),
                 curr_motion_ids=t2n(lib._curr_motion_ids), motion_aa=t2n(lib._motion_aa))
        if extra:
            d.update(extra)
        np.savez_compressed(os.path.join(OUT, f"motion_lib_{tag}.npz"), **d)

    def load(lib_flags_test):
        flags.test = lib_flags_test
        flags.im_eval = False
        lib = MotionLibSMPL(cfg)
        lib.load_motions(skeleton_trees=[tree] * N, gender_betas=torch.zeros(N, 17), limb_weights=np.zeros((N, 10)),
                         random_sample=False, start_idx=0, max_len=-1)
        return lib

    lib = load(True)  # no random heading (flags.test)
    # get_motion_state at seeded ids/times incl. edge cases (t<0, t>len, exact frame times)
    gq = torch.Generator().manual_seed(99)
    M = 64
    ids = torch.randint(0, N, (M,), generator=gq)
    times = torch.rand(M, generator=gq) * lib._motion_lengths[ids] * 1.2 - 0.05
    times[0] = 0.0
    times[1] = lib._motion_lengths[ids[1]]
    times[2] = lib._motion_lengths[ids[2]] + 1.0
    times[3] = -0.3
    times[4] = 7 * (1 / 30)
    times[5] = 1 / 30
    offs = torch.randn(M, 3, generator=gq) * 0.3
    offs[:, 2] = 0
    res = lib.get_motion_state(ids, times, offset=offs)
    f0, f1, bl = lib._calc_frame_blend(times, lib._motion_lengths[ids], lib._motion_num_frames[ids], lib._motion_dt[ids])
    ms = {f"ms_{k}": t2n(v) for k, v in res.items()}
    ms.update(ms_ids=t2n(ids), ms_times=t2n(times), ms_offset=t2n(offs), ms_idx0=t2n(f0), ms_idx1=t2n(f1), ms_blend=t2n(bl))
    # sample_time_interval: the rand draw is reproduced by reseeding (motion_lib_base.py:414-423)
    torch.manual_seed(5)
    st = lib.sample_time_interval(ids)
    torch.manual_seed(5)
    phase = torch.rand(ids.shape)
    ms.update(sti_phase=t2n(phase), sti_time=t2n(st))
    ms.update(num_steps=t2n(lib.get_motion_num_steps()))
    dump_lib(lib, "eval", ms)

    lib_h = load(False)  # random heading: np.random.seed(randint*pid) with pid 0 -> seed 0 (motion_lib_smpl.py:106,138-146)
    dump_lib(lib_h, "heading")
    os.chdir(cwd)

    # ---------------- reward / reset / observations (R1-R9) on body state = ref + noise ----------------
    gr = torch.Generator().manual_seed(2024)
    E = 48
    env_motion = torch.arange(E) % N
    progress = torch.randint(0, 40, (E,), generator=gr)
    progress[:4] = torch.tensor([0, 1, 2, 3])
    dt = 1 / 30
    start_times = lib.sample_time_interval(env_motion)
    mt = progress * dt + start_times + torch.zeros(E)
    goff = torch.zeros(E, 3)
    r0 = lib.get_motion_state(env_motion, mt, offset=goff)
    r1 = lib.get_motion_state(env_motion, (progress + 1) * dt + start_times + torch.zeros(E), offset=goff)
    noise = lambda shape, s: torch.randn(*shape, generator=gr) * s
    body_pos = r0["rg_pos"] + noise((E, 24, 3), 0.03)
    body_pos[5:9] += noise((4, 24, 3), 0.25)  # some envs far enough to terminate
    body_rot = itu.quat_mul(itu.exp_map_to_quat(noise((E * 24, 3), 0.15)).view(E, 24, 4), r0["rb_rot"])
    body_vel = r0["body_vel"] + noise((E, 24, 3), 0.3)
    body_ang_vel = r0["body_ang_vel"] + noise((E, 24, 3), 0.5)
    dof_pos = r0["dof_pos"] + noise((E, 69), 0.1)
    dof_vel = r0["dof_vel"] + noise((E, 69), 0.5)
    dof_force = noise((E, 69), 40.0)

    specs = {"k_pos": 100., "k_rot": 10., "k_vel": 0.1, "k_ang_vel": 0.1, "w_pos": 0.5, "w_rot": 0.3, "w_vel": 0.1, "w_ang_vel": 0.1}
    rew, rew_raw = him.compute_imitation_reward(body_pos[:, 0], body_rot[:, 0], body_pos, body_rot, body_vel, body_ang_vel,
                                                r0["rg_pos"], r0["rb_rot"], r0["body_vel"], r0["body_ang_vel"], specs)
    power = torch.abs(torch.multiply(dof_force, dof_vel)).sum(dim=-1)  # humanoid_im.py:939-946
    power_reward = -0.0005 * power
    power_reward[progress <= 3] = 0
    rid = torch.tensor([names.index(b) for b in RESET_BODIES])
    pass_time = mt >= lib._motion_lengths[env_motion]
    term_dist = torch.full((E, 24), 0.25)
    reset_buf = torch.zeros(E, dtype=torch.long)
    reset, term = him.compute_humanoid_im_reset(reset_buf, progress, torch.zeros(E, 24, 3), torch.zeros(4, dtype=torch.long),
                                                body_pos[:, rid], r0["rg_pos"][:, rid], pass_time, True, term_dist[:, rid], False, False)
    reset_m, term_m = him.compute_humanoid_im_reset(reset_buf, progress, torch.zeros(E, 24, 3), torch.zeros(4, dtype=torch.long),
                                                    body_pos[:, rid], r0["rg_pos"][:, rid], pass_time, True, term_dist[:, rid], False, True)
    self_obs = hum.compute_humanoid_observations_smpl_max(body_pos, body_rot, body_vel, body_ang_vel, torch.zeros(E, 11), torch.zeros(E, 10),
                                                          True, True, True, False, False)
    task_obs = him.compute_imitation_observations_v6(body_pos[:, 0], body_rot[:, 0], body_pos, body_rot, body_vel, body_ang_vel,
                                                     r1["rg_pos"], r1["rb_rot"], r1["body_vel"], r1["body_ang_vel"], 1, True)
    dof_names = names[1:]
    remove = ["L_Hand", "R_Hand", "L_Toe", "R_Toe"]  # humanoid.py:388-413
    dof_subset = torch.from_numpy(np.concatenate([np.arange(i * 3, i * 3 + 3) for i, nm in enumerate(dof_names) if nm not in remove]))
    kid = torch.tensor([names.index(b) for b in KEY_BODIES])
    amp = hamp.build_amp_observations_smpl(body_pos[:, 0], body_rot[:, 0], body_vel[:, 0], body_ang_vel[:, 0], dof_pos, dof_vel,
                                           body_pos[:, kid], torch.zeros(E, 11), torch.zeros(E, 10), dof_subset,
                                           True, True, True, False, False, True)
    np.savez_compressed(os.path.join(OUT, "task_fns.npz"), env_motion=t2n(env_motion), progress=t2n(progress), start_times=t2n(start_times),
                        motion_times=t2n(mt), body_pos=t2n(body_pos), body_rot=t2n(body_rot), body_vel=t2n(body_vel), body_ang_vel=t2n(body_ang_vel),
                        dof_pos=t2n(dof_pos), dof_vel=t2n(dof_vel), dof_force=t2n(dof_force),
                        ref_pos=t2n(r0["rg_pos"]), ref_rot=t2n(r0["rb_rot"]), ref_vel=t2n(r0["body_vel"]), ref_ang_vel=t2n(r0["body_ang_vel"]),
                        ref1_pos=t2n(r1["rg_pos"]), ref1_rot=t2n(r1["rb_rot"]), ref1_vel=t2n(r1["body_vel"]), ref1_ang_vel=t2n(r1["body_ang_vel"]),
                        ref_dof_pos=t2n(r0["dof_pos"]), ref_dof_vel=t2n(r0["dof_vel"]),
                        reward=t2n(rew), reward_raw=t2n(rew_raw), power_reward=t2n(power_reward), pass_time=t2n(pass_time),
                        reset=t2n(reset), terminate=t2n(term), reset_mean=t2n(reset_m), terminate_mean=t2n(term_m),
                        reset_body_ids=t2n(rid), key_body_ids=t2n(kid), dof_subset=t2n(dof_subset),
                        self_obs=t2n(self_obs), task_obs=t2n(task_obs), amp_obs=t2n(amp))
    print("golden vectors written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
