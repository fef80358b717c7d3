"""TEST INFRASTRUCTURE ONLY -- golden vectors for the robot path (H1: config 5; G1: `python oracle/gen_golden_h1.py g1`, 38 bodies ->
the 64-lane kernels) from the reference's own code:

  * `Humanoid_Batch.fk_batch` + `MotionLibReal.load_motions / get_motion_state`  (phc/utils/torch_humanoid_batch.py,
    phc/utils/motion_lib_real.py) on synthetic robot clips -> tests/golden/motion_lib_h1.npz
  * reward with the extended bodies (humanoid_im.py:916-923), self / task observations on 20 bodies, the robot AMP
    observation (humanoid_amp.py:1063-1104), reset -> tests/golden/task_fns_h1.npz
  * the skeleton constants Humanoid_Batch reads from h1.xml (model-compiler pin) -> tests/golden/skeleton_h1.npz

Run in the build container:  python oracle/gen_golden_h1.py [h1|g1]
(lxml / open3d / stl are absent here: ref_shim provides an ElementTree-based lxml stand-in and mocks the mesh loaders;
 the mesh-based start-height fix is therefore switched off -- FixHeightMode.no_fix.)
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
import yaml  # noqa: E402

from gen_golden import OUT, t2n  # noqa: E402
from phc_amd.model import load_model  # noqa: E402
from phc_amd.utils.synthetic_motion import make_robot_motion_dict  # noqa: E402

KEY_BODIES = {"h1": ["left_ankle_link", "right_ankle_link", "left_elbow_link", "right_elbow_link"],                # env_im_h1_phc.yaml
              "g1": ["left_ankle_roll_link", "right_ankle_roll_link", "left_zero_link", "right_zero_link"]}      # env_im_g1_phc.yaml


def main(rb="h1"):
    torch.set_num_threads(1)
    from easydict import EasyDict
    cwd = os.getcwd()
    os.chdir(ref_shim.REFERENCE_ROOT)   # robot.asset.assetFileName is relative to the reference root
    robot = EasyDict(yaml.safe_load(open(f"phc/data/cfg/robot/unitree_{rb}.yaml")))
    him = ref_shim.ref_module("phc.env.tasks.humanoid_im")
    hum = ref_shim.ref_module("phc.env.tasks.humanoid")
    hamp = ref_shim.ref_module("phc.env.tasks.humanoid_amp")
    from phc.utils.flags import flags
    from phc.utils.motion_lib_base import FixHeightMode
    from phc.utils.motion_lib_real import MotionLibReal
    from phc.utils.torch_humanoid_batch import Humanoid_Batch
    from poselib.poselib.skeleton.skeleton3d import SkeletonTree

    hb = Humanoid_Batch(robot)
    NB = len(hb.body_names)
    ND = NB - 1
    np.savez_compressed(os.path.join(OUT, f"skeleton_{rb}.npz"), node_names=np.array(hb.body_names), parents=t2n(hb._parents[:NB]).astype(np.int32),
                        local_translation=t2n(hb._offsets[0, :NB]), local_rotation=t2n(hb._local_rotation[0, :NB]), dof_axis=t2n(hb.dof_axis),
                        joints_range=t2n(hb.joints_range), ext_parents=t2n(hb._parents[NB:]).astype(np.int32),
                        ext_offsets=t2n(hb._offsets[0, NB:]), body_names_augment=np.array(hb.body_names_augment))

    model = load_model(f"{rb}_humanoid")
    clips = make_robot_motion_dict(model, 3, seed=9, lengths=[33, 47, 40])
    tmp = tempfile.mkdtemp()
    pkl = os.path.join(tmp, f"{rb}_clips.pkl")
    joblib.dump({k: dict(v, root_trans_offset=torch.from_numpy(v["root_trans_offset"]), pose_aa=v["pose_aa"]) for k, v in clips.items()}, pkl)
    np.savez_compressed(os.path.join(OUT, f"motion_clips_{rb}.npz"), keys=np.array(list(clips.keys())),
                        **{f"{k}/pose_aa": v["pose_aa"] for k, v in clips.items()},
                        **{f"{k}/root_trans_offset": v["root_trans_offset"] for k, v in clips.items()})
    tree = SkeletonTree.from_mjcf(robot.asset.assetFileName)
    N = 6
    cfg = EasyDict({"motion_file": pkl, "device": torch.device("cpu"), "fix_height": FixHeightMode.no_fix, "min_length": -1, "max_length": -1,
                    "im_eval": False, "multi_thread": False, "smpl_type": rb, "randomrize_heading": True, "robot": robot, "step_dt": 1 / 50})
    flags.test, flags.im_eval, flags.real_traj = False, False, False
    lib = MotionLibReal(cfg)
    lib.load_motions(skeleton_trees=[tree] * N, gender_betas=torch.zeros(N, 17), limb_weights=np.zeros((N, 10)), random_sample=False, start_idx=0, max_len=-1)
    d = {k: t2n(getattr(lib, k)) for k in ("gts", "grs", "gvs", "gavs", "dvs", "dof_pos", "gts_t", "grs_t")}
    d.update(motion_lengths=t2n(lib._motion_lengths), motion_fps=t2n(lib._motion_fps), motion_dt=t2n(lib._motion_dt),
             motion_num_frames=t2n(lib._motion_num_frames), length_starts=t2n(lib.length_starts), num_steps=t2n(lib.get_motion_num_steps()))
    gq = torch.Generator().manual_seed(123)
    M = 48
    ids = torch.randint(0, N, (M,), generator=gq)
    times = torch.rand(M, generator=gq) * lib._motion_lengths[ids] * 1.2 - 0.05
    times[0], times[1], times[2] = 0.0, lib._motion_lengths[ids[1]], -0.2
    offs = torch.randn(M, 3, generator=gq) * 0.3
    offs[:, 2] = 0
    res = lib.get_motion_state(ids, times, offset=offs)
    d.update({f"ms_{k}": t2n(v) for k, v in res.items()})
    d.update(ms_ids=t2n(ids), ms_times=t2n(times), ms_offset=t2n(offs))
    np.savez_compressed(os.path.join(OUT, f"motion_lib_{rb}.npz"), **d)

    # ---------------- task functions on H1 shapes ----------------
    gr = torch.Generator().manual_seed(555)
    E, dt = 40, 4 * (1 / 200)
    env_motion = torch.arange(E) % N
    progress = torch.randint(0, 40, (E,), generator=gr)
    progress[:4] = torch.tensor([0, 1, 2, 3])
    torch.manual_seed(3)
    start_times = lib.sample_time_interval(env_motion)
    mt = progress * dt + start_times
    goff = torch.zeros(E, 3)
    r0 = lib.get_motion_state(env_motion, mt, offset=goff)
    r1 = lib.get_motion_state(env_motion, (progress + 1) * dt + start_times, offset=goff)
    itu = ref_shim.ref_module("phc.utils.isaacgym_torch_utils")
    noise = lambda shape, s: torch.randn(*shape, generator=gr) * s
    body_pos = r0["rg_pos"] + noise((E, NB, 3), 0.03)
    body_pos[5:9] += noise((4, NB, 3), 0.25)
    body_rot = itu.quat_mul(itu.exp_map_to_quat(noise((E * NB, 3), 0.15)).view(E, NB, 4), r0["rb_rot"])
    body_vel = r0["body_vel"] + noise((E, NB, 3), 0.3)
    body_ang_vel = r0["body_ang_vel"] + noise((E, NB, 3), 0.5)
    dof_pos = r0["dof_pos"] + noise((E, ND), 0.1)
    dof_vel = r0["dof_vel"] + noise((E, ND), 0.5)
    dof_force = noise((E, ND), 40.0)
    names = list(hb.body_names)
    ext_parent = torch.tensor([names.index(e["parent_name"]) for e in robot.extend_config])
    ext_pos = torch.tensor([e["pos"] for e in robot.extend_config]).float().repeat(E, 1, 1)
    # humanoid_im.py:916-923
    extend_curr_pos = itu.my_quat_rotate(body_rot[:, ext_parent].reshape(-1, 4), ext_pos.reshape(-1, 3)).view(E, -1, 3) + body_pos[:, ext_parent]
    body_pos_extend = torch.cat([body_pos, extend_curr_pos], dim=1)
    body_rot_extend = torch.cat([body_rot, body_rot[:, ext_parent]], dim=1)
    ref_pos_extend = torch.cat([r0["rg_pos"], r0["rg_pos_t"][:, NB:]], dim=1)
    ref_rot_extend = torch.cat([r0["rb_rot"], r0["rg_rot_t"][:, NB:]], dim=1)
    specs = {"k_pos": 100., "k_rot": 10., "k_vel": 0.1, "k_ang_vel": 0.1, "w_pos": 0.5, "w_rot": 0.3, "w_vel": 0.1, "w_ang_vel": 0.1}
    rew, rew_raw = him.compute_imitation_reward(body_pos[:, 0], body_rot[:, 0], body_pos_extend, body_rot_extend, body_vel, body_ang_vel,
                                                ref_pos_extend, ref_rot_extend, r0["body_vel"], r0["body_ang_vel"], specs)
    power_reward = -0.0005 * torch.abs(dof_force * dof_vel).sum(dim=-1)
    power_reward[progress <= 3] = 0
    pass_time = mt >= lib._motion_lengths[env_motion]
    td = torch.full((E, NB), 0.25)
    reset, term = him.compute_humanoid_im_reset(torch.zeros(E, dtype=torch.long), progress, torch.zeros(E, NB, 3), torch.zeros(2, dtype=torch.long),
                                                body_pos, r0["rg_pos"], pass_time, True, td, False, False)
    self_obs = hum.compute_humanoid_observations_smpl_max(body_pos, body_rot, body_vel, body_ang_vel, torch.zeros(E, 17), torch.zeros(E, 10),
                                                          True, True, True, False, False)
    task_obs = him.compute_imitation_observations_v6(body_pos[:, 0], body_rot[:, 0], body_pos, body_rot, body_vel, body_ang_vel,
                                                     r1["rg_pos"], r1["rb_rot"], r1["body_vel"], r1["body_ang_vel"], 1, True)
    kid = torch.tensor([names.index(b) for b in KEY_BODIES[rb]])
    amp = hamp.build_amp_observations_robot(body_pos[:, 0], body_rot[:, 0], body_vel[:, 0], body_ang_vel[:, 0], dof_pos, dof_vel, body_pos[:, kid],
                                            torch.zeros(E, 17), torch.zeros(E, 10), torch.zeros(0, dtype=torch.long), True, True, True, False, False, True)
    np.savez_compressed(os.path.join(OUT, f"task_fns_{rb}.npz"), env_motion=t2n(env_motion), progress=t2n(progress), start_times=t2n(start_times),
                        body_pos=t2n(body_pos), body_rot=t2n(body_rot), body_vel=t2n(body_vel), body_ang_vel=t2n(body_ang_vel), dof_pos=t2n(dof_pos),
                        dof_vel=t2n(dof_vel), dof_force=t2n(dof_force), reward=t2n(rew), reward_raw=t2n(rew_raw), power_reward=t2n(power_reward),
                        reset=t2n(reset), terminate=t2n(term), self_obs=t2n(self_obs), task_obs=t2n(task_obs), amp_obs=t2n(amp), key_body_ids=t2n(kid),
                        ext_parent=t2n(ext_parent), ext_pos=t2n(ext_pos[0]), ref1_pos=t2n(r1["rg_pos"]), ref1_dof_pos=t2n(r1["dof_pos"]))
    os.chdir(cwd)
    print(rb, "goldens:", {f: os.path.getsize(os.path.join(OUT, f)) // 1024 for f in sorted(os.listdir(OUT)) if f"_{rb}." in f}, "KiB;",
          "self_obs", tuple(self_obs.shape), "task_obs", tuple(task_obs.shape), "amp", tuple(amp.shape), "terminated", int(term.sum()))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "h1")
