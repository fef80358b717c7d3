"""TEST INFRASTRUCTURE ONLY -- time the *reference's own* CPU reward / FK / observation path (SURVEY.md 8d "CPU baseline
beside it"): the unmodified reference functions imported through ``oracle/ref_shim.py``, on the host cores of the machine
this runs on.  /root/reference does not exist on the GPU box; since round 5 a git-ignored travel archive of exactly the imported reference files
(`oracle/_ref/reference_modules.zip`, packed by oracle/make_ref.py in the build container; zipimport) rides along with the gpurun snapshot, so bench.py's `cpu_reference` / `config0`
legs run this script LIVE on the GPU box's host cores (`--stdout`); without either copy bench.py falls back to the newest committed file:

    python oracle/time_reference.py [rNN]      # -> profiles/rNN_reference_cpu_stages.json (default r05)
    python oracle/time_reference.py --stdout   # one line `REFERENCE_JSON{...}` on stdout (bench.py)

Stages (5 warm-up + 50 timed iterations, median), at N = 64 and N = 4096 on the synthetic AMASS-shaped clips:
  (1) MotionLibSMPL.load_motions (poselib FK + finite differences; start-up cost, timed once for 64 clips)
  (2) 2 x get_motion_state                                   (motion_lib_base.py:437-520)
  (3) compute_imitation_reward + compute_humanoid_im_reset   (humanoid_im.py:1524-1554, 1581-1608)
  (4) compute_humanoid_observations_smpl_max + compute_imitation_observations_v6 + build_amp_observations_smpl
  (c0) BASELINE.json configs[0] as written -- "poselib FK + imitation-reward on 64 envs, CPU PyTorch, single AMASS clip": one
       `SkeletonState.from_rotation_and_root_translation(...).global_translation / .global_rotation` (poselib FK, skeleton3d.py:390-426) over
       64 poses of ONE clip + one `compute_imitation_reward` on the 64 resulting body states (round 4)
There is no reference CPU number for the dynamics (Isaac Gym is a closed GPU binary).
"""
import json
import os
import sys
import tempfile
import time

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

MJCF = ref_shim.data_path("phc/data/assets/mjcf/smpl_0_humanoid.xml")
KEY_BODIES = ["R_Ankle", "L_Ankle", "R_Wrist", "L_Wrist"]


def median_ms(fn, warm=5, it=50):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(it):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts) * 1e3)


def main():
    threads = len(os.sched_getaffinity(0))
    try:   # container CPU quota (cgroup v2), which sched_getaffinity does not show: 256 torch threads on a 16-core quota would time the scheduler
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            threads = max(1, min(threads, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    threads = int(os.environ.get("PHC_REFERENCE_THREADS", threads))
    torch.set_num_threads(threads)
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
    parents = tree.parent_indices.numpy().astype(np.int32)
    clips = make_motion_dict(parents, 64, seed=0, body_names=names)
    for v in clips.values():
        v["root_trans_offset"] = torch.from_numpy(np.asarray(v["root_trans_offset"], np.float64))
    tmp = tempfile.mkdtemp()
    pkl = os.path.join(tmp, "clips.pkl")
    joblib.dump(clips, pkl)
    os.chdir(tmp)
    cfg = EasyDict({"motion_file": pkl, "device": torch.device("cpu"), "fix_height": FixHeightMode.full_fix, "min_length": -1,
                    "max_length": -1, "im_eval": False, "multi_thread": False, "smpl_type": "smpl", "randomrize_heading": True,
                    "step_dt": 1 / 30})
    flags.test, flags.im_eval = False, False
    lib = MotionLibSMPL(cfg)
    t0 = time.perf_counter()
    lib.load_motions(skeleton_trees=[tree] * 64, gender_betas=torch.zeros(64, 17), limb_weights=np.zeros((64, 10)), random_sample=False,
                     start_idx=0, max_len=-1)
    load_s = time.perf_counter() - t0
    frames = int(lib.gts.shape[0])
    torch.set_num_threads(threads)  # the reference's loader pins torch to one thread (motion_lib_smpl.py); undo for the stage timings
    out = {"host": {"threads": threads, "cpu_count": os.cpu_count(), "torch_threads": torch.get_num_threads(),
                    "model": next((l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")},
           "load_motions": {"clips": 64, "frames": frames, "seconds": load_s, "ms_per_clip": load_s / 64 * 1e3}, "stages": {}}
    specs = {"k_pos": 100., "k_rot": 10., "k_vel": 0.1, "k_ang_vel": 0.1, "w_pos": 0.5, "w_rot": 0.3, "w_vel": 0.1, "w_ang_vel": 0.1}
    dof_names = names[1:]
    remove = ["L_Hand", "R_Hand", "L_Toe", "R_Toe"]
    dof_subset = torch.from_numpy(np.concatenate([np.arange(i * 3, i * 3 + 3) for i, nm in enumerate(dof_names) if nm not in remove]))
    kid = torch.tensor([names.index(b) for b in KEY_BODIES])
    for N in (64, 4096):
        g = torch.Generator().manual_seed(0)
        ids = torch.arange(N) % 64
        progress = torch.randint(0, 40, (N,), generator=g)
        st = lib.sample_time_interval(ids)
        mt = progress / 30 + st
        goff = torch.zeros(N, 3)
        r0 = lib.get_motion_state(ids, mt, offset=goff)
        r1 = lib.get_motion_state(ids, mt + 1 / 30, offset=goff)
        bp = r0["rg_pos"] + torch.randn(N, 24, 3, generator=g) * 0.03
        br, bv, bav = r0["rb_rot"], r0["body_vel"], r0["body_ang_vel"]
        dp, dv = r0["dof_pos"], r0["dof_vel"]
        td = torch.full((N, 24), 0.25)
        rb = torch.zeros(N, dtype=torch.long)
        z11, z10 = torch.zeros(N, 11), torch.zeros(N, 10)

        def s2():
            lib.get_motion_state(ids, mt, offset=goff)
            lib.get_motion_state(ids, mt + 1 / 30, offset=goff)

        def s3():
            him.compute_imitation_reward(bp[:, 0], br[:, 0], bp, br, bv, bav, r0["rg_pos"], r0["rb_rot"], r0["body_vel"], r0["body_ang_vel"], specs)
            him.compute_humanoid_im_reset(rb, progress, torch.zeros(N, 24, 3), torch.zeros(4, dtype=torch.long), bp, r0["rg_pos"],
                                          mt >= lib._motion_lengths[ids], True, td, False, False)

        def s4():
            hum.compute_humanoid_observations_smpl_max(bp, br, bv, bav, z11, z10, True, True, True, False, False)
            him.compute_imitation_observations_v6(bp[:, 0], br[:, 0], bp, br, bv, bav, r1["rg_pos"], r1["rb_rot"], r1["body_vel"], r1["body_ang_vel"], 1, True)
            hamp.build_amp_observations_smpl(bp[:, 0], br[:, 0], bv[:, 0], bav[:, 0], dp, dv, bp[:, kid], z11, z10, dof_subset, True, True, True, False, False, True)

        a, b, c = median_ms(s2), median_ms(s3), median_ms(s4)
        out["stages"][str(N)] = {"get_motion_state_x2_ms": a, "reward_reset_ms": b, "observations_ms": c, "sum_ms": a + b + c,
                                 "env_steps_per_s_reward_obs_only": N / ((a + b + c) * 1e-3)}
    # ---- BASELINE configs[0]: poselib FK + imitation reward, 64 envs, single clip ----
    from poselib.poselib.skeleton.skeleton3d import SkeletonState
    clip0 = list(clips.values())[0]
    T0 = len(clip0["root_trans_offset"])
    fr = torch.arange(64) % T0
    pq = torch.from_numpy(np.asarray(clip0["pose_quat_global"], np.float32))[fr]          # [64, 24, 4] global rotations of 64 frames
    rt = clip0["root_trans_offset"][fr].float()
    state0 = SkeletonState.from_rotation_and_root_translation(tree, pq, rt, is_local=False)
    local_rot = state0.local_rotation.clone()
    ids0 = torch.zeros(64, dtype=torch.long)
    t64 = fr.float() / 30
    ref64 = lib.get_motion_state(ids0, t64, offset=torch.zeros(64, 3))

    def c0_fk():
        s_ = SkeletonState.from_rotation_and_root_translation(tree, local_rot, rt, is_local=True)
        return s_.global_translation, s_.global_rotation

    def c0_reward():
        gp, gr = c0_fk()
        him.compute_imitation_reward(gp[:, 0], gr[:, 0], gp, gr, ref64["body_vel"], ref64["body_ang_vel"], ref64["rg_pos"], ref64["rb_rot"],
                                     ref64["body_vel"], ref64["body_ang_vel"], specs)
    fk_ms, both_ms = median_ms(c0_fk), median_ms(c0_reward)
    out["config0"] = {"what": "BASELINE configs[0]: poselib FK (SkeletonState.from_rotation_and_root_translation -> global_translation / global_rotation) "
                              "of 64 poses of one clip + compute_imitation_reward on them, CPU PyTorch fp32", "envs": 64, "poselib_fk_ms": fk_ms,
                      "fk_plus_reward_ms": both_ms, "env_evaluations_per_s": 64 / (both_ms * 1e-3)}
    out["date"] = time.strftime("%Y-%m-%d")
    out["protocol"] = "BASELINE.md section 2: 5 warm-up + 50 timed iterations, median; torch.set_num_threads(all usable cores); fp32"
    import socket
    out["measured_in"] = (f"this host ({socket.gethostname()}): the reference's modules from " +
                          ("the travel archive oracle/_ref/reference_modules.zip (oracle/make_ref.py)" if ref_shim.is_archive() else ref_shim.REFERENCE_ROOT + " (build container)"))
    out["has_gpu"] = bool(torch.cuda.is_available())
    if "--stdout" in sys.argv:      # bench.py's live `cpu_reference` leg: one JSON line, nothing written
        print("REFERENCE_JSON" + json.dumps(out))
        return
    tag = next((a for a in sys.argv[1:] if not a.startswith("--")), "r05")
    dst = os.path.join(ROOT, "profiles", f"{tag}_reference_cpu_stages.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
