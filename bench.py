#!/usr/bin/env python
"""bench.py -- the path's headline metric (BASELINE.json): env-steps/sec at 4096 humanoid envs per GPU.

    python bench.py --gpus N --steps K --warmup W          (N > 1: one rank per GPU over RCCL -- either launched by
                                                            torch.distributed.run, or, when WORLD_SIZE is not set, bench.py
                                                            re-executes itself under torch.distributed.run with N ranks)

A "step" is one pass of the hot path over one batch of synthetic input: `VecEnv.step()` for 4096 SMPL-humanoid envs
(BASELINE.json configs[1]: 69 DoF, 4096 envs, single reference motion) = action -> PD targets -> 4 ABA sub-steps with
contact -> body-state publication -> reference-motion lookup -> imitation reward -> reset test -> observations -> AMP
observation, plus the reset of the envs that finished on the previous step (rollout idiom).
Inputs (motion buffer, state, the fixed random action tensor) are resident in HBM before the timed region.

Envs shard trivially: every rank owns 4096 envs and its own motion-library shard; the env step has
no data-path collective, so scaling is weak and `value` = sum over ranks of env-steps / max-over-ranks time.
The one collective of the path (gradient all-reduce per optimizer step) belongs to the PPO update and is timed by
`--ppo-epochs` (reported as `ppo_samples_per_s` in the same JSON line).

One JSON line is printed by rank 0, with the `roofline` object of the dominant kernel (k_sim_step, timed live with
HIP events on the launch stream) and the `cpu_baseline` object (CPU port of the same path on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # cpu_baseline leg: no spinning OpenMP workers next to the GPU driver threads

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (6.3 TB/s achievable)
# SURVEY.md 8(d) algorithmic bytes per env-step of the stepper launch (SMPL, J=24, D=69, fp32):
#   ABA sweep 1484 B / env-sub-step x 4 sub-steps (canonical un-fused accounting) + 1812 B state publication
ABA_BYTES_PER_ENV_SUBSTEP = 1484
PUBLISH_BYTES_PER_ENV_STEP = 1812
FLOPS_PER_ENV_SUBSTEP = 40e3   # SURVEY.md 8(d) "algorithmic flops (secondary)"
FP32_VECTOR_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: peak FP32 (vector)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU (BASELINE configs[1]: 4096)")
    ap.add_argument("--no-update-graph", action="store_true", help="PPO update with eager launches instead of the captured hipGraph "
                    "(learning.params.config.hip_graph; single-GPU runs only)")
    ap.add_argument("--multi-gpu-update-graph", action="store_true", help="(default since round 2; kept for compatibility) captured update "
                    "graph also with more than one rank: the graph holds no collective, the all-reduce follows each replay eagerly")
    ap.add_argument("--force-rccl", action="store_true", help="single-GPU self-test of the multi-GPU code path: a ONE-rank nccl (=RCCL) "
                    "process group, the gradient all-reduce issued after every optimizer step's captured forward / backward")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="process-group backend: nccl (= RCCL, the product path); gloo "
                    "exists to exercise the multi-rank logic of this script with several ranks SHARING one GPU (tests on 1-GPU boxes)")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 5], help="BASELINE.json configs preset (1-based): 2 = SMPL 4096 envs single "
                    "clip (the bench line), 3 = 8192 envs + AMASS-sized synthetic library (--motion-clips, default 11313), 5 = H1 4096 envs")
    ap.add_argument("--ppo-epochs", type=int, default=3, help="timed PPO epochs (rollout 32 steps + 48 optimizer steps: 6 mini-epochs x 8 minibatches); 0 = skip")
    ap.add_argument("--learning", default="im", help="learner config of the PPO part (phc/data/cfg/learning/*.yaml): im (the bench line: 1024-512 ReLU), "
                    "im_big / im_pnn_big (the reference's flagship 2048-1536-1024-1024-512-512 SiLU networks; im_pnn* also selects env=env_im_pnn)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the two extra env-step measurements (tracking actions, Unitree H1) "
                    "that the default single-GPU run appends as `other_workloads`")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live HBM-traffic measurement (two rocprofv3 --pmc passes of a 20-step "
                    "child run, FETCH_SIZE and WRITE_SIZE separately as MI355X_MICROARCH.md prescribes); `roofline.traffic` then falls "
                    "back to the newest committed profiles/*_pmc_traffic.json")
    ap.add_argument("--robot", choices=["smpl", "h1", "g1"], default="smpl", help="smpl: BASELINE configs[1] (the bench line); h1: configs[4] morphology "
                    "(Unitree H1, 19 revolute DoFs, 200 Hz x 4 pd-torque control); g1: Unitree G1 (38 bodies, the 64-lane kernels) -- parity-test "
                    "configurations, timed for reference only")
    ap.add_argument("--lane-mapping", type=int, default=0, help="stepper thread mapping (phc_sim_params_t.lane_mapping): 0 auto, 1 one body "
                    "per lane (32 lanes/env, or 64 above 32 bodies) -- the only mapping the stepper has")
    ap.add_argument("--burn-in", type=int, default=64, help="untimed env steps in FRONT of the W warm-up steps (reported as `protocol_burn_in_steps`): "
                    "right after env.reset() every env falls, and is reset, in the same few steps; 0 = the bare driver protocol (W warm-up steps only)")
    ap.add_argument("--self-collision", type=int, default=-1, help="-1: as the robot yaml says (has_self_collision: True); 0/1 force")
    ap.add_argument("--solver", action="append", default=[], metavar="KEY=VALUE", help="stepper switch passed on as `+solver.KEY=VALUE` (repeatable), "
                    "e.g. --solver inertia_lag=1 --solver force_average=1 --solver contact=tgs")
    ap.add_argument("--learner", action="append", default=[], metavar="KEY=VALUE", help="learner switch passed on as `+learning.params.config.KEY=VALUE` (repeatable), "
                    "e.g. --learner actor_precision=split_bf16")
    ap.add_argument("--motion-clips", type=int, default=1, help="synthetic clips in the motion library (configs[1]: 1; configs[2]/[3] shape: thousands)")
    ap.add_argument("--actions", choices=["random", "tracking"], default="random",
                    help="random: fixed a ~ U(-1,1)*0.1 tensor (SURVEY 8d protocol; zero-pose targets -> episodes end after a few steps); "
                         "tracking: PD target = reference pose of the next frame (episodes last like a trained policy's)")
    return ap.parse_args()


def cpu_baseline(num_envs=4096, budget_s=12.0, max_steps=4000):
    """CPU port of the same path on the host cores: oracle/hostemu = the kernels' per-lane functions compiled with
    g++ -O2 -fopenmp, one env per OpenMP iteration.  (The reference has no CPU dynamics at all -- Isaac Gym is a GPU
    binary -- and its reward/obs path is Python/torch; this port is the builder's CPU restatement)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hostemu_util as hu
    from hostemu_util import P, emu
    from phc_amd import abi
    from phc_amd.motion_lib import process_clip
    from phc_amd.utils.synthetic_motion import make_motion_dict
    F = np.float32
    model, mstruct, keep = hu.np_model()
    nb, nd = model.num_bodies, model.num_dof
    clip = list(make_motion_dict(model.parent, 1, seed=0, body_names=model.body_names).values())[0]
    proc = process_clip(model.parent, model.local_translation, clip["pose_quat_global"], clip["root_trans_offset"], 30)
    T = proc["gts"].shape[0]
    lib = {k: proc[k].astype(F) for k in ("gts", "grs", "gvs", "gavs", "lrs", "dvs")}
    lib.update(motion_lengths=np.full(num_envs, (T - 1) / 30, F), motion_dt=np.full(num_envs, 1 / 30, F),
               motion_num_frames=np.full(num_envs, T, np.int64), length_starts=np.zeros(num_envs, np.int64))
    lstruct, lkeep = hu.np_motion_lib(lib)
    tabs = abi.task_index_tables(model, model.body_names, [b for b in model.body_names if "Toe" not in b and "Ankle" not in b],
                                 ["R_Ankle", "L_Ankle", "R_Wrist", "L_Wrist"])
    td = np.full(64, 0.25, F)
    prm = abi.im_params_struct(dt=1 / 30, max_episode_length=300, reward_specs=dict(k_pos=100, k_rot=10, k_vel=0.1, k_ang_vel=0.1, w_pos=0.5,
                               w_rot=0.3, w_vel=0.1, w_ang_vel=0.1), power_reward=True, power_coefficient=0.0005, enable_early_termination=True,
                               use_mean_termination=False, disable_collision_check=False, local_root_obs=True, root_height_obs=True,
                               num_track_bodies=nb, track_slot=tabs[0], reset_mask=tabs[1], num_reset_bodies=20, first_reset_body=0,
                               termination_distances=td, num_key_bodies=4, key_body_ids=tabs[2], num_amp_joints=tabs[4], amp_joint_slot=tabs[3],
                               num_amp_obs_steps=10, num_amp_obs_per_step=196, num_self_obs=358, num_task_obs=576)
    N = num_envs
    a = dict(root=np.zeros((N, 13), F), dof=np.zeros((N, nd, 2), F), rbs=np.zeros((N, nb, 13), F), cf=np.zeros((N, nb, 3), F),
             df=np.zeros((N, nd), F), pd=np.zeros((N, nd), F))
    sim = abi.sim_state_struct(N, a["root"], a["dof"], a["rbs"], a["cf"], a["df"], a["pd"])
    amp = [np.zeros((N, 10, 196), F), np.zeros((N, 10, 196), F)]
    b = dict(progress=np.zeros(N, np.int64), reset=np.ones(N, np.int64), term=np.zeros(N, np.int64), rew=np.zeros(N, F), raw=np.zeros((N, 5), F),
             obs=np.zeros((N, 934), F), mids=np.arange(N, dtype=np.int64), st=np.zeros(N, F), so=np.zeros(N, F), goff=np.zeros((N, 3), F))
    params = abi.sim_params_struct(self_collision=1, inertia_lag=1)   # as the GPU run: robot.has_self_collision is True in smpl_humanoid.yaml, lagged inertias (the task's default)
    rng = np.random.default_rng(0)
    actions = ((rng.random((N, nd)) * 2 - 1) * 0.1).astype(F)
    off, scale = model.pd_action_offset_scale()
    e = emu()

    def one(cur):
        buf_r = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp[cur], amp[cur], b["mids"], b["st"], b["so"], b["goff"])
        phase = rng.random(N).astype(F)
        e.emu_im_reset(P(mstruct), P(lstruct), P(prm), P(sim), P(buf_r), N, None, abi.ptr(phase), 0)
        e.emu_sim_step(P(mstruct), P(params), P(sim), abi.ptr(actions), abi.ptr(off), abi.ptr(scale), None, 2, 1)
        buf = abi.im_buffers_struct(b["progress"], b["reset"], b["term"], b["rew"], b["raw"], b["obs"], amp[cur], amp[1 - cur], b["mids"], b["st"], b["so"], b["goff"])
        e.emu_im_post_physics(P(mstruct), P(lstruct), P(prm), P(sim), P(buf))
        return 1 - cur

    def usable_cores():
        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:  # container CPU quota (cgroup v2), which sched_getaffinity does not show
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                n = max(1, min(n, int(float(q) / float(per) + 0.5)))
        except Exception:
            pass
        return n

    def rate(threads, nsteps, cur):
        e.emu_set_threads(int(threads))
        t0 = time.perf_counter()
        for _ in range(nsteps):
            cur = one(cur)
        return N * nsteps / (time.perf_counter() - t0), cur

    cur = one(0)
    cores = usable_cores()
    cands = sorted({c for c in (1, 4, 8, 16, 32, 64, 128, cores) if c <= cores})
    best, best_rate = 1, 0.0
    for c in cands:  # short calibration: oversubscribed or quota-limited hosts are common in containers
        r, cur = rate(c, 2, cur)
        if r > best_rate:
            best, best_rate = c, r
    e.emu_set_threads(int(best))
    steps, t0 = 0, time.perf_counter()
    while steps < max_steps and time.perf_counter() - t0 < budget_s:  # bounded sample: ~budget_s of CPU work
        for _ in range(10):
            cur = one(cur)
        steps += 10
    dt = time.perf_counter() - t0
    val = N * steps / dt
    return {"value": val, "unit": "env-steps/s", "cores": best, "kind": "port",
            "sample": f"{steps} steps x {N} envs (reset+stepper+post-physics), g++ -O2 -fopenmp build of the kernels' per-lane code "
                      f"(oracle/hostemu), {dt:.1f} s wall on {best} OpenMP threads (best of {cands}; host reports {os.cpu_count()} cpus)"}


_REF_CACHE = {}


def reference_timings():
    """The reference's own CPU stage timings (oracle/time_reference.py).  LIVE on this host when a copy of the reference is importable -- /root/reference in the
    build container, or the git-ignored travel archive oracle/_ref/reference_modules.zip that oracle/make_ref.py packed and that rides along with the gpurun snapshot (the checker
    being timed as a baseline; nothing of it is in the timed region of the path) -- else the newest committed profiles/*_reference_cpu_stages.json, with
    `measured_in` saying which.  -> (dict, source string) or (None, None)."""
    if "v" in _REF_CACHE:
        return _REF_CACHE["v"]
    import subprocess
    res = (None, None)
    have = os.path.isdir("/root/reference/phc") or os.path.isfile(os.path.join(ROOT, "oracle", "_ref", "reference_modules.zip"))
    if have and not os.environ.get("PHC_BENCH_NO_LIVE_REFERENCE"):
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "time_reference.py"), "--stdout"], capture_output=True, text=True, timeout=420,
                               env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
            line = next(l for l in r.stdout.splitlines() if l.startswith("REFERENCE_JSON"))
            res = (json.loads(line[len("REFERENCE_JSON"):]), "live: oracle/time_reference.py --stdout on this host")
        except Exception as exc:   # noqa: BLE001
            print(f"[bench] live reference timing failed ({type(exc).__name__}: {exc}); falling back to the committed file", file=sys.stderr)
    if res[0] is None:
        files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_reference_cpu_stages.json"))
        if files:
            d = json.load(open(os.path.join(ROOT, "profiles", files[-1])))
            d["measured_in"] = "NOT this host -- committed file from " + d.get("measured_in", "the build container")
            res = (d, "profiles/" + files[-1])
    _REF_CACHE["v"] = res
    return res


def cpu_reference():
    """The REFERENCE'S OWN CPU path (stages C1-C3: motion lookup x2, imitation reward + reset, self / task / AMP
    observations; C0 load_motions) as measured by oracle/time_reference.py -- on THIS host's cores when the reference (or its travel archive) is present,
    see reference_timings().  There is no reference CPU number for C4 (dynamics: closed Isaac Gym binary); `cpu_baseline` (kind "port") covers it."""
    d, src = reference_timings()
    if d is None:
        return None
    st = d["stages"]
    return {"kind": "reference", "what": "phc MotionLibSMPL.get_motion_state x2 + compute_imitation_reward + compute_humanoid_im_reset + "
            "compute_humanoid_observations_smpl_max + compute_imitation_observations_v6 + build_amp_observations_smpl (no dynamics: C4 has no CPU code)",
            "value": st["4096"]["env_steps_per_s_reward_obs_only"], "unit": "env-steps/s", "cores": d["host"]["threads"], "host": d["host"]["model"],
            "stages_ms": {n: {k: v for k, v in st[n].items() if k.endswith("_ms")} for n in st}, "load_motions_ms_per_clip": d["load_motions"]["ms_per_clip"],
            "protocol": d.get("protocol", "5 warm-up + 50 timed iterations, median"), "measured_in": d.get("measured_in", "build container"),
            "sample": "50 timed iterations per stage at 64 and 4096 envs (about 10 s of CPU work)", "source": src}


def config0_line(dev):
    """BASELINE.json configs[0] -- "poselib FK + imitation-reward on 64 envs, CPU PyTorch, single AMASS clip (plumbing, no GPU)" -- next to the same
    work on the MI355X: `phc_fk` over 64 poses of one clip + the post-physics launch (reference lookup x2, imitation reward, reset test,
    observations, AMP frame) of a 64-env task, timed with HIP events over 200 repetitions; the reference side is its own poselib
    `SkeletonState.from_rotation_and_root_translation` + `compute_imitation_reward` timed by oracle/time_reference.py in the build container."""
    from phc_amd.config import compose
    from phc_amd.env.tasks.vec_task import parse_task
    from phc_amd import _lib as L
    d_ref, ref_src = reference_timings()
    ref = d_ref.get("config0") if d_ref else None
    dev = torch.device(dev)
    di = dev.index if dev.index is not None else torch.cuda.current_device()
    task, env = parse_task(compose(["env.num_envs=64", "env.motion_file=synthetic:1:0", f"device_id={di}", f"rl_device=cuda:{di}"]), device_id=di)
    env.reset()
    lib = task._motion_lib
    lr = lib.lrs[:64].contiguous()
    rt = lib.gts[:64, 0].contiguous()
    grot, gpos = torch.empty(64, task.num_bodies, 4, device=dev), torch.empty(64, task.num_bodies, 3, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def once():
        L.check(task._lib.phc_fk(task._model_struct, 64, lr.data_ptr(), rt.data_ptr(), grot.data_ptr(), gpos.data_ptr(), stream), "phc_fk")
        task.post_physics_step()
    for _ in range(20):
        once()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(200):
        once()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 200
    out = {"what": "BASELINE configs[0] shape: FK of 64 poses of one clip + imitation reward (+ the rest of post-physics) on 64 envs",
           "mi355x_ms": ms, "mi355x_env_evaluations_per_s": 64 / (ms * 1e-3), "mi355x_method": "HIP events around 200 x (phc_fk + phc_im_post_physics), 64 envs "
           "(launch-latency bound at this size: two launches)"}
    if ref is not None:
        out.update(reference_cpu_ms=ref["fk_plus_reward_ms"], reference_cpu_poselib_fk_ms=ref["poselib_fk_ms"], reference_cpu_env_evaluations_per_s=ref["env_evaluations_per_s"],
                   reference_source=ref_src, reference_measured_in=d_ref.get("measured_in"), reference_cores=d_ref["host"]["threads"])
    return out


def other_workloads():
    """The same env step on the other single-GPU workloads of BASELINE.json, measured NOW by child runs of this script (so that the driver's
    line carries them too): configs[1] with tracking actions (no reset storm), configs[4] Unitree H1.  -> {name: {value, ms_per_step, ...}}."""
    import subprocess
    env = dict(os.environ, PHC_BENCH_CHILD="1")
    out = {}
    for name, tail in (("configs1_tracking_actions", ["--actions", "tracking"]), ("configs4_unitree_h1", ["--config", "5"]),
                       ("configs1_fresh_inertias", ["--solver", "inertia_lag=0"])):   # (round 6: the lagged scheme is the default; this is the every-sub-step-fresh one)
        cmd = [sys.executable, os.path.abspath(__file__), "--steps", "300", "--warmup", "30", "--ppo-epochs", "0", "--no-cpu-baseline", "--no-pmc"] + tail
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
            d = json.loads(line)
            out[name] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "stepper_kernel_ms": d["roofline"]["kernel_ms"],
                         "workload": d["config"]["workload"], "envs_per_gpu": d["config"]["envs_per_gpu"]}
        except Exception as exc:   # noqa: BLE001
            out[name] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    # the regime a trained policy produces: the 64-clip locomotion library tracked by a policy trained on the spot (<= 90 s), < 2 % of the envs reset per step
    try:
        r = subprocess.run([sys.executable, "-m", "phc_amd.learning.bench_policy", "--train-s", "90", "--steps", "300"], env=env, capture_output=True, text=True, timeout=400, cwd=ROOT)
        out["trained_policy"] = json.loads(next(l for l in r.stdout.splitlines() if l.startswith("POLICY_JSON"))[len("POLICY_JSON"):])
    except Exception as exc:   # noqa: BLE001
        out["trained_policy"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    # the path's collective on ONE rank at the two bucket sizes (its per-step floor; over xGMI: unmeasured)
    try:
        r = subprocess.run([sys.executable, "-m", "phc_amd.learning.bench_collective"], env=env, capture_output=True, text=True, timeout=180, cwd=ROOT)
        out["collective_us"] = json.loads(next(l for l in r.stdout.splitlines() if l.startswith("COLLECTIVE_JSON"))[len("COLLECTIVE_JSON"):])
    except Exception as exc:   # noqa: BLE001
        out["collective_us"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    # the PPO half on the reference's flagship learner config (phc/data/cfg/learning/im_pnn_big.yaml: 2048-1536-1024-1024-512-512, SiLU, PNN actor)
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "20", "--warmup", "5", "--ppo-epochs", "3", "--learning", "im_pnn_big", "--no-cpu-baseline", "--no-pmc"]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        out["ppo_learning_im_pnn_big"] = {k: d[k] for k in ("ppo_samples_per_s", "ppo_epoch_ms", "ppo_play_ms", "ppo_update_ms", "ppo_roofline", "ppo_config") if k in d}
    except Exception as exc:   # noqa: BLE001
        out["ppo_learning_im_pnn_big"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
    return out


def live_pmc(argv_tail, counter_groups):
    """PMC counters per kernel measured NOW: one child run of this script per counter group under `rocprofv3 --pmc <group> --kernel-trace` (never
    together with another trace domain; FETCH_SIZE and WRITE_SIZE in separate passes: TCC slot limit -- MI355X_MICROARCH.md).
    -> ({kernel: {counter: (median per dispatch, dispatches)}}, None) or (None, reason)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    per, failed = {}, []
    env = dict(os.environ, PHC_BENCH_CHILD="1", TMPDIR="/tmp")
    for group in counter_groups:     # a group that fails (counter not offered on a box, time-out) does not take the others with it
        d = tempfile.mkdtemp(prefix="phc_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", *group.split(), "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
               os.path.abspath(__file__), "--steps", "20", "--warmup", "5", "--ppo-epochs", "0", "--no-cpu-baseline", "--no-pmc"] + argv_tail
        try:
            r = subprocess.run(cmd, env=env, cwd="/tmp", capture_output=True, text=True, timeout=240)
            path = next((os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")), None)
            if r.returncode != 0 or path is None:
                raise RuntimeError(f"rc={r.returncode}")
            vals = {}
            for row in csv.DictReader(open(path)):
                vals.setdefault((row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"]), []).append(float(row["Counter_Value"]))
            for (k, c), v in vals.items():
                v = sorted(v)
                per.setdefault(k, {})[c] = (v[len(v) // 2], len(v))   # median: the first dispatches (everything reset at once) are not the steady state
        except Exception as exc:   # noqa: BLE001
            failed.append(f"{group}: {type(exc).__name__} {exc}"[:120])
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if failed:
        print(f"[bench] PMC passes that failed: {failed}", file=sys.stderr)
    return (per, None) if per else (None, "; ".join(failed) or "no counters")


def pmc_traffic_of(per, prefix):
    """HBM bytes per launch of the kernel whose name starts with `prefix`, corrected as MI355X_MICROARCH.md prescribes for gfx950 (read bytes =
    2 x FETCH_SIZE KiB, WRITE_SIZE KiB as is; calibrated on k_im_post_physics in profiles/pmc_summary.py) -> (bytes, detail) or (None, reason)."""
    k = next((k for k in per if k.startswith(prefix) and "FETCH_SIZE" in per[k] and "WRITE_SIZE" in per[k]), None)
    if k is None:
        return None, f"no {prefix} dispatch in the FETCH_SIZE / WRITE_SIZE passes"
    rd, wr = 2.0 * per[k]["FETCH_SIZE"][0] * 1024.0, per[k]["WRITE_SIZE"][0] * 1024.0
    return rd + wr, {"kernel": k, "read_bytes": rd, "write_bytes": wr, "dispatches": per[k]["FETCH_SIZE"][1],
                     "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace), median per dispatch; gfx950: read = 2 x FETCH_SIZE"}


def device_state(dev):
    """What kind of box is this?  The pool shows two populations (profiles/r01_notes.md, r02_notes.md: memory-bound kernels 1.6 x apart);
    a 1 GiB device-to-device copy rate and the driver's clock / power readings make a line interpretable without a second run."""
    import shutil
    import subprocess
    p = torch.cuda.get_device_properties(dev)
    out = {"name": p.name, "compute_units": p.multi_processor_count, "hbm_GiB": round(p.total_memory / 2 ** 30, 1), "hip": torch.version.hip}
    try:
        a = torch.empty(2 ** 28, dtype=torch.float32, device=dev)
        b = torch.empty_like(a)
        for _ in range(3):
            b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        out["d2d_copy_GBs_read_plus_write"] = 10 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b
    except Exception as exc:   # noqa: BLE001
        out["d2d_copy_error"] = type(exc).__name__
    smi = shutil.which("rocm-smi")
    if smi:
        try:
            r = subprocess.run([smi, "--showclocks", "--showpower", "--showperflevel", "--json"], capture_output=True, text=True, timeout=20)
            card = next(iter(json.loads(r.stdout).values()))
            out["rocm_smi"] = {k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "power", "performance"))}
        except Exception as exc:   # noqa: BLE001
            out["rocm_smi_error"] = type(exc).__name__
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU (RCCL)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and "gloo" not in sys.argv:
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible on this node -- refusing to report an {n}-GPU line from fewer devices")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (the host driver's only mode): RCCL needs it across processes
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Exactly ONE line on stdout, the JSON the driver parses: everything else that writes to descriptor 1 -- RCCL prints a version banner there when
    # its first communicator comes up, after Python has flushed its own buffer -- is sent to stderr; the line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the path has no CPU fallback")
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.backend == "nccl" and torch.cuda.device_count() < min(world, int(os.environ.get("LOCAL_WORLD_SIZE", world))):
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
    if args.backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())     # ranks share the visible GPU(s)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.force_rccl:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 400))
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    if args.config == 3:      # configs[2]: full-AMASS-sized library, 8192 envs (defaults only: explicit flags win)
        if "--envs" not in " ".join(sys.argv):
            args.envs = 8192
        if "--motion-clips" not in " ".join(sys.argv):
            args.motion_clips = 11313
    elif args.config == 5:    # configs[4]: H1 morphology
        args.robot = "h1"

    from phc_amd.config import compose
    from phc_amd.env.tasks.vec_task import parse_task
    torch.manual_seed(rank)  # per-rank seed offset, as the reference's horovod path does (run_hydra.py:121)
    robot_over = [f"robot=unitree_{args.robot}", f"env=env_im_{args.robot}_phc", "sim=robot_sim", "control=robot_control"] if args.robot != "smpl" else []
    # the captured update graph holds no collective (the flat-gradient all-reduce follows each replay eagerly), so one and many ranks
    # replay the same graph; --no-update-graph falls back to eager launches
    graph_over = ["+learning.params.config.hip_graph=True"] if not args.no_update_graph else []
    if args.force_rccl:
        graph_over.append("+learning.params.config.force_collectives=True")
    learn_over = ([f"learning={args.learning}"] + (["env=env_im_pnn"] if "pnn" in args.learning and args.robot == "smpl" else [])) if args.learning != "im" else []
    cfg = compose(learn_over + robot_over + graph_over + [f"env.num_envs={args.envs}", f"env.motion_file=synthetic:{args.motion_clips}:0", f"device_id={local_rank}",
                                f"rl_device=cuda:{local_rank}", f"+solver.lane_mapping={args.lane_mapping}"] + ([f"+solver.self_collision={args.self_collision}"] if args.self_collision >= 0 else [])
                  + [f"+solver.{kv}" for kv in args.solver] + [f"+learning.params.config.{kv}" for kv in args.learner])
    t_build = time.perf_counter()
    task, env = parse_task(cfg, device_id=local_rank)
    t_build = time.perf_counter() - t_build
    dev = task.device
    N = task.num_envs
    env.reset()
    cfg3 = None
    if args.config == 3:   # configs[2]: time what the reference does every 500 epochs (amp_agent.py:511-515): re-sample one clip per env
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        task.resample_motions()
        torch.cuda.synchronize()
        lib_ = task._motion_lib
        cfg3 = {"library_clips": int(lib_._num_unique_motions), "distinct_clips_sampled": int(torch.unique(lib_._curr_motion_ids).numel()),
                "task_build_s_incl_synthetic_library_and_first_load": t_build, "resample_motions_s": time.perf_counter() - t0,
                "motion_frames": int(lib_.frames.shape[0]), "motion_bytes_on_device": int(lib_.frames.numel() * 4),
                "host_workers": int(lib_.m_cfg.get("num_workers", 0)) or __import__("phc_amd.motion_lib", fromlist=["x"]).default_pool_workers()}
    actions = (torch.rand(N, task.num_actions, device=dev) * 2 - 1) * 0.1  # fixed a ~ U(-1,1)*0.1

    inv_scale = 1.0 / task._pd_action_scale

    def env_step(ev=None, ev_post=None):
        """One step THROUGH THE B1 BOUNDARY the metric names (`VecTaskPython.step`, phc/env/tasks/vec_task.py:145-162 -> phc_amd/env/tasks/vec_task.py): the rollout
        idiom's reset of the envs that finished on the previous step, then `env.step(actions)` -> (obs, reward, done, info).  The HIP events around the stepper /
        post-physics launches are recorded inside `task.step` (HumanoidIm._launch_events) on the steps that carry them."""
        task.reset_done()                 # envs that finished on the previous step (device-side mask, no host sync)
        if args.actions == "random":
            a = actions
        elif args.robot != "smpl":
            a = task.ref_dof_pos - task.default_dof_pos
        else:
            a = (task.ref_dof_pos - task._pd_action_offset) * inv_scale
        if ev is not None or ev_post is not None:
            task._launch_events = (ev, ev_post)
        return env.step(a)

    # The SURVEY protocol is a STEADY state (fixed random actions, ~98 % of the envs within 5 steps of a reset, resets spread over the steps).  Right
    # after env.reset() all envs are in lockstep -- they fall, and are reset, in the same few steps -- so a short run (the driver's K = 20, W = 5)
    # would time that transient instead.  Untimed burn-in in front of the W warm-up steps; reported in the line.
    BURN_IN = max(0, args.burn_in)
    for _ in range(BURN_IN):
        env_step()
    for _ in range(args.warmup):
        env_step()
    # HIP events around the stepper launch of every EVENT_STRIDE-th timed step (an event pair costs the stream two extra packets per step; the
    # average launch duration does not need all of them)
    EVENT_STRIDE = 4
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if k % EVENT_STRIDE == 0 else None for k in range(args.steps)]
    events_post = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if k % EVENT_STRIDE == 2 else None for k in range(args.steps)]   # (the post-physics launch, on other steps)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        env_step(events[k], events_post[k])
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        gathered = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)                      # every rank's own time: weak scaling readable per rank
        per_rank = [float(x.item()) for x in gathered]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean([ev[0].elapsed_time(ev[1]) for ev in events if ev is not None]))  # torch's current stream == the launch stream
    post_ms = float(np.mean([ev[0].elapsed_time(ev[1]) for ev in events_post if ev is not None])) if any(e is not None for e in events_post) else None
    resets = float((task.progress_buf < 5).float().mean().item())

    ppo = None
    if args.ppo_epochs > 0:
        try:
            from phc_amd.learning.bench_ppo import time_ppo_epochs
            ppo = time_ppo_epochs(task, env, cfg, args.ppo_epochs, dist)
            ppo["ppo_config"]["learning"] = args.learning
        except ImportError:
            ppo = None

    if rank == 0:
        nsub = task.control_freq_inv * int(cfg.sim.get("substeps", 2))
        D_, NB_ = task.num_dof, task.num_bodies
        aba_bytes = 4 * ((13 + 2 * D_ + D_) + (13 + 2 * D_))          # read root+dof+target, write root+dof (1484 B for SMPL)
        pub_bytes = 4 * (NB_ * 13 + D_ + NB_ * 3)                      # body state + dof force + contact force (1812 B for SMPL)
        assert args.robot != "smpl" or (aba_bytes, pub_bytes) == (ABA_BYTES_PER_ENV_SUBSTEP, PUBLISH_BYTES_PER_ENV_STEP)
        bytes_per_launch = (aba_bytes * nsub + pub_bytes) * N
        traffic, traffic_src = None, None  # HBM bytes per launch of the stepper from the PMC counters
        per, post_traffic, post_src, sq = None, None, None, None
        if world == 1 and not args.no_pmc and not os.environ.get("PHC_BENCH_CHILD"):
            tail = ["--envs", str(args.envs), "--robot", args.robot, "--lane-mapping", str(args.lane_mapping), "--actions", args.actions,
                    "--motion-clips", str(args.motion_clips), "--self-collision", str(args.self_collision)] + [x for kv in args.solver for x in ("--solver", kv)]
            # four child passes: HBM read / write traffic (all kernels), then the SQ counters behind the VALU roofline of the stepper
            per, why = live_pmc(tail, ["FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU"])
            if per is None:
                print(f"[bench] live PMC counters unavailable ({why}); falling back to the committed profile", file=sys.stderr)
            else:
                traffic, detail = pmc_traffic_of(per, "k_sim_step<true")
                traffic_src = {"live": detail} if traffic is not None else None
                post_traffic, post_src = pmc_traffic_of(per, "k_im_post_physics")
                k = next((k for k in per if k.startswith("k_sim_step<true") and "SQ_INSTS_VALU" in per[k] and "SQ_ACTIVE_INST_VALU" in per[k]), None)
                if k is not None:
                    c = {n: v[0] for n, v in per[k].items()}
                    sq = {"valu_instructions_per_wavefront": c["SQ_INSTS_VALU"] / max(c.get("SQ_WAVES", 0.0), 1.0),
                          "valu_active_share_of_wavefront_cycles": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None,
                          "active_lane_share": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]) if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU") else None,
                          "method": "rocprofv3 --pmc (SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU | SQ_THREAD_CYCLES_VALU, separate passes, --kernel-trace), "
                                    "median per dispatch of " + k}
        pmc = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))
        if traffic is None and pmc and N == 4096 and args.robot == "smpl":
            tab = json.load(open(os.path.join(ROOT, "profiles", pmc[-1])))
            rec = next((v for k, v in tab.items() if k.startswith("k_sim_step<true")), None)
            if rec:
                traffic, traffic_src = rec["traffic_bytes"], "profiles/" + pmc[-1]
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        tflops = FLOPS_PER_ENV_SUBSTEP * nsub * N / (kern_ms * 1e-3) / 1e12
        kname = traffic_src["live"]["kernel"] if isinstance(traffic_src, dict) else "k_sim_step (one body per lane)"
        out = {
            "metric": "env-steps/sec at 4096 humanoid envs per GPU (VecEnv.step incl. resets)",
            "value": N * world * args.steps / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (AMASS-shaped smooth random clip, seed 0; random-init state from the reference motion)",
            "config": {"workload": (("BASELINE configs[1]: SMPL humanoid 69-DoF, 4096 envs per GPU, single reference motion, " if args.config != 3 else
                                     f"BASELINE configs[2] shape: SMPL humanoid, {N} envs, AMASS-sized synthetic library of {args.motion_clips} clips, ") +
                                    "30 Hz control = 2 x simulate @60 Hz x 2 sub-steps") if args.robot == "smpl" else
                                   (("BASELINE configs[4]: Unitree H1 19-DoF" if args.robot == "h1" else "env_im_g1_phc: Unitree G1 37-DoF, 38 bodies") +
                                    ", envs per GPU as given, synthetic retargeted-shape clips, 50 Hz control = 4 x simulate @200 Hz x 2 sub-steps, pd torque mode"),
                       "timed_call": "task.reset_done() + VecTaskPythonWrapper.step(actions) -> (obs, reward, done, info): the B1 boundary (phc/env/tasks/vec_task.py:145-162)",
                       "envs_per_gpu": N, "num_bodies": task.num_bodies,
                       "obs": task.num_obs, "amp_obs": task.get_num_amp_obs(), "parallelism": f"env-sharded x{world}",
                       "self_collision": bool(task._sim_params.self_collision)},
            # The dominant kernel is the stepper, and what bounds it is the fp32 VECTOR pipe seen through one wavefront's serial instruction stream -- not HBM
            # (PMC traffic = 1.2 x the fused launch's compulsory bytes).  `roofline` therefore prices it against the fp32 vector peak; the HBM accounting of
            # SURVEY 8(d) (the north star's "fraction of HBM roofline") is kept in full under `roofline.hbm`.
            "roofline": {"kernel": "phc_sim_step -> %s (A2 + %d ABA sub-steps + S7 publication)" % (kname, nsub),
                         "bound": "valu", "achieved": tflops, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / FP32_VECTOR_PEAK_TFLOPS,
                         "algorithmic_flops_per_launch": FLOPS_PER_ENV_SUBSTEP * nsub * N,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel_ms": kern_ms, "kernel_ms_method": "HIP events around the launch of every %d-th of the %d timed steps, launch stream" % (EVENT_STRIDE, args.steps),
                         "sq_counters": sq,
                         "hbm": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                 "algorithmic_bytes_per_launch": bytes_per_launch, "compulsory_bytes_per_launch_fused": (aba_bytes + pub_bytes) * N,
                                 "note": "SURVEY 8(d) canonical un-fused accounting: %d B x %d sub-steps + %d B per env; the fused launch reads the state once and "
                                         "writes it once (compulsory %d B/env)" % (aba_bytes, nsub, pub_bytes, aba_bytes + pub_bytes)},
                         "note": "SURVEY 8(d): 40 kflop per env and sub-step against the fp32 vector peak (MI355X_MICROARCH.md: 157.3 TFLOP/s).  At 4096 envs all 2048 "
                                 "wavefronts are resident at once (two per SIMD) and the launch lasts as long as ONE wavefront's serial stream: ~15.5 k VALU "
                                 "instructions at a per-wavefront issue interval of ~4.4 cycles + LDS / scalar / waits; one lane per body leaves `active_lane_share` "
                                 "of the lanes working during the level-synchronous tree sweeps (profiles/r03_stepper/README.md, profiles/microbench/valu_issue_mi355x.txt)"},
        }
        if post_ms is not None:
            # the HBM-bound kernel of the env step: bytes per env as DESIGN.md section 4 counts them (7 774 read + 6 522 written for the SMPL task)
            post_bytes = 14296 * N if (args.robot == "smpl" and task.num_obs == 934) else None
            out["roofline_post_physics"] = {"kernel": "phc_im_post_physics -> k_im_post_physics (reference lookups x2, reward, reset test, observations, AMP frame)",
                                            "bound": "hbm", "achieved": (post_bytes / (post_ms * 1e-3) / 1e9) if post_bytes else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": (post_bytes / (post_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if post_bytes else None, "traffic": post_traffic,
                                            "traffic_source": {"live": post_src} if post_traffic is not None else None, "kernel_ms": post_ms,
                                            "kernel_ms_method": "HIP events around phc_im_post_physics on every %d-th timed step (steps the stepper events skip)" % EVENT_STRIDE,
                                            "algorithmic_bytes_per_launch": post_bytes}
        if ppo is not None:
            out.update(ppo)
        if world == 1 and not args.no_cpu_baseline and args.robot == "smpl":
            out["cpu_baseline"] = cpu_baseline()
            ref = cpu_reference()
            if ref is not None:
                out["cpu_reference"] = ref
            if not os.environ.get("PHC_BENCH_CHILD"):
                try:
                    out["config0"] = config0_line(dev)
                except Exception as e:   # never lose the line over the plumbing comparison
                    out["config0"] = {"error": repr(e)}
        if cfg3 is not None:
            out["config3_motion_library"] = cfg3
        if (world == 1 and not args.no_other_workloads and not os.environ.get("PHC_BENCH_CHILD") and args.config == 2 and args.robot == "smpl"
                and args.actions == "random" and args.envs == 4096):
            out["other_workloads"] = other_workloads()
        out["per_rank"] = [{"rank": r, "elapsed_s": e, "env_steps_per_s": N * args.steps / e} for r, e in enumerate(per_rank)]
        out["actions"] = args.actions
        if args.solver:
            out["solver_overrides"] = list(args.solver)
        if not os.environ.get("PHC_BENCH_CHILD"):
            out["device"] = device_state(dev)
        out["envs_within_5_steps_of_a_reset"] = resets
        out["protocol_burn_in_steps"] = BURN_IN   # untimed, before the W warm-up steps: desynchronises the resets after env.reset()
        print(json.dumps(out), file=real_stdout, flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
