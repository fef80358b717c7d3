"""-m gpu: parity at every BASELINE.json configuration's OWN size -- configs[1] stays in
tests/test_env_gpu.py::test_step_matches_oracle[4096].

  configs[2]  8192 SMPL envs, multi-clip library (512 synthetic clips), a NON-identity `sampled_motion_ids` table, `cycle_motion` on, clips
              running out inside the test; the stepper against the fp64 dense oracle on envs of BOTH occupancy rounds of the launch
              (4096 wavefronts at 2 resident per SIMD x 1024 SIMDs: envs 0 .. 4095 start first, 4096 .. 8191 behind them)
  configs[4]  Unitree H1 at 4096 envs: reward over the 20 + 3 extended bodies, 298 + 480 observation floats, 63 x 10 AMP history,
              `pd` torque control; the explicit torque the stepper holds == the reference's `_compute_torques` at 4096 rows
  G1          38 bodies at 4096 envs: the 64-lane instantiation of every env kernel (one env per wavefront)

Everything is compared with the numpy oracle (pinned to reference-generated goldens on the CPU, tests/test_oracle_golden.py) driven with
the task's own tensors: indices / flags / start times bit-exact, floats <= 1e-4 (north_star), stepper <= 1e-3 m vs fp64."""
import numpy as np
import pytest
import torch

import phc_oracle as po
from step_oracle import StepChecker

pytestmark = pytest.mark.gpu
F = np.float32
H1_OVER = {"robot": "unitree_h1", "env": "env_im_h1_phc", "sim": "robot_sim", "control": "robot_control"}
G1_OVER = {"robot": "unitree_g1", "env": "env_im_g1_phc", "sim": "robot_sim", "control": "robot_control"}


def make_task(num_envs, motion, seed=0, **over):
    from phc_amd.config import compose
    from phc_amd.env.tasks.vec_task import parse_task
    torch.manual_seed(seed)
    cfg = compose([f"env.num_envs={num_envs}", f"env.motion_file={motion}"] + [f"{k}={v}" for k, v in over.items()])
    return parse_task(cfg)


def test_config3_8192_envs_multi_clip_library_non_identity_ids_cycle_motion():
    n = 8192
    task, env = make_task(n, "synthetic:512:3:2.0", **{"env.cycle_motion": True})
    assert task.num_envs == n and task.cycle_motion
    lib_ = task._motion_lib
    assert int(lib_._num_unique_motions) == 512 and torch.unique(lib_._curr_motion_ids).numel() > 400   # a real multi-clip draw
    dev = task.device
    # NON-identity clip table: env i tracks the library slot perm[i] (the reference indexes every lookup with _sampled_motion_ids,
    # humanoid_im.py:880,1151; the kernels skip the table only while it is the identity)
    g = torch.Generator().manual_seed(11)
    perm = torch.randperm(n, generator=g).to(dev)
    task._sampled_motion_ids.copy_(perm)
    env.reset()
    assert not task._motion_ids_are_identity()
    np.testing.assert_array_equal(task._sampled_motion_ids.cpu().numpy(), perm.cpu().numpy())
    # reset state == the reference state of the env's OWN clip (through the table)
    res = lib_.get_motion_state(task._sampled_motion_ids, task._motion_start_times)
    np.testing.assert_allclose(task._rigid_body_pos.cpu().numpy(), res["rg_pos"].cpu().numpy(), atol=2e-5)
    # every 8th env sits 2.5 steps in front of its clip's end: those clips run out (and restart: cycle_motion) inside the test
    ends = torch.arange(0, n, 8, device=dev)
    lens = lib_._motion_lengths[task._sampled_motion_ids[ends]]
    task._motion_start_times[ends] = (lens - 2.5 * task.dt).clamp_min(0.0)
    chk = StepChecker(task)
    n_wrapped = 0
    both_rounds = sorted({0, 1, 63, 2047, 4095, 4096, 4097, 6000, n - 64, n - 1})
    for it in range(5):
        chk.before()
        actions = (torch.rand(n, 69, device=dev) * 2 - 1) * 0.3
        obs, rew, done, info = env.step(actions)
        if it in (0, 4):
            chk.dynamics(actions, both_rounds)
        out = chk.after(obs, rew, done, info)
        n_wrapped += int(out["wrapped"].sum())
        assert not out["reset"][out["wrapped"]].any(), "a restarted clip is in its recovery window: no reset"
        if it % 2 == 0:
            env.reset(done.nonzero(as_tuple=False).squeeze(-1))
        else:
            task.reset_done()
    assert n_wrapped >= 300, n_wrapped     # (many of the envs placed at a clip end terminate first: their pose is the old start time's)
    assert obs.shape == (n, 934) and info["amp_obs"].shape == (n, 1960)


@pytest.mark.parametrize("rb,over,nb,obs_dim,amp", [("h1", H1_OVER, 20, 298 + 480, 63), ("g1", G1_OVER, 38, 568 + 912, 99)])
def test_robot_env_step_matches_oracle_at_4096_envs(rb, over, nb, obs_dim, amp):
    """BASELINE configs[4] (H1, 32-lane groups, two envs per wavefront) and G1 (64-lane instantiations) at 4096 envs."""
    n = 4096
    task, env = make_task(n, "synthetic:3:2:2.0", **over)
    assert task.humanoid_type == rb and task.num_bodies == nb and task.num_obs == obs_dim and task.get_num_amp_obs() == amp * 10
    assert task.control_mode == "pd" and task.control_freq_inv == 4
    dev = task.device
    env.reset()
    chk = StepChecker(task)
    spread = sorted({0, 1, 2, 63, 64, n // 2 + 5, n - 65, n - 1})
    tot_done = 0
    for it in range(5):
        chk.before()
        # PD targets around the next reference pose: episodes last, and a tenth of the envs gets large random actions (terminations, saturated drives)
        act = task.ref_dof_pos - task.default_dof_pos + torch.randn(n, task.num_dof, device=dev) * 0.05
        act[::10] = (torch.rand(len(act[::10]), task.num_dof, device=dev) * 2 - 1) * 1.5
        obs, rew, done, info = env.step(act)
        if it in (0, 4):
            fresh = np.flatnonzero(chk.prog_before == 0)
            chk.dynamics(act, sorted({*spread, *fresh[:2].tolist()}), pos_atol=2e-3, root_atol=4e-3)
        out = chk.after(obs, rew, done, info)
        tot_done += int(out["reset"].sum())
        if it % 2 == 0:
            env.reset(done.nonzero(as_tuple=False).squeeze(-1))
        else:
            task.reset_done()
    assert obs.shape == (n, obs_dim) and info["amp_obs"].shape == (n, amp * 10)
    assert tot_done > 0


@pytest.mark.parametrize("tag,asset", [("h1_pdv1", "h1_humanoid"), ("g1_pdv1", "g1_humanoid")])
def test_pd_torque_held_by_the_stepper_at_4096_rows(golden, tag, asset):
    """S9 at the configuration's own size: 4096 rows -- the 6 golden rows of the reference's `_compute_torques` (pd_torques.npz) tiled, plus
    random states against the numpy restatement that the CPU suite pins to that golden bit for bit."""
    from backends import get_backend
    from phc_amd import abi
    from phc_amd.model import load_model
    from phc_amd.robots import ROBOTS, apply_robot_gains
    g = {k.split("/", 1)[1]: v for k, v in golden("pd_torques").items() if k.startswith(tag + "/")}
    rb, pd_v = tag.split("_")[0], int(tag[-1])
    be = get_backend("hip")
    m = load_model(asset)
    apply_robot_gains(m, ROBOTS[rb], pd_v)
    ints, floats = m.pack()
    keep = (be.arr(ints), be.arr(floats))
    ms = abi.model_struct(keep[0], keep[1], m.num_bodies, m.num_dof, m.max_level, len(m.contact_body))
    n, nd, nbod = 4096, m.num_dof, m.num_bodies
    rng = np.random.default_rng(5)
    reps = -(-n // g["actions"].shape[0])
    acts = np.tile(g["actions"], (reps, 1))[:n].astype(F)
    qs = np.tile(g["dof_pos"], (reps, 1))[:n].astype(F)
    qd = np.tile(g["dof_vel"], (reps, 1))[:n].astype(F)
    gold = np.tile(g["torques"], (reps, 1))[:n]
    half = n // 2                                   # second half: fresh random states (other rows of every wavefront pattern)
    acts[half:] = rng.uniform(-3, 3, (n - half, nd)).astype(F)
    qs[half:] = rng.uniform(-1, 1, (n - half, nd)).astype(F)
    qd[half:] = rng.normal(0, 3, (n - half, nd)).astype(F)
    want = po.compute_torques_pd(acts, qs, qd, g["p_gains"], g["d_gains"], g["default_dof_pos"][0], g["torque_limits"])
    np.testing.assert_array_equal(want[:half], gold[:half])
    root = np.zeros((n, 13), F)
    root[:, 2], root[:, 6] = 3.0, 1.0
    a = dict(root=be.arr(root), dof=be.arr(np.stack([qs, qd], -1)), rbs=be.zeros((n, nbod, 13)), cf=be.zeros((n, nbod, 3)), df=be.zeros((n, nd)),
             pd=be.zeros((n, nd)))
    sim = abi.sim_state_struct(n, a["root"], a["dof"], a["rbs"], a["cf"], a["df"], a["pd"])
    params = abi.sim_params_struct(sim_dt=1 / 200, substeps=1, control_freq_inv=1, control_mode=1)
    off, scale = be.arr(g["default_dof_pos"][0].astype(F)), be.arr(np.ones(nd, F))
    assert be.sim_step(ms, params, sim, be.arr(acts), off, scale, be.arr(np.zeros(nd, np.int32)), 1) == 0
    be.sync()
    np.testing.assert_allclose(be.np(a["df"]), want, rtol=2e-5, atol=2e-4)
