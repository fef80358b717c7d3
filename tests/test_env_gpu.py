"""-m gpu: the host-side task (phc_amd.env.tasks.HumanoidIm) end to end on an MI355X, through the C ABI.

Small sizes are checked against the numpy oracle driven with the task's own tensors; BASELINE.json's full size
(4096 envs) is checked through size-independent properties: determinism, env-permutation equivariance,
reset idempotence, episode bookkeeping."""
import numpy as np
import pytest
import torch

import phc_oracle as po

pytestmark = pytest.mark.gpu
F = np.float32


def make_task(num_envs, motion="synthetic:3:1", seed=0, **over):
    from phc_amd.config import compose
    from phc_amd.env.tasks.vec_task import parse_task
    torch.manual_seed(seed)
    ov = [f"env.num_envs={num_envs}", f"env.motion_file={motion}"] + [f"{k}={v}" for k, v in over.items()]
    cfg = compose(ov)
    return parse_task(cfg)


def lib_dict(task):
    ml = task._motion_lib
    d = {k: getattr(ml, k).cpu().numpy() for k in ("gts", "grs", "gvs", "gavs", "lrs", "dvs")}
    d.update(motion_lengths=ml._motion_lengths.cpu().numpy(), motion_dt=ml._motion_dt.cpu().numpy(),
             motion_num_frames=ml._motion_num_frames.cpu().numpy(), length_starts=ml.length_starts.cpu().numpy())
    return d


@pytest.mark.parametrize("n,motion,iters", [(64, "synthetic:3:1", 3), (4096, "synthetic:1:0", 6)])
def test_step_matches_oracle(n, motion, iters):
    """Full env steps at N=64 and at BASELINE configs[1]'s own size (4096 envs, single clip -- the benchmarked configuration, incl. the
    grid tail and the reset sub-lists that only exist at scale): post-physics outputs recomputed by the numpy oracle from the task's own
    post-step simulator tensors (reference order: reward/reset at t, observations at t+dt); at 4096 the stepper itself is also checked
    against the fp64 dense oracle on envs spread over the launch (first / last wavefront, freshly reset ones)."""
    import dyn_oracle as do
    task, env = make_task(n, motion=motion)
    env.reset()
    n = task.num_envs
    dt = F(task.dt)
    for it in range(iters):
        amp_before = task._amp_obs_buf.clone().cpu().numpy()
        prog_before = task.progress_buf.cpu().numpy()
        actions = (torch.rand(n, 69, device=task.device) * 2 - 1) * 0.3
        root0 = task._root_states.cpu().numpy().copy()
        dof0 = task._dof_state.view(n, task.num_dof, 2).cpu().numpy().copy()
        obs, rew, done, info = env.step(actions)
        if n >= 4096 and it in (0, iters - 1):
            tgt = (task._pd_action_offset + task._pd_action_scale * actions).cpu().numpy()
            tgt[:, task._freeze_mask.cpu().numpy() != 0] = 0
            fresh = np.flatnonzero(prog_before == 0)
            envs = sorted({0, 1, 63, n // 2 + 5, n - 64, n - 1, *fresh[:2].tolist()})
            assert task._sim_params.inertia_lag == 1      # (round 6 default: the reference of the lagged scheme is the fp64 build of the recursion, tests/step_oracle.py)
            import hostemu_util as hu
            ref = hu.sim_step_f64(task.model, task._sim_params, root0[envs], dof0[envs], tgt[envs], task.control_freq_inv)
            for k, e in enumerate(envs):
                np.testing.assert_allclose(task._rigid_body_pos[e].cpu().numpy(), ref["rbs"][k][:, 0:3], atol=1e-3, err_msg=f"env {e}")
                np.testing.assert_allclose(task._root_states[e].cpu().numpy(), ref["root"][k], atol=2e-3, rtol=1e-3, err_msg=f"env {e}")
            if it == 0:   # ... and the every-sub-step-fresh scheme at this size against the dense oracle, through the C ABI on a copy of the pre-step state
                from phc_amd import abi
                a = {k_: torch.from_numpy(np.ascontiguousarray(v)).to(task.device) for k_, v in dict(root=root0[envs], dof=dof0[envs], pd=tgt[envs].astype(F)).items()}
                z = lambda *sh: torch.zeros(*sh, device=task.device)
                ne, nb_, nd_ = len(envs), task.num_bodies, task.num_dof
                rbs_d, cf_d, df_d = z(ne, nb_, 13), z(ne, nb_, 3), z(ne, nd_)
                sim = abi.sim_state_struct(ne, a["root"], a["dof"], rbs_d, cf_d, df_d, a["pd"])
                prm = abi.sim_params_struct(self_collision=int(task._sim_params.self_collision), inertia_lag=0)
                assert task._lib.phc_sim_step(task._model_struct, prm, sim, None, None, None, None, task.control_freq_inv, torch.cuda.current_stream().cuda_stream) == 0
                torch.cuda.synchronize()
                for k, e in enumerate(envs):
                    r, d, rbs, tau, fc = do.sim_step(task.model, root0[e], dof0[e], tgt[e], params=dict(self_collision=int(task._sim_params.self_collision)),
                                                     sim_dt=task.sim_dt, substeps=2, num_sim_calls=task.control_freq_inv)
                    np.testing.assert_allclose(rbs_d[k, :, 0:3].cpu().numpy(), rbs[:, 0:3], atol=1e-3, err_msg=f"env {e} (fresh scheme vs dense oracle)")
                    np.testing.assert_allclose(a["root"][k].cpu().numpy(), r, atol=2e-3, rtol=1e-3, err_msg=f"env {e} (fresh scheme vs dense oracle)")
        torch.cuda.synchronize()
        lib = lib_dict(task)
        prog = task.progress_buf.cpu().numpy()
        np.testing.assert_array_equal(prog, prog_before + 1)
        st = task._motion_start_times.cpu().numpy()
        mids = task._sampled_motion_ids.cpu().numpy()
        t0 = (prog.astype(F) * dt + st + F(0)).astype(F)
        t1 = ((prog + 1).astype(F) * dt + st + F(0)).astype(F)
        goff = task._global_offset.cpu().numpy()
        r0 = po.get_motion_state(lib, mids, t0, goff)
        r1 = po.get_motion_state(lib, mids, t1, goff)
        bp, br = task._rigid_body_pos.cpu().numpy(), task._rigid_body_rot.cpu().numpy()
        bv, bav = task._rigid_body_vel.cpu().numpy(), task._rigid_body_ang_vel.cpu().numpy()
        assert np.isfinite(bp).all() and np.isfinite(bv).all()
        rw, raw = po.compute_imitation_reward(bp, br, bv, bav, r0["rg_pos"], r0["rb_rot"], r0["body_vel"], r0["body_ang_vel"])
        pr = po.power_reward(task.dof_force_tensor.cpu().numpy(), task._dof_vel.cpu().numpy(), prog)
        np.testing.assert_allclose(info["reward_raw"].cpu().numpy()[:, :4], raw, atol=1e-4)
        np.testing.assert_allclose(rew.cpu().numpy(), rw + pr, atol=1e-4, rtol=1e-4)
        rid = task._reset_bodies_id.cpu().numpy()
        td = np.broadcast_to(task._termination_distances.cpu().numpy()[rid], (n, len(rid)))
        reset, term = po.compute_humanoid_im_reset(prog, bp[:, rid], r0["rg_pos"][:, rid], t0 >= lib["motion_lengths"][mids], td)
        np.testing.assert_array_equal(done.cpu().numpy(), reset)
        np.testing.assert_array_equal(info["terminate"].cpu().numpy(), term)
        so = po.compute_humanoid_observations_smpl_max(bp, br, bv, bav)
        to = po.compute_imitation_observations_v6(bp[:, 0], br[:, 0], bp, br, bv, bav, r1["rg_pos"], r1["rb_rot"], r1["body_vel"], r1["body_ang_vel"])
        np.testing.assert_allclose(obs.cpu().numpy(), np.concatenate([so, to], -1), atol=1e-4)
        kid = task._key_body_ids.cpu().numpy()
        amp = po.build_amp_observations_smpl(bp[:, 0], br[:, 0], bv[:, 0], bav[:, 0], task._dof_pos.cpu().numpy(), task._dof_vel.cpu().numpy(),
                                             bp[:, kid], task.dof_subset.numpy())
        a = info["amp_obs"].cpu().numpy().reshape(n, 10, 196)
        np.testing.assert_allclose(a[:, 0], amp, atol=1e-4)
        np.testing.assert_array_equal(a[:, 1:], amp_before[:, :-1])
        if it % 2 == 0:
            env.reset(done.nonzero(as_tuple=False).squeeze(-1))     # the reference's call
        else:
            task.reset_done()                                        # the rollout loop's device-side form (16 sub-lists at scale)
    assert obs.shape == (n, 934) and info["amp_obs"].shape == (n, 1960)


def test_body_state_is_fk_of_joint_state():
    """S7: published rigid-body poses == poselib-style FK of (root, exp-map joints) (oracle fp64), <= 1e-4."""
    import dyn_oracle as do
    task, env = make_task(16)
    env.reset()
    for _ in range(5):
        env.step(torch.zeros(16, 69, device=task.device))
    torch.cuda.synchronize()
    root = task._root_states.cpu().numpy()
    dof = task._dof_state.view(16, 69, 2).cpu().numpy()
    bp, br = task._rigid_body_pos.cpu().numpy(), task._rigid_body_rot.cpu().numpy()
    for e in range(16):
        st = do.State(root[e], dof[e])
        Q, R, p = do.kinematics(task.model, st)
        np.testing.assert_allclose(bp[e], p, atol=1e-4)
        np.testing.assert_allclose(np.abs((br[e] * np.array(Q)).sum(-1)), 1, atol=1e-5)


@pytest.mark.parametrize("n", [4096])
def test_full_size_properties(n):
    """BASELINE configs[1] size.  Determinism, env-permutation equivariance (bitwise), sane episode statistics."""
    task, env = make_task(n, motion="synthetic:1:0")
    dev = task.device
    torch.manual_seed(1)
    env.reset()
    perm = torch.randperm(n, device=dev)

    def snapshot():
        return {k: getattr(task, k).clone() for k in ("_root_states", "_dof_state", "_rigid_body_state", "_pd_target", "progress_buf",
                                                      "reset_buf", "_motion_start_times", "obs_buf", "rew_buf", "_terminate_buf")}

    s0 = snapshot()
    amp0 = task._amp_obs_buf.clone()
    actions = (torch.rand(n, 69, device=dev) * 2 - 1) * 0.2
    env.step(actions)
    torch.cuda.synchronize()
    s1 = snapshot()
    amp1 = task._amp_obs_buf.clone()
    assert all(torch.isfinite(v).all() for v in s1.values() if v.is_floating_point())
    # determinism: restore, step again, bitwise identical
    for k, v in s0.items():
        getattr(task, k).copy_(v)
    task._amp_obs_buf.copy_(amp0)
    env.step(actions)
    torch.cuda.synchronize()
    for k, v in s1.items():
        assert torch.equal(getattr(task, k), v), k
    assert torch.equal(task._amp_obs_buf, amp1)
    # permutation equivariance: envs are independent (one clip per env, no cross-env term)
    N, NB, D = n, task.num_bodies, task.num_dof
    for k, v in s0.items():
        t = getattr(task, k)
        if k == "_dof_state":
            t.view(N, D, 2).copy_(v.view(N, D, 2)[perm])
        elif k == "_rigid_body_state":
            t.view(N, NB, 13).copy_(v.view(N, NB, 13)[perm])
        else:
            t.copy_(v[perm])
    task._amp_obs_buf.copy_(amp0[perm])
    # every env uses clip 0 here, but per-env motion copies differ by their random heading -> permute the ids too
    task._sampled_motion_ids.copy_(perm)
    env.step(actions[perm])
    torch.cuda.synchronize()
    assert torch.equal(task.obs_buf, s1["obs_buf"][perm])
    assert torch.equal(task.rew_buf, s1["rew_buf"][perm])
    assert torch.equal(task.reset_buf, s1["reset_buf"][perm])
    assert torch.equal(task._root_states, s1["_root_states"][perm])
    assert torch.equal(task._amp_obs_buf, amp1[perm])
    task._sampled_motion_ids.copy_(torch.arange(n, device=dev))


def test_rollout_episode_bookkeeping_and_reset_done():
    """A 40-step rollout with resets: progress counts, terminated envs restart on the reference motion
    (zero tracking error right after reset), masked reset == indexed reset."""
    task, env = make_task(256, motion="synthetic:4:2:1.5")
    env.reset()
    n = task.num_envs
    ends = 0
    for it in range(40):
        obs, rew, done, info = env.step((torch.rand(n, 69, device=task.device) * 2 - 1))
        ids = done.nonzero(as_tuple=False).squeeze(-1)
        ends += len(ids)
        if it % 2 == 0:
            env.reset(ids)
        else:
            task.reset_done()
        torch.cuda.synchronize()
        if len(ids):
            assert (task.progress_buf[ids] == 0).all()
            if it % 2 == 0:
                assert (task.reset_buf[ids] == 0).all()      # reset(env_ids) clears the flags; reset_done() leaves them to the next step
            t = task._motion_start_times[ids]
            lens = task._motion_lib._motion_lengths[task._sampled_motion_ids[ids]]
            assert (t >= 0).all() and (t < lens).all() and torch.allclose(t * 30, torch.round(t * 30), atol=1e-3)   # sample_time_interval grid
            if len(ids) > 8:
                assert t.unique().numel() > 2, "in-kernel phase draw must vary across envs"
            res = task._motion_lib.get_motion_state(task._sampled_motion_ids[ids], t)
            assert torch.allclose(task._rigid_body_pos[ids], res["rg_pos"], atol=1e-5)
            assert torch.allclose(task._dof_pos[ids], res["dof_pos"], atol=1e-5)
        assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert ends > 0, "random actions must terminate some episodes within 40 steps"
    assert (task.progress_buf <= 40).all()


def test_fetch_amp_obs_demo_and_motion_state_api():
    task, env = make_task(32)
    demo = env.fetch_amp_obs_demo(64)
    assert demo.shape == (64, 1960) and torch.isfinite(demo).all()
    lib = lib_dict(task)
    ids = torch.randint(0, 32, (50,), device=task.device)
    times = torch.rand(50, device=task.device) * task._motion_lib._motion_lengths[ids]
    res = task._motion_lib.get_motion_state(ids, times)
    want = po.get_motion_state(lib, ids.cpu().numpy(), times.cpu().numpy())
    for k in ("root_pos", "root_rot", "dof_pos", "dof_vel", "rg_pos", "rb_rot", "body_vel", "body_ang_vel"):
        np.testing.assert_allclose(res[k].cpu().numpy(), want[k], atol=2e-5, err_msg=k)
    assert set(res) >= {"root_pos", "root_rot", "dof_pos", "root_vel", "root_ang_vel", "dof_vel", "motion_aa", "rg_pos", "rb_rot",
                        "body_vel", "body_ang_vel", "motion_bodies", "motion_limb_weights"}


def test_gae_kernel_vs_oracle():
    from phc_amd import _lib
    lib = _lib.load()
    T, n = 32, 513
    g = torch.Generator(device="cpu").manual_seed(0)
    fd = (torch.rand(T, n, generator=g) < 0.1).float().cuda()
    v, r, nv = (torch.randn(T, n, generator=g).cuda() for _ in range(3))
    adv = torch.empty(T, n, device="cuda")
    rc = lib.phc_gae(T, n, fd.data_ptr(), v.data_ptr(), r.data_ptr(), nv.data_ptr(), 0.99, 0.95, adv.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    want = po.discount_values(fd.cpu().numpy(), v.cpu().numpy(), r.cpu().numpy(), nv.cpu().numpy())
    np.testing.assert_allclose(adv.cpu().numpy(), want, atol=1e-4, rtol=1e-4)


def test_fk_kernel_vs_oracle(golden):
    """M5 on device: phc_fk == poselib FK (oracle fp64) within 1e-4."""
    from backends import get_backend, model_on
    be = get_backend("hip")
    model, mstruct, keep = model_on(be)
    sk = golden("skeleton_smpl")
    c = golden("motion_clips")
    k = c["keys"][1]
    g = c[f"{k}/pose_quat_global"]
    trans = c[f"{k}/root_trans_offset"]
    lrs, gts = po.poselib_fk_from_global(sk["parent_indices"], sk["local_translation"], g, trans)
    T = g.shape[0]
    lr, rt = be.arr(lrs.astype(F)), be.arr(trans.astype(F))
    grot, gpos = be.zeros((T, 24, 4)), be.zeros((T, 24, 3))
    assert be.lib.phc_fk(mstruct, T, lr.data_ptr(), rt.data_ptr(), grot.data_ptr(), gpos.data_ptr(), be._s()) == 0
    be.sync()
    np.testing.assert_allclose(be.np(gpos), gts, atol=1e-4)


def test_ppo_train_epoch_on_device():
    """P4-P9 on the device: two full train_epochs (rollout with policy inference, disc rewards, phc_gae, bf16 MFMA
    GEMMs, Adam) at 256 envs; finite losses, parameters move, statistics update."""
    from phc_amd.learning.amp_agent import IMAmpAgent
    task, env = make_task(256, motion="synthetic:2:3", **{"learning.params.config.minibatch_size": 2048,
                                                         "learning.params.config.amp_obs_demo_buffer_size": 4096,
                                                         "learning.params.config.amp_replay_buffer_size": 4096})
    agent = IMAmpAgent(env, task.cfg)
    agent.init_train()
    w0 = agent.model.a2c_network.mu.weight.detach().clone()
    for _ in range(2):
        info = agent.train_epoch()
        assert np.isfinite([info["actor_loss"], info["critic_loss"], info["disc_loss"], info["kl"], info["mean_task_reward"]]).all(), info
    assert not torch.equal(w0, agent.model.a2c_network.mu.weight)
    assert agent.batch_size == 256 * 32 and agent.num_minibatches == 4
    assert info["total_fps"] > 0


def test_update_graph_equals_eager_launches():
    """The captured optimizer step (hipGraphs, replayed per minibatch with the row-index buffer and the device-side Adam step
    count) trains like the eager launch sequence: same parameters and normaliser statistics after three epochs.  Runs in a child
    process (tests/graph_equivalence_main.py): a failed stream capture takes the process down on this ROCm stack."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "graph_equivalence_main.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["graph_used"] == [True, False] and out["steps"][0] == out["steps"][1] == out["expected_steps"]
    assert out["count_equal"] and out["mean_maxdiff"] < 1e-9 and out["var_relmaxdiff"] < 1e-6
    assert out["param_maxdiff"] < 2e-4 and out["param_maxdiff"] < 0.05 * out["param_update_size"], out
    for k, (a, b) in out["info"].items():
        assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (k, a, b)
    # round 4: the discriminator pass is its own graph on a second stream (three linear graphs per step); the eager steps fork the same way;
    # `branch_streams=False` is the one-graph, one-stream update of round 3 -- same kernels on the same operands
    assert out["three_graphs"] == [True, False] and out["two_streams"] == [True, True, False]
    assert out["count_equal_one_stream"] and out["param_maxdiff_one_stream"] < 2e-4 and out["param_maxdiff_one_stream"] < 0.05 * out["param_update_size"], out


@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_two_rank_update_on_one_gpu(mode):
    """The learner's multi-rank path ON THE DEVICE (tests/two_rank_gpu_main.py: two gloo ranks sharing cuda:0, different seeds and
    clips per rank): initial broadcast, one flat-gradient all-reduce per optimizer step between the (optionally captured) forward /
    backward and the optimizer kernels, normaliser sync -- the replicas stay bit-identical."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "two_rank_gpu_main.py")] + (["graph"] if mode == "graph" else []),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert out["exitcodes"] == [0, 0] and out["same_params"] and out["same_stats"] and out["finite"] and out["graph"] == (mode == "graph"), out
    assert out["collectives"] == out["expected_collectives"]


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_split_gradient_allreduce_keeps_replicas_identical(mode, backend):
    """`+learning.params.config.split_allreduce=True` (round 6): the bucket cut at the policy / discriminator boundary, the discriminator's all-reduce issued on its
    stream right behind its pass -- two gloo ranks sharing cuda:0 (different seeds and clips) and a one-rank RCCL communicator: two collectives per optimizer step,
    replicas bit-identical, finite parameters."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    extra = ["nccl", "world=1"] if backend == "nccl" else []
    r = subprocess.run([sys.executable, os.path.join(here, "two_rank_gpu_main.py"), "split"] + extra + (["graph"] if mode == "graph" else []),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert all(c == 0 for c in out["exitcodes"]) and out["same_params"] and out["same_stats"] and out["finite"] and out["graph"] == (mode == "graph"), out
    assert out["collectives"] == out["expected_collectives"] > 0, out


@pytest.mark.parametrize("world", [1, 2])
@pytest.mark.parametrize("mode", ["eager", "graph"])
def test_rccl_gradient_allreduce_next_to_the_update(mode, world):
    """The same under the REAL backend (`nccl` = RCCL): world 2 needs two GPUs (skipped on a 1-GPU box -- the driver's scaling run is
    the multi-GPU evidence); world 1 runs everywhere: a one-rank RCCL communicator with the all-reduce forced after every optimizer
    step's forward / backward, i.e. RCCL's kernels, its watchdog thread and the captured update graph side by side on one device."""
    import json
    import os
    import subprocess
    import sys
    if world > torch.cuda.device_count():
        pytest.skip(f"needs {world} GPUs")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "two_rank_gpu_main.py"), "nccl", f"world={world}"] + (["graph"] if mode == "graph" else []),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert out["exitcodes"] == [0] * world and out["backend"] == "nccl" and out["world"] == world, out
    assert out["same_params"] and out["same_stats"] and out["finite"] and out["graph"] == (mode == "graph"), out
    assert out["collectives"] == out["expected_collectives"] > 0


def test_im_eval_sweep_and_auto_pmcp():
    """P10: evaluation sweep over all clips (5 clips, 2 envs -> 3 batches), metrics, failed keys, sampler re-weighting,
    training state restored afterwards."""
    from phc_amd.learning.amp_agent import IMAmpAgent
    from phc_amd.utils.flags import flags
    task, env = make_task(2, motion="synthetic:5:4:1.2", **{"learning.params.config.minibatch_size": 64, "env.auto_pmcp_soft": True,
                                                           "learning.params.config.amp_obs_demo_buffer_size": 512,
                                                           "learning.params.config.amp_replay_buffer_size": 512})
    agent = IMAmpAgent(env, task.cfg)
    td0 = task._termination_distances.clone()
    lib0 = task._motion_lib
    info, failed = agent.eval()
    assert 0.0 <= info["eval/success_rate"] <= 1.0 and np.isfinite(info["eval/mpjpe_all"]) and info["eval/mpjpe_all"] > 0
    assert len(failed) == round((1 - info["eval/success_rate"]) * 5)
    assert torch.equal(task._termination_distances, td0) and task._motion_lib is lib0 and not flags.im_eval and not flags.test
    if len(failed):  # soft auto-PMCP: sampling mass moved onto the failed clips (motion_lib_base.py:365-387)
        p = task._motion_lib._sampling_prob.cpu().numpy()
        keys = task._motion_lib._motion_data_keys
        assert p[[list(keys).index(k) for k in failed]].sum() > 0.99
    obs, rew, done, _ = env.step(torch.zeros(2, 69, device=task.device))
    assert torch.isfinite(obs).all()


def test_metrics_lite_procrustes():
    from phc_amd.learning.im_eval import compute_metrics_lite
    rng = np.random.default_rng(0)
    gt = rng.normal(size=(6, 24, 3))
    c, s = np.cos(0.3), np.sin(0.3)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    pred = 1.1 * gt @ R.T + np.array([0.5, -0.2, 0.1])
    m = compute_metrics_lite([pred], [gt])
    assert m["mpjpe_g"][0] > 100 and m["mpjpe_pa"][0] < 1e-6


# ---------------------------------------------------------------------------------------------------------------
# config 3: getup / MCP composer tasks (env_im_getup_mcp.yaml)
# ---------------------------------------------------------------------------------------------------------------
def _pnn_checkpoint(task, num_prim, seed=0):
    """A PNN checkpoint in the reference's key layout (what `learning=im_pnn` training saves)."""
    from phc_amd.learning.network import PNN
    torch.manual_seed(seed)
    pnn = PNN(task.num_obs, [64, 32], "silu", task.num_dof, num_prim)
    model = {f"a2c_network.pnn.{k}": v.clone() for k, v in pnn.state_dict().items()}
    return {"model": model, "running_mean_std": {"running_mean": torch.zeros(task.num_obs, dtype=torch.float64),
                                                 "running_var": torch.ones(task.num_obs, dtype=torch.float64)}}, pnn


def test_getup_task_fall_states_and_recovery_gating():
    """HumanoidImGetup: fall states come out of the stepper lying on the ground; a reset mixes fall / recovery / reference
    episodes; while the recovery counter runs an env is not reset and its motion clock does not advance."""
    task, env = make_task(256, motion="synthetic:3:2", **{"env.task": "HumanoidImGetup", "env.recoveryEpisodeProb": 0.5, "env.recoverySteps": 8,
                                                         "env.fallInitProb": 0.5, "env.getup_schedule": True})
    assert type(task).__name__ == "HumanoidImGetup"
    fall = task._fall_root_states.cpu().numpy()
    assert np.isfinite(fall).all() and np.isfinite(task._fall_dof_pos.cpu().numpy()).all()
    assert (fall[:, 2] < 0.6).mean() > 0.8, "after 2.5 s of random torques from a random orientation most humanoids lie on the ground"
    assert (fall[:, 7:13] == 0).all()
    env.reset()
    rc = task._recovery_counter.cpu().numpy()
    assert ((rc == 0) | (rc == 8)).all() and 0.2 < (rc == 8).mean() < 0.9
    fall_envs = np.nonzero(rc == 8)[0]
    amp = task._amp_obs_buf.cpu().numpy()
    assert np.abs(amp[fall_envs] - amp[fall_envs][:, :1]).max() == 0, "fall episodes: AMP history = current obs repeated"
    root_h = task._rigid_body_pos[:, 0, 2].cpu().numpy()
    np.testing.assert_allclose(root_h[fall_envs], task._humanoid_root_states[fall_envs, 2].cpu().numpy(), atol=1e-6)
    prog0 = task.progress_buf.clone()
    for k in range(5):
        env.step(torch.zeros(256, 69, device=task.device))
    rc2 = task._recovery_counter.cpu().numpy()
    np.testing.assert_array_equal(rc2[fall_envs], 3)
    assert (task.progress_buf[fall_envs] == prog0[fall_envs]).all(), "progress frozen during recovery"
    assert (task.reset_buf[fall_envs] == 0).all() and (task._terminate_buf[fall_envs] == 0).all()
    for k in range(40):
        task.reset_done()
        env.step((torch.rand(256, 69, device=task.device) - 0.5) * 0.4)
    assert torch.isfinite(task.obs_buf).all() and torch.isfinite(task.rew_buf).all()
    task.update_getup_schedule(1, getup_udpate_epoch=5)
    assert task._recovery_episode_prob == 0 and task._fall_init_prob == 1
    task.update_getup_schedule(6, getup_udpate_epoch=5)
    assert task._recovery_episode_prob == 0.5 and task._fall_init_prob == 0.5


def test_mcp_getup_task_composes_primitives_and_trains():
    """HumanoidImMCPGetup (env_im_getup_mcp + learning=im_mcp): actions are mixing weights over the frozen PNN columns;
    cycle_motion keeps episodes alive past the clip end; one PPO epoch of the composer runs on the device."""
    from phc_amd.learning.amp_agent import IMAmpAgent
    task, env = make_task(128, motion="synthetic:2:5:1.5", **{"env": "env_im_getup_mcp", "learning": "im_mcp", "env.num_prim": 3,
                                                             "learning.params.config.minibatch_size": 1024,
                                                             "learning.params.config.amp_obs_demo_buffer_size": 2048,
                                                             "learning.params.config.amp_replay_buffer_size": 2048})
    assert type(task).__name__ == "HumanoidImMCPGetup" and task.num_actions == 3 and task.cycle_motion and task.zero_out_far
    ck, pnn = _pnn_checkpoint(task, 3)
    task.load_primitives(ck)
    env.reset()
    w = torch.softmax(torch.randn(128, 3, device=task.device), dim=-1)
    obs_n = torch.clamp(task.obs_buf, -5, 5)
    want = sum(w[:, k:k + 1] * pnn.to(task.device).actors[k](obs_n) for k in range(3))
    np.testing.assert_allclose(task.compose_actions(w).cpu().numpy(), want.detach().cpu().numpy(), atol=1e-5)
    # one-hot weights select one primitive
    np.testing.assert_allclose(task.compose_actions(torch.eye(3, device=task.device)[torch.ones(128, dtype=torch.long)]).cpu().numpy(),
                               pnn.actors[1](obs_n).detach().cpu().numpy(), atol=1e-5)
    lengths = task._motion_lib._motion_lengths[task._sampled_motion_ids]
    cycled = torch.zeros(128, dtype=torch.bool, device=task.device)
    for k in range(70):   # clips are ~1.5 s = 45 steps: with cycle_motion the clock wraps instead of ending the episode
        st0 = task._motion_start_times_offset.clone()
        task.reset_done()
        obs, rew, done, info = env.step(w)
        cycled |= (task._motion_start_times_offset != st0) & (task.progress_buf > 1)
        t = task.progress_buf * task.dt + task._motion_start_times + task._motion_start_times_offset
        assert (t < lengths + 1e-4).all()
    assert cycled.float().mean() > 0.02 and torch.isfinite(obs).all() and torch.isfinite(rew).all()
    assert (task._point_goal >= 0).all()
    agent = IMAmpAgent(env, task.cfg)
    agent.init_train()
    c0 = agent.model.a2c_network.composer[0].weight.detach().clone()
    info = agent.train_epoch()
    assert np.isfinite([info["actor_loss"], info["critic_loss"], info["disc_loss"]]).all(), info
    assert not torch.equal(c0, agent.model.a2c_network.composer[0].weight)
    assert agent._task_reward_w == 0 and agent._disc_reward_w == 1   # getup schedule warm-up (amp_agent.py:518-525)


# ---------------------------------------------------------------------------------------------------------------
# config 5: Unitree H1 (robot=unitree_h1 env=env_im_h1_phc sim=robot_sim control=robot_control, README.MD:311)
# ---------------------------------------------------------------------------------------------------------------
H1_OVER = {"robot": "unitree_h1", "env": "env_im_h1_phc", "sim": "robot_sim", "control": "robot_control"}


def test_h1_env_end_to_end():
    """H1 task on the device: 20 bodies / 19 revolute DoFs, 200 Hz x 4 `pd` torque control, obs 298 + 480, AMP obs 63 x 10,
    extended-body reward, MotionLibReal lookups; body state == FK of the joint state (fp64 oracle kinematics)."""
    import dyn_oracle as do
    task, env = make_task(256, motion="synthetic:3:2:2.0", **H1_OVER)
    assert task.humanoid_type == "h1" and task.num_bodies == 20 and task.num_dof == 19 and task.num_actions == 19
    assert task.num_obs == 298 + 480 and task.get_num_amp_obs() == 630 and task.control_freq_inv == 4 and abs(task.dt - 0.02) < 1e-9
    assert task.num_extend_bodies == 3 and task._motion_lib.num_ext_bodies == 3 and task._motion_lib.dofs_per_joint == 1
    obs = env.reset()
    assert obs.shape == (256, 778) and torch.isfinite(obs).all()
    # reset state == reference state: joint angles straight from the clip, body state from the clip's FK
    res = task._motion_lib.get_motion_state(task._sampled_motion_ids, task._motion_start_times)
    np.testing.assert_allclose(task._dof_pos.cpu().numpy(), res["dof_pos"].cpu().numpy(), atol=1e-6)
    np.testing.assert_allclose(task._rigid_body_pos.cpu().numpy(), res["rg_pos"].cpu().numpy(), atol=1e-5)
    assert res["rg_pos_t"].shape == (256, 23, 3) and res["rg_rot_t"].shape == (256, 23, 4)
    lo, hi = task.model.dof_limits()
    rew_sum = torch.zeros(256, device=task.device)
    n_done = 0
    for k in range(40):
        task.reset_done()
        # PD target = next reference pose (+ noise): torques = p_gains * (action + default_dof_pos - q) - d_gains * qd (humanoid.py:1590)
        act = task.ref_dof_pos - task.default_dof_pos + torch.randn(256, 19, device=task.device) * 0.05
        obs, rew, done, info = env.step(act)
        rew_sum += rew
        n_done += int(done.sum())
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(rew_sum).all() and info["amp_obs"].shape == (256, 630)
    assert (task.progress_buf.max() > 5) and rew_sum.mean() > 0.2 * 40 * 0.5, (n_done, float(rew_sum.mean()))
    assert (task._dof_pos.cpu().numpy() > lo - 0.35).all() and (task._dof_pos.cpu().numpy() < hi + 0.35).all(), "joint limits hold the joints"
    assert task.dof_force_tensor.abs().max() <= 350.0 + 2000.0 * 0.4 + 1e-3   # torque limit (+ limit spring when outside the range)
    root, dof = task._root_states.cpu().numpy(), task._dof_state.view(256, 19, 2).cpu().numpy()
    bp, br = task._rigid_body_pos.cpu().numpy(), task._rigid_body_rot.cpu().numpy()
    for e in (0, 100, 255):
        st = do.State(root[e], dof[e], task.model)
        Q, R, p = do.kinematics(task.model, st)
        np.testing.assert_allclose(bp[e], p, atol=2e-5)
        np.testing.assert_allclose(np.abs((br[e] * np.array(Q)).sum(-1)), 1.0, atol=1e-5)
    demo = task.fetch_amp_obs_demo(64)
    assert demo.shape == (64, 630) and torch.isfinite(demo).all()


def test_g1_env_end_to_end():
    """Unitree G1 (env_im_g1_phc): 38 bodies -> one env per wavefront in the task kernels and the stepper; obs 568 + 912, AMP obs 99 x 10, one extended body; body state == FK of the joint state."""
    import dyn_oracle as do
    over = {"robot": "unitree_g1", "env": "env_im_g1_phc", "sim": "robot_sim", "control": "robot_control"}
    task, env = make_task(192, motion="synthetic:3:2:2.0", **over)
    assert task.humanoid_type == "g1" and task.num_bodies == 38 and task.num_dof == 37 and task.num_actions == 37
    assert task.num_obs == 568 + 912 and task.get_num_amp_obs() == 990 and task.num_extend_bodies == 1
    obs = env.reset()
    assert obs.shape == (192, 1480) and torch.isfinite(obs).all()
    res = task._motion_lib.get_motion_state(task._sampled_motion_ids, task._motion_start_times)
    np.testing.assert_allclose(task._dof_pos.cpu().numpy(), res["dof_pos"].cpu().numpy(), atol=1e-6)
    np.testing.assert_allclose(task._rigid_body_pos.cpu().numpy(), res["rg_pos"].cpu().numpy(), atol=1e-5)
    rew_sum = torch.zeros(192, device=task.device)
    for k in range(30):
        task.reset_done()
        act = task.ref_dof_pos - task.default_dof_pos + torch.randn(192, 37, device=task.device) * 0.05
        obs, rew, done, info = env.step(act)
        rew_sum += rew
    torch.cuda.synchronize()
    assert torch.isfinite(obs).all() and torch.isfinite(rew_sum).all() and info["amp_obs"].shape == (192, 990)
    assert (task.progress_buf.max() > 5) and rew_sum.mean() > 0.2 * 30 * 0.5, float(rew_sum.mean())
    root, dof = task._root_states.cpu().numpy(), task._dof_state.view(192, 37, 2).cpu().numpy()
    bp, br = task._rigid_body_pos.cpu().numpy(), task._rigid_body_rot.cpu().numpy()
    for e in (0, 100, 191):
        st = do.State(root[e], dof[e], task.model)
        Q, R, p = do.kinematics(task.model, st)
        np.testing.assert_allclose(bp[e], p, atol=2e-5)
        np.testing.assert_allclose(np.abs((br[e] * np.array(Q)).sum(-1)), 1.0, atol=1e-5)
    demo = task.fetch_amp_obs_demo(64)
    assert demo.shape == (64, 990) and torch.isfinite(demo).all()


def test_h1_ppo_epoch():
    """README.MD:311 training line at small size: PNN actor on the H1 task, one PPO + AMP epoch on the device."""
    from phc_amd.learning.amp_agent import IMAmpAgent
    task, env = make_task(128, motion="synthetic:2:1:2.0", **dict(H1_OVER, **{"learning": "im_pnn", "learning.params.config.minibatch_size": 1024,
                                                                           "learning.params.config.amp_obs_demo_buffer_size": 2048,
                                                                           "learning.params.config.amp_replay_buffer_size": 2048,
                                                                           "learning.params.network.space.continuous.sigma_init.val": -1.7}))
    agent = IMAmpAgent(env, task.cfg)
    agent.init_train()
    w0 = agent.model.a2c_network.pnn.actors[0][0].weight.detach().clone()
    info = agent.train_epoch()
    assert np.isfinite([info["actor_loss"], info["critic_loss"], info["disc_loss"], info["mean_task_reward"]]).all(), info
    assert not torch.equal(w0, agent.model.a2c_network.pnn.actors[0][0].weight)
    assert abs(float(agent.model.a2c_network.sigma[0]) + 1.7) < 1e-6


def test_reset_done_resets_exactly_the_finished_envs():
    """reset_done() consumes the device-built list of the envs the last post-physics launch flagged: after it exactly those envs
    (and no stale list entries from earlier steps) sit at progress 0; a masked fallback serves calls with no fresh list."""
    task, env = make_task(1024, motion="synthetic:4:2:1.5")
    env.reset()
    total = 0
    for it in range(60):
        obs, rew, done, info = env.step((torch.rand(1024, 69, device=task.device) * 2 - 1))
        done_ids = set(done.nonzero(as_tuple=False).flatten().tolist())
        total += len(done_ids)
        task.reset_done()
        zero_ids = set((task.progress_buf == 0).nonzero(as_tuple=False).flatten().tolist())
        assert zero_ids == done_ids, (it, len(zero_ids), len(done_ids))
        if it == 30:   # a second call without a step in between: no fresh list -> masked sweep over the (unchanged) flags, same envs
            st = task._motion_start_times.clone()
            task.reset_done()
            assert set((task.progress_buf == 0).nonzero(as_tuple=False).flatten().tolist()) == done_ids
            assert (task._motion_start_times != st).sum() <= len(done_ids)
    assert total > 100


def test_env_vr_three_point_tracking_config():
    """env=env_vr (trackBodies = reset_bodies = Head, L_Hand, R_Hand): obs 358 + 72, episodes run, resets follow the three bodies."""
    task, env = make_task(128, motion="synthetic:2:3:2.0", **{"env": "env_vr"})
    assert task.num_obs == 358 + 72 and task._track_bodies == ["Head", "L_Hand", "R_Hand"]
    obs = env.reset()
    assert obs.shape == (128, 430) and torch.isfinite(obs).all()
    for _ in range(20):
        task.reset_done()
        obs, rew, done, info = env.step((task.ref_dof_pos - task._pd_action_offset) / task._pd_action_scale)
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and task.progress_buf.max() > 5


def test_run_entry_trains_saves_and_plays(tmp_path):
    """`python -m phc_amd.run ...` the way run_hydra.py is used: a short training run saves Humanoid.pth; `test=True epoch=-1`
    restores it and plays (deterministic policy); `test=True im_eval=True` runs the evaluation sweep."""
    from phc_amd.run import main
    from phc_amd.utils.flags import flags
    common = ["env.num_envs=64", "env.motion_file=synthetic:3:4:1.5", f"output_path={tmp_path}", "learning.params.config.minibatch_size=512",
              "learning.params.config.amp_obs_demo_buffer_size=1024", "learning.params.config.amp_replay_buffer_size=1024"]
    try:
        info = main(common + ["max_epochs=2"])
        assert np.isfinite(info["actor_loss"]) and (tmp_path / "Humanoid.pth").exists()
        played = main(common + ["test=True", "epoch=-1", "+games=40"])
        assert played["steps"] == 40 and played["mean_episode_length"] > 0 and np.isfinite(played["mean_episode_reward"])
        ev = main(common + ["test=True", "im_eval=True", "epoch=-1"])
        assert 0.0 <= ev["eval/success_rate"] <= 1.0
    finally:
        flags.test = flags.im_eval = False


def test_force_sensors_and_self_obs_v3():
    """S6 + self_obs_v 3 end to end: standing still under zero actions, the foot sensors (L_Ankle, R_Ankle) read the ground-contact
    wrench of their body in the body frame -- rotated to the world it equals the stepper's net contact force on those bodies (steady
    state) and carries a good part of the weight (the toes carry the rest); the readings are the last 12 floats of the self observation."""
    task, env = make_task(64, motion="stand:4", **{"env.self_obs_v": 3})
    assert task.get_self_obs_size() == 370 and task.num_obs == 370 + 576
    env.reset()
    task._motion_start_times[:] = 0
    for _ in range(45):
        obs, rew, done, info = env.step(torch.zeros(64, 69, device=task.device))
    torch.cuda.synchronize()
    s = task.vec_sensor_tensor.view(64, 2, 6)
    assert torch.equal(obs[:, 358:370], task.vec_sensor_tensor)
    ids = [task._body_names.index(b) for b in task.force_sensor_joints]
    rot = task._rigid_body_rot[:, ids]                                # [N, 2, 4] xyzw
    q, f = rot.reshape(-1, 4), s[..., :3].reshape(-1, 3)
    qv, qw = q[:, :3], q[:, 3:]
    t = 2 * torch.cross(qv, f, dim=-1)
    f_world = (f + qw * t + torch.cross(qv, t, dim=-1)).view(64, 2, 3)
    weight = float(task.model.mass.sum()) * 9.81
    fz = f_world[..., 2].sum(-1)
    assert (fz > 0.3 * weight).all() and (fz < 1.05 * weight).all(), (float(fz.min()), float(fz.max()), weight)
    np.testing.assert_allclose(f_world.cpu().numpy(), task._contact_forces[:, ids].cpu().numpy(), rtol=0.05, atol=0.02 * weight)
    assert torch.isfinite(s).all() and float(s[..., 3:].abs().max()) < 200.0     # torques about the ankle origin: N m scale


@pytest.mark.parametrize("world", [2, 8])
def test_bench_multi_rank_logic_ranks_sharing_one_gpu(world):
    """`python bench.py --gpus 2` end to end on a 1-GPU box: bench.py spawns two ranks itself (torch.distributed.run), here over gloo with
    both ranks sharing cuda:0 (`--backend gloo`; RCCL refuses two ranks on one device) -- barriers, max-over-ranks timing, the whole-job
    value, the PPO epochs with one gradient all-reduce per optimizer step next to the captured update, the rank table of the JSON line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    envs = 512 if world == 2 else 256
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--backend", "gloo", "--envs", str(envs), "--steps", "20", "--warmup", "5",
                        "--ppo-epochs", "2", "--no-cpu-baseline", "--no-pmc"], capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]            # rank 0 alone prints
    d = json.loads(line[0])
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["config"]["envs_per_gpu"] == envs
    assert d["value"] > 0 and abs(d["value"] - world * envs * 20 / (d["ms_per_step"] * 1e-3 * 20)) < 1e-6 * d["value"]
    assert [x["rank"] for x in d["per_rank"]] == list(range(world)) and all(x["env_steps_per_s"] > 0 for x in d["per_rank"])
    assert max(x["elapsed_s"] for x in d["per_rank"]) == pytest.approx(d["ms_per_step"] * 1e-3 * 20, rel=1e-9)      # the line's time is the slowest rank's
    comm = d["ppo_comm"]
    assert [x["rank"] for x in comm["ranks"]] == list(range(world)) and all(x["world"] == world for x in comm["ranks"])
    assert comm["replicas_identical"] is True
    n_opt = d["ppo_config"]["optimizer_steps_per_epoch"]
    assert comm["grad_allreduces_per_epoch"] == n_opt and all(x["collectives"] == 2 * n_opt for x in comm["ranks"])
    assert d["ppo_config"]["collectives_per_epoch"] == n_opt and d["ppo_samples_per_s"] > 0


def test_trained_policy_and_collective_workloads_of_the_bench_line():
    """Round 6: the two new child workloads of `bench.py` at a small size -- `bench_policy.run` (train a policy for a few seconds on the 64-clip library, then time the env step with
    its actions; the policy's inference replayed from ONE captured graph so that the loop is GPU-bound) and `bench_collective.run` (a one-rank RCCL all-reduce at the two bucket sizes),
    the second in a child process (it owns a process group)."""
    import json
    import os
    import subprocess
    import sys
    from phc_amd.learning import bench_policy
    out = bench_policy.run(envs=256, train_s=4.0, steps=24, target_episode_len=1e9)
    assert out["policy_inference_as_one_graph"] and out["envs_per_gpu"] == 256 and out["policy_train_epochs"] >= 2
    assert 0.0 <= out["resets_per_step_share"] <= 1.0 and out["env_step_us"] > 0
    assert out["env_step_us"] >= out["stepper_launch_us"] and out["stepper_launch_us"] > out["reset_launch_us"] > 0 and out["post_physics_launch_us"] > 0
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "phc_amd.learning.bench_collective"], capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    c = json.loads(next(l for l in r.stdout.splitlines() if l.startswith("COLLECTIVE_JSON"))[len("COLLECTIVE_JSON"):])
    assert c["world_size"] == 1 and set(c["allreduce"]) == {"23.2MB", "149.0MB"} and all(v["median_us"] > 0 for v in c["allreduce"].values())


def test_shape_variation_env_end_to_end():
    """robot.has_shape_variation (smpl_humanoid_shape.yaml) end to end: 3 compiled shapes (the reference's gender assets, its own fallback
    when smpl_sim cannot write per-env MJCFs, humanoid.py:748), env i wears shape i % 3.  Checks: the reference clips are run through each
    env's own skeleton (reset = reference state); the stepper keeps every env's OWN link lengths (rigid links: |knee - hip| is the shape's
    offset norm) and carries its own weight; shape / limb columns sit at the end of the self observation and of every AMP step."""
    task, env = make_task(48, motion="stand:4", **{"robot.has_shape_variation": True, "robot.has_shape_obs": True, "robot.has_shape_obs_disc": True,
                                                   "robot.has_weight_obs": True})
    N = 48
    assert task._model_struct.num_shapes == 3 and len(task.shape_models) == 3
    assert task.get_self_obs_size() == 358 + 11 + 10 and task._num_amp_obs_per_step == 196 + 11
    shape = (torch.arange(N) % 3).numpy()
    env.reset()
    task._motion_start_times[:] = 0
    names = task._body_names
    pairs = [(names.index(a), names.index(b)) for a, b in (("L_Hip", "L_Knee"), ("R_Knee", "R_Ankle"), ("L_Shoulder", "L_Elbow"), ("Chest", "Neck"))]
    want = np.array([[np.linalg.norm(task.shape_models[s].local_translation[b]) for a, b in pairs] for s in shape])
    assert np.ptp(want[:3], axis=0).min() > 2e-3          # the three shapes do differ in every one of these links

    def link_lengths():
        p = task._rigid_body_pos.cpu().numpy()
        return np.stack([np.linalg.norm(p[:, b] - p[:, a], axis=-1) for a, b in pairs], axis=1)
    np.testing.assert_allclose(link_lengths(), want, atol=2e-5)              # reference state: FK of the clip through the env's skeleton
    for it in range(40):
        obs, rew, done, info = env.step(torch.zeros(N, 69, device=task.device))
    torch.cuda.synchronize()
    np.testing.assert_allclose(link_lengths(), want, atol=1e-4)              # the stepper used the env's own offsets for 40 steps
    fz = task._contact_forces[..., 2].sum(-1).cpu().numpy()
    weight = np.array([task.shape_models[s].total_mass for s in shape]) * 9.81
    np.testing.assert_allclose(fz, weight, rtol=0.03)                        # standing: the ground carries each env's own weight
    assert abs(weight[0] - weight[1]) > 50
    hs = task.humanoid_shapes[:, :11]
    assert torch.equal(obs[:, 358:369], hs) and torch.equal(obs[:, 369:379], task.humanoid_limb_and_weights)
    amp = info["amp_obs"].view(N, 10, 207)
    assert torch.equal(amp[:, :, 196:], hs[:, None].expand(N, 10, 11))
    assert float(hs[:, 0].max()) == 2.0 and torch.isfinite(obs).all() and torch.isfinite(amp).all()
    demo = env.fetch_amp_obs_demo(64)
    assert demo.shape == (64, 2070) and torch.isfinite(demo).all()
    # random actions + resets keep everything finite and the links rigid
    for it in range(20):
        obs, rew, done, info = env.step((torch.rand(N, 69, device=task.device) * 2 - 1) * 0.5)
    np.testing.assert_allclose(link_lengths(), want, atol=2e-4)
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all()


def test_non_upright_asset_env():
    """robot.has_upright_start False (smplx_humanoid.yaml) end to end on the reference's y-up SMPL asset (mjcf/smpl_humanoid.xml): clips whose
    root rests at (0.5, 0.5, 0.5, 0.5); the observations strip that base rotation, so a standing, forward-facing humanoid sees the same
    heading-local picture as the upright asset does (root rotation observation ~ identity tan-norm), and PD offsets follow
    humanoid.py:1398."""
    task, env = make_task(32, motion="synthetic:3:1", **{"robot.has_upright_start": False, "model_asset": "smpl_yup_humanoid"})
    assert task._im_params.remove_base_rot == 1
    env.reset()
    obs = task.obs_buf.clone()
    root_rot = task._rigid_body_rot[:, 0]
    base_c = torch.tensor([-0.5, -0.5, -0.5, 0.5], device=task.device).expand(32, 4)
    stripped = torch.from_numpy(po.quat_mul(root_rot.cpu().numpy(), base_c.cpu().numpy())).to(task.device)
    # heading-local root rotation observation (local_root_obs True): tan-norm of hinv * body_rot[0]; with the base rotation stripped from the
    # heading the recomputation below must agree with the kernel's columns
    hinv = po.calc_heading_quat_inv(stripped.cpu().numpy())
    want = po.quat_to_tan_norm(po.quat_mul(hinv, root_rot.cpu().numpy()))
    off = 1 + 23 * 3
    np.testing.assert_allclose(obs[:, off:off + 6].cpu().numpy(), want, atol=1e-5)
    # the stripped root is near upright (small tilt): its z axis points up
    zc = po.my_quat_rotate(stripped.cpu().numpy(), np.tile(np.array([[0, 0, 1.0]], F), (32, 1)))[:, 2]
    assert (zc > 0.9).all()
    for it in range(10):
        obs, rew, done, info = env.step((torch.rand(32, 69, device=task.device) * 2 - 1) * 0.3)
    assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and torch.isfinite(info["amp_obs"]).all()


def test_k_padded_first_layers_keep_shapes_checkpoints_and_zero_pad():
    """K-padded storage of the first-layer weights (FlatGradBucket; obs 934 -> 1024, AMP obs 1960 -> 2048 columns for the GEMMs): the
    module parameters, the state dict and the per-parameter optimizer state keep the reference's shapes; the pad elements of the
    parameter, of its gradient and of the Adam moments stay exactly zero through training; a checkpoint written by the padded agent
    loads into an un-padded one (PHC_NO_K_PAD) and back, giving the same actions."""
    import os
    from phc_amd.learning.amp_agent import IMAmpAgent
    over = {"learning.params.config.minibatch_size": 2048, "learning.params.config.amp_obs_demo_buffer_size": 4096,
            "learning.params.config.amp_replay_buffer_size": 4096}
    task, env = make_task(256, motion="synthetic:2:3", **over)
    agent = IMAmpAgent(env, task.cfg)
    net = agent.model.a2c_network
    w_a, w_c, w_d = net.actor_mlp[0].weight, net.critic_mlp[0].weight, net._disc_mlp[0].weight
    assert tuple(w_a.shape) == (1024, 934) and w_a._padded.shape == (1024, 1024) and w_a.stride() == (1024, 1)
    assert tuple(w_d.shape) == (1024, 1960) and w_d._padded.shape == (1024, 2048) and agent._obs_pad_cols == 1024 and agent._amp_pad_cols == 2048
    agent.init_train()
    for _ in range(2):
        info = agent.train_epoch()
    assert np.isfinite([info["actor_loss"], info["disc_loss"]]).all()
    for w in (w_a, w_c, w_d):
        k = w.shape[1]
        assert float(w._padded[:, k:].abs().max()) == 0.0 and float(w._grad_padded[:, k:].abs().max()) == 0.0
        assert float(w._bf16_shadow[:, k:].abs().max()) == 0.0
    ck = agent.get_full_state_weights()
    sd = ck["model"]
    assert tuple(sd["a2c_network.actor_mlp.0.weight"].shape) == (1024, 934) and tuple(sd["a2c_network._disc_mlp.0.weight"].shape) == (1024, 1960)
    names = [n for n, _ in agent.model.named_parameters()]
    i_a = names.index("a2c_network.actor_mlp.0.weight")
    st = ck["optimizer"]["state"][i_a]
    assert tuple(st["exp_avg"].shape) == (1024, 934) and float(st["exp_avg"].abs().max()) > 0
    obs = torch.randn(64, task.num_obs, device=task.device)
    agent.set_eval()
    a0 = agent.get_action_values(obs)["mus"].float().clone()
    # the same checkpoint in an agent without padding, and back
    os.environ["PHC_NO_K_PAD"] = "1"
    try:
        task2, env2 = make_task(256, motion="synthetic:2:3", **over)
        plain = IMAmpAgent(env2, task2.cfg)
    finally:
        del os.environ["PHC_NO_K_PAD"]
    assert getattr(plain.model.a2c_network.actor_mlp[0].weight, "_padded", None) is None
    plain.set_full_state_weights(ck)
    plain.set_eval()
    a1 = plain.get_action_values(obs)["mus"].float()
    np.testing.assert_allclose(a1.cpu().numpy(), a0.cpu().numpy(), atol=2e-2, rtol=2e-2)     # (bf16 / fp32 inference paths)
    assert torch.equal(plain.model.a2c_network.actor_mlp[0].weight, w_a.detach())
    back = plain.get_full_state_weights()
    agent.set_full_state_weights(back)
    assert torch.equal(net.actor_mlp[0].weight, plain.model.a2c_network.actor_mlp[0].weight) and float(w_a._padded[:, 934:].abs().max()) == 0.0
    st2 = agent.optimizer.state_dict()["state"][0]
    np.testing.assert_array_equal(agent.grads.param_view(st2["exp_avg"], agent.grads.params.index(w_a)).cpu().numpy(), st["exp_avg"].cpu().numpy())


def test_self_obs_v2_env_keeps_a_body_state_history():
    """env.self_obs_v = 2 end to end: observation = 6 x 358 self columns + task block; after a reset all six blocks agree (history = the reset
    state); stepping makes the newest block the current state's v1 block and moves the previous one back by one slot each step."""
    task, env = make_task(64, motion="synthetic:3:1", **{"env.self_obs_v": 2})
    assert task.get_self_obs_size() == 6 * 358 and task.num_obs == 6 * 358 + 576
    env.reset()
    blocks = task.obs_buf[:, :6 * 358].view(64, 6, 358)
    for k in range(5):
        assert torch.equal(blocks[:, k], blocks[:, 5])
    hist0 = task._body_state_hist.clone()
    assert torch.equal(hist0[:, 0], task._rigid_body_state.view(64, 24, 13)) and torch.equal(hist0[:, 4], hist0[:, 0])
    state0 = task._rigid_body_state.view(64, 24, 13).clone()
    task.progress_buf[:] = 0
    task._motion_start_times[:] = 0.2
    obs, rew, done, info = env.step(torch.zeros(64, 69, device=task.device))
    state1 = task._rigid_body_state.view(64, 24, 13).clone()
    keep = done == 0
    assert keep.sum() > 32
    h = task._body_state_hist
    assert torch.equal(h[keep][:, 4], state1[keep]) and torch.equal(h[keep][:, 3], state0[keep]) and torch.equal(h[keep][:, 0], state0[keep])
    # the newest block is what self_obs_v 1 computes from the same state (same root / heading)
    task1, env1 = make_task(64, motion="synthetic:3:1")
    task1._rigid_body_state.copy_(task._rigid_body_state)
    task1._root_states.copy_(task._root_states); task1._dof_state.copy_(task._dof_state)
    task1.progress_buf.copy_(task.progress_buf - 1); task1._motion_start_times.copy_(task._motion_start_times)
    task1._sampled_motion_ids.copy_(task._sampled_motion_ids)
    task1.post_physics_step()
    torch.cuda.synchronize()
    got = obs[:, 5 * 358:6 * 358]
    np.testing.assert_allclose(got.cpu().numpy(), task1.obs_buf[:, :358].cpu().numpy(), atol=1e-6)
    assert torch.isfinite(obs).all()


@pytest.mark.parametrize("over", [
    {"learning": "im_pnn", "env": "env_im_pnn"},                                                         # PNN columns (frozen + training)
    {"learning": "im_big"},                                                                              # the six-layer SiLU networks (2048-1536-1024-1024-512-512)
    {"learning": "im_pnn_big", "env": "env_im_pnn"},                                                     # ... with the PNN actor: the reference's flagship learner
    {"robot": "smpl_humanoid_shape"},                                                                    # per-env shapes + shape columns in obs / AMP obs
    {"env.self_obs_v": 2, "env.obs_v": 8},                                                               # history self obs + v8 task obs (wide inputs)
])
def test_captured_update_trains_the_other_network_and_observation_variants(over):
    """The captured update (optimizer inside the graph on one rank, K-padded first layers, row-restricted gradient penalty) on the variants the
    bench line does not exercise: three epochs, graph active, finite losses, parameters move, pad columns stay zero."""
    from phc_amd.learning.amp_agent import IMAmpAgent
    groups = [f"{k}={v}" for k, v in over.items() if "." not in k]
    from phc_amd.config import compose
    from phc_amd.env.tasks.vec_task import parse_task
    torch.manual_seed(0)
    cfg = compose(groups + ["env.num_envs=256", "env.motion_file=synthetic:2:3", "learning.params.config.minibatch_size=2048",
                            "learning.params.config.amp_minibatch_size=1024", "learning.params.config.amp_obs_demo_buffer_size=4096",
                            "learning.params.config.amp_replay_buffer_size=4096", "+learning.params.config.hip_graph=True"]
                  + [f"{k}={v}" for k, v in over.items() if "." in k])
    task, env = parse_task(cfg)
    agent = IMAmpAgent(env, cfg)
    agent.init_train()
    p0 = agent.grads.flat_param.clone()
    for _ in range(3):
        info = agent.train_epoch()
        assert np.isfinite([info["actor_loss"], info["critic_loss"], info["disc_loss"], info["kl"]]).all(), info
    assert agent._graph is not None and not torch.equal(p0, agent.grads.flat_param)
    #  the default form: THREE linear graphs per step -- policy pass, discriminator pass captured on its own stream, tail -- not one graph
    assert isinstance(agent._graph, tuple) and len(agent._graph) == 3 and agent._branches is not None
    st = agent.optimizer.state[agent.grads.flat_param]
    assert int(st["step"]) == 3 * agent.mini_epochs_num * agent.num_minibatches
    for p in agent.grads.params:
        if getattr(p, "_padded", None) is not None:
            assert float(p._padded[:, p.shape[1]:].abs().max()) == 0.0


@pytest.mark.parametrize("obs_v", [6, 9])
def test_fut_tracks_env_samples_future_reference_frames(obs_v):
    """env.fut_tracks with numTrajSamples = 3 end to end (humanoid_im.py:39-47,741-747): the task block is three standard blocks; block 0 is the
    plain env's block (reset and step paths), block k the plain env's block with the clip start moved k * (1 / trajSampleTimestepInv) s on."""
    N = 64
    over = {"env.obs_v": obs_v}
    task, env = make_task(N, motion="synthetic:3:1", **over, **{"env.fut_tracks": True, "env.numTrajSamples": 3})
    task1, env1 = make_task(N, motion="synthetic:3:1", **over)
    blk = task1.get_task_obs_size()
    so = task.get_self_obs_size()
    assert task.get_task_obs_size() == 3 * blk and task.num_obs == so + 3 * blk
    assert task.get_task_obs_size_detail()["num_traj_samples"] == 3
    def same_rng(fn, *a):   # both envs draw their reset samples from the same generator state
        torch.manual_seed(11)
        return fn(*a)
    o, o1 = same_rng(env.reset), same_rng(env1.reset)
    assert torch.equal(task._sampled_motion_ids, task1._sampled_motion_ids) and torch.equal(task._motion_start_times, task1._motion_start_times)
    assert torch.equal(o[:, :so + blk], o1) and torch.isfinite(o).all()
    assert (o[:, so + blk:so + 2 * blk] - o[:, so:so + blk]).abs().max() > 1e-3       # a different reference frame
    act = torch.zeros(N, task.get_action_size(), device=task.device)
    for _ in range(2):
        o, _, done, _ = same_rng(env.step, act)
        o1, _, done1, _ = same_rng(env1.step, act)
    assert torch.equal(done, done1) and torch.equal(o[:, :so + blk], o1)
    keep = (done == 0).cpu().numpy()
    assert keep.sum() >= 8
    # block k == the plain env's block with the clip start k sample intervals later, on the same simulator state
    ts = task._traj_sample_timestep
    assert ts == 1 / task.cfg["env"]["trajSampleTimestepInv"]
    for k in (1, 2):
        task1.progress_buf.copy_(task.progress_buf - 1)
        task1._motion_start_times.copy_(task._motion_start_times + k * ts)
        task1.post_physics_step()
        torch.cuda.synchronize()
        np.testing.assert_allclose(o[:, so + k * blk:so + (k + 1) * blk].cpu().numpy()[keep], task1.obs_buf[:, so:].cpu().numpy()[keep], atol=1e-3)   # the sample time rounds differently (sum order)
        task1._motion_start_times.copy_(task._motion_start_times)


def test_res_action_targets_are_residuals_on_the_reference_pose():
    """env.res_action (humanoid_im.py:1094-1099): pd_tar = ref_dof_pos + scale * action, limited to the current joint position +- pi / 2, where
    ref_dof_pos is the reference pose the last observation was computed against; bit-exact against the same torch expression."""
    N = 64
    task, env = make_task(N, motion="synthetic:3:1", **{"+env.res_action": True})
    env.reset()
    ref = task.ref_dof_pos.clone()
    q = task._dof_state.view(N, -1, 2)[..., 0].clone()
    assert ref.abs().max() > 0.05
    torch.manual_seed(3)
    a = 4 * torch.randn(N, task.get_action_size(), device=task.device)     # large enough for the +- pi / 2 window to bind
    env.step(a)
    want = torch.maximum(torch.minimum(ref + task._pd_action_scale * a, q + np.pi / 2), q - np.pi / 2)
    assert torch.equal(task._pd_target, want)
    bound = (want == q + np.pi / 2) | (want == q - np.pi / 2)
    assert 0.02 < bound.float().mean() < 0.98
    # the plain map on the same state for contrast
    task0, env0 = make_task(N, motion="synthetic:3:1")
    env0.reset()
    env0.step(a)
    assert torch.equal(task0._pd_target, task0._pd_action_offset + task0._pd_action_scale * a)


def test_occl_training_env_masks_the_lower_body_reference():
    """env.occl_training end to end: after the first step the mask the reference ends up with (bodies 0..8, humanoid_im.py:1091-1092) is in force:
    the position / velocity differences of those bodies in the v6 task block are exactly zero, the other bodies' columns are the plain env's."""
    N, J = 64, 24
    task, env = make_task(N, motion="synthetic:3:1", **{"+env.occl_training": True})
    task0, env0 = make_task(N, motion="synthetic:3:1")
    so = task.get_self_obs_size()
    torch.manual_seed(11)
    o = env.reset()
    torch.manual_seed(11)
    o0 = env0.reset()
    assert torch.equal(o, o0)                                     # the mask starts empty (:96)
    act = torch.zeros(N, task.get_action_size(), device=task.device)
    o, _, done, _ = env.step(act)
    o0, _, done0, _ = env0.step(act)
    assert task._occl_mask[:, :9].all() and not task._occl_mask[:, 9:].any()
    keep = (done == 0) & (done0 == 0)
    assert keep.sum() > N // 2
    t, t0 = o[keep][:, so:], o0[keep][:, so:]
    # v6 layout: dpos 3J | drot 6J | dvel 3J | dangvel 3J | local ref pos 3J | local ref rot 6J
    assert (t[:, :27] == 0).all() and (t[:, 9 * J:9 * J + 27] == 0).all() and (t0[:, :27].abs().max() > 1e-4)
    assert torch.equal(t[:, 27:3 * J], t0[:, 27:3 * J]) and torch.equal(t[:, 15 * J + 27:18 * J], t0[:, 15 * J + 27:18 * J])
    assert torch.equal(task.rew_buf, task0.rew_buf)               # the reward sees the true reference


def test_add_obs_noise_perturbs_every_fresh_observation_row_once():
    """env.add_obs_noise (humanoid_im.py:710-711): the observation is the clean one + N(0, 0.1) per element -- on the step's rows and on reset
    rows alike -- and nothing is added in test mode."""
    from phc_amd.utils.flags import flags
    N = 256
    task, env = make_task(N, motion="synthetic:3:1", **{"env.add_obs_noise": True})
    task0, env0 = make_task(N, motion="synthetic:3:1")
    torch.manual_seed(11)
    o = env.reset()
    torch.manual_seed(11)
    o0 = env0.reset()
    d = (o - o0).flatten()
    assert abs(float(d.std()) - 0.1) < 0.003 and abs(float(d.mean())) < 0.002      # reset(env_ids) rows
    act = torch.zeros(N, task.get_action_size(), device=task.device)
    task.step(act); task0.step(act)
    d = (task.obs_buf - task0.obs_buf).flatten()
    assert abs(float(d.std()) - 0.1) < 0.003                                          # the step's rows
    done = task.reset_buf != 0
    assert torch.equal(task.reset_buf, task0.reset_buf) and 0 < int(done.sum()) < N
    before = task.obs_buf.clone()
    task.reset_done()
    assert torch.equal(task.obs_buf[~done], before[~done])                           # only the reset rows change ...
    task0.reset_done()
    d = (task.obs_buf - task0.obs_buf)[done]                                         # ... (same hash-drawn start phases in both envs)
    assert abs(float(d.std()) - 0.1) < 0.01
    flags.test = True
    try:
        torch.manual_seed(5)
        o = env.reset()
        torch.manual_seed(5)
        o0 = env0.reset()
        assert torch.equal(o, o0)
    finally:
        flags.test = False


def test_fut_tracks_dropout_zeroes_a_tenth_of_the_reference_samples():
    """env.fut_tracks_dropout (humanoid_im.py:824-830): each of the numTrajSamples blocks of a fresh task observation is zeroed with probability
    0.1 -- on the step's rows and on reset rows -- and never in test mode."""
    from phc_amd.utils.flags import flags
    N, T = 2048, 3
    task, env = make_task(N, motion="synthetic:3:1", **{"env.fut_tracks": True, "env.numTrajSamples": T, "+env.fut_tracks_dropout": True})
    so = task.get_self_obs_size()

    def zero_share():
        blocks = task.obs_buf[:, so:].view(N, T, -1)
        return float((blocks.abs().amax(dim=-1) == 0).float().mean())
    env.reset()
    assert abs(zero_share() - 0.1) < 0.02                 # reset(env_ids) rows
    task.step(torch.zeros(N, task.get_action_size(), device=task.device))
    assert abs(zero_share() - 0.1) < 0.02                 # the step's rows
    flags.test = True
    try:
        env.reset()
        assert zero_share() == 0.0
    finally:
        flags.test = False


def test_reset_amp_history_from_the_per_frame_table_equals_the_lookups():
    """phc_im_params_t.amp_ref_table (ABI 33): a reset fills the AMP history from rows of the per-frame table instead of S lookups + observation builds.
    Start times are multiples of 1/30 s and the history steps back by dt = 1/30 s on 30 fps clips, so every history time is a frame up to the rounding
    of the blend factor: exactly 0 for ~5 lookups in 6 -- row f is built from the pair (f, f + 1) at blend 0, the very lookup, so those are bit-equal --
    and <= 1e-4 for most others (first-order blend of two rows; tolerance 5e-6 here, the full build itself is pinned to the reference at 1e-5); lookups
    that land just below the next frame are built in full.  Both reset idioms, resets at t = 0 (negative history times clamp to frame 0) included;
    the table is rebuilt when the motions are re-sampled; a 50 Hz robot gets no table."""
    from phc_amd.utils.flags import flags
    ta, ea = make_task(256, motion="synthetic:5:1")
    tb, eb = make_task(256, motion="synthetic:5:1", **{"+env.amp_ref_table": False})
    torch.manual_seed(3); ea.reset()
    torch.manual_seed(3); eb.reset()
    assert ta._motion_lib._amp_ref_cache[1] is not None and ta._motion_lib._amp_ref_cache[1].shape == (ta._motion_lib.frames.shape[0], 196) and tb._motion_lib._amp_ref_cache[1] is None
    assert torch.isfinite(ta._motion_lib._amp_ref_cache[1]).all()
    close = lambda x, y: np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=0, atol=5e-6)
    close(ta._amp_obs_buf, tb._amp_obs_buf)
    assert float((ta._amp_obs_buf == tb._amp_obs_buf).float().mean()) > 0.7      # (the lookups with blend factor 0: bit-equal)
    assert torch.equal(ta.obs_buf, tb.obs_buf) and torch.equal(ta._root_states, tb._root_states)
    a = (torch.rand(256, 69, device=ta.device) * 2 - 1) * 0.5
    n_reset = 0
    for step in range(30):
        ea.step(a); eb.step(a)
        if step % 2:
            torch.manual_seed(70 + step); ta.reset_done()
            torch.manual_seed(70 + step); tb.reset_done()
        else:
            ids = ta.reset_buf.nonzero().flatten()
            n_reset += len(ids)
            torch.manual_seed(70 + step); ea.reset(ids)
            torch.manual_seed(70 + step); eb.reset(ids)
        close(ta._amp_obs_buf, tb._amp_obs_buf)
        assert torch.equal(ta.obs_buf, tb.obs_buf) and torch.equal(ta.progress_buf, tb.progress_buf)
    assert n_reset > 20
    flags.test = True          # episodes start at t = 0: history times below zero
    try:
        ea.reset(); eb.reset()
        close(ta._amp_obs_buf, tb._amp_obs_buf)
        assert float(ta._motion_start_times.abs().max()) == 0.0
    finally:
        flags.test = False
    old = ta._motion_lib._amp_ref_cache[1]
    torch.manual_seed(9); ta.resample_motions()
    torch.manual_seed(9); tb.resample_motions()
    assert ta._motion_lib._amp_ref_cache[1] is not old
    close(ta._amp_obs_buf, tb._amp_obs_buf)
    th, eh = make_task(64, motion="synthetic:3:2:2.0", **H1_OVER)
    eh.reset()
    assert th._motion_lib._amp_ref_cache[1] is None and abs(th.dt - 0.02) < 1e-9


def test_enable_hist_obs_appends_the_amp_history_as_it_stood_before_the_step():
    """env.enableHistObs (humanoid_amp.py:98-101,327-328,546-557): `_compute_humanoid_obs` appends `_amp_obs_buf` (flattened, newest first) behind the
    self observation.  post_physics_step (:193-204) forms the observation BEFORE `_update_hist_amp_obs`, and `_reset_envs` (:378-385) before
    `_init_amp_obs`: the block is the history of the PREVIOUS step -- for a freshly reset env the finished episode's.  Checked against a task
    without the switch stepped in lockstep."""
    torch.manual_seed(0)
    a = (torch.rand(64, 69, device="cuda") * 2 - 1) * 0.3
    hist, env = make_task(64, motion="synthetic:3:1", **{"+env.enableHistObs": True})
    plain, penv = make_task(64, motion="synthetic:3:1")
    H = 1960
    assert hist.get_self_obs_size() == 358 + H and hist.num_obs == 934 + H and not hist.whole_step_capturable()
    torch.manual_seed(5); env.reset()
    torch.manual_seed(5); penv.reset()
    assert torch.equal(hist._root_states, plain._root_states)
    seen_reset = False
    for step in range(12):
        before = plain._amp_obs_buf.reshape(64, H).clone()
        obs, rew, done, _ = env.step(a)
        pobs, prew, pdone, _ = penv.step(a)
        assert torch.equal(done, pdone) and torch.equal(rew, prew)
        assert torch.equal(obs[:, :358], pobs[:, :358]) and torch.equal(obs[:, 358 + H:], pobs[:, 358:])      # the kernel's own columns, task block moved back
        assert torch.equal(obs[:, 358:358 + H], before)                                                     # the history the step started with
        assert not torch.equal(before, plain._amp_obs_buf.reshape(64, H))
        if bool(done.any()):
            seen_reset = True
            after_step = plain._amp_obs_buf.reshape(64, H).clone()
            torch.manual_seed(100 + step); hist.reset_done()
            torch.manual_seed(100 + step); plain.reset_done()
            rows = done.bool()
            assert torch.equal(hist.obs_buf[rows][:, 358:358 + H], after_step[rows])      # the finished episode's history, not the re-initialised one
            assert torch.equal(hist.obs_buf[rows][:, :358], plain.obs_buf[rows][:, :358])
            assert torch.equal(hist._amp_obs_buf, plain._amp_obs_buf)
    assert seen_reset


def test_obs_v4_v5_remove_disc_rot_and_action_noise_switches():
    """The remaining env switches of VERDICT r2 #9, each against what the reference's code does with it:
      * obs_v 4 (humanoid_im.py:496-501,713-722): with past_track_steps = 1 -- the only value its row stacking fits -- the observation IS the v6 one;
        past_track_steps > 1 raises (the reference's own assignment fails there);
      * obs_v 5 (:503-504,812-815; motion_lib_base.py:214): v6 + the one-hot id of the env's clip, 30 hard-coded columns -> a 30-clip library; fewer raise;
      * remove_disc_rot (humanoid.py:405-413, humanoid_amp.py:996-998): dof_subset empty -> AMP frame = root block (13) + key bodies (12);
      * add_action_noise (humanoid.py:1530-1535): N(0, action_noise_std) on the actions, ONLY while collect_dataset is set."""
    import phc_oracle as po
    torch.manual_seed(0)
    a = None
    ref = {}
    for tag, over in (("v6", {}), ("v4", {"env.obs_v": 4, "+env.past_track_steps": 1})):
        task, env = make_task(64, motion="synthetic:3:1", **over)
        env.reset()
        if a is None:
            a = (torch.rand(64, 69, device=task.device) * 2 - 1) * 0.1
        for _ in range(3):
            obs, rew, done, info = env.step(a)
        ref[tag] = (obs.clone(), rew.clone())
    assert ref["v4"][0].shape == (64, 934) and torch.equal(ref["v4"][0], ref["v6"][0]) and torch.equal(ref["v4"][1], ref["v6"][1])
    with pytest.raises(NotImplementedError, match="past_track_steps"):
        make_task(8, **{"env.obs_v": 4, "+env.past_track_steps": 5})
    # obs_v 5 on a 30-clip library
    task, env = make_task(64, motion="synthetic:30:1", **{"env.obs_v": 5})
    assert task.num_obs == 934 + 30
    obs = env.reset()
    ids = task._motion_lib._curr_motion_ids.cpu()
    for _ in range(2):
        obs, rew, done, info = env.step(a)
        task.reset_done()
    onehot = torch.nn.functional.one_hot(ids, 30).float()
    assert torch.equal(task.obs_buf[:, 934:].cpu(), onehot) and len(set(ids.tolist())) > 5
    bp, br, bv, bav = (t.cpu().numpy() for t in (task._rigid_body_pos, task._rigid_body_rot, task._rigid_body_vel, task._rigid_body_ang_vel))
    np.testing.assert_allclose(task.obs_buf[:, :358].cpu().numpy()[~task.reset_buf.bool().cpu().numpy()],
                               po.compute_humanoid_observations_smpl_max(bp, br, bv, bav)[~task.reset_buf.bool().cpu().numpy()], atol=1e-4)
    with pytest.raises(ValueError, match="30"):
        t5, e5 = make_task(8, motion="synthetic:3:1", **{"env.obs_v": 5})
        e5.reset()
    # remove_disc_rot
    task, env = make_task(32, **{"+env.remove_disc_rot": True})
    full, _ = make_task(32)
    assert task._num_amp_obs_per_step == 25 and task.get_num_amp_obs() == 250 and len(task.dof_subset) == 0
    torch.manual_seed(11); env.reset()
    torch.manual_seed(11); full.reset()          # the same start-time draws: the two tasks hold the same state
    assert torch.equal(task._root_states, full._root_states)
    obs, rew, done, info = env.step(a[:32])
    assert info["amp_obs"].shape == (32, 250) and torch.isfinite(info["amp_obs"]).all()
    full.step(a[:32])
    fa = full.extras["amp_obs"].view(32, 10, 196)
    np.testing.assert_allclose(info["amp_obs"].view(32, 10, 25)[:, 0, :13].cpu().numpy(), fa[:, 0, :13].cpu().numpy(), atol=1e-6)     # root block
    np.testing.assert_allclose(info["amp_obs"].view(32, 10, 25)[:, 0, 13:].cpu().numpy(), fa[:, 0, 184:].cpu().numpy(), atol=1e-6)   # key bodies
    # add_action_noise: inert without collect_dataset, N(0, std) with it
    plain, _ = make_task(256)
    inert, _ = make_task(256, **{"env.add_action_noise": True})
    noisy, _ = make_task(256, **{"env.add_action_noise": True, "+env.action_noise_std": 0.05, "+collect_dataset": True})
    a256 = (torch.rand(256, 69, device=plain.device) * 2 - 1) * 0.1
    for t in (plain, inert, noisy):
        t.pre_physics_step(a256)
    assert torch.equal(inert.actions, plain.actions) and torch.equal(noisy.clean_actions, a256)
    d = (noisy.actions - a256).flatten()
    assert abs(float(d.std()) - 0.05) < 0.003 and abs(float(d.mean())) < 0.002


def test_whole_rollout_step_in_one_graph_equals_eager_launches():
    """reset_done() + step() captured into ONE hipGraph per (step index, reset-list slot, AMP window position) and replayed with
    `replay_step_host()` keeping the task's host state in step -- bit-identical simulator state, observations, AMP windows, reset flags and start
    times to the eager launches over 45 steps (several passes through the 11 window positions and 3 slots, resets included), and the start-time
    draws of a replayed reset launch are fresh (device-side call counter), not those of the captured one."""
    torch.manual_seed(3)
    a = (torch.rand(512, 69, device="cuda") * 2 - 1) * 0.1
    tasks = []
    for mode in ("eager", "graph"):
        task, env = make_task(512, motion="synthetic:3:1", seed=7)
        torch.manual_seed(11)
        env.reset()
        tasks.append((task, env))
    (te, ee), (tg, eg) = tasks
    assert tg.whole_step_capturable()
    tg.align_amp_window(); te.align_amp_window()
    graphs, pool, starts = {}, torch.cuda.graph_pool_handle(), []
    for k in range(45):
        te.reset_done(); ee.step(a)
        key = tg.rollout_step_key()
        if key in graphs:
            graphs[key].replay()
            tg.replay_step_host()
        else:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                tg.reset_done(); eg.step(a)
            g.replay()
            graphs[key] = g
        torch.cuda.synchronize()
        for name in ("_root_states", "_dof_state", "obs_buf", "reset_buf", "progress_buf", "_motion_start_times", "rew_buf", "_terminate_buf"):
            assert torch.equal(getattr(te, name), getattr(tg, name)), (k, name)
        assert torch.equal(te.extras["amp_obs"], tg.extras["amp_obs"]), k
        assert te._amp_head == tg._amp_head and te._reset_slot == tg._reset_slot
        starts.append(tg._motion_start_times.clone())
    assert len(graphs) < 45 and sum(1 for _ in graphs) >= 11          # replays happened
    assert int(tg._reset_rng_dev.item()) == 45
    fresh = torch.stack(starts)                                        # [45, N]: envs that reset several times drew different start times
    resets = (fresh[1:] != fresh[:-1]).sum(0)
    assert int((resets >= 2).sum()) > 50 and int(torch.unique(fresh).numel()) > 100


def _check_task_tracks_its_current_library(task, tag):
    """After a rollout: the reference side buffers (written by the last post-physics launch) and the AMP history of the envs the last reset launch
    touched must come from the task's CURRENT motion library -- a captured launch replayed against a library that has since been re-loaded or
    swapped reads freed memory instead."""
    torch.cuda.synchronize()
    t = (task.progress_buf + 1).float() * task.dt + task._motion_start_times + task._motion_start_times_offset
    res = task._motion_lib.get_motion_state(task._sampled_motion_ids, t, task._global_offset)
    np.testing.assert_allclose(task.ref_body_pos.cpu().numpy(), res["rg_pos"].cpu().numpy(), atol=2e-5, err_msg=f"{tag}: reference bodies of the last step")
    e = (task.progress_buf == 1).nonzero(as_tuple=False).squeeze(-1)
    assert len(e) > 0, tag
    demo = task.build_amp_obs_demo(task._sampled_motion_ids[e], task._motion_start_times[e])   # row k: the clip at start - k dt
    np.testing.assert_allclose(task._amp_obs_buf[e][:, 1:].cpu().numpy(), demo[:, :-1].cpu().numpy(), atol=2e-5, err_msg=f"{tag}: AMP history of freshly reset envs")


def test_rollout_graphs_follow_resample_motions_and_evaluate():
    """the whole-rollout-step hipGraphs bake `phc_motion_lib_t`, `phc_im_params_t` (with the AMP reference table) and the buffer
    struct into the captured env kernels.  `resample_motions()` re-allocates the library, `evaluate()` swaps it (and the table) and flips the
    evaluation flags: `launch_generation()` moves, the learner drops its graphs and captures new ones -- and the replayed launches keep tracking the
    task's current library, exactly like eager launches do."""
    from phc_amd.learning.amp_agent import IMAmpAgent
    over = {"learning.params.config.minibatch_size": 2048, "learning.params.config.amp_obs_demo_buffer_size": 4096,
            "learning.params.config.amp_replay_buffer_size": 4096}
    runs = {}
    for mode in ("graph", "eager"):
        task, env = make_task(256, motion="synthetic:6:1:1.5", seed=5, **dict(over, **{"+learning.params.config.hip_graph": mode == "graph"}))
        agent = IMAmpAgent(env, task.cfg)
        agent.init_train()
        for _ in range(3):
            agent.train_epoch()                      # rollout graphs are captured from the third epoch on
        if mode == "graph":
            assert any(k[0] == "step" for k in agent._roll_graphs), "whole-step graphs in use"
        _check_task_tracks_its_current_library(task, f"{mode}: before")
        gens = [task.launch_generation()]
        frames0 = task._motion_lib.frames
        task.resample_motions()                      # new frames / lengths / starts tensors, new AMP table
        assert task._motion_lib.frames is not frames0
        gens.append(task.launch_generation())
        for _ in range(2):
            agent.train_epoch()
        _check_task_tracks_its_current_library(task, f"{mode}: after resample_motions()")
        lib_train = task._motion_lib
        info, failed = agent.eval()                  # eval library in, flags flipped, params rebuilt, table rebuilt -- and everything back
        assert task._motion_lib is lib_train
        gens.append(task.launch_generation())
        for _ in range(2):
            agent.train_epoch()
        _check_task_tracks_its_current_library(task, f"{mode}: after evaluate()")
        assert len(set(gens)) == 3, gens
        if mode == "graph":
            assert agent._roll_generation == gens[-1] and any(k[0] == "step" for k in agent._roll_graphs)
        runs[mode] = np.isfinite(info["eval/mpjpe_all"])
    assert all(runs.values())


def test_rollout_graphs_survive_a_resample_on_a_robot():
    """Round 5: a Unitree H1 learning run crashed at epoch 101 (`use_count > 0` INTERNAL ASSERT in HIPCachingAllocator, from `torch.cuda.graph(..., pool=...)`): the H1 task
    resamples its motions every 100 epochs, the learner dropped every rollout graph -- and then captured into the dead graphs' pool id.  With the interval cut to 4 epochs
    the same sequence runs here in seconds: graphs from epoch 3, a resample at epochs 5 and 9, fresh graphs in a fresh pool each time."""
    from phc_amd.learning.amp_agent import IMAmpAgent
    over = {"robot": "unitree_h1", "env": "env_im_h1_phc", "sim": "robot_sim", "control": "robot_control", "env.shape_resampling_interval": 4,
            "learning.params.config.minibatch_size": 2048, "learning.params.config.amp_obs_demo_buffer_size": 4096,
            "learning.params.config.amp_replay_buffer_size": 4096, "+learning.params.config.hip_graph": True}
    task, env = make_task(256, motion="stand:4", seed=2, **over)
    agent = IMAmpAgent(env, task.cfg)
    agent.init_train()
    gens, pools = set(), set()
    for _ in range(10):
        agent.train_epoch()                          # pre_epoch(): resample_motions() when epoch_num % 4 == 1
        gens.add(task.launch_generation())
        if getattr(agent, "_roll_pool", None) is not None:
            pools.add(tuple(agent._roll_pool))
    assert len(gens) >= 3, gens
    assert any(k[0] == "step" for k in agent._roll_graphs) and len(pools) >= 2, pools
    assert torch.isfinite(task.obs_buf).all()


def test_tgs_contact_option_env_steps_match_the_dense_oracle_and_the_humanoid_stands():
    """`+solver.contact=tgs` (phc_sim_params_t.contact_model 1, ABI 34): the task installs the rigid ground-contact model with the PhysX parameters of
    sim/default_sim.yaml (4 passes, max_depenetration_velocity 10, bounce threshold 0.2); whole env steps agree with the fp64 dense oracle of the
    same model, every post-physics output with the numpy oracle, and a humanoid under zero actions keeps standing ON the plane (the penalty model
    rests ~1 mm inside it)."""
    from step_oracle import StepChecker
    task, env = make_task(256, motion="synthetic:2:1", **{"+solver.contact": "tgs"})
    sp = task._sim_params
    assert sp.contact_model == 1 and sp.contact_iterations == 4 and abs(sp.max_depenetration_velocity - 10.0) < 1e-6 and abs(sp.bounce_threshold_velocity - 0.2) < 1e-6
    env.reset()
    chk = StepChecker(task)
    for it in range(4):
        chk.before()
        actions = (torch.rand(256, 69, device=task.device) * 2 - 1) * 0.2
        obs, rew, done, info = env.step(actions)
        chk.dynamics(actions, [0, 1, 100, 255], pos_atol=2e-3, root_atol=4e-3)
        chk.after(obs, rew, done, info)
        task.reset_done()
    stand, env2 = make_task(64, motion="stand:4", **{"+solver.contact": "tgs"})
    env2.reset()
    zero = (torch.zeros(64, 69, device=stand.device) - stand._pd_action_offset) / stand._pd_action_scale   # PD target = the rest pose
    for _ in range(60):
        obs, rew, done, info = env2.step(zero)
    torch.cuda.synchronize()
    assert int(stand.progress_buf.min()) >= 60 and float(stand._rigid_body_pos[:, 0, 2].min()) > 0.85, "still standing after 2 s"
    fz = stand._contact_forces[..., 2].sum(-1)
    weight = float(stand.model.mass.sum()) * 9.81
    assert float((fz / weight - 1).abs().max()) < 0.15, (fz / weight)


def test_stepper_switches_through_the_task_config():
    """Round 5 / 6: `solver.inertia_lag` (default on for penalty contact since round 6) and `+solver.force_average=1` reach the kernel through the hydra-style overrides.  With the lagged scheme every
    post-physics output still equals the numpy oracle recomputed from the task's own tensors (the checker does not depend on the stepper's scheme; the scheme
    itself is gated in tests/test_stepper_options.py) and a humanoid under zero actions keeps standing.  With force_average the state trajectory is bit-identical and the published forces differ."""
    from step_oracle import StepChecker
    task, env = make_task(256, motion="synthetic:2:1")
    assert task._sim_params.inertia_lag == 1 and task._sim_params.force_average == 0          # round 6: the lagged scheme is the default of the penalty contact model ...
    assert make_task(8, motion="synthetic:2:1", **{"+solver.inertia_lag": 0})[0]._sim_params.inertia_lag == 0     # ... `+solver.inertia_lag=0` = every sub-step fresh ...
    assert make_task(8, motion="synthetic:2:1", **{"+solver.contact": "tgs"})[0]._sim_params.inertia_lag == 0      # ... and the rigid model always runs fresh
    env.reset()
    chk = StepChecker(task)
    for it in range(4):
        chk.before()
        actions = (torch.rand(256, 69, device=task.device) * 2 - 1) * 0.2
        obs, rew, done, info = env.step(actions)
        chk.after(obs, rew, done, info)
        task.reset_done()
    stand, env2 = make_task(64, motion="stand:4")
    env2.reset()
    zero = (torch.zeros(64, 69, device=stand.device) - stand._pd_action_offset) / stand._pd_action_scale
    for _ in range(60):
        env2.step(zero)
    torch.cuda.synchronize()
    assert int(stand.progress_buf.min()) >= 60 and float(stand._rigid_body_pos[:, 0, 2].min()) > 0.85, "still standing after 2 s"
    # force_average: same states, other published forces
    outs = {}
    for avg in (0, 1):
        t, e = make_task(64, motion="synthetic:2:1", seed=3, **{"+solver.force_average": avg})
        assert t._sim_params.force_average == avg
        e.reset()
        torch.manual_seed(11)
        for _ in range(3):
            e.step((torch.rand(64, 69, device=t.device) * 2 - 1) * 0.2)
        torch.cuda.synchronize()
        outs[avg] = (t._rigid_body_state.clone(), t._contact_forces.clone(), t.dof_force_tensor.clone())
    assert torch.equal(outs[0][0], outs[1][0])
    assert not torch.equal(outs[0][2], outs[1][2]) and float((outs[0][1] - outs[1][1]).abs().max()) > 0.0


def test_multi_clip_acceptance_pipeline_mechanics(tmp_path):
    """Round 5 (f-2 + f-3 together): scripts/multi_clip_acceptance.py end to end in a child process with tiny budgets -- the 16-clip locomotion library, the PNN learner,
    sweeps with soft auto-PMCP re-weighting, the `forward_pmcp` column copy, a hard-mined second stage with column 0 frozen, per-clip report.  Mechanics only (the learning
    result of the real run is under profiles/r05_multi_clip/): the copy is exact, the frozen column does not move, every clip has its row."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = str(tmp_path / "mc.json")
    r = subprocess.run([sys.executable, os.path.join(here, "..", "scripts", "multi_clip_acceptance.py"), "--stage1-s", "6", "--stage2-s", "5", "--envs", "512", "--clips", "16",
                        "--eval-every", "12", "--out", out, "learning.params.config.minibatch_size=4096", "learning.params.config.amp_minibatch_size=1024"],
                       capture_output=True, text=True, timeout=600, cwd=os.path.join(here, ".."))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.load(open(out))
    assert d["primitive1_equals_primitive0_after_the_copy"] is True
    p0 = d["primitive0_after_stage1"]
    assert len(p0["per_clip"]) == 16 and set(p0["by_class"]) == {"stand", "armswing", "stepinplace", "walk", "squat"}
    assert all(np.isfinite(v["mpjpe_g_mm"]) for v in p0["per_clip"].values())
    if "primitive1_after_stage2" in d:      # (always, unless 11 s of training tracked everything)
        assert d["primitive0_frozen_unchanged"] is True and d["stage2_epochs"] > 0
        assert set(d["covered_by_some_primitive"]) == {"rate", "uncovered"}
    assert any("sweep_success_rate" in row for row in d["curve"])
