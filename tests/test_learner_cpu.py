"""P1-P9 on the CPU: the learner is PyTorch and runs without a GPU against a small synthetic VecEnv.
Covers: RunningMeanStd vs the reference's class (imported through the shim when the reference is present),
GAE vs the oracle, loss terms, the replay buffer, one full train_epoch, checkpoint key names (B4), and the
world_size-2 gloo path: flat-bucket gradient all-reduce keeps replicas bit-identical."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import phc_oracle as po
from phc_amd.config import compose
from phc_amd.learning.amp_agent import FlatGradBucket, IMAmpAgent, discount_values, swap_and_flatten01
from phc_amd.learning.network import A2CNetwork, ModelAMPContinuous, policy_kl
from phc_amd.learning.replay_buffer import ReplayBuffer
from phc_amd.learning.running_mean_std import RunningMeanStd


class FakeTask:
    """Tiny stand-in with the attributes the agent reads (no physics): obs 20, amp 3x6, actions 5."""
    device = "cpu"
    temp_running_mean = True
    shape_resampling_interval = 500

    def __init__(self, n, seed=0):
        self.num_envs = n
        self.g = torch.Generator().manual_seed(seed)
        self.obs_buf = torch.randn(n, 20, generator=self.g)
        self.reset_buf = torch.zeros(n, dtype=torch.long)

    def get_num_amp_obs(self):
        return 18

    def reset_done(self):
        self.reset_buf.zero_()


class FakeVecEnv:
    clip_obs = np.inf

    def __init__(self, n, seed=0):
        self.task = FakeTask(n, seed)
        self.num_envs, self.num_obs, self.num_actions = n, 20, 5

    def reset(self, env_ids=None):
        return self.task.obs_buf

    def step(self, actions):
        t = self.task
        t.obs_buf = 0.9 * t.obs_buf + 0.1 * torch.randn(t.num_envs, 20, generator=t.g)
        rew = torch.exp(-actions.pow(2).mean(-1))
        done = (torch.rand(t.num_envs, generator=t.g) < 0.1).long()
        t.reset_buf = done
        info = {"amp_obs": torch.randn(t.num_envs, 18, generator=t.g), "terminate": done * (torch.rand(t.num_envs, generator=t.g) < 0.5).long(),
                "reward_raw": torch.rand(t.num_envs, 5, generator=t.g)}
        return t.obs_buf, rew, done, info

    def fetch_amp_obs_demo(self, n):
        return torch.randn(n, 18, generator=self.task.g) + 0.5


def small_cfg():
    cfg = compose(["learning.params.config.horizon_length=8", "learning.params.config.minibatch_size=64", "learning.params.config.mini_epochs=2",
                   "learning.params.config.amp_minibatch_size=32", "learning.params.config.amp_batch_size=16",
                   "learning.params.config.amp_obs_demo_buffer_size=256", "learning.params.config.amp_replay_buffer_size=256",
                   "learning.params.network.mlp.units=[32,16]", "learning.params.network.disc.units=[32,16]"])
    return cfg


def test_running_mean_std_matches_reference():
    torch.manual_seed(0)
    ours = RunningMeanStd((7,))
    ours.train()
    ref = None
    if os.path.isdir("/root/reference"):
        import ref_shim
        ref = ref_shim.ref_module("phc.utils.running_mean_std").RunningMeanStd((7,))
        ref.train()
    xs = [torch.randn(50, 7) * (i + 1) + i for i in range(4)]
    for x in xs:
        y = ours(x)
        if ref is not None:
            assert torch.equal(y, ref(x))
    assert ours.running_mean.dtype == torch.float64 and ours.count.item() == 201
    allx = torch.cat(xs).double()
    np.testing.assert_allclose(ours.running_mean.numpy(), allx.sum(0).numpy() / 201, rtol=1e-4, atol=1e-5)  # count starts at 1 with mean 0; batch moments are fp32
    if ref is not None:
        assert torch.equal(ours.running_mean, ref.running_mean) and torch.equal(ours.running_var, ref.running_var)
    ours.eval()
    m = ours.running_mean.clone()
    y = ours(xs[0])
    assert torch.equal(m, ours.running_mean) and y.abs().max() <= 5.0
    np.testing.assert_allclose(ours(ours(xs[1]), unnorm=True).numpy(), xs[1].clamp(-1e9, 1e9).numpy(), atol=2e-2, rtol=1e-2)
    ours.train(); ours.freeze()
    ours(xs[2] * 100)
    assert torch.equal(m, ours.running_mean)


def test_gae_matches_oracle():
    g = torch.Generator().manual_seed(0)
    T, N = 16, 33
    fd = (torch.rand(T, N, generator=g) < 0.15).float()
    v, r, nv = (torch.randn(T, N, 1, generator=g) for _ in range(3))
    adv = discount_values(fd, v, r, nv, 0.99, 0.95)
    want = po.discount_values(fd.numpy(), v.numpy(), r.numpy(), nv.numpy())
    np.testing.assert_allclose(adv.numpy(), want, atol=1e-5)
    x = torch.arange(24).reshape(2, 3, 4)
    assert torch.equal(swap_and_flatten01(x)[1], x[1, 0])  # env-major flattening


def test_network_shapes_names_and_policy_math():
    cfg = compose([])
    net = A2CNetwork(cfg.learning.params.network, 69, (934,), (1960,))
    keys = set(ModelAMPContinuous(net).state_dict().keys())
    for k in ("a2c_network.actor_mlp.0.weight", "a2c_network.actor_mlp.2.bias", "a2c_network.mu.weight", "a2c_network.sigma",
              "a2c_network.critic_mlp.0.weight", "a2c_network.value.weight", "a2c_network._disc_mlp.0.weight", "a2c_network._disc_logits.bias"):
        assert k in keys, k
    n_params = sum(p.numel() for p in net.parameters() if p.requires_grad)
    assert n_params == 1517637 - 0 + 1482753 + 2533377 - 0, n_params  # actor + critic + disc of `learning=im`
    assert not net.sigma.requires_grad and torch.allclose(net.sigma, torch.full((69,), -2.9))
    model = ModelAMPContinuous(net)
    obs = torch.randn(4, 934)
    out = model({"is_train": False, "obs": obs})
    mu, sigma, a = out["mus"], out["sigmas"], out["actions"]
    nlp = -torch.distributions.Normal(mu, sigma).log_prob(a).sum(-1)
    np.testing.assert_allclose(out["neglogpacs"].detach().numpy(), nlp.detach().numpy(), rtol=1e-5)
    assert policy_kl(mu, sigma, mu, sigma).abs() < 0.2  # rl_games formula: the 1e-5 epsilons give -0.11 at sigma=exp(-2.9), D=69


def test_replay_buffer_wraps_and_samples():
    torch.manual_seed(0)
    rb = ReplayBuffer(10, "cpu")
    rb.store({"amp_obs": torch.arange(6).float()[:, None]})
    s = rb.sample(50)["amp_obs"]
    assert s.max() <= 5 and rb.get_total_count() == 6
    rb.store({"amp_obs": 10 + torch.arange(7).float()[:, None]})
    assert rb.get_total_count() == 13 and rb._head == 3
    vals = set(rb._data_buf["amp_obs"].flatten().tolist())
    assert vals == {14., 15., 16., 3., 4., 5., 10., 11., 12., 13.}


def test_train_epoch_runs_and_learns_something():
    torch.manual_seed(0)
    env = FakeVecEnv(32)
    agent = IMAmpAgent(env, small_cfg(), bf16=False)
    agent.init_train()
    w0 = agent.model.a2c_network.mu.weight.clone()
    infos = [agent.train_epoch() for _ in range(3)]
    for i in infos:
        assert np.isfinite([i["actor_loss"], i["critic_loss"], i["disc_loss"], i["kl"]]).all()
    assert not torch.equal(w0, agent.model.a2c_network.mu.weight)
    assert agent.running_mean_std.count.item() > 1 and agent._amp_input_mean_std.count.item() > 1
    assert agent._amp_replay_buffer.get_total_count() >= 2 * 32 * 8  # third store is thinned by amp_replay_keep_prob (amp_agent.py:880-894)
    st = agent.get_full_state_weights()
    assert {"model", "running_mean_std", "reward_mean_std", "amp_input_mean_std"} <= set(st)  # amp_agent.py:69-108
    agent2 = IMAmpAgent(FakeVecEnv(32), small_cfg(), bf16=False)
    agent2.set_full_state_weights(st)
    assert torch.equal(agent2.model.a2c_network.mu.weight, agent.model.a2c_network.mu.weight)


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)  # different init + different env data per rank
    env = FakeVecEnv(32, seed=rank)
    agent = IMAmpAgent(env, small_cfg(), dist=dist, bf16=False)
    agent.init_train()
    for _ in range(2):
        agent.train_epoch()
    flat = torch.cat([p.data.flatten() for p in agent.model.parameters()])
    gather = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gather, flat)
    stats = torch.cat([agent.running_mean_std.running_mean, agent.running_mean_std.running_var])
    sg = [torch.zeros_like(stats) for _ in range(world)]
    dist.all_gather(sg, stats)
    # the optional split collective (amp_agent.py `split_allreduce`): the same bucket reduced in two ranges == reduced in one piece
    b = agent.grads
    net = agent.model.a2c_network
    own, rest = b.spans_of([p for mod in (net._disc_mlp, net._disc_logits) for p in mod.parameters()])
    g0 = torch.randn(b.flat.numel(), generator=torch.Generator().manual_seed(7 + rank))
    b.flat.copy_(g0)
    n_one = b.all_reduce_mean(dist)
    whole = b.flat.clone()
    b.flat.copy_(g0)
    n_two = b.all_reduce_mean(dist, spans=own) + b.all_reduce_mean(dist, spans=rest)
    split_ok = bool(torch.equal(whole, b.flat)) and n_one == 1 and n_two == len(own) + len(rest) == 2
    if rank == 0:
        q.put((bool(torch.equal(gather[0], gather[1])) and split_ok, bool(torch.allclose(sg[0], sg[1])), float(flat.abs().sum())))
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_allreduce_keeps_replicas_identical():
    """N>1 path on CPU: broadcast of initial params, one flat-bucket all-reduce(mean) per optimizer step, running-stat sync."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    same_params, same_stats, norm = q.get(timeout=10)
    assert same_params, "replicas diverged: the gradient all-reduce is not keeping them in lock-step"
    assert same_stats and norm > 0


def test_flat_grad_bucket_is_one_buffer():
    lin = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    b = FlatGradBucket(lin.parameters())
    lin(torch.randn(5, 4)).sum().backward()
    # one buffer; every parameter's segment starts 16-byte aligned (12 | 3 + 1 pad | 6 + 2 pad | 2)
    assert b.flat.numel() == 12 + 4 + 8 + 2 and [o for o, _ in b.segments] == [0, 12, 16, 24] and b.flat.abs().sum() > 0
    assert lin[0].weight.grad.data_ptr() == b.flat.data_ptr()
    b.zero()
    assert lin[1].bias.grad.abs().sum() == 0


def test_flat_grad_bucket_spans_and_partial_zero():
    """Round 4: the two passes of an optimizer step (policy, discriminator) each zero THEIR part of the flat gradient (`spans_of`, `zero(spans=)`)."""
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2), torch.nn.Linear(2, 5))
    b = FlatGradBucket(net.parameters())
    own, rest = b.spans_of(list(net[1].parameters()))
    assert own == [(16, 26)] and rest == [(0, 16), (26, b.flat.numel())]          # (weight 6 + 2 pad, bias 2 | before | after)
    own_last, rest_last = b.spans_of(list(net[2].parameters()))
    assert own_last == [(28, b.flat.numel())] and rest_last == [(0, 28)]
    with pytest.raises(ValueError):
        b.spans_of([net[0].weight, net[2].weight])                                 # not one consecutive run
    b.flat.fill_(1.0)
    g0 = b.gen
    b.zero(decay=[(net[1].weight, 2.0)], spans=own)
    assert b.gen == g0 + 1 and torch.equal(net[1].weight.grad, 2.0 * net[1].weight.data) and net[1].bias.grad.abs().sum() == 0
    assert (net[0].weight.grad == 1).all() and (net[2].bias.grad == 1).all()       # the other pass's gradients are not touched
    b.zero(spans=rest)
    assert net[0].weight.grad.abs().sum() == 0 and net[2].bias.grad.abs().sum() == 0 and torch.equal(net[1].weight.grad, 2.0 * net[1].weight.data)


def test_pnn_network_and_forward_pmcp():
    """P3: PNN actor columns (key names, freezing, column selection) and the forward_pmcp column copy."""
    from phc_amd.learning.network import A2CPNNNetwork, forward_pmcp
    cfg = compose(["learning=im_pnn", "env=env_im_pnn"])
    detail = {"num_prim": 3, "training_prim": 1, "has_lateral": False}
    net = A2CPNNNetwork(cfg.learning.params.network, 69, (934,), (1960,), detail)
    model = ModelAMPContinuous(net)
    keys = set(model.state_dict())
    assert "a2c_network.pnn.actors.2.4.weight" in keys and "a2c_network.actor_mlp.0.weight" not in keys and "a2c_network.mu.weight" in keys  # the reference keeps mu (amp_network_pnn_builder.py:51)
    assert not any(p.requires_grad for p in net.pnn.actors[0].parameters())          # columns < training_prim frozen (pnn.py:40-45)
    assert all(p.requires_grad for p in net.pnn.actors[1].parameters())
    obs = torch.randn(5, 934)
    mu, logstd = net.eval_actor(obs)
    assert torch.equal(mu, net.pnn.actors[1](obs)) and mu.shape == (5, 69)
    ck = {"model": {k: v.clone() for k, v in model.state_dict().items()}}
    forward_pmcp(ck, 1)
    assert torch.equal(ck["model"]["a2c_network.pnn.actors.2.0.weight"], ck["model"]["a2c_network.pnn.actors.1.0.weight"])
    assert not torch.equal(ck["model"]["a2c_network.pnn.actors.0.0.weight"], ck["model"]["a2c_network.pnn.actors.1.0.weight"])
    # a PNN agent trains only the active column (and critic / disc)
    class T(FakeTask):
        def get_task_obs_size_detail(self):
            return {"num_prim": 2, "training_prim": 1, "has_lateral": False}
    env = FakeVecEnv(32)
    env.task = T(32)
    c = small_cfg()
    c.learning.params.network.name = "amp_pnn"
    agent = IMAmpAgent(env, c, bf16=False)
    agent.init_train()
    w_frozen = agent.model.a2c_network.pnn.actors[0][0].weight.clone()
    w_active = agent.model.a2c_network.pnn.actors[1][0].weight.clone()
    agent.train_epoch()
    assert torch.equal(w_frozen, agent.model.a2c_network.pnn.actors[0][0].weight)
    assert not torch.equal(w_active, agent.model.a2c_network.pnn.actors[1][0].weight)


def test_stale_gradient_accumulator_probe():
    """The guard in front of the update-graph capture: an autograd-tracked copy of a parameter held by the caller keeps that parameter's
    AccumulateGrad node alive (bound to the stream it was created on) and must be detected; detached snapshots are fine."""
    agent = IMAmpAgent(FakeVecEnv(32), small_cfg(), bf16=False)
    assert not agent._stale_grad_accumulators()
    snap = agent.model.a2c_network.mu.weight.detach().clone()
    assert not agent._stale_grad_accumulators()
    tracked = agent.model.a2c_network.mu.weight.clone()
    assert agent._stale_grad_accumulators()
    del tracked
    assert not agent._stale_grad_accumulators() and snap is not None


def test_first_write_protocol_of_the_flat_bucket():
    """fast_ops._first_write: a layer may store the first gradient of a step straight into `p.grad` exactly once per `zero()`; a
    second contribution in the same step, a parameter outside a bucket, or one without a gradient buffer take the accumulate path."""
    from phc_amd.learning.fast_ops import _first_write
    lin = torch.nn.Linear(4, 3)
    assert not _first_write(lin.weight)                      # not in a bucket
    b = FlatGradBucket(lin.parameters())
    b.zero()
    assert _first_write(lin.weight) and not _first_write(lin.weight)
    assert _first_write(lin.bias)
    b.zero()
    assert _first_write(lin.weight) and _first_write(lin.bias) and not _first_write(lin.bias)
    assert b.shadow is None and lin.weight.grad.data_ptr() == b.flat.data_ptr()


@pytest.mark.skipif(not os.path.exists("/root/reference/phc/learning/replay_buffer.py"), reason="reference checkout not present")
def test_replay_buffer_equals_the_reference_class():
    """P9: the same store / sample sequence (wrap-around stores, sampling before and after the buffer fills, index re-shuffles) through
    the reference's own `ReplayBuffer` (phc/learning/replay_buffer.py, pure torch: imported as is) and through ours, same torch seed:
    identical storage, identical samples."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_replay_buffer", "/root/reference/phc/learning/replay_buffer.py")
    ref_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_mod)
    outs = []
    for cls in (ref_mod.ReplayBuffer, ReplayBuffer):
        torch.manual_seed(42)
        rb = cls(37, "cpu")
        g = torch.Generator().manual_seed(1)
        seq = []
        for n_store, n_sample in ((10, 5), (20, 16), (15, 30), (37, 8), (3, 50), (9, 37)):
            rb.store({"amp_obs": torch.randn(n_store, 4, generator=g)})
            seq.append(rb.sample(n_sample)["amp_obs"].clone())
        outs.append((rb._data_buf["amp_obs"].clone(), seq, rb.get_total_count(), rb._head))
    (da, sa, ca, ha), (db, sb, cb, hb) = outs
    assert ca == cb and ha == hb and torch.equal(da, db)
    for x, y in zip(sa, sb):
        assert torch.equal(x, y)


def test_train_checkpoint_cadence_and_termination_history_restore(tmp_path):
    """The reference's save cadence (common_agent.py:142-150: `Humanoid.pth` every min(50, save_best_after) epochs, numbered
    checkpoints every save_frequency) and `IMAmpAgent.restore` re-installing the newest `failed_*.pkl` termination history into the
    motion library's sampling probabilities (im_amp.py:101-117)."""
    import joblib
    cfg = small_cfg()
    cfg.learning.params.config.save_frequency = 2
    cfg.learning.params.config.save_best_after = 3
    cfg.learning.params.config.save_intermediate = True
    agent = IMAmpAgent(FakeVecEnv(32), cfg, bf16=False)
    agent.train(4, log=None, output_dir=str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == ["Humanoid.pth", "Humanoid_00000002.pth", "Humanoid_00000004.pth", "summaries"]   # Humanoid.pth written at epoch 3
    # the reference's per-epoch scalars (common_agent.py:603-635, amp_agent.py:900-933: writer.add_scalar(tag, value, epoch)) as a jsonl stream, its tags
    import json
    rows = [json.loads(l) for l in open(tmp_path / "summaries" / "scalars.jsonl")]
    assert [r["step"] for r in rows] == [1, 2, 3, 4] and rows[-1]["frame"] == agent.frame
    for tag in ("performance/update_time", "performance/play_time", "learning_rate/last_lr", "learning_rate/e_clip", "loss/actor_loss", "loss/critic_loss", "loss/bounds_loss",
                "loss/entropy", "loss/kl", "disc/loss", "disc/agent_acc", "disc/demo_acc", "disc/grad_penalty", "disc/logit_loss", "disc/reward_mean", "rewards/mb_rewards",
                "rewards/body_pos", "rewards/body_rot", "rewards/lin_vel", "rewards/ang_vel", "rewards/power"):
        assert all(tag in r and np.isfinite(r[tag]) for r in rows), tag

    class Lib:
        def __init__(self):
            self._termination_history = torch.zeros(5)
            self._sampling_prob = torch.ones(5) / 5

        def update_sampling_prob(self, h):
            if len(h) == len(self._termination_history) and h.sum() > 0:
                self._sampling_prob[:] = h / h.sum()
                self._termination_history = h
                return True
            return False
    joblib.dump({"failed_keys": ["a"], "termination_history": torch.tensor([0., 1, 0, 0, 0])}, tmp_path / "failed_0000000002.pkl")
    joblib.dump({"failed_keys": ["a", "c"], "termination_history": torch.tensor([0., 2, 0, 2, 0])}, tmp_path / "failed_0000000004.pkl")
    agent2 = IMAmpAgent(FakeVecEnv(32), cfg, bf16=False)
    agent2.task._motion_lib = Lib()
    agent2.restore(str(tmp_path / "Humanoid_00000004.pth"))
    assert agent2.epoch_num == 4
    assert torch.equal(agent2.task._motion_lib._sampling_prob, torch.tensor([0., 0.5, 0, 0.5, 0]))
    assert torch.equal(agent2.model.a2c_network.mu.weight, agent.model.a2c_network.mu.weight)


def test_checkpoint_optimizer_state_is_per_parameter_and_loads_plain_adam_checkpoints(tmp_path):
    """B4: the checkpoint's optimizer entry has the reference's layout -- `Adam(model.parameters())` (common_agent.py:67), one state
    entry per parameter -- although the agent steps ONE flat parameter.  (a) save -> load round trip restores the flat moments
    bit for bit; (b) a checkpoint written by a plain torch Adam over `model.parameters()` loads and continues identically to an
    agent that kept training; (c) a state that belongs to another set of trainable parameters is skipped, not mis-applied."""
    torch.manual_seed(0)
    env = FakeVecEnv(16)
    agent = IMAmpAgent(env, small_cfg())
    agent.init_train()
    agent.train_epoch()
    w = agent.get_full_state_weights()
    params = list(agent.model.parameters())
    trainable = [i for i, p in enumerate(params) if p.requires_grad]
    assert w["optimizer"]["param_groups"][0]["params"] == list(range(len(params)))
    assert sorted(w["optimizer"]["state"]) == trainable
    for i in trainable:
        assert tuple(w["optimizer"]["state"][i]["exp_avg"].shape) == tuple(params[i].shape)
    path = str(tmp_path / "Humanoid.pth")
    agent.save(path)
    # (a)
    torch.manual_seed(1)
    other = IMAmpAgent(FakeVecEnv(16), small_cfg())
    other.restore(path)
    sa, sb = agent.optimizer.state[agent.grads.flat_param], other.optimizer.state[other.grads.flat_param]
    assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]) and float(sa["step"]) == float(sb["step"])
    assert torch.equal(agent.grads.flat_param, other.grads.flat_param)
    # (b) the reference's own optimizer object over the same model: its state_dict must load
    ref_model = ModelAMPContinuous(A2CNetwork(small_cfg()["learning"]["params"]["network"], 5, (20,), (18,)))
    ref_model.load_state_dict(w["model"])
    ref_opt = torch.optim.Adam(ref_model.parameters(), 2e-5, eps=1e-8)
    ref_opt.load_state_dict(w["optimizer"])       # our file is readable by the reference's optimizer ...
    g = torch.Generator().manual_seed(5)
    grads = [torch.randn(p.shape, generator=g) for p in ref_model.parameters()]
    for p, gr in zip(ref_model.parameters(), grads):
        p.grad = gr.clone() if p.requires_grad else None
    ref_opt.step()
    ck = {"model": w["model"], "optimizer": ref_opt.state_dict(), "epoch": 3, "frame": 7}
    third = IMAmpAgent(FakeVecEnv(16), small_cfg())
    third.set_full_state_weights(ck)              # ... and the reference's file by us
    st = third.optimizer.state[third.grads.flat_param]
    ref_moments = [ref_opt.state[p]["exp_avg"] for p in ref_model.parameters() if p.requires_grad]
    for i, m in enumerate(ref_moments):       # (the flat state has 16-byte alignment gaps between the parameters' segments)
        assert torch.equal(third.grads.param_view(st["exp_avg"], i), m)
    assert float(st["exp_avg"].abs().sum()) == float(sum(m.abs().sum() for m in ref_moments)) and float(st["step"]) == float(sa["step"]) + 1
    # one more step on both sides with the same gradient: same parameters
    for p, gr in zip(third.model.parameters(), grads):
        if p.requires_grad:
            p.grad.copy_(gr)
    third.optimizer.step()
    third2 = torch.cat([p.detach().reshape(-1) for p in third.model.parameters() if p.requires_grad])
    for p, gr in zip(ref_model.parameters(), grads):
        p.grad = gr.clone() if p.requires_grad else None
    ref_opt.step()
    # (third loaded the pre-step weights `w["model"]`, ref_model has stepped twice) -> compare the moments instead of the weights
    st = third.optimizer.state[third.grads.flat_param]
    ref_sq = [ref_opt.state[p]["exp_avg_sq"] for p in ref_model.parameters() if p.requires_grad]
    for i, m in enumerate(ref_sq):
        assert torch.allclose(third.grads.param_view(st["exp_avg_sq"], i), m, rtol=1e-6, atol=1e-12)
    assert third2.numel() == sum(m.numel() for m in ref_sq)
    # (c) state of a different trainable set: skipped
    bad = {"state": {0: w["optimizer"]["state"][trainable[0]]}, "param_groups": [dict(w["optimizer"]["param_groups"][0])]}
    fresh = IMAmpAgent(FakeVecEnv(16), small_cfg())
    assert fresh._load_optimizer_state_dict(bad) is False and len(fresh.optimizer.state) == 0
    # players do not touch the optimizer
    fresh.restore(path, load_optimizer=False)
    assert len(fresh.optimizer.state) == 0 and torch.equal(fresh.grads.flat_param, agent.grads.flat_param)


@pytest.mark.parametrize("clip", [True, False])
def test_env_is_stepped_with_clamped_actions_and_the_buffer_keeps_the_samples(clip):
    """rl_games' A2CBase.env_step -> preprocess_actions (clip_actions, default True; common_agent.py:47): the env sees clamp(action, -1, 1) (the
    action space is Box(-1, 1): the rescale is the identity) while the experience buffer -- and so the PPO ratio -- keeps the raw sample."""
    torch.manual_seed(0)
    env = FakeVecEnv(32)
    cfg = small_cfg()
    cfg["learning"]["params"]["config"]["clip_actions"] = clip
    agent = IMAmpAgent(env, cfg, bf16=False)
    agent.init_train()
    with torch.no_grad():
        agent.model.a2c_network.mu.bias.add_(1.5)    # push the policy mean outside the action space
    seen = []
    step = env.step
    env.step = lambda a: (seen.append(a.clone()), step(a))[1]
    agent.play_steps()
    raw = agent.exp["actions"]
    assert len(seen) == 8 and float(raw.abs().max()) > 1.2
    for n, a in enumerate(seen):
        assert torch.equal(a, raw[n].clamp(-1, 1) if clip else raw[n])
    assert torch.equal(agent.preprocess_actions(torch.tensor([-3.0, 0.25, 2.0])), torch.tensor([-1.0, 0.25, 1.0]) if clip else torch.tensor([-3.0, 0.25, 2.0]))
