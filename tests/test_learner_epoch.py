"""P4 / P7 / P9 (and P5, P6, P8, P1 in sequence): THREE WHOLE EPOCHS of the learner against the reference's own `AMPAgent.train_epoch`.

`tests/golden/learner_epoch.npz` (oracle/gen_golden_epoch.py) is a recording of the unmodified reference methods -- `play_steps`, `_update_amp_demos`, the replay /
demo `ReplayBuffer`s, `prepare_dataset`, `AMPDataset._get_item / _shuffle_idx_buf`, 2 x 4 `calc_gradients` per epoch, `_store_replay_amp_obs` (all three of its
branches), `pre_epoch / post_epoch`, `_assemble_train_info` -- on a scripted vec-env, with every random draw (policy noise, `randperm`, `bernoulli`) taken from the
fixture.  Here `IMAmpAgent.train_epoch` runs on the same scripted env with the same draws and has to reproduce, epoch by epoch:

  the experience buffer (obses, actions, mus, sigmas, neglogpacs, values, rewards, dones, next_obses, next_values with the `terminate` mask, amp_obs),
  the actions the env was stepped with (clamped), discriminator / combined rewards, returns, advantages, normalised old values / returns,
  the value normaliser after `prepare_dataset`, which rows each minibatch held (through the per-minibatch losses), every optimizer step's scalars
  (incl. clip fraction, KL, mean logits), parameters + the three normalisers + the frozen copy after each epoch, the contents and counters of both ring buffers,
  the Adam state at the end, and the epoch's logged scalars under the reference's tags.

CPU: the agent's torch path, both reset conventions (`faithful_reset`: the reference's `env.reset(done_indices)` call sequence; default: `task.reset_done()`).
`-m gpu`: the product path on the device (HIP kernels through the C ABI), fp32 GEMMs, eager launches and captured hipGraphs (rollout segments + update)."""
import numpy as np
import pytest
import torch

from phc_amd.config import compose
from phc_amd.learning.amp_agent import IMAmpAgent


def _dims(g):
    names = ("O", "M", "A", "T", "N", "MB", "AMB", "MINI_EPOCHS", "AMP_BATCH", "DEMO_BUF", "REPLAY_BUF", "EPOCHS")
    return dict(zip(names, (int(v) for v in g["dims"])))


def _sub(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(np.asarray(g[k])) for k in g if k.startswith(prefix)}


class _Task:
    temp_running_mean = True
    shape_resampling_interval = 500

    def __init__(self, env, with_reset_done):
        self.env, self.device, self.num_envs = env, env.device, env.d["N"]
        if with_reset_done:
            self.reset_done = self._reset_done

    obs_buf = property(lambda self: self.env.obs_buf)
    reset_buf = property(lambda self: self.env.reset_buf)

    def get_num_amp_obs(self):
        return self.env.d["M"]

    def _reset_done(self):
        """The device convention: finished envs (reset_buf) are reset in place, no index list crosses to the host."""
        self.env.reset(self.env.reset_buf.nonzero(as_tuple=False)[:, 0])
        self.env.reset_buf.zero_()


class ScriptedVecEnv:
    """Plays the fixture's streams through the B1 surface; the returned tensors are views of buffers the next call overwrites."""
    clip_obs = np.inf

    def __init__(self, g, device, with_reset_done):
        d = self.d = _dims(g)
        self.device = device
        self.s = {k: v.to(device) for k, v in _sub(g, "script/").items()}
        self.noise = torch.from_numpy(g["noise"]).to(device)
        self.k = self.demo_k = 0
        N, M = d["N"], d["M"]
        self.obs_buf = self.s["obs0"].clone()
        self.rew_buf, self.reset_buf = torch.zeros(N, device=device), torch.zeros(N, dtype=torch.long, device=device)
        self.terminate_buf, self.raw_buf = torch.zeros(N, dtype=torch.long, device=device), torch.zeros(N, 5, device=device)
        self.amp_buf, self.noise_buf = torch.zeros(N, M, device=device), self.noise[0].clone()
        self.actions_seen, self.reset_counts = [], []
        self.num_envs, self.num_obs, self.num_actions = N, d["O"], d["A"]
        self.task = _Task(self, with_reset_done)

    def reset(self, env_ids=None):
        if env_ids is not None and len(env_ids) > 0:
            self.reset_counts.append(len(env_ids))
            self.obs_buf[env_ids] = self.s["reset_obs"][self.k - 1][env_ids]
        return self.obs_buf

    def step(self, actions):
        s, k = self.s, self.k
        self.actions_seen.append(actions.detach().clone())
        for buf, key in ((self.obs_buf, "obs"), (self.rew_buf, "rewards"), (self.reset_buf, "dones"), (self.terminate_buf, "terminate"), (self.raw_buf, "reward_raw"),
                         (self.amp_buf, "amp_obs")):
            buf.copy_(s[key][k])
        self.k += 1
        if self.k < self.noise.shape[0]:
            self.noise_buf.copy_(self.noise[self.k])          # the draw the policy takes next (see _Draws.randn)
        return self.obs_buf, self.rew_buf, self.reset_buf, {"terminate": self.terminate_buf, "reward_raw": self.raw_buf, "amp_obs": self.amp_buf}

    def fetch_amp_obs_demo(self, n):
        assert n == self.d["AMP_BATCH"]
        self.demo_k += 1
        return self.s["demo"][self.demo_k - 1]


class _Draws:
    """torch.randn / randn_like of the policy's shape, torch.randperm and torch.bernoulli served from the fixture (per size, in the order the reference drew them)."""

    def __init__(self, g, env):
        self.env, self.shape = env, (env.d["N"], env.d["A"])
        self.perms = {int(k.split("/")[1]): list(torch.from_numpy(g[k])) for k in g if k.startswith("perm/")}
        self.masks = [torch.from_numpy(g[f"mask/{i}"]) for i in range(int(g["n_masks"]))]
        self.noise_used = 0

    def __enter__(self):
        self.real = (torch.randn, torch.randn_like, torch.randperm, torch.bernoulli)
        torch.randn, torch.randn_like, torch.randperm, torch.bernoulli = self.randn, self.randn_like, self.randperm, self.bernoulli
        return self

    def __exit__(self, *a):
        torch.randn, torch.randn_like, torch.randperm, torch.bernoulli = self.real

    def randn(self, *size, **kw):
        shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
        if shape == self.shape:
            self.noise_used += 1
            return self.env.noise_buf
        return self.real[0](*size, **kw)

    def randn_like(self, x, **kw):
        if tuple(x.shape) == self.shape:
            self.noise_used += 1
            return self.env.noise_buf
        return self.real[1](x, **kw)

    def randperm(self, n, **kw):
        assert self.perms.get(int(n)), f"randperm({n}): the reference drew no (further) permutation of this size"
        return self.perms[int(n)].pop(0).to(kw.get("device", "cpu"))

    def bernoulli(self, p, **kw):
        m = self.masks.pop(0)
        assert m.shape == p.shape
        return m.to(p.device, p.dtype)

    def all_consumed(self):
        return not any(self.perms.values()) and not self.masks


def _cfg(d, g, extra=()):
    return compose(["learning=im", f"learning.params.config.horizon_length={d['T']}", f"learning.params.config.minibatch_size={d['MB']}",
                    f"learning.params.config.mini_epochs={d['MINI_EPOCHS']}", f"learning.params.config.amp_minibatch_size={d['AMB']}",
                    f"learning.params.config.amp_batch_size={d['AMP_BATCH']}", f"learning.params.config.amp_obs_demo_buffer_size={d['DEMO_BUF']}",
                    f"learning.params.config.amp_replay_buffer_size={d['REPLAY_BUF']}", f"learning.params.config.amp_replay_keep_prob={float(g['keep_prob'])}",
                    f"learning.params.network.mlp.units=[{','.join(str(int(u)) for u in g['units'])}]",
                    f"learning.params.network.disc.units=[{','.join(str(int(u)) for u in g['disc_units'])}]", "+learning.params.config.trace_minibatches=True"]
                   + list(extra))


def _near(got, want, rtol, atol, msg):
    got = got.detach().float().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    np.testing.assert_allclose(got, np.asarray(want), rtol=rtol, atol=atol, err_msg=msg)


def run_epochs(g, device="cpu", faithful_reset=False, with_reset_done=True, extra=(), tol=None, graphs=False, bf16=False):
    """-> worst relative parameter error over the epochs (for the caller's report)."""
    d = _dims(g)
    t = dict(exp=2e-5, scal_r=2e-4, scal_a=2e-6, stat=1e-6, stat_a=1e-9, param_tail=1e-3)
    t.update(tol or {})
    env = ScriptedVecEnv(g, device, with_reset_done)
    draws = _Draws(g, env)
    with draws:
        torch.manual_seed(0)
        agent = IMAmpAgent(env, _cfg(d, g, extra), faithful_reset=faithful_reset, bf16=bf16)
        assert agent.bf16 == bf16
        agent.model.load_state_dict(_sub(g, "model/"), strict=True)
        if agent.grads.shadow is not None:
            agent.grads.shadow.copy_(agent.grads.flat_param)
        agent.running_mean_std.load_state_dict(_sub(g, "running_mean_std/"))
        agent.value_mean_std.load_state_dict(_sub(g, "reward_mean_std/"))
        agent._amp_input_mean_std.load_state_dict(_sub(g, "amp_input_mean_std/"))
        assert agent.num_minibatches == d["T"] * d["N"] // d["MB"] and agent.mini_epochs_num == d["MINI_EPOCHS"]
        agent.init_train()
        lr, steps_per_epoch = agent.last_lr, agent.num_minibatches * agent.mini_epochs_num
        names = [str(n) for n in g["param_names"]]
        worst_param = 0.0
        for e in range(1, d["EPOCHS"] + 1):
            seen = {}
            orig_prepare = agent.prepare_dataset

            def prepare(b, _orig=orig_prepare, _seen=seen):
                _seen["batch"] = {k: b[k].detach().clone() for k in ("returns", "disc_rewards", "mb_rewards", "terminated_flags", "reward_raw", "amp_obs_demo", "amp_obs_replay")}
                _orig(b)
                _seen["dataset"] = {k: agent.dataset[k].detach().clone() for k in ("old_values", "advantages", "returns", "old_logp_actions")}
                _seen["vms"] = {k: v.detach().clone() for k, v in agent.value_mean_std.state_dict().items()}
            agent.prepare_dataset = prepare
            info = agent.train_epoch()
            agent.prepare_dataset = orig_prepare
            tag = f"ep{e}"
            # ---- P4: the rollout
            for k in ("obses", "next_obses", "amp_obs", "rewards", "sigmas"):        # what the env handed over + the fixed sigma: no network arithmetic in them
                _near(agent.exp[k], g[f"{tag}/exp/{k}"], t["exp"], t["exp"], f"{tag} exp/{k}")
            for k in ("actions", "mus", "values", "next_values"):
                _near(agent.exp[k], g[f"{tag}/exp/{k}"], t.get("net", t["exp"]), t.get("net", t["exp"]), f"{tag} exp/{k}")
            _near(agent.exp["neglogpacs"], g[f"{tag}/exp/neglogpacs"], t.get("net", t["exp"]), t.get("nlp", 10 * t["exp"]), f"{tag} exp/neglogpacs")
            assert np.array_equal(agent.exp["dones"].cpu().numpy(), g[f"{tag}/exp/dones"]), f"{tag} exp/dones"
            _near(agent.current_rewards, g[f"{tag}/current_rewards"], 1e-5, 1e-5, f"{tag} current_rewards")
            assert np.array_equal(agent.current_lengths.cpu().numpy(), g[f"{tag}/current_lengths"]), f"{tag} current_lengths"
            for k in ("returns", "disc_rewards", "mb_rewards", "reward_raw"):
                bt = t["exp"] if k == "reward_raw" else t.get("net", t["exp"])
                _near(seen["batch"][k], g[f"{tag}/batch/{k}"], 10 * bt, 10 * bt, f"{tag} batch/{k}")
            assert np.array_equal(seen["batch"]["terminated_flags"].cpu().numpy(), g[f"{tag}/batch/terminated_flags"])
            # ---- P9: what the two ring buffers handed out (bit-exact: index arithmetic on stored rows)
            for k in ("amp_obs_demo", "amp_obs_replay"):
                assert np.array_equal(seen["batch"][k].cpu().numpy(), g[f"{tag}/batch/{k}"]), f"{tag} batch/{k}"
            # ---- P7: the dataset
            for k in ("old_values", "advantages", "returns", "old_logp_actions"):
                want = g[f"{tag}/dataset/{k}"]
                dt_ = t.get("nlp", 20 * t["exp"]) if k == "old_logp_actions" else 20 * t.get("net", t["exp"])
                _near(seen["dataset"][k].reshape(want.shape), want, 20 * t.get("net", t["exp"]), dt_, f"{tag} dataset/{k}")
            for k, v in _sub(g, f"{tag}/reward_mean_std_after_prepare/").items():
                _near(seen["vms"][k], v.numpy(), t.get("vstat", t["stat"]), t.get("vstat", t["stat_a"]), f"{tag} value normaliser after prepare_dataset: {k}")
            # ---- P8 x 8: every optimizer step's scalars, in order (a wrong minibatch slice or shuffle would show here)
            keys = [str(k) for k in g["step_keys"]]
            ours = {"actor_loss": "actor_loss", "critic_loss": "critic_loss", "b_loss": "b_loss", "entropy": "entropy", "kl": "kl", "actor_clip_frac": "actor_clip_frac",
                    "disc_loss": "disc_loss", "disc_grad_penalty": "disc_grad_penalty", "disc_logit_loss": "disc_logit_loss", "disc_agent_acc": "disc_agent_acc",
                    "disc_demo_acc": "disc_demo_acc", "disc_agent_logit_mean": "disc_agent_logit", "disc_demo_logit_mean": "disc_demo_logit"}
            trace = info["minibatch_trace"]
            want = g[f"{tag}/steps"]
            assert want.shape == (steps_per_epoch, len(keys)) and len(trace["kl"]) == steps_per_epoch
            for j, k in enumerate(keys):
                if k in t.get("skip_scalars", ()):
                    continue
                got = np.asarray(trace[ours[k]])
                if k in ("actor_clip_frac", "disc_agent_acc", "disc_demo_acc"):     # counts over 64 / 32 rows: at most one row on the other side of its threshold
                    assert np.abs(got - want[:, j]).max() <= t.get("count_slack", 0.0) + 1e-7, f"{tag} step {k}: {got} vs {want[:, j]}"
                else:
                    scale = np.abs(want[:, j]).max()
                    _near(got, want[:, j], t["scal_r"], t["scal_a"] + t["scal_r"] * 0.05 * scale, f"{tag} step scalars: {k}")
            # ---- state after the epoch
            after = _sub(g, f"{tag}/model/")
            sd = agent.model.state_dict()
            assert list(dict(agent.model.named_parameters())) == names
            for n, p in sd.items():
                err = (p.detach().float().cpu() - after[n]).abs()
                if bf16:   # bf16 GEMMs: every gradient element carries a few per cent of rounding noise; the UPDATE has to agree in the mean, element for element it cannot
                    moved = (after[n] - _sub(g, "model/")[n]).abs().mean()
                    assert float(err.max()) <= 2.0 * lr * steps_per_epoch * e + 1e-7 and float(err.mean()) <= t["param_mean"] * float(moved) + 1e-9, \
                        f"{tag} param {n}: mean error {float(err.mean()):.3e} vs mean update {float(moved):.3e}"
                    worst_param = max(worst_param, float(err.mean()) / max(float(moved), 1e-12))
                    continue
                # Adam moves every element by ~lr per step whatever the gradient's size: an element whose gradient is within rounding of zero may step the other way
                assert float(err.max()) <= 2.0 * lr * steps_per_epoch * e + 1e-7, f"{tag} param {n}: worst {float(err.max()):.3e}"
                frac_off = float((err > 0.05 * lr).float().mean())
                assert frac_off <= t["param_tail"], f"{tag} param {n}: {frac_off:.4f} of the elements differ by more than 5 % of one Adam step"
                worst_param = max(worst_param, float(err.max()) / lr)
            for nm, mod in (("running_mean_std", agent.running_mean_std), ("running_mean_std_temp", agent.running_mean_std_temp), ("reward_mean_std", agent.value_mean_std),
                            ("amp_input_mean_std", agent._amp_input_mean_std)):
                for k, v in _sub(g, f"{tag}/{nm}/").items():
                    st_ = (t.get("vstat", t["stat"]), t.get("vstat", t["stat_a"])) if nm == "reward_mean_std" else (t["stat"], t["stat_a"])
                    _near(getattr(mod, k), v.numpy(), st_[0], st_[1], f"{tag} {nm}.{k}")
            for nm, buf in (("replay", agent._amp_replay_buffer), ("demo", agent._amp_obs_demo_buffer)):
                assert np.array_equal(buf._data_buf["amp_obs"].cpu().numpy(), g[f"{tag}/{nm}/data"]), f"{tag} {nm} buffer contents"
                assert [buf._head, buf._total_count, buf._sample_head] == list(g[f"{tag}/{nm}/head_count_samplehead"]), f"{tag} {nm} buffer counters"
            # ---- the scalars of the epoch under the reference's tags (amp_agent.py:900-933, common_agent.py:603-626)
            tags = agent.assemble_train_info(info)
            for k, v in zip((str(x) for x in g[f"{tag}/tags"]), g[f"{tag}/tag_values"]):
                assert k in tags, f"tag {k} of the reference is not logged"
                if k.split("/")[1] in t.get("skip_scalars", ()):
                    continue
                slack = t.get("count_slack", 0.0) if k in ("loss/clip_frac", "disc/agent_acc", "disc/demo_acc") else 0.0
                assert abs(tags[k] - v) <= t["scal_a"] + slack + t["scal_r"] * max(abs(v), 0.05), f"{tag} {k}: {tags[k]} vs {v}"
        # ---- the whole run
        got_actions = torch.stack(env.actions_seen).cpu().numpy()
        np.testing.assert_allclose(got_actions, g["actions_seen"], rtol=t.get("net", t["exp"]), atol=t.get("net", t["exp"]))
        assert np.abs(got_actions).max() <= 1.0
        assert env.k == d["EPOCHS"] * d["T"] and env.demo_k == env.s["demo"].shape[0]
        # (replayed rollout segments read the noise buffer without a host-side draw: epoch 1 eager + epoch 2 capture make the calls)
        assert draws.noise_used >= (2 if graphs else d["EPOCHS"]) * d["T"] and draws.all_consumed(), "the agent drew fewer permutations / masks than the reference"
        if faithful_reset or not with_reset_done:
            assert env.reset_counts == list(g["reset_counts"])
        opt = agent.get_full_state_weights()["optimizer"]
        assert sorted(opt["state"]) == list(g["opt/state_ids"])
        for i in g["opt/state_ids"]:
            assert float(opt["state"][int(i)]["step"]) == float(g[f"opt/{i}/step"]) == d["EPOCHS"] * steps_per_epoch
            if bf16:
                continue
            m_ref = g[f"opt/{i}/exp_avg"]
            _near(opt["state"][int(i)]["exp_avg"], m_ref, 50 * t["scal_r"], 50 * t["scal_r"] * np.abs(m_ref).max() + 1e-9, f"Adam exp_avg {i}")
    return worst_param


@pytest.mark.parametrize("mode", ["reset_done", "faithful_reset", "no_reset_done"])
def test_three_epochs_equal_the_reference_agent_cpu(golden, mode):
    g = golden("learner_epoch")
    assert (np.abs(g["ep1/exp/actions"]) > 1.0).any(), "fixture: some sampled actions must leave [-1, 1] so that the clamp is exercised"
    worst = run_epochs(g, "cpu", faithful_reset=(mode == "faithful_reset"), with_reset_done=(mode != "no_reset_done"))
    print(f"{mode}: worst parameter difference after three epochs = {worst:.3f} Adam steps")


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
def test_three_epochs_equal_the_reference_agent_on_hip(golden, graph):
    """The product path: phc_running_norm, FastLinear, phc_policy_sample, phc_rollout_bookkeeping, phc_gae, phc_ppo_loss, phc_disc_bce, phc_weighted_sumsq,
    phc_adam_clip_step through the C ABI, fp32 GEMMs; `graph`: the rollout's policy / bookkeeping segments (from epoch 2) and the optimizer step replayed from hipGraphs."""
    g = golden("learner_epoch")
    extra = ["+learning.params.config.hip_graph=True", "+learning.params.config.hip_graph_min_rows=1"] if graph else ["+learning.params.config.hip_graph=False"]
    worst = run_epochs(g, "cuda", extra=extra, graphs=graph, tol=dict(exp=2e-4, scal_r=2e-3, scal_a=2e-5, stat=2e-6, stat_a=1e-7, param_tail=2e-2, count_slack=1.0 / 32))   # (statistics: batch moments summed in another order)
    torch.cuda.synchronize()
    print(f"hip graph={graph}: worst parameter difference after three epochs = {worst:.3f} Adam steps")


@pytest.mark.gpu
def test_three_epochs_of_the_reference_agent_on_hip_with_bf16_gemms(golden):
    """The BENCHMARKED path -- bf16 MFMA GEMMs under autocast, captured graphs -- through the same three epochs: everything that involves no network arithmetic stays exact (what the env handed over, done
    flags, both ring buffers' contents and counters, the observation / AMP normalisers, which permutations and masks were drawn), network outputs agree within bf16 accuracy against sigma = 0.055 (mu / values
    3e-2, neglogp 2.0 of ~10, the value normaliser 5e-2), every optimizer step's critic / bound / discriminator scalars within 15 %, and the parameter UPDATE of each epoch agrees with the reference's in the
    mean (mean |error| <= 0.35 x mean |update|; measured 0.195).  The actor loss and the KL of a 64-row minibatch are not compared in bf16 (a mean of signed terms that nearly cancels: tests/test_learner_parity.py)."""
    g = golden("learner_epoch")
    worst = run_epochs(g, "cuda", extra=["+learning.params.config.hip_graph=True", "+learning.params.config.hip_graph_min_rows=1"], graphs=True, bf16=True,
                       tol=dict(exp=2e-4, net=3e-2, nlp=2.0, scal_r=0.15, scal_a=2e-3, stat=2e-6, stat_a=1e-7, vstat=5e-2, param_mean=0.35, count_slack=6.0 / 32,
                                skip_scalars=("actor_loss", "kl")))
    torch.cuda.synchronize()
    print(f"bf16 GEMMs: worst mean parameter error / mean update over the three epochs = {worst:.3f}")
