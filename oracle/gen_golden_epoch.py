"""TEST INFRASTRUCTURE -- golden for ONE WHOLE LEARNER EPOCH (SURVEY.md 8a rows P4, P7, P9, and P5 / P6 / P8 in sequence) from the reference's OWN method bodies.

Run in the build container (needs /root/reference):  python oracle/gen_golden_epoch.py      -> tests/golden/learner_epoch.npz
Deterministic (every draw comes from seeded generators): re-running reproduces the committed file bit for bit.

What runs, unmodified, on a `__new__`-made `phc.learning.im_amp.IMAmpAgent` (constructors need a simulator; the instance gets exactly the attributes the bodies read):

  CommonAgent.init_tensors / AMPAgent.init_tensors -> _build_amp_buffers (common_agent.py:92-98, amp_agent.py:130-135,810-823)
  CommonAgent.env_reset (:507-510), AMPAgent._init_amp_demo_buf (:825-833)
  three times  AMPAgent.train_epoch (amp_agent.py:413-504):
     pre_epoch (:506-528)                     frozen copy of the observation normaliser
     play_steps (:309-397)                    get_action_values (common_agent.py:262-288), _eval_critic (:552-562), next_values zeroed on `terminate`,
                                              _calc_amp_rewards / _combine_rewards (:848-878), discount_values (common_agent.py:493-505), swap_and_flatten01
     _update_amp_demos (:835-838), ReplayBuffer.sample / store (replay_buffer.py:3-84) of the demo and the replay buffer
     prepare_dataset (:399-411 + common_agent.py:357-398)   _calc_advs, value / return normalisation IN TRAINING MODE (two moment updates), AMPDataset.update_values_dict
     2 mini-epochs x 4 minibatches of train_actor_critic -> calc_gradients (:554-688) on AMPDataset._get_item / _shuffle_idx_buf (amp_datasets.py:72-99) slices
     _store_replay_amp_obs (:880-894)         all three branches over the three epochs: plain store, random subset when the rollout exceeds the buffer,
                                              Bernoulli keep mask once the buffer has wrapped
     post_epoch (:530-532)

against a SCRIPTED vec-env: pre-drawn streams of observations, rewards, dones, `terminate`, `reward_raw`, AMP observations (the returned tensors are views of
buffers that the next step overwrites, like the real task's), reset observations for the envs in `done_indices`, and the demo batches `fetch_amp_obs_demo`
hands out.  The env ignores the actions' values but RECORDS them (what `preprocess_actions` clamped).

Randomness is taken from the fixture instead of the global generator, so the other side can replay it: `torch.normal` (the policy's `Normal.sample()`) returns
mean + std * noise[k]; `torch.randperm(n)` and `torch.bernoulli(p)` draw from a seeded generator and every result is recorded per size n, in call order.

What is NOT the reference's code here: the rl-games 1.1.4 pieces of oracle/rl_games_stub.py (ExperienceBuffer, PPODataset, env_step, ... -- restated, unpinned) and the scripted env."""
import copy
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

ref_shim.install()
import gen_golden_learner as gl  # noqa: E402
import rl_games_stub as rg  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

O, M, A = 40, 36, 9           # obs, amp obs (3 steps x 12), actions
T, N = 8, 32                  # horizon, envs  -> batch 256
MB, AMB = 64, 32              # minibatch (4 per mini-epoch), amp minibatch
MINI_EPOCHS = 2
AMP_BATCH, DEMO_BUF, REPLAY_BUF, KEEP_PROB = 24, 300, 200, 0.5
EPOCHS = 3
UNITS, DISC_UNITS = (64, 32), (48, 24)


class ScriptedEnv:
    """The `vec_env` the reference agent talks to (B1 surface: reset / step; `.env.fetch_amp_obs_demo`, `.env.task.*`)."""

    def __init__(self, s):
        self.s, self.k, self.demo_k = s, 0, 0
        self.obs_buf = s["obs0"].clone()
        self.rew_buf, self.reset_buf = torch.zeros(N), torch.zeros(N, dtype=torch.long)
        self.terminate_buf, self.raw_buf, self.amp_buf = torch.zeros(N, dtype=torch.long), torch.zeros(N, 5), torch.zeros(N, M)
        self.actions_seen, self.reset_ids_seen = [], []
        task = types.SimpleNamespace(humanoid_type="smpl", shape_resampling_interval=500, getup_schedule=False, viewer=None, _num_amp_obs_steps=3,
                                     temp_running_mean=True)
        self.env = types.SimpleNamespace(task=task, fetch_amp_obs_demo=self.fetch_amp_obs_demo)

    def reset(self, env_ids=None):
        if env_ids is not None and len(env_ids) > 0:
            self.reset_ids_seen.append(torch.as_tensor(env_ids).clone())
            self.obs_buf[env_ids] = self.s["reset_obs"][self.k - 1][env_ids]
        return self.obs_buf

    def step(self, actions):
        s, k = self.s, self.k
        self.actions_seen.append(actions.clone())
        self.obs_buf.copy_(s["obs"][k])
        self.rew_buf.copy_(s["rewards"][k])
        self.reset_buf.copy_(s["dones"][k])
        self.terminate_buf.copy_(s["terminate"][k])
        self.raw_buf.copy_(s["reward_raw"][k])
        self.amp_buf.copy_(s["amp_obs"][k])
        self.k += 1
        return self.obs_buf, self.rew_buf, self.reset_buf, {"terminate": self.terminate_buf, "reward_raw": self.raw_buf, "amp_obs": self.amp_buf}

    def fetch_amp_obs_demo(self, n):
        assert n == AMP_BATCH
        self.demo_k += 1
        return self.s["demo"][self.demo_k - 1]


def make_script(g):
    K = EPOCHS * T
    dones = (torch.rand(K, N, generator=g) < 0.12).long()
    dones[3, 5] = dones[3, 6] = 1                                   # some envs finish mid-rollout in every epoch, whatever the draw
    dones[T + 2, 0] = dones[2 * T + 5, 7] = 1
    terminate = dones * (torch.rand(K, N, generator=g) < 0.6).long()   # `terminate` only where done; some dones are time-outs (not terminated)
    terminate[3, 5], terminate[3, 6] = 1, 0
    n_demo = int(np.ceil(DEMO_BUF / AMP_BATCH)) + EPOCHS
    return {"obs0": torch.randn(N, O, generator=g) * 2 + 0.5, "obs": torch.randn(K, N, O, generator=g) * 2 + 0.5,
            "reset_obs": torch.randn(K, N, O, generator=g) * 1.5 - 0.3, "rewards": torch.rand(K, N, generator=g),
            "dones": dones, "terminate": terminate, "reward_raw": torch.rand(K, N, 5, generator=g),
            "amp_obs": torch.randn(K, N, M, generator=g) * 1.5 - 0.2, "demo": torch.randn(n_demo, AMP_BATCH, M, generator=g) * 1.2 + 0.3}


class Draws:
    """torch.normal / randperm / bernoulli from the fixture's generator, recorded."""

    def __init__(self, g):
        self.g, self.noise, self.perms, self.masks = g, [], {}, []

    def normal(self, mean, std, *a, **k):
        eps = torch.randn(mean.shape, generator=self.g)
        self.noise.append(eps)
        return mean + std * eps

    def randperm(self, n, *a, **k):
        p = self._randperm(n, generator=self.g)
        self.perms.setdefault(int(n), []).append(p.clone())
        return p

    def bernoulli(self, p, *a, **k):
        m = (torch.rand(p.shape, generator=self.g) < p).to(p.dtype)
        self.masks.append(m.clone())
        return m

    def __enter__(self):
        self._normal, self._randperm, self._bernoulli = torch.normal, torch.randperm, torch.bernoulli
        torch.normal, torch.randperm, torch.bernoulli = self.normal, self.randperm, self.bernoulli
        return self

    def __exit__(self, *a):
        torch.normal, torch.randperm, torch.bernoulli = self._normal, self._randperm, self._bernoulli


def stats(m):
    return gl.np_state(m.state_dict())


def main():
    ia = ref_shim.ref_module("phc.learning.im_amp")
    ds = ref_shim.ref_module("learning.amp_datasets")
    gl.O, gl.M, gl.A, gl.N = O, M, A, N
    torch.manual_seed(31)
    params = gl.net_params("im.yaml", UNITS, DISC_UNITS)
    model, rms, rms_mod = gl.build_reference_model(params, "phc.learning.amp_network_builder", "AMPBuilder")
    g = torch.Generator().manual_seed(404)
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith("bias"):     # (the policy head's bias large enough that some sampled actions leave [-1, 1]: the env must see them clamped, the buffer unclamped)
                p.copy_(torch.randn(p.shape, generator=g) * (0.8 if n_ == "a2c_network.mu.bias" else 0.05))
    vms, ams = rms_mod.RunningMeanStd((1,)), rms_mod.RunningMeanStd((M,))
    gl.warm_stats(rms, O, g)
    gl.warm_stats(vms, 1, g, scale=0.5, shift=0.2)
    gl.warm_stats(ams, M, g, scale=1.5, shift=-0.2)
    script = make_script(g)
    env = ScriptedEnv(script)

    c = params["config"]
    c.update(horizon_length=T, minibatch_size=MB, mini_epochs=MINI_EPOCHS, amp_batch_size=AMP_BATCH, amp_minibatch_size=AMB,
             amp_obs_demo_buffer_size=DEMO_BUF, amp_replay_buffer_size=REPLAY_BUF, amp_replay_keep_prob=KEEP_PROB)
    a = ia.IMAmpAgent.__new__(ia.IMAmpAgent)
    a.config, a.vec_env = c, env
    a.model, a.running_mean_std, a.value_mean_std, a._amp_input_mean_std = model, rms, vms, ams
    a.normalize_input, a.normalize_value, a._normalize_amp_input, a._disc_reward_mean_std = True, True, True, None
    a.ppo_device = a.device = "cpu"
    a.num_agents, a.num_actors, a.value_size, a.obs_shape, a.actions_num = 1, N, 1, (O,), A
    a.horizon_length, a.gamma, a.tau = T, c["gamma"], c["tau"]
    a.batch_size, a.minibatch_size, a.mini_epochs_num = T * N, MB, MINI_EPOCHS
    a.e_clip, a.critic_coef, a.entropy_coef, a.bounds_loss_coef = c["e_clip"], c["critic_coef"], c["entropy_coef"], c["bounds_loss_coef"]
    a.clip_value, a.truncate_grads, a.grad_norm, a.normalize_advantage = c["clip_value"], c["truncate_grads"], c["grad_norm"], c["normalize_advantage"]
    a.last_lr = float(c["learning_rate"])
    a.clip_actions, a.actions_low, a.actions_high = True, -torch.ones(A), torch.ones(A)
    a.is_rnn, a.rnn_states, a.mixed_precision, a.multi_gpu, a.has_central_value, a.use_action_masks = False, None, False, False, False, False
    a.schedule_type, a.scheduler = "legacy", rg.IdentityScheduler()      # rl-games default schedule_type; `lr_schedule: constant` (im.yaml:60)
    a.rewards_shaper = rg.DefaultRewardsShaper(**c["reward_shaper"])
    a.game_rewards, a.game_lengths = rg.AverageMeter(1, 100), rg.AverageMeter(1, 100)
    a.algo_observer = types.SimpleNamespace(process_infos=lambda *x: None, after_steps=lambda: None)
    a.temp_running_mean = env.env.task.temp_running_mean
    a.scaler = torch.cuda.amp.GradScaler(enabled=False)
    a.optimizer = torch.optim.Adam(model.parameters(), a.last_lr, eps=1e-08, weight_decay=c.get("weight_decay", 0.0))     # common_agent.py:67
    a.env_info = {"amp_observation_space": types.SimpleNamespace(shape=(M,))}
    ia.amp_agent.AMPAgent._load_config_params(a, c)                    # the AMP coefficients, as the reference reads them from the yaml (amp_agent.py:690-707)
    a.last_lr = float(a.last_lr)                                       # common_agent.py:64 (the yaml's `2e-5` is a string)
    a.epoch_num = 0

    out = {"model/" + k: v for k, v in gl.np_state(model.state_dict()).items()}
    for nm, m in (("running_mean_std", rms), ("reward_mean_std", vms), ("amp_input_mean_std", ams)):
        out.update({f"{nm}/" + k: v for k, v in stats(m).items()})
    out.update({"script/" + k: v.numpy() for k, v in script.items()})
    out["param_names"] = np.array([n_ for n_, _ in model.named_parameters()])

    with Draws(g) as draws:
        a.dataset = ds.AMPDataset(a.batch_size, a.minibatch_size, False, False, "cpu", 1)      # common_agent.py:89
        a.init_tensors()
        a.obs = a.env_reset()
        a._init_amp_demo_buf()
        rec = {}
        cls = type(a)

        def play_steps():
            b = cls.play_steps(a)
            e = rec["e"]
            for k, v in a.experience_buffer.tensor_dict.items():
                out[f"ep{e}/exp/{k}"] = v.numpy().copy()
            for k in ("returns", "disc_rewards", "mb_rewards", "terminated_flags", "reward_raw"):
                out[f"ep{e}/batch/{k}"] = b[k].numpy().copy()
            out[f"ep{e}/current_rewards"], out[f"ep{e}/current_lengths"] = a.current_rewards.numpy().copy(), a.current_lengths.numpy().copy()
            return b

        def prepare_dataset(batch_dict):
            e = rec["e"]
            out[f"ep{e}/batch/amp_obs_demo"], out[f"ep{e}/batch/amp_obs_replay"] = batch_dict["amp_obs_demo"].numpy().copy(), batch_dict["amp_obs_replay"].numpy().copy()
            cls.prepare_dataset(a, batch_dict)
            for k in ("old_values", "advantages", "returns", "old_logp_actions"):
                out[f"ep{e}/dataset/{k}"] = a.dataset.values_dict[k].numpy().copy()
            out.update({f"ep{e}/reward_mean_std_after_prepare/" + k: v for k, v in stats(vms).items()})

        def train_actor_critic(d):
            e = rec["e"]
            rec["mb"].append({k: v.numpy().copy() for k, v in d.items() if k in ("obs", "advantages")})
            tr = cls.train_actor_critic(a, d)
            rec["steps"].append([float(tr[k]) for k in STEP_KEYS] + [float(tr["disc_agent_logit"].mean()), float(tr["disc_demo_logit"].mean())])
            return tr

        a.play_steps, a.prepare_dataset, a.train_actor_critic = play_steps, prepare_dataset, train_actor_critic
        for e in range(1, EPOCHS + 1):
            rec.update(e=e, steps=[], mb=[])
            a.epoch_num += 1                                         # A2CBase.update_epoch
            info = a.train_epoch()
            out[f"ep{e}/steps"] = np.array(rec["steps"], dtype=np.float64)                # [mini_epochs * minibatches, len(STEP_KEYS) + 2]
            out[f"ep{e}/minibatch_obs_row0"] = np.stack([m["obs"][0] for m in rec["mb"]])   # which rows each minibatch started with
            out.update({f"ep{e}/model/" + k: v for k, v in gl.np_state(model.state_dict()).items()})
            for nm, m in (("running_mean_std", rms), ("running_mean_std_temp", a.running_mean_std_temp), ("reward_mean_std", vms), ("amp_input_mean_std", ams)):
                out.update({f"ep{e}/{nm}/" + k: v for k, v in stats(m).items()})
            for nm, buf in (("replay", a._amp_replay_buffer), ("demo", a._amp_obs_demo_buffer)):
                out[f"ep{e}/{nm}/data"] = buf._data_buf["amp_obs"].numpy().copy()
                out[f"ep{e}/{nm}/head_count_samplehead"] = np.array([buf._head, buf._total_count, buf._sample_head])
            assert np.isfinite(out[f"ep{e}/steps"]).all()
            # the scalars the reference would log for this epoch (amp_agent.py:900-933, common_agent.py:603-626)
            tags = a._assemble_train_info(info, 0)
            keep = [k for k in sorted(tags) if k.split("/")[0] in ("loss", "disc", "rewards")]
            out[f"ep{e}/tags"], out[f"ep{e}/tag_values"] = np.array(keep), np.array([tags[k] for k in keep], dtype=np.float64)
        sd = a.optimizer.state_dict()
        for i, st in sd["state"].items():
            out[f"opt/{i}/exp_avg"], out[f"opt/{i}/exp_avg_sq"], out[f"opt/{i}/step"] = st["exp_avg"].numpy(), st["exp_avg_sq"].numpy(), np.asarray(float(st["step"]))
        out["opt/state_ids"] = np.array(sorted(sd["state"]))

    out["actions_seen"] = torch.stack(env.actions_seen).numpy()
    out["noise"] = torch.stack(draws.noise).numpy()
    for n_, ps in draws.perms.items():
        out[f"perm/{n_}"] = torch.stack(ps).numpy()
    for i, m in enumerate(draws.masks):
        out[f"mask/{i}"] = m.numpy()
    out["n_masks"] = np.asarray(len(draws.masks))
    out["reset_counts"] = np.array([len(x) for x in env.reset_ids_seen])
    out["step_keys"] = np.array(STEP_KEYS + ["disc_agent_logit_mean", "disc_demo_logit_mean"])
    out["dims"] = np.array([O, M, A, T, N, MB, AMB, MINI_EPOCHS, AMP_BATCH, DEMO_BUF, REPLAY_BUF, EPOCHS])
    out["keep_prob"], out["units"], out["disc_units"] = np.asarray(KEEP_PROB), np.array(UNITS), np.array(DISC_UNITS)
    assert env.k == EPOCHS * T and env.demo_k == script["demo"].shape[0]
    assert len(draws.masks) >= 1, "the Bernoulli branch of _store_replay_amp_obs was not reached"
    np.savez_compressed(os.path.join(OUT, "learner_epoch.npz"), **out)
    print("wrote learner_epoch.npz:", {k: len(v) for k, v in draws.perms.items()}, "perms,", len(draws.masks), "masks,", len(draws.noise), "noise draws")


STEP_KEYS = ["actor_loss", "critic_loss", "b_loss", "entropy", "kl", "actor_clip_frac", "disc_loss", "disc_grad_penalty", "disc_logit_loss", "disc_agent_acc",
             "disc_demo_acc"]

if __name__ == "__main__":
    main()
