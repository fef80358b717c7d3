"""TEST INFRASTRUCTURE ONLY -- the few pieces of `rl-games==1.1.4` (reference `requirement.txt:27`, not vendored under
/root/reference and not installed here) that the reference's learner needs in order to RUN its own method bodies.

PARITY UNPINNED for what is in THIS file: these are restatements of the published rl-games 1.1.4 sources
(`rl_games/common/object_factory.py`, `rl_games/algos_torch/torch_ext.py: policy_kl, shape_whc_to_cwh`,
`rl_games/algos_torch/models.py: BaseModel, ModelA2CContinuousLogStd`) -- nothing under /root/reference can pin them.
Everything PHC itself owns on the learner path is NOT restated: with these stand-ins registered by `ref_shim.install()`,
`phc/learning/network_builder.py`, `amp_network_builder.py`, `amp_network_pnn_builder.py`, `amp_network_mcp_builder.py`,
`pnn.py`, `amp_models.py`, `common_agent.py`, `amp_agent.py`, `im_amp.py`, `network_loader.py` import and run UNMODIFIED,
and `oracle/gen_golden_learner.py` calls their method bodies (losses, GAE, discriminator reward, `calc_gradients`, the
checkpoint loaders) to produce `tests/golden/learner_*.npz`.

The agent base classes (`a2c_continuous.A2CAgent`, `a2c_common.A2CBase`, players) carry no constructor: the reference's
constructors are never run (they need a simulator); instances are made with `__new__` and given exactly the attributes the
called method reads.  For the whole-epoch fixture (oracle/gen_golden_epoch.py: the reference's `play_steps`, `prepare_dataset`,
`train_epoch`, `AMPDataset`, `ReplayBuffer`, `_store_replay_amp_obs` bodies run unmodified) the base class restates the handful of
rl-games methods those bodies call -- `init_tensors` + `ExperienceBuffer.update_data / get_transformed_list` (rl_games/common/experience.py),
`env_step`, `preprocess_actions`, `obs_to_tensors`, `train_actor_critic`, `update_lr` (a2c_common.py / a2c_continuous.py), `PPODataset`
(common/datasets.py), `DefaultRewardsShaper` (common/tr_helpers.py), `AverageMeter` (algos_torch/torch_ext.py), `IdentityScheduler`
(common/schedulers.py) -- all PARITY UNPINNED like the rest of this file."""
import sys
import types

import numpy as np
import torch
import torch.nn as nn


class ObjectFactory:
    """rl_games/common/object_factory.py: name -> builder registry."""

    def __init__(self):
        self._builders = {}

    def register_builder(self, name, builder):
        self._builders[name] = builder

    def set_builders(self, builders):
        self._builders = builders

    def create(self, name, **kwargs):
        builder = self._builders.get(name)
        if not builder:
            raise ValueError(name)
        return builder(**kwargs)


def policy_kl(p0_mu, p0_sigma, p1_mu, p1_sigma, reduce=True):
    """rl_games/algos_torch/torch_ext.py."""
    c1 = torch.log(p1_sigma / p0_sigma + 1e-5)
    c2 = (p0_sigma ** 2 + (p1_mu - p0_mu) ** 2) / (2.0 * (p1_sigma ** 2 + 1e-5))
    c3 = -1.0 / 2.0
    kl = c1 + c2 + c3
    kl = kl.sum(dim=-1)  # returning mean between all steps of sum between all actions
    if reduce:
        return kl.mean()
    return kl


def shape_whc_to_cwh(shape):
    if len(shape) == 3:
        return (shape[2], shape[0], shape[1])
    return shape


def mean_list(val):
    return torch.mean(torch.stack(val))


def load_checkpoint(filename):
    return torch.load(filename, map_location="cpu", weights_only=False)


class BaseModel:
    def __init__(self):
        pass

    def is_rnn(self):
        return False

    def is_separate_critic(self):
        return False


class ModelA2CContinuousLogStd(BaseModel):
    """rl_games/algos_torch/models.py: Normal(mu, exp(logstd)) policy head around an a2c network."""

    def __init__(self, network):
        BaseModel.__init__(self)
        self.network_builder = network

    def build(self, config):
        net = self.network_builder.build("a2c", **config)
        return ModelA2CContinuousLogStd.Network(net)

    class Network(nn.Module):
        def __init__(self, a2c_network):
            nn.Module.__init__(self)
            self.a2c_network = a2c_network

        def is_rnn(self):
            return self.a2c_network.is_rnn()

        def get_default_rnn_state(self):
            return self.a2c_network.get_default_rnn_state()

        def forward(self, input_dict):
            is_train = input_dict.get("is_train", True)
            prev_actions = input_dict.get("prev_actions", None)
            mu, logstd, value, states = self.a2c_network(input_dict)
            sigma = torch.exp(logstd)
            distr = torch.distributions.Normal(mu, sigma)
            if is_train:
                entropy = distr.entropy().sum(dim=-1)
                prev_neglogp = self.neglogp(prev_actions, mu, sigma, logstd)
                return {"prev_neglogp": torch.squeeze(prev_neglogp), "values": value, "entropy": entropy, "rnn_states": states, "mus": mu,
                        "sigmas": sigma}
            selected_action = distr.sample()
            neglogp = self.neglogp(selected_action, mu, sigma, logstd)
            return {"neglogpacs": torch.squeeze(neglogp), "values": value, "actions": selected_action, "rnn_states": states, "mus": mu,
                    "sigmas": sigma}

        def neglogp(self, x, mean, std, logstd):
            return 0.5 * (((x - mean) / std) ** 2).sum(dim=-1) + 0.5 * np.log(2.0 * np.pi) * x.size()[-1] + logstd.sum(dim=-1)


class _Empty:
    """Stand-in base class: the reference's subclasses only inherit the name."""

    def __init__(self, *a, **k):
        pass


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []   # lets the mock finder of ref_shim serve submodules that are not listed here
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent and parent in sys.modules:
        setattr(sys.modules[parent], leaf, m)
    return m


def register():
    """Put the stand-ins into sys.modules (before ref_shim's catch-all mock finder gets to see `rl_games.*`)."""
    if isinstance(sys.modules.get("rl_games.common.object_factory"), types.ModuleType) and hasattr(sys.modules["rl_games.common.object_factory"], "ObjectFactory") \
            and sys.modules["rl_games.common.object_factory"].ObjectFactory is ObjectFactory:
        return
    for k in [k for k in sys.modules if k == "rl_games" or k.startswith("rl_games.")]:
        del sys.modules[k]
    _module("rl_games")
    _module("rl_games.common")
    _module("rl_games.algos_torch")
    _module("rl_games.common.object_factory", ObjectFactory=ObjectFactory)
    _module("rl_games.algos_torch.torch_ext", policy_kl=policy_kl, shape_whc_to_cwh=shape_whc_to_cwh, mean_list=mean_list,
            load_checkpoint=load_checkpoint)
    _module("rl_games.algos_torch.models", BaseModel=BaseModel, ModelA2CContinuousLogStd=ModelA2CContinuousLogStd)

    class A2CBase(_Empty):
        # rl_games/common/a2c_common.py: A2CBase.set_eval / set_train (the reference's AMPAgent.set_train calls super())
        # ... and, for oracle/gen_golden_epoch.py (the reference's play_steps / prepare_dataset / train_epoch run on a `__new__`-made agent), the
        # few base-class methods those bodies call: obs_to_tensors, preprocess_actions, env_step, train_actor_critic, update_lr
        def obs_to_tensors(self, obs):
            return obs if isinstance(obs, dict) else {"obs": obs}

        def preprocess_actions(self, actions):
            if self.clip_actions:
                return _rescale_actions(self.actions_low, self.actions_high, torch.clamp(actions, -1.0, 1.0))
            return actions

        def env_step(self, actions):
            actions = self.preprocess_actions(actions)
            obs, rewards, dones, infos = self.vec_env.step(actions)
            if self.value_size == 1:
                rewards = rewards.unsqueeze(1)
            return self.obs_to_tensors(obs), rewards.to(self.ppo_device), dones.to(self.ppo_device), infos

        def train_actor_critic(self, input_dict):   # rl_games/algos_torch/a2c_continuous.py
            self.calc_gradients(input_dict)
            return self.train_result

        def update_lr(self, lr):
            for g in self.optimizer.param_groups:
                g["lr"] = lr

        def init_tensors(self):   # A2CBase.init_tensors + ContinuousA2CBase.init_tensors
            n = self.num_agents * self.num_actors
            self.experience_buffer = ExperienceBuffer(self.horizon_length, n, self.obs_shape, self.actions_num, self.value_size, self.ppo_device)
            self.current_rewards = torch.zeros((n, self.value_size), dtype=torch.float32, device=self.ppo_device)
            self.current_lengths = torch.zeros(n, dtype=torch.float32, device=self.ppo_device)
            self.dones = torch.ones((n,), dtype=torch.uint8, device=self.ppo_device)
            self.update_list = ["actions", "neglogpacs", "values", "mus", "sigmas"]
            self.tensor_list = self.update_list + ["obses", "states", "dones"]

        def set_eval(self):
            self.model.eval()
            if self.normalize_input:
                self.running_mean_std.eval()
            if self.normalize_value:
                self.value_mean_std.eval()

        def set_train(self):
            self.model.train()
            if self.normalize_input:
                self.running_mean_std.train()
            if self.normalize_value:
                self.value_mean_std.train()

    class ContinuousA2CBase(A2CBase):
        pass

    class A2CAgent(ContinuousA2CBase):
        pass

    class DiscreteA2CAgent(A2CBase):
        pass

    class PPODataset:
        """rl_games/common/datasets.py: the minibatch view of one rollout (the reference's AMPDataset overrides _get_item / the shuffling)."""

        def __init__(self, batch_size, minibatch_size, is_discrete, is_rnn, device, seq_len):
            self.is_rnn, self.seq_len, self.batch_size, self.minibatch_size, self.device = is_rnn, seq_len, batch_size, minibatch_size, device
            self.length = self.batch_size // self.minibatch_size
            self.is_discrete, self.is_continuous = is_discrete, not is_discrete
            self.special_names = ["rnn_states"]

        def update_values_dict(self, values_dict):
            self.values_dict = values_dict

        def __len__(self):
            return self.length

        def __getitem__(self, idx):
            return self._get_item_rnn(idx) if self.is_rnn else self._get_item(idx)

    class BasePlayer(_Empty):
        pass

    class PpoPlayerContinuous(BasePlayer):
        pass

    _module("rl_games.common.a2c_common", A2CBase=A2CBase, ContinuousA2CBase=ContinuousA2CBase, swap_and_flatten01=_swap_and_flatten01)
    _module("rl_games.algos_torch.a2c_continuous", A2CAgent=A2CAgent)
    _module("rl_games.algos_torch.a2c_discrete", DiscreteA2CAgent=DiscreteA2CAgent)
    _module("rl_games.common.datasets", PPODataset=PPODataset)
    _module("rl_games.common.player", BasePlayer=BasePlayer)
    _module("rl_games.algos_torch.players", PpoPlayerContinuous=PpoPlayerContinuous, rescale_actions=_rescale_actions)


class ExperienceBuffer:
    """rl_games/common/experience.py, the part a continuous-action, single-agent, non-recurrent rollout uses: [T, N, ...] tensors by name."""

    def __init__(self, horizon_length, num_actors, obs_shape, actions_num, value_size, device):
        self.obs_base_shape = (horizon_length, num_actors)
        z = lambda *s, dtype=torch.float32: torch.zeros(self.obs_base_shape + s, dtype=dtype, device=device)
        self.tensor_dict = {"obses": z(*obs_shape), "rewards": z(value_size), "values": z(value_size), "neglogpacs": z(),
                            "dones": z(dtype=torch.uint8), "actions": z(actions_num), "mus": z(actions_num), "sigmas": z(actions_num)}

    def update_data(self, name, index, val):
        self.tensor_dict[name][index, :] = val

    def get_transformed_list(self, transform_op, tensor_list):
        res = {}
        for k in tensor_list:
            v = self.tensor_dict.get(k)
            if v is None:
                continue
            res[k] = transform_op(v)
        return res


class DefaultRewardsShaper:
    """rl_games/common/tr_helpers.py."""

    def __init__(self, scale_value=1, shift_value=0, min_val=-np.inf, max_val=np.inf, is_torch=True):
        self.scale_value, self.shift_value, self.min_val, self.max_val, self.is_torch = scale_value, shift_value, min_val, max_val, is_torch

    def __call__(self, reward):
        reward = reward + self.shift_value
        reward = reward * self.scale_value
        if self.is_torch:
            import torch
            reward = torch.clamp(reward, self.min_val, self.max_val)
        else:
            reward = np.clip(reward, self.min_val, self.max_val)
        return reward


class AverageMeter(nn.Module):
    """rl_games/algos_torch/torch_ext.py: running mean of the last `max_size` episode scores."""

    def __init__(self, in_shape, max_size):
        super().__init__()
        self.max_size, self.current_size = max_size, 0
        self.register_buffer("mean", torch.zeros(in_shape, dtype=torch.float32))

    def update(self, values):
        size = values.size()[0]
        if size == 0:
            return
        new_mean = torch.mean(values.float(), dim=0)
        size = np.clip(size, 0, self.max_size)
        old_size = min(self.max_size - size, self.current_size)
        size_sum = old_size + size
        self.current_size = size_sum
        self.mean = (self.mean * old_size + new_mean * size) / size_sum

    def get_mean(self):
        return self.mean.squeeze(0).cpu().numpy()


class IdentityScheduler:
    """rl_games/common/schedulers.py (`lr_schedule: constant`, im.yaml:60)."""

    def update(self, current_lr, entropy_coef, epoch, frames, kl_dist, **kwargs):
        return current_lr, entropy_coef


def _swap_and_flatten01(arr):
    """rl_games/common/a2c_common.py."""
    if arr is None:
        return arr
    s = arr.size()
    return arr.transpose(0, 1).reshape(s[0] * s[1], *s[2:])


def _rescale_actions(low, high, action):
    """rl_games/algos_torch/players.py."""
    d = (high - low) / 2.0
    m = (high + low) / 2.0
    return action * d + m
