"""P2: actor / critic / discriminator networks of the AMP agent.

Architecture and parameter NAMES follow the reference (`AMPBuilder.Network`, phc/learning/amp_network_builder.py:17-249,
on rl_games' A2CBuilder / ModelA2CContinuousLogStd): separate actor and critic MLPs, a linear `mu` head, a fixed
non-trainable log-sigma (-2.9, learn_sigma False), a linear `value` head, and the `_disc_mlp` + `_disc_logits`
discriminator.  The state-dict keys are the ones the reference's checkpoints use
(`a2c_network.actor_mlp.0.weight`, `a2c_network.mu.weight`, `a2c_network.sigma`, `a2c_network.critic_mlp.*`,
`a2c_network.value.*`, `a2c_network._disc_mlp.*`, `a2c_network._disc_logits.*`).

MI355X: weights are fp32 masters; the GEMMs run as bf16 MFMA under torch.autocast (hipBLASLt) -- the only
dense contraction on the path."""
import math
import os

import torch
from torch import nn

from .fast_ops import FastLinear, FastLinear1DD, FastLinearDD, FusedReLU

DISC_LOGIT_INIT_SCALE = 1.0  # amp_network_builder.py:12

_ACT = {"relu": nn.ReLU, "silu": nn.SiLU, "tanh": nn.Tanh, "elu": nn.ELU, "gelu": nn.GELU, "None": nn.Identity}


def pad_cols(k):
    """GEMM-friendly K for a first-layer input width (obs 934 -> 1024, AMP obs 1960 -> 2048: hipBLASLt's forward / weight-gradient kernels run
    20-30 % faster on multiples of 128 at these sizes, scripts/probes/gemm_pad_probe.py); 0 = leave as is."""
    if os.environ.get("PHC_NO_K_PAD") or k < 512 or k % 128 == 0:
        return 0
    return (k + 127) // 128 * 128


def build_mlp(input_size, units, activation, linear=nn.Linear):
    """network_builder.py:126-137 `_build_sequential_mlp`: Linear, act, Linear, act ... (indices 0,2,4,..)."""
    layers, n = [], input_size
    for u in units:
        lin = linear(n, u)
        if not layers and isinstance(lin, (FastLinear, FastLinearDD)):
            lin.weight._pad_cols = pad_cols(n)   # first layer: store the weight K-padded on the device (FlatGradBucket)
        if activation == "relu" and isinstance(lin, (FastLinear, FastLinearDD)) and not os.environ.get("PHC_NO_RELU_FUSION"):   # the ReLU rides in the GEMM epilogue of the device passes
            lin.fuse_relu = True
            layers += [lin, FusedReLU(lin)]
        else:
            layers += [lin, _ACT[activation]()]
        n = u
    return nn.Sequential(*layers)


class A2CNetwork(nn.Module):
    def __init__(self, params, actions_num, input_shape, amp_input_shape, value_size=1):
        super().__init__()
        mlp, disc = params["mlp"], params["disc"]
        assert params.get("separate", True), "the shipped PHC configs use separate actor / critic networks"
        space = params["space"]["continuous"]
        assert space["fixed_sigma"] and not space["learn_sigma"]
        self.units, self.activation = list(mlp["units"]), mlp["activation"]
        # actor / critic: nn.Linear with a device training pass of its own (fast_ops.FastLinear); the discriminator stays nn.Linear
        self.actor_mlp = build_mlp(input_shape[0], self.units, self.activation, FastLinear)
        self.critic_mlp = build_mlp(input_shape[0], self.units, self.activation, FastLinear)
        self.value = FastLinear(self.units[-1], value_size)
        self.mu = FastLinear(self.units[-1], actions_num)
        self.sigma = nn.Parameter(torch.full((actions_num,), float(space["sigma_init"]["val"]), dtype=torch.float32), requires_grad=False)
        self._disc_mlp = build_mlp(amp_input_shape[0], list(disc["units"]), disc["activation"], FastLinearDD)
        self._disc_logits = FastLinear1DD(list(disc["units"])[-1], 1)
        # initializer "default" leaves the weights at torch's Linear default; the builder then zeroes the bias of EVERY nn.Linear that
        # exists at that point -- actor_mlp, critic_mlp, value, mu (network_builder.py:277-284) -- and the AMP builder those of the
        # discriminator (amp_network_builder.py:230-249), logit layer U(-1,1).  PNN columns and the MCP composer are created after that
        # loop and keep torch's default biases (amp_network_pnn_builder.py:42-55, amp_network_mcp_builder.py:38-52).
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.zeros_(m.bias)
        nn.init.uniform_(self._disc_logits.weight, -DISC_LOGIT_INIT_SCALE, DISC_LOGIT_INIT_SCALE)
        nn.init.zeros_(self._disc_logits.bias)

    def eval_actor(self, obs):
        mu = self.mu(self.actor_mlp(obs))
        return mu, self.sigma.expand_as(mu)  # amp_network_builder.py:143-151 (`mu * 0.0 + sigma`: the logstd broadcast, as a view)

    def eval_critic(self, obs):
        return self.value(self.critic_mlp(obs))

    def eval_disc(self, amp_obs):
        return self._disc_logits(self._disc_mlp(amp_obs))

    def get_disc_logit_weights(self):
        return torch.flatten(self._disc_logits.weight)

    def get_disc_weights_raw(self):
        """The weight tensors `get_disc_weights` flattens (hidden layers, then the logit layer)."""
        return [m.weight for m in self._disc_mlp.modules() if isinstance(m, nn.Linear)] + [self._disc_logits.weight]

    def get_disc_weights(self):
        w = [torch.flatten(m.weight) for m in self._disc_mlp.modules() if isinstance(m, nn.Linear)]
        w.append(torch.flatten(self._disc_logits.weight))
        return w


class PNN(nn.Module):
    """Progressive-network actor (phc/learning/pnn.py:10-131): `numCols` full actor MLPs ("primitives") sharing the input;
    column k's output layer IS the action head (no separate `mu`).  `freeze_pnn(idx)` freezes columns < idx (:40-45).
    `has_lateral` (off in every shipped yaml, env_im_pnn.yaml): bias-free adapters `u[i][j]` from the first hidden activation of every
    earlier column j <= i into the second hidden layer of column i + 1 (:25-38,84-126; the output-layer adapters `u[i][j][1]` exist in
    the state dict but the reference's forward leaves them unused: "disable action space transfer")."""

    def __init__(self, input_size, units, activation, output_size, num_cols, has_lateral=False):
        super().__init__()
        self.numCols = num_cols
        self.has_lateral = bool(has_lateral)
        self.actors = nn.ModuleList()
        for _ in range(num_cols):
            mlp = build_mlp(input_size, units, activation, FastLinear)
            mlp.append(FastLinear(units[-1], output_size))
            self.actors.append(mlp)
        if self.has_lateral:
            assert len(units) == 2, "lateral connections: the reference supports two hidden layers (pnn.py:98)"
            # the lateral term enters BEFORE the second hidden layer's activation (pnn.py:103: relu(W a1 + b + lateral)): that layer must
            # not apply its ReLU in the GEMM epilogue (the FusedReLU behind it then does the activation itself)
            for mlp in self.actors:
                mlp[2].fuse_relu = False
            self.u = nn.ModuleList()
            for i in range(num_cols - 1):
                self.u.append(nn.ModuleList())
                for _ in range(i + 1):
                    seq = nn.Sequential()
                    n = units[0]
                    for unit in units[1:]:
                        seq.append(nn.Linear(n, unit, bias=False))
                        n = unit
                    seq.append(nn.Linear(units[-1], output_size, bias=False))
                    self.u[i].append(seq)

    def freeze_pnn(self, idx):
        for p in self.actors[:idx].parameters():
            p.requires_grad = False

    def load_actor(self, checkpoint, idx=0):
        """pnn.py:52-59: initialise column idx from a plain (non-PNN) checkpoint's actor + mu head."""
        sd, m = self.actors[idx].state_dict(), checkpoint["model"]
        n_hidden = (len(sd) - 2) // 2
        for k in range(n_hidden):
            sd[f"{2 * k}.weight"].copy_(m[f"a2c_network.actor_mlp.{2 * k}.weight"])
            sd[f"{2 * k}.bias"].copy_(m[f"a2c_network.actor_mlp.{2 * k}.bias"])
        sd[f"{2 * n_hidden}.weight"].copy_(m["a2c_network.mu.weight"])
        sd[f"{2 * n_hidden}.bias"].copy_(m["a2c_network.mu.bias"])

    def forward(self, x, idx=-1):
        if self.has_lateral:   # pnn.py:85-126
            if idx == 0:
                a = self.actors[0](x)
                return a, [a]
            if idx == -1:
                idx = self.numCols - 1
            first, outs = [], []
            for k in range(idx + 1):
                col = self.actors[k]
                a1 = col[1](col[0](x))
                lateral = sum(self.u[k - 1][j][0](first[j]) for j in range(len(first))) if first else 0
                a2 = col[3](col[2](a1) + lateral)
                actions = col[4](a2)
                first.append(a1)
                outs.append(actions)
            return actions, outs
        if idx != -1:
            a = self.actors[idx](x)
            return a, [a]
        acts = [actor(x) for actor in self.actors]
        return acts, acts


class A2CPNNNetwork(A2CNetwork):
    """`AMPPNNBuilder.Network` (phc/learning/amp_network_pnn_builder.py:25-87): the actor MLP + mu head are replaced by a
    PNN whose column `training_prim` produces the action mean; columns below it are frozen.  State-dict keys:
    `a2c_network.pnn.actors.{k}.{0,2,4}.*` (what scripts/pmcp/forward_pmcp.py copies between columns) next to the parent's
    `critic_mlp / value / mu / sigma / _disc_*` -- exactly the reference's key set (tests/test_learner_parity.py)."""

    def __init__(self, params, actions_num, input_shape, amp_input_shape, task_obs_size_detail, value_size=1):
        super().__init__(params, actions_num, input_shape, amp_input_shape, value_size)
        d = task_obs_size_detail
        self.num_prim, self.training_prim = d["num_prim"], d["training_prim"]
        del self.actor_mlp
        # `mu` STAYS in the module (and in the state dict) although the PNN columns carry their own output layer: the reference deletes
        # only actor_mlp (amp_network_pnn_builder.py:51), and its env-side loader reads `a2c_network.mu.bias` for the action size
        # (network_loader.py:65).  It takes no part in the forward pass, so it does not train.
        for p in self.mu.parameters():
            p.requires_grad = False
        self.pnn = PNN(input_shape[0], self.units, self.activation, actions_num, self.num_prim, d.get("has_lateral", False))
        self.pnn.freeze_pnn(self.training_prim)

    def eval_actor(self, obs):
        mu, _ = self.pnn(obs, idx=self.training_prim)
        return mu, self.sigma.expand_as(mu)


class A2CMCPNetwork(A2CNetwork):
    """`AMPMCPBuilder.Network` (phc/learning/amp_network_mcp_builder.py:25-91): the action is the vector of mixing weights over
    `num_prim` frozen primitives, produced by a `composer` MLP = units + [num_prim], every layer (also the last, `ending_act`)
    followed by the activation, optionally a softmax (`has_softmax`; False in im_mcp.yaml).  The plain actor_mlp / mu of the
    parent stay in the state dict (as in the reference) but are not used."""

    def __init__(self, params, actions_num, input_shape, amp_input_shape, task_obs_size_detail, value_size=1):
        super().__init__(params, actions_num, input_shape, amp_input_shape, value_size)
        self.num_primitive = task_obs_size_detail.get("num_prim", 4)
        assert actions_num == self.num_primitive, "the MCP task's action space is the primitive weights"
        self.composer = build_mlp(input_shape[0], self.units + [self.num_primitive], self.activation, FastLinear)
        if params.get("has_softmax", True):
            self.composer.append(nn.Softmax(dim=1))
        if not params.get("ending_act", True):
            self.composer = self.composer[:-1]
        for p in list(self.actor_mlp.parameters()) + list(self.mu.parameters()):
            p.requires_grad = False

    def eval_actor(self, obs):
        mu = self.composer(obs)
        return mu, self.sigma.expand_as(mu)


def forward_pmcp(checkpoint, trained_idx):
    """scripts/pmcp/forward_pmcp.py:44-51: copy PNN column `trained_idx` into column `trained_idx + 1` (the next primitive
    starts from the previous one).  Operates on a checkpoint dict in place and returns it."""
    prefix = "a2c_network.pnn.actors"
    src = [k for k in checkpoint["model"] if k.startswith(f"{prefix}.{trained_idx}.")]
    dst = [k for k in checkpoint["model"] if k.startswith(f"{prefix}.{trained_idx + 1}.")]
    assert len(src) == len(dst) and len(src) > 0, "checkpoint has no such PNN columns"
    for s_key, d_key in zip(src, dst):
        checkpoint["model"][d_key].copy_(checkpoint["model"][s_key])
    return checkpoint


class ModelAMPContinuous(nn.Module):
    """`ModelAMPContinuous.Network` (phc/learning/amp_models.py:6-59) over rl_games' ModelA2CContinuousLogStd:
    Normal(mu, exp(logstd)) policy; neglogp / entropy as rl_games computes them."""

    def __init__(self, a2c_network):
        super().__init__()
        self.a2c_network = a2c_network

    @staticmethod
    def neglogp(x, mean, std, logstd):
        return 0.5 * (((x - mean) / std) ** 2).sum(dim=-1) + 0.5 * math.log(2.0 * math.pi) * x.size()[-1] + logstd.sum(dim=-1)

    def forward(self, input_dict):
        is_train = input_dict.get("is_train", True)
        obs = input_dict["obs"]
        obs_actor = input_dict.get("obs_actor")     # (split-precision actor: its fp32 input, IMAmpAgent._actor_obs)
        mu, logstd = self.a2c_network.eval_actor(obs if obs_actor is None else obs_actor)
        value = self.a2c_network.eval_critic(obs)
        mu, logstd, value = mu.float(), logstd.float(), value.float()
        sigma = torch.exp(logstd)
        if is_train:
            prev_actions = input_dict["prev_actions"]
            entropy = (0.5 + 0.5 * math.log(2 * math.pi) + logstd).sum(dim=-1)
            res = {"prev_neglogp": self.neglogp(prev_actions, mu, sigma, logstd), "values": value, "entropy": entropy, "mus": mu, "sigmas": sigma}
            # one discriminator pass over [agent; replay; demo] (the reference runs three, amp_models.py:40-48): same logits,
            # a third of the GEMM launches; the gradient penalty still differentiates w.r.t. the demo rows only
            a, r, d = input_dict["amp_obs"], input_dict["amp_obs_replay"], input_dict["amp_obs_demo"]
            logits = self.a2c_network.eval_disc(torch.cat([a, r, d], dim=0)).float()
            res["disc_agent_logit"], res["disc_agent_replay_logit"], res["disc_demo_logit"] = torch.split(logits, [a.shape[0], r.shape[0], d.shape[0]], dim=0)
            return res
        action = mu + sigma * torch.randn_like(mu)
        return {"neglogpacs": self.neglogp(action, mu, sigma, logstd), "values": value, "actions": action, "mus": mu, "sigmas": sigma}

    def forward_heads(self, input_dict):
        """Training pass that stops at the network heads -- `mu` [B, D] and `value` [B, 1] as the GEMMs produce them (bf16 under
        autocast), `logstd` [D] and the three discriminator logit blocks -- for the fused loss (fast_ops.ppo_loss)."""
        obs = input_dict["obs"]
        obs_actor = input_dict.get("obs_actor")
        mu, logstd = self.a2c_network.eval_actor(obs if obs_actor is None else obs_actor)
        value = self.a2c_network.eval_critic(obs)
        if "amp_obs_cat" in input_dict:   # [agent; replay; demo] already assembled in one buffer (fast_ops.rows_with_grad)
            x = input_dict["amp_obs_cat"]
        else:
            a, r, d = input_dict["amp_obs"], input_dict["amp_obs_replay"], input_dict["amp_obs_demo"]
            x = torch.cat([a, r, d], dim=0)
        # one discriminator pass over [agent; replay; demo] (the reference runs three, amp_models.py:40-48); a separate demo pass would
        # shrink the gradient penalty's double backward to a third of the rows but adds nine launches: no gain measured (scripts/probes/gpu_ab.sh)
        logits_raw = self.a2c_network.eval_disc(x)
        if input_dict.get("raw_disc_logits", False):   # the fused discriminator loss takes the [3m, 1] logits as the GEMM wrote them
            la = lr_ = ld = None
        else:
            la, lr_, ld = torch.split(logits_raw.float(), [a.shape[0], r.shape[0], d.shape[0]], dim=0)
        return {"mu": mu, "value": value, "logstd": logstd[0] if logstd.dim() == 2 else logstd, "disc_logits": logits_raw, "disc_agent_logit": la,
                "disc_agent_replay_logit": lr_, "disc_demo_logit": ld}


def policy_kl(p0_mu, p0_sigma, p1_mu, p1_sigma):
    """rl_games torch_ext.policy_kl (mean over the batch)."""
    c1 = torch.log(p1_sigma / p0_sigma + 1e-5)
    c2 = (p0_sigma ** 2 + (p1_mu - p0_mu) ** 2) / (2.0 * (p1_sigma ** 2 + 1e-5))
    return (c1 + c2 - 0.5).sum(dim=-1).mean()
