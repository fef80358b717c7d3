"""HumanoidImMCP: the policy outputs mixing weights over frozen motor primitives (reference:
phc/env/tasks/humanoid_im_mcp.py:14-114).  The primitives are the columns of a trained PNN checkpoint
(`env.models[0]`, loaded like phc/learning/network_loader.py:54-72); `step(weights)` normalises the observation with the
checkpoint's running statistics, runs every column, mixes `sum_k w_k * a_k` and feeds the result to the usual
pre-physics / physics / post-physics launches.  The PNN forward is `num_prim` small bf16-free fp32 GEMM chains in torch.
"""
import torch

from ...learning.network import PNN
from .humanoid_im import HumanoidIm


def load_pnn(checkpoint, num_prim, has_lateral, activation="relu", device="cpu"):
    """network_loader.py:54-72: rebuild the PNN from `a2c_network.pnn.actors.*` and freeze every column."""
    sd = checkpoint["model"]
    biases = [k for k in sd if k.startswith("a2c_network.pnn.actors.0") and k.endswith("bias")]
    sizes = [sd[k].shape[0] for k in biases]
    pnn = PNN(sd["a2c_network.pnn.actors.0.0.weight"].shape[1], sizes[:-1], activation, sizes[-1], num_prim, has_lateral)
    own = pnn.state_dict()
    for k, v in sd.items():
        if "pnn." in k:
            own[k.split("pnn.")[1]].copy_(v)
    pnn.freeze_pnn(num_prim)
    return pnn.to(device).eval()


_ACTIVATIONS = {"relu": torch.nn.ReLU, "silu": torch.nn.SiLU, "tanh": torch.nn.Tanh, "elu": torch.nn.ELU, "gelu": torch.nn.GELU,
                "selu": torch.nn.SELU, "sigmoid": torch.nn.Sigmoid}


def load_mcp_mlp(checkpoint, activation="relu", device="cpu", mlp_name="actor_mlp"):
    """network_loader.py:11-52: a frozen copy of one plain (non-PNN) policy's actor -- `a2c_network.{mlp_name}.*` followed by the
    `mu` head (or, for `composer`, a trailing activation) -- as one nn.Sequential.  One primitive of the `has_pnn=False` form of
    the MCP task: `env.models` then lists one such checkpoint per primitive."""
    sd = checkpoint["model"]
    keys = [k for k in sd if k.startswith(f"a2c_network.{mlp_name}")]
    if mlp_name != "composer":
        keys += ["a2c_network.mu.weight", "a2c_network.mu.bias"]
    wkeys = [k for k in keys if k.endswith("weight")]
    act = _ACTIVATIONS[activation]
    layers = []
    for i, k in enumerate(wkeys):
        w = sd[k]
        if w.dim() == 1:
            layers.append(torch.nn.LayerNorm(w.shape[0]))
        elif w.dim() == 2:
            layers.append(torch.nn.Linear(w.shape[1], w.shape[0]))
            if i < len(wkeys) - 1:
                layers.append(act())
        else:
            raise NotImplementedError(k)
    mlp = torch.nn.Sequential(*layers)
    if mlp_name == "composer":
        mlp.append(act())
    own = mlp.state_dict()
    for dst, src in zip(own.keys(), keys):
        own[dst].copy_(sd[src])
    for p in mlp.parameters():
        p.requires_grad = False
    return mlp.to(device).eval()


class BypassMLP(torch.nn.Module):
    """The distilled policy of `env.mlp_bypass` (phc/learning/mlp.py:4-58, built at humanoid_im_mcp.py:32 as MLP(num_obs, num_dof,
    [2048, 1024, 512], "silu")): Linear + activation per hidden width, then a Linear head; parameters live under `model.*` as in the
    reference's class, so its checkpoints load as they are."""

    def __init__(self, input_dim, output_dim, units=(2048, 1024, 512), activation="silu"):
        super().__init__()
        act = {"none": None}.get(activation, _ACTIVATIONS.get(activation))
        layers, n = [], input_dim
        for u in units:
            layers.append(torch.nn.Linear(n, u))
            if act is not None:
                layers.append(act())
            n = u
        layers.append(torch.nn.Linear(n, output_dim))
        self.model = torch.nn.Sequential(*layers)

    def forward(self, x):
        return self.model(x)


def load_bypass_mlp(path_or_state, num_obs, num_dof, device="cpu"):
    """humanoid_im_mcp.py:31-38: `model_state_dict` entry of a training checkpoint, or a bare state dict."""
    ck = torch.load(path_or_state, map_location=device, weights_only=False) if isinstance(path_or_state, str) else path_or_state
    mlp = BypassMLP(num_obs, num_dof)
    mlp.load_state_dict(ck["model_state_dict"] if "model_state_dict" in ck else ck)
    for p in mlp.parameters():
        p.requires_grad = False
    return mlp.to(device).eval()


class MCPMixin:
    """Shared by HumanoidImMCP and HumanoidImMCPGetup."""

    def _mcp_config(self, cfg):
        env = cfg["env"]
        self.num_prim = env.get("num_prim", 3)
        self.discrete_mcp = env.get("discrete_moe", False)
        self.has_pnn = env.get("has_pnn", False)
        self.has_lateral = env.get("has_lateral", False)
        self.z_activation = env.get("z_activation", "relu")
        self.mlp_bypass = env.get("mlp_bypass", False)           # humanoid.py:338-341: a distilled MLP acts in place of the mixed primitives
        self.mlp_model_path = env.get("mlp_model_path", "")
        self.mlp_model = None

    def _mcp_load(self):
        self.pnn, self.actors = None, None
        if self.mlp_bypass and self.mlp_model_path:
            self.mlp_model = load_bypass_mlp(self.mlp_model_path, self.num_obs, self.num_dof, self.device)
        if not self.has_pnn:
            # one plain checkpoint per primitive, each rebuilt by load_mcp_mlp (network_loader.py:11-52); the observation statistics
            # are those of the first one
            if self.models_path:
                cks = [torch.load(p, map_location=self.device, weights_only=False) for p in self.models_path]
                self.load_primitive_mlps(cks)
            return
        if len(self.models_path) == 1:
            self.load_primitives(torch.load(self.models_path[0], map_location=self.device, weights_only=False))
        elif len(self.models_path) > 1:
            raise AssertionError("exactly one PNN checkpoint is expected in env.models (humanoid_im_mcp.py:26)")

    def load_primitive_mlps(self, checkpoints):
        assert len(checkpoints) == self.num_prim, "one checkpoint per primitive"
        self.actors = [load_mcp_mlp(ck, activation=self.z_activation, device=self.device) for ck in checkpoints]
        rms = checkpoints[0]["running_mean_std"]
        self.running_mean = rms["running_mean"].float().to(self.device)
        self.running_var = rms["running_var"].float().to(self.device)

    def load_primitives(self, checkpoint):
        """Install the frozen primitives from a PNN checkpoint dict (`model` + `running_mean_std`)."""
        self.pnn = load_pnn(checkpoint, num_prim=self.num_prim, has_lateral=self.has_lateral, activation=self.z_activation, device=self.device)
        rms = checkpoint["running_mean_std"]
        self.running_mean = rms["running_mean"].float().to(self.device)
        self.running_var = rms["running_var"].float().to(self.device)

    def get_action_size(self):
        return self.num_prim                      # humanoid_im_mcp.py:45-48

    def get_task_obs_size_detail(self):
        d = super().get_task_obs_size_detail()
        d["num_prim"] = self.num_prim
        return d

    def compose_actions(self, weights):
        if self.pnn is None and self.actors is None and not self.mlp_bypass:
            raise RuntimeError("no primitives loaded: set env.models=[<pnn checkpoint>] or call load_primitives()")
        with torch.no_grad():
            obs = torch.clamp((self.obs_buf - self.running_mean) / torch.sqrt(self.running_var + 1e-05), min=-5.0, max=5.0)
            if self.mlp_bypass:                                   # humanoid_im_mcp.py:83-84: the weights are ignored
                if self.mlp_model is None:
                    raise RuntimeError("env.mlp_bypass needs env.mlp_model_path (or load_bypass_mlp() installed as task.mlp_model)")
                return self.mlp_model(obs)
            if self.discrete_mcp:
                weights = torch.nn.functional.one_hot(torch.argmax(weights, dim=1), num_classes=self.num_prim).float()
            if self.pnn is not None:
                _, acts = self.pnn(obs)
            else:
                acts = [net(obs) for net in self.actors]      # humanoid_im_mcp.py:78-79
            return torch.sum(weights[:, :, None] * torch.stack(acts, dim=1), dim=1)

    def step(self, weights):
        actions = self.compose_actions(weights.to(self.device))
        self.pre_physics_step(actions)
        self._physics_step()
        self.post_physics_step()


class HumanoidImMCP(MCPMixin, HumanoidIm):

    def __init__(self, cfg, sim_params=None, physics_engine=None, device_type="cuda", device_id=0, headless=True):
        self._mcp_config(cfg)
        super().__init__(cfg=cfg, sim_params=sim_params, physics_engine=physics_engine, device_type=device_type, device_id=device_id,
                         headless=headless)
        self._mcp_load()
