"""VecEnv boundary seen by the learner.

Mirrors the reference's `VecTask / VecTaskPython / VecTaskPythonWrapper`
(phc/env/tasks/vec_task.py:40-162, vec_task_wrappers.py:45-81) and `RLGPUEnv`
(phc/run_hydra.py:187-244) without the gym / rl_games imports: `Box` below carries the three
attributes (`low`, `high`, `shape`) the learner reads from the spaces.
"""
import numpy as np
import torch


class Box:
    """Minimal stand-in for gym.spaces.Box (shape / low / high / dtype)."""

    def __init__(self, low, high):
        self.low = np.asarray(low, dtype=np.float32)
        self.high = np.asarray(high, dtype=np.float32)
        self.shape = self.low.shape
        self.dtype = np.float32

    def __repr__(self):
        return f"Box{self.shape}"


class VecTask:
    def __init__(self, task, rl_device, clip_observations=5.0):
        self.task = task
        self.num_environments = task.num_envs
        self.num_agents = 1
        self.num_observations = task.num_obs
        self.num_states = task.num_states
        self.num_actions = task.num_actions
        self.obs_space = Box(np.ones(self.num_obs) * -np.inf, np.ones(self.num_obs) * np.inf)
        self.state_space = Box(np.ones(self.num_states) * -np.inf, np.ones(self.num_states) * np.inf)
        self.act_space = Box(np.ones(self.num_actions) * -1., np.ones(self.num_actions) * 1.)
        self.clip_obs = clip_observations
        self.rl_device = rl_device

    def step(self, actions):
        raise NotImplementedError

    def reset(self):
        raise NotImplementedError

    def get_number_of_agents(self):
        return self.num_agents

    @property
    def observation_space(self):
        return self.obs_space

    @property
    def action_space(self):
        return self.act_space

    @property
    def num_envs(self):
        return self.num_environments

    @property
    def num_acts(self):
        return self.num_actions

    @property
    def num_obs(self):
        return self.num_observations


class VecTaskPython(VecTask):
    def get_state(self):
        return torch.clamp(self.task.states_buf, -self.clip_obs, self.clip_obs).to(self.rl_device)

    def _clipped_obs(self):
        # clip_observations = inf (parse_task.py:61, the default): the task's own buffer, no [N, obs] copy per step and a STABLE address (the
        # learner's captured rollout segments are keyed by it; a fresh clamp output made the graph cache depend on the allocator)
        if not np.isfinite(self.clip_obs):
            return self.task.obs_buf.to(self.rl_device)
        return torch.clamp(self.task.obs_buf, -self.clip_obs, self.clip_obs).to(self.rl_device)

    def step(self, actions):
        self.task.step(actions)
        return (self._clipped_obs(), self.task.rew_buf.to(self.rl_device), self.task.reset_buf.to(self.rl_device), self.task.extras)

    def reset(self):
        actions = 0.01 * (1 - 2 * torch.rand([self.task.num_envs, self.task.num_actions], dtype=torch.float32, device=self.rl_device))
        self.task.step(actions)
        return self._clipped_obs()


class VecTaskPythonWrapper(VecTaskPython):
    def __init__(self, task, rl_device, clip_observations=5.0):
        super().__init__(task, rl_device, clip_observations)
        self._amp_obs_space = Box(np.ones(task.get_num_amp_obs()) * -np.inf, np.ones(task.get_num_amp_obs()) * np.inf)
        self._enc_amp_obs_space = Box(np.ones(task.get_num_enc_amp_obs()) * -np.inf, np.ones(task.get_num_enc_amp_obs()) * np.inf)

    def reset(self, env_ids=None):
        self.task.reset(env_ids)
        return torch.clamp(self.task.obs_buf, -self.clip_obs, self.clip_obs).to(self.rl_device)

    @property
    def amp_observation_space(self):
        return self._amp_obs_space

    @property
    def enc_amp_observation_space(self):
        return self._enc_amp_obs_space

    def fetch_amp_obs_demo(self, num_samples):
        return self.task.fetch_amp_obs_demo(num_samples)


class RLGPUEnv:
    """phc/run_hydra.py:187-240: what rl_games' vecenv factory returns under the name 'rlgpu'."""

    def __init__(self, env, **kwargs):
        self.env = env
        self.use_global_obs = (self.env.num_states > 0)
        self.full_state = {"obs": self.reset()}
        if self.use_global_obs:
            self.full_state["states"] = self.env.get_state()

    def step(self, action):
        next_obs, reward, is_done, info = self.env.step(action)
        self.full_state["obs"] = next_obs
        if self.use_global_obs:
            self.full_state["states"] = self.env.get_state()
            return self.full_state, reward, is_done, info
        return self.full_state["obs"], reward, is_done, info

    def reset(self, env_ids=None):
        self.full_state = {"obs": self.env.reset(env_ids)}
        if self.use_global_obs:
            self.full_state["states"] = self.env.get_state()
            return self.full_state
        return self.full_state["obs"]

    def get_number_of_agents(self):
        return self.env.get_number_of_agents()

    def get_env_info(self):
        info = {"action_space": self.env.action_space, "observation_space": self.env.observation_space,
                "amp_observation_space": self.env.amp_observation_space,
                "enc_amp_observation_space": self.env.enc_amp_observation_space}
        info["task_obs_size"] = self.env.task.get_task_obs_size() if hasattr(self.env.task, "get_task_obs_size") else 0
        if self.use_global_obs:
            info["state_space"] = self.env.state_space
        return info

    # pass-throughs the agent uses
    def fetch_amp_obs_demo(self, num_samples):
        return self.env.fetch_amp_obs_demo(num_samples)

    @property
    def task(self):
        return self.env.task


TASKS = {}


def register_task(name, cls):
    TASKS[name] = cls


def parse_task(cfg, rl_device=None, device_id=None, headless=True):
    """phc/utils/parse_task.py:50-63: build the task named cfg.env.task and wrap it."""
    from .humanoid_im import HumanoidIm
    from .humanoid_im_getup import HumanoidImGetup
    from .humanoid_im_mcp import HumanoidImMCP
    from .humanoid_im_mcp_getup import HumanoidImMCPGetup
    for cls in (HumanoidIm, HumanoidImGetup, HumanoidImMCP, HumanoidImMCPGetup):
        TASKS.setdefault(cls.__name__, cls)
    name = cfg["env"]["task"]
    if name not in TASKS:
        raise Exception(f"Unrecognized task {name!r}! Built so far: {sorted(TASKS)}")
    device_id = cfg.get("device_id", 0) if device_id is None else device_id
    rl_device = rl_device or cfg.get("rl_device", f"cuda:{device_id}")
    task = TASKS[name](cfg=cfg, sim_params=None, physics_engine=None, device_type=cfg.get("device", "cuda"), device_id=device_id,
                       headless=headless)
    learn_cfg = cfg.get("learning", {}).get("params", {}).get("config", {})
    env = VecTaskPythonWrapper(task, rl_device, learn_cfg.get("clip_observations", np.inf))
    return task, env
