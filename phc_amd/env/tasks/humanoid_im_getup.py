"""HumanoidImGetup: fall-recovery episodes on top of HumanoidIm (reference: phc/env/tasks/humanoid_im_getup.py:42-216).

A reset env becomes, with the reference's probabilities,
  * a RECOVERY episode  -- it had terminated (fallen) and simply keeps its state for `recoverySteps` steps (:160-164),
  * a FALL episode      -- it is teleported to one of the pre-generated fall states (:166-183), or
  * a normal reference-state-init episode (HumanoidIm reset kernel).
While `_recovery_counter > 0` the env is neither reset nor does its motion clock advance (:203-216) -- that gating
runs inside the post-physics kernel (`phc_im_buffers_t.recovery_counter`).  Fall states come from the HIP stepper:
random root orientation, zero joint state, one random action held for 150 `simulate` calls (:83-129), in ONE launch.
"""
import torch

from ... import _lib as L
from ...utils.flags import flags
from .humanoid_im import HumanoidIm, _stream


class HumanoidImGetup(HumanoidIm):

    def __init__(self, cfg, sim_params=None, physics_engine=None, device_type="cuda", device_id=0, headless=True):
        env = cfg["env"]
        self._recovery_episode_prob_tgt = self._recovery_episode_prob = env["recoveryEpisodeProb"]
        self._recovery_steps_tgt = self._recovery_steps = env["recoverySteps"]
        self._fall_init_prob_tgt = self._fall_init_prob = env["fallInitProb"]
        if flags.server_mode:
            self._recovery_episode_prob_tgt = self._recovery_episode_prob = 1
            self._fall_init_prob_tgt = self._fall_init_prob = 0
        self.getup_udpate_epoch = env.get("getup_udpate_epoch", 10000)
        dev = f"cuda:{device_id}"
        n = env["num_envs"]
        self._recovery_counter = torch.zeros(n, device=dev, dtype=torch.int)  # before super(): the kernel buffers point at it
        self.availalbe_fall_states = torch.zeros(n, device=dev, dtype=torch.long)
        self.fall_id_assignments = torch.zeros(n, device=dev, dtype=torch.long)
        self._reset_fall_env_ids = []
        self._use_reset_list = False   # episode kinds are decided on the host: the done envs are listed there
        super().__init__(cfg=cfg, sim_params=sim_params, physics_engine=physics_engine, device_type=device_type, device_id=device_id,
                         headless=headless)
        self._generate_fall_states()

    # ------------------------------------------------------------------ schedule (:71-78)
    def update_getup_schedule(self, epoch_num, getup_udpate_epoch=5000):
        warm = epoch_num > getup_udpate_epoch
        self._recovery_episode_prob = self._recovery_episode_prob_tgt if warm else 0
        self._fall_init_prob = self._fall_init_prob_tgt if warm else 1

    # ------------------------------------------------------------------ fall states (:83-129)
    def _generate_fall_states(self, max_steps=150):
        N = self.num_envs
        root = self._initial_humanoid_root_states.clone()
        root[:, 3:7] = torch.nn.functional.normalize(torch.randn(N, 4, device=self.device), dim=-1)  # random root orientation
        self._root_states.copy_(root)
        self._dof_state.zero_()
        rand_actions = torch.rand(N, self.get_dof_action_size(), device=self.device) - 0.5        # U(-0.5, 0.5)
        self.pre_physics_step(rand_actions)
        L.check(self._lib.phc_sim_step(self._model_struct, self._sim_params, self._sim_struct, self.actions.data_ptr(),
                                       self._pd_action_offset.data_ptr(), self._pd_action_scale.data_ptr(), self._freeze_mask.data_ptr(),
                                       max_steps, _stream()), "phc_sim_step")
        self._fall_root_states = self._humanoid_root_states.clone()
        self._fall_root_states[:, 7:13] = 0
        self._fall_dof_pos = self._dof_pos.clone()
        self._fall_dof_vel = torch.zeros_like(self._dof_vel)
        self.availalbe_fall_states[:] = 0
        self.fall_id_assignments[:] = 0
        self._recovery_counter.zero_()

    def resample_motions(self):
        self._reload_motions()
        if not flags.test:
            self._generate_fall_states()
        self.reset()

    # ------------------------------------------------------------------ reset (:131-196)
    def _reset_envs(self, env_ids):
        if len(env_ids) == 0:
            return
        env_ids = torch.as_tensor(env_ids, device=self.device).to(torch.long)
        self.availalbe_fall_states[self.fall_id_assignments[env_ids]] = 0
        n = env_ids.shape[0]
        want_recovery = torch.bernoulli(torch.full((n,), float(self._recovery_episode_prob), device=self.device)) == 1.0
        recovery_mask = want_recovery & (self._terminate_buf[env_ids] == 1)    # only envs that really fell (:140-142)
        recovery_ids = env_ids[recovery_mask]
        rest = env_ids[~recovery_mask]
        fall_mask = torch.bernoulli(torch.full((rest.shape[0],), float(self._fall_init_prob), device=self.device)) == 1.0
        fall_ids, normal_ids = rest[fall_mask], rest[~fall_mask]
        self._reset_fall_env_ids = fall_ids
        if len(fall_ids) > 0:
            self._reset_fall_episode(fall_ids)
        if len(recovery_ids) > 0:
            self._reset_from_state(recovery_ids, refresh=False, fill_history=False)
        if len(normal_ids) > 0:
            super()._reset_envs(normal_ids)      # zeroes _recovery_counter of these envs (:155)
        if len(fall_ids) + len(recovery_ids) > 0:
            self._recovery_counter[torch.cat([fall_ids, recovery_ids])] = self._recovery_steps

    def _reset_fall_episode(self, env_ids):
        free = (self.availalbe_fall_states == 0).nonzero().squeeze(-1)
        assert free.shape[0] >= env_ids.shape[0]
        pick = free[torch.randperm(free.shape[0], device=self.device)][:env_ids.shape[0]]
        self._humanoid_root_states[env_ids] = self._fall_root_states[pick]
        self._dof_pos[env_ids] = self._fall_dof_pos[pick]
        self._dof_vel[env_ids] = self._fall_dof_vel[pick]
        self.availalbe_fall_states[pick] = 1
        self.fall_id_assignments[env_ids] = pick
        self._reset_from_state(env_ids, refresh=True, fill_history=True)

    def _reset_from_state(self, env_ids, refresh, fill_history):
        ids = env_ids.contiguous()
        n = ids.shape[0]
        if refresh:
            L.check(self._lib.phc_refresh_body_state_indexed(self._model_struct, self._sim_struct, n, ids.data_ptr(), _stream()),
                    "phc_refresh_body_state_indexed")
        cur = self._amp_obs_buf
        self._refresh_hist_obs()
        L.check(self._lib.phc_im_reset_from_state(self._model_struct, self._motion_lib.struct, self._im_params, self._sim_struct,
                                                  self._buffers(cur, cur), n, ids.data_ptr(), int(fill_history), _stream()),
                "phc_im_reset_from_state")

    def reset_done(self):
        """Three kinds of episode need host-side bookkeeping, so the done envs are listed (as the reference's rollout does)."""
        ids = self.reset_buf.nonzero(as_tuple=False).flatten()
        if len(ids) > 0:
            self._reset_envs(ids)
