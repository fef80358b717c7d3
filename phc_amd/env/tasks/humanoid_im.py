"""HumanoidIm: host-side mirror of the reference's imitation task, backed by the HIP hot path.

The reference builds this task as the class chain
`BaseTask -> Humanoid -> HumanoidAMP -> HumanoidAMPTask -> HumanoidIm`
(phc/env/tasks/{base_task,humanoid,humanoid_amp,humanoid_amp_task,humanoid_im}.py) on top of the
closed Isaac Gym tensor API.  This class keeps the SAME public surface -- constructor signature,
buffer names, `step / reset / fetch_amp_obs_demo / resample_motions`, the attributes the learner
reaches into (`phc/learning/amp_agent.py:54-59,509-519`) -- but one env step is three device
launches behind the C ABI of include/phc_amd.h:

    pre_physics_step + _physics_step  -> phc_sim_step        (A2, S8, S10, S7)
    post_physics_step                 -> phc_im_post_physics (R1,R2,R5,R6,R7,R9 + progress_buf)
    reset(env_ids)                    -> phc_im_reset        (R11)

All simulator tensors keep Isaac Gym's layouts (humanoid.py:201-235), allocated by torch in HBM.
There is no CPU fallback: constructing the task without a HIP device raises.
"""
from collections import OrderedDict
from enum import Enum
import itertools

import os

import numpy as np
import torch

from ... import _lib as L
from ... import abi
from ...model import load_model, pack_shapes
from ...motion_lib import FixHeightMode, MotionLibReal, MotionLibSMPL
from ... import robots
from ...utils.flags import flags
from ...utils.synthetic_motion import make_motion_dict, make_robot_motion_dict

_GENERATION = itertools.count(1)   # serial numbers behind HumanoidIm.launch_generation()

SMPL_MUJOCO_NAMES = ['Pelvis', 'L_Hip', 'L_Knee', 'L_Ankle', 'L_Toe', 'R_Hip', 'R_Knee', 'R_Ankle', 'R_Toe', 'Torso', 'Spine',
                     'Chest', 'Neck', 'Head', 'L_Thorax', 'L_Shoulder', 'L_Elbow', 'L_Wrist', 'L_Hand', 'R_Thorax', 'R_Shoulder',
                     'R_Elbow', 'R_Wrist', 'R_Hand']


class SkeletonTree:
    """The three fields of poselib's SkeletonTree the path uses (skeleton3d.py:149-193)."""

    def __init__(self, node_names, parent_indices, local_translation):
        self.node_names = list(node_names)
        self.parent_indices = torch.as_tensor(np.asarray(parent_indices), dtype=torch.int32)
        self.local_translation = torch.as_tensor(np.asarray(local_translation), dtype=torch.float32)

    def __len__(self):
        return len(self.node_names)

    @property
    def num_joints(self):
        return len(self.node_names)


def _stream():
    return torch.cuda.current_stream().cuda_stream


class HumanoidIm:

    class StateInit(Enum):  # humanoid_amp.py:73-78
        Default = 0
        Start = 1
        Random = 2
        Hybrid = 3

    # ------------------------------------------------------------------ construction
    def __init__(self, cfg, sim_params=None, physics_engine=None, device_type="cuda", device_id=0, headless=True):
        self.cfg = cfg
        self.sim_params = sim_params
        self.physics_engine = physics_engine
        self.headless = headless
        self.device_type, self.device_id = device_type, device_id
        if device_type not in ("cuda", "GPU"):
            raise RuntimeError("phc_amd.HumanoidIm runs on the HIP device only (device_type='cuda'); there is no CPU path")
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: phc_amd has no CPU fallback")
        self.device = f"cuda:{device_id}"
        self._lib = L.load()  # fails loudly if libphc_amd.so is missing
        torch.cuda.set_device(device_id)
        env = cfg["env"]
        robot = cfg["robot"]

        # ---- load_humanoid_configs (humanoid.py:250-420) ----
        self.humanoid_type = robot.get("humanoid_type", "smpl")
        if self.humanoid_type not in ("smpl", "h1", "g1"):
            raise NotImplementedError(f"humanoid_type={self.humanoid_type!r}: built so far: smpl, h1, g1")
        self._is_robot = self.humanoid_type in ("h1", "g1")
        # options of the reference this path does not build: refuse them rather than run without them
        #   divide_group: several humanoids in ONE collision group, i.e. inter-env contact (humanoid.py:261-266,1051-1052): the stepper's envs are independent
        unsupported = dict(kin_loss=False, z_readout=False, distill=False, divide_group=False, is_discrete=False)
        for k, off in unsupported.items():
            v = env.get(k, robot.get(k, off))
            if v != off:
                raise NotImplementedError(f"config option {k}={v!r} is outside the hot path built so far")
        # (env.group_obs / disable_group_obs: read by load_common_humanoid_configs (humanoid.py:262-263) and used nowhere else in the reference)
        self._group_obs, self._disable_group_obs = env.get("group_obs", False), env.get("disable_group_obs", False)
        if env.get("obs_v", 1) not in (1, 2, 3, 4, 5, 6, 7, 8, 9) or env.get("self_obs_v", 1) not in (1, 2, 3) or env.get("amp_obs_v", 1) not in (1, 2):
            raise NotImplementedError("built: obs_v 1 - 9, self_obs_v 1 / 2 / 3 (force sensors), amp_obs_v 1 / 2 (key-body velocities)")
        self.has_task = True
        self.obs_v, self.self_obs_v, self.amp_obs_v = int(env.get("obs_v", 6)), int(env.get("self_obs_v", 1)), int(env.get("amp_obs_v", 1))
        self.past_track_steps = int(env.get("past_track_steps", 5))   # humanoid.py:331
        if self.obs_v == 4 and self.past_track_steps != 1:
            # obs_v 4 = the v6 observation of the last `past_track_steps` steps: `_compute_observations` (humanoid_im.py:713-722) stacks the WHOLE row
            # (self + task, width n) into rows of width n * past_track_steps, while the sizes count only the task block that many times
            # (:496-501; the self-observation factor is commented out, :479-484): for past_track_steps > 1 the reference's own assignment fails with
            # a shape error.  With past_track_steps = 1 the stacking is the identity and the observation is the v6 one.
            raise NotImplementedError("obs_v=4 with past_track_steps > 1: the reference's row stacking (humanoid_im.py:713-722) does not fit its own "
                                      "observation size (:496-501,479-484); past_track_steps=1 (== obs_v 6) is what it can run")
        if self.obs_v in (4, 5) and env.get("fut_tracks", False):
            raise NotImplementedError("fut_tracks: built for the time-major task observations obs_v 6 / 7 / 9")
        self._add_action_noise = bool(env.get("add_action_noise", False))     # humanoid.py:337,1533-1535: applied only while collect_dataset is on
        self._action_noise_std = float(env.get("action_noise_std", 0.05))
        self._remove_disc_rot = bool(env.get("remove_disc_rot", False))       # humanoid.py:309,405-406: no joint rotations / rates in the AMP observation
        self._enable_hist_obs = bool(env.get("enableHistObs", False))         # humanoid_amp.py:98-101,327-328,546-557: the AMP history behind the self observation
        # future reference tracks in the task observation (humanoid_im.py:39-47): numTrajSamples frames, 1 / trajSampleTimestepInv apart
        self._fut_tracks = bool(env.get("fut_tracks", False))
        self._num_traj_samples = int(env["numTrajSamples"]) if self._fut_tracks else 1
        self._traj_sample_timestep = 1 / env.get("trajSampleTimestepInv", 30)
        self._fut_tracks_dropout = bool(env.get("fut_tracks_dropout", False)) and self._num_traj_samples > 1   # humanoid_im.py:40,824-830
        if self._fut_tracks and self._num_traj_samples > 1:
            if self.obs_v not in (6, 7, 9):
                raise NotImplementedError("fut_tracks: built for the time-major task observations obs_v 6 / 7 / 9")
            if env.get("fut_tracks_dropout", False) and self.obs_v == 7:
                raise NotImplementedError("fut_tracks_dropout: the reference drops samples in the obs_v 4 / 5 / 6 / 8 / 9 branch only (humanoid_im.py:824-830)")
            if env.get("zero_out_far", False):
                raise NotImplementedError("fut_tracks with zero_out_far: the reference's far-mask indexes a single-sample block (humanoid_im.py:785-800)")
        if self.self_obs_v == 2 and (self._is_robot or robot.get("has_shape_obs", False) or robot.get("has_weight_obs", False)):
            raise NotImplementedError("self_obs_v=2: SMPL family, without shape / limb-weight columns (the reference raises for them, humanoid.py:2101-2105)")
        # S6: force sensors at the feet (humanoid.py:268,1031-1040), read by self_obs_v 3 only (:683,1449,1481)
        self.force_sensor_joints = list(env.get("force_sensor_joints", ["L_Ankle", "R_Ankle"]))
        if self._is_robot:  # load_robot_configs, humanoid.py:422-439
            self._body_names_orig = list(robot["body_names"])
            self._body_names = self._body_names_orig
            self._dof_names = list(robot["dof_names"])
            self._full_track_bodies = self._body_names_orig.copy()
            self._eval_bodies = self._body_names_orig.copy()
        else:
            self._body_names_orig = list(SMPL_MUJOCO_NAMES)
            self._body_names = self._body_names_orig
            self._dof_names = self._body_names[1:]
            self._full_track_bodies = self._body_names_orig.copy()
            self._eval_bodies = [b for b in self._body_names_orig if b not in ("L_Toe", "R_Toe", "L_Hand", "R_Hand")]
        self._has_upright_start = robot.get("has_upright_start", True)   # False: observations strip the asset's base rotation (humanoid.py:1936-1939)
        self._has_shape_obs = robot.get("has_shape_obs", False)
        self._has_shape_obs_disc = robot.get("has_shape_obs_disc", False)
        self._has_limb_weight_obs = robot.get("has_weight_obs", False)
        self._has_limb_weight_obs_disc = robot.get("has_weight_obs_disc", False)
        self.has_shape_variation = bool(robot.get("has_shape_variation", False))   # humanoid.py:275
        if self._is_robot and (self.has_shape_variation or self._has_shape_obs or self._has_shape_obs_disc or self._has_limb_weight_obs
                               or self._has_limb_weight_obs_disc):
            raise NotImplementedError("body-shape variation / shape observations are SMPL-family options (humanoid.py:669-676,735)")
        self._has_dof_subset = robot.get("has_dof_subset", False)
        self._has_self_collision = robot.get("has_self_collision", False)
        self._freeze_toe = robot.get("freeze_toe", True)
        self._freeze_hand = robot.get("freeze_hand", True)
        self._bias_offset = robot.get("bias_offset", False)
        self._has_smpl_pd_offset = robot.get("has_smpl_pd_offset", False)
        self.shape_resampling_interval = env.get("shape_resampling_interval", 100)
        self.getup_schedule = env.get("getup_schedule", False)
        self._kp_scale = env.get("kp_scale", 1.0)
        self._kd_scale = env.get("kd_scale", self._kp_scale)
        self.hard_negative = env.get("hard_negative", False)
        self.cycle_motion = env.get("cycle_motion", False)       # humanoid.py:316
        self.cycle_motion_xp = env.get("cycle_motion_xp", False)  # humanoid.py:317 (a clip restart shifts the reference by up to a metre)
        self.power_reward = env.get("power_reward", False)
        self.power_coefficient = env.get("power_coefficient", 0.0005)
        self.kin_lr = env.get("kin_lr", 5e-4)
        self.fitting = env.get("fitting", False)
        self.z_readout = self.z_read = self.z_uniform = self.z_model = self.distill = self.kin_loss = False
        self.zero_out_far = env.get("zero_out_far", False)       # humanoid.py:325-330
        self.zero_out_far_train = env.get("zero_out_far_train", True)   # humanoid.py:315 (effective with zero_out_far only; the yamls set False)
        self.close_distance = env.get("close_distance", 0.25)
        self.far_distance = env.get("far_distance", 3)
        self._zero_out_far_steps = env.get("zero_out_far_steps", 90)
        self.max_len = env.get("max_len", -1)
        self.models_path = env.get("models", [])
        self.eval_full = env.get("eval_full", False)
        self.auto_pmcp = env.get("auto_pmcp", False)
        self.auto_pmcp_soft = env.get("auto_pmcp_soft", False)
        self.strict_eval = env.get("strict_eval", False)
        self.add_obs_noise = env.get("add_obs_noise", False)
        self._add_amp_input_noise = bool(env.get("add_amp_input_noise", False))   # humanoid_amp.py:135
        self.start_idx = env.get("start_idx", 0)
        self.seq_motions = env.get("seq_motions", False)
        self.collect_dataset = cfg.get("collect_dataset", False)
        self.temp_running_mean = env.get("temp_running_mean", True)
        self.partial_running_mean = env.get("partial_running_mean", False)
        self._full_body_reward = env.get("full_body_reward", True)   # False: reward over the tracked bodies only (humanoid_im.py:925-936)
        self._min_motion_len = env.get("min_length", -1)
        self.reward_specs = dict(env.get("reward_specs", {"k_pos": 100, "k_rot": 10, "k_vel": 0.1, "k_ang_vel": 0.1,
                                                          "w_pos": 0.5, "w_rot": 0.3, "w_vel": 0.1, "w_ang_vel": 0.1}))

        # ---- Humanoid.__init__ (humanoid.py:81-131) ----
        self.control_mode = cfg["control"]["control_mode"]
        if self.control_mode not in (("isaac_pd", "pd") if self._is_robot else ("isaac_pd",)):
            raise NotImplementedError(f"control_mode={self.control_mode!r} for humanoid_type={self.humanoid_type!r} is not built "
                                      "(isaac_pd: implicit position drive; pd: explicit torques, revolute robots only)")
        self._pd_control = self.control_mode == "isaac_pd"   # humanoid.py:96-99
        self.max_episode_length = env["episode_length"]
        self._local_root_obs = env["local_root_obs"]
        self._root_height_obs = env.get("root_height_obs", True)
        self._enable_early_termination = env["enableEarlyTermination"]
        self.key_bodies = env["key_bodies"]
        self._enable_task_obs = env["enableTaskObs"]
        self._state_init = HumanoidIm.StateInit[env["stateInit"]]
        if self._state_init not in (HumanoidIm.StateInit.Random, HumanoidIm.StateInit.Start):
            raise NotImplementedError("stateInit must be Random or Start on the imitation path (humanoid_im.py:1003-1008)")
        self._hybrid_init_prob = env["hybridInitProb"]
        self._num_amp_obs_steps = env["numAMPObsSteps"]
        self._amp_root_height_obs = env.get("ampRootHeightObs", self._root_height_obs)
        if self._amp_root_height_obs != self._root_height_obs:
            raise NotImplementedError("ampRootHeightObs != root_height_obs is not built")
        self._num_amp_obs_enc_steps = env.get("numAMPEncObsSteps", self._num_amp_obs_steps)
        self.num_envs = env["num_envs"]

        # ---- model (replaces create_sim / load_asset, humanoid.py:528-535,768-990) ----
        asset = robot.get("asset", {}).get("assetFileName", "mjcf/smpl_humanoid.xml")
        self.model = load_model(cfg.get("model_asset", f"{self.humanoid_type}_humanoid"))
        assert self.model.body_names == self._body_names, f"asset {asset} does not have the body order of robot.body_names"
        if self._is_robot:
            # gains / default pose / torque limit live in the reference's task code (humanoid.py:1112-1121,1016-1022)
            self._robot_consts = robots.ROBOTS[self.humanoid_type]
            robots.apply_robot_gains(self.model, self._robot_consts, env.get("pd_v", 1))
            self.p_gains = torch.tensor(self._robot_consts["p_gains"][env.get("pd_v", 1)], dtype=torch.float32, device=self.device)
            self.d_gains = torch.tensor(self._robot_consts["d_gains"][env.get("pd_v", 1)], dtype=torch.float32, device=self.device)
            self.default_dof_pos = torch.tensor([self._robot_consts["default_dof_pos"]], dtype=torch.float32, device=self.device)
        robots.apply_collision_filter(self.model, self.humanoid_type)   # humanoid.py:1205-1226
        self.num_bodies, self.num_dof = self.model.num_bodies, self.model.num_dof
        # ---- per-env body shapes (humanoid.py:726-766,824-866): env i wears gender_betas[i % K].  The reference writes one MJCF per env
        # with smpl_sim's SMPL_Robot (needs the SMPL model files) and falls back to `smpl_{gender}_humanoid.xml` (:748) without it; here
        # `robot.shape_assets` names one compiled model per row of `robot.shape_gender_betas` (default: the three gender assets, i.e. the
        # reference's fallback) -- K compiled shapes, an int32 shape id per env, one launch for all of them ----
        self.shape_models = [self.model]
        self._env_shape = None
        gender_betas = np.zeros((1, 17), dtype=np.float32)
        if self.has_shape_variation and not flags.im_eval:     # (:743: evaluation runs on the mean shape)
            gb = robot.get("shape_gender_betas", None)
            if isinstance(gb, str):
                import joblib
                gb = joblib.load(gb)
                gb = list(gb.values()) if isinstance(gb, dict) else gb
            gender_betas = np.asarray(gb if gb is not None else [[g] + [0.0] * 16 for g in (0, 1, 2)], dtype=np.float32).reshape(-1, 17)
            assets = robot.get("shape_assets", None) or [f"smpl_{int(row[0])}_humanoid" for row in gender_betas]
            assert len(assets) == len(gender_betas), "one compiled model per row of shape_gender_betas"
            loaded = {}
            self.shape_models = []
            for a in assets:
                if a not in loaded:
                    loaded[a] = load_model(a)
                    robots.apply_collision_filter(loaded[a], self.humanoid_type)
                self.shape_models.append(loaded[a])
            self._env_shape = (torch.arange(self.num_envs, dtype=torch.int32) % len(self.shape_models)).to(self.device)
        K = len(self.shape_models)
        trees = [SkeletonTree(m.body_names, m.parent, m.local_translation) for m in self.shape_models]
        self.skeleton_trees = [trees[i % K] for i in range(self.num_envs)]
        if K > 1:
            ints, floats = pack_shapes(self.shape_models, self._kp_scale, self._kd_scale)
        else:
            ints, floats = self.model.pack(self._kp_scale, self._kd_scale)
        self._model_ints = torch.from_numpy(ints).to(self.device)
        self._model_floats = torch.from_numpy(floats).to(self.device)
        self._model_struct = abi.model_struct(self._model_ints, self._model_floats, self.num_bodies, self.num_dof,
                                              self.model.max_level, max(len(m.contact_body) for m in self.shape_models),
                                              num_shapes=K)
        self.humanoid_masses = [self.shape_models[i % K].total_mass for i in range(min(self.num_envs, 10))]
        groups = robot.get("limb_weight_group", []) if self._is_robot else (
            ['L_Hip', 'L_Knee', 'L_Ankle', 'L_Toe'], ['R_Hip', 'R_Knee', 'R_Ankle', 'R_Toe'],
            ['Pelvis', 'Torso', 'Spine', 'Chest', 'Neck', 'Head'], ['L_Thorax', 'L_Shoulder', 'L_Elbow', 'L_Wrist', 'L_Hand'],
            ['R_Thorax', 'R_Shoulder', 'R_Elbow', 'R_Wrist', 'R_Hand'])
        self.limb_weight_group = [[self._body_names.index(g) for g in grp] for grp in groups]
        lw = torch.from_numpy(np.stack([m.limb_lengths_and_weights(self.limb_weight_group) for m in self.shape_models])).to(self.device)
        shape_of_env = torch.arange(self.num_envs, device=self.device) % K
        self.humanoid_limb_and_weights = lw[shape_of_env].contiguous()
        self.humanoid_shapes = torch.from_numpy(gender_betas).to(self.device)[shape_of_env % len(gender_betas)].contiguous()   # [N, 17]
        # constant per-env observation columns (humanoid.py:1469-1473,2043-2047; humanoid_amp.py:1005-1008).  The self observation takes
        # humanoid_shapes[:, :-6] (gender + 10 betas: `_num_self_obs += 11`, :669-671); for the discriminator the reference counts the
        # same 11 columns (humanoid_amp.py:310-311) but hands the function all 17 (:691) -- a shape mismatch as shipped; 11 are used here
        se = ([self.humanoid_shapes[:, :-6]] if self._has_shape_obs else []) + ([self.humanoid_limb_and_weights] if self._has_limb_weight_obs else [])
        ae = ([self.humanoid_shapes[:, :-6]] if self._has_shape_obs_disc else []) + ([self.humanoid_limb_and_weights] if self._has_limb_weight_obs_disc else [])
        self._self_obs_extra = torch.cat(se, dim=-1).float().contiguous() if se else None
        self._amp_obs_extra = torch.cat(ae, dim=-1).float().contiguous() if ae else None

        # ---- _setup_character_props (humanoid.py:636-706, humanoid_amp.py:290-329) ----
        self._dof_body_ids = np.arange(1, len(self._body_names))
        if self._is_robot:  # humanoid.py:684-687: one DoF per joint
            self._dof_obs_size = len(self._dof_names)
            self._dof_offsets = np.arange(len(self._dof_names) + 1)
            self._dof_size = len(self._dof_names)
        else:
            self._dof_offsets = np.linspace(0, len(self._dof_names) * 3, len(self._body_names)).astype(int)
            self._dof_obs_size = len(self._dof_names) * 6
            self._dof_size = len(self._dof_names) * 3
        self._num_actions = self._dof_size
        self._num_self_obs = 1 + len(self._body_names) * (3 + 6 + 3 + 3) - 3
        if self._self_obs_extra is not None:   # humanoid.py:669-676
            self._num_self_obs += self._self_obs_extra.shape[1]
        if not self._root_height_obs:
            self._num_self_obs -= 1
        if self.amp_obs_v == 2 and self._is_robot:
            raise NotImplementedError("amp_obs_v=2 routes to build_amp_observations_smpl_v2 (humanoid_amp.py:716-717): SMPL family only")
        if self.self_obs_v == 3:   # humanoid.py:683
            if self._is_robot:
                raise NotImplementedError("self_obs_v=3 (foot force sensors) is an SMPL-family option (humanoid.py:1449-1481)")
            self._num_self_obs += 6 * len(self.force_sensor_joints)
        self._track_bodies = env.get("trackBodies", self._full_track_bodies)
        self._reset_bodies = env.get("reset_bodies", self._track_bodies)
        if self._remove_disc_rot and (self._is_robot or not self._has_dof_subset):
            raise NotImplementedError("remove_disc_rot empties dof_subset, which only the SMPL family with robot.has_dof_subset reads (humanoid.py:405-413, "
                                      "humanoid_amp.py:996-998)")
        track_slot, reset_mask, key_ids, amp_slot, n_amp_joints = abi.task_index_tables(
            self.model, self._track_bodies, self._reset_bodies, self.key_bodies, has_dof_subset=self._has_dof_subset and not self._is_robot,
            **({"amp_remove_names": tuple(self._body_names_orig)} if self._remove_disc_rot else {}))
        if self._is_robot:  # humanoid_amp.py:315-322: [root_h, root_rot 6, root_vel 3, root_ang_vel 3, dof_pos, dof_vel, key_body_pos]
            self._num_amp_obs_per_step = 13 + self._dof_obs_size + len(self._dof_names) + 3 * len(self.key_bodies) - (0 if self._amp_root_height_obs else 1)
            self.dof_subset = torch.tensor([]).long()
        else:
            self._num_amp_obs_per_step = 13 + n_amp_joints * 9 + 3 * len(self.key_bodies) - (0 if self._amp_root_height_obs else 1)
            if self.amp_obs_v == 2:   # + key-body velocities (humanoid_amp.py:303)
                self._num_amp_obs_per_step += 3 * len(self.key_bodies)
            if self._amp_obs_extra is not None:   # humanoid_amp.py:310-313
                self._num_amp_obs_per_step += self._amp_obs_extra.shape[1]
            dof_sub = [np.arange(3 * (j - 1), 3 * j) for j in range(1, self.num_bodies) if amp_slot[j] >= 0]
            self.dof_subset = torch.from_numpy(np.concatenate(dof_sub)) if self._has_dof_subset and dof_sub else torch.tensor([]).long()
        # env.enableHistObs (humanoid_amp.py:327-328,546-557): HumanoidAMP._compute_humanoid_obs appends `_amp_obs_buf` -- flattened, newest frame first, AS
        # IT STANDS when the observation is formed: post_physics_step (:193-204) and _reset_envs (:378-385) both form the observation BEFORE they
        # update / re-initialise the history, so the policy sees the history of the previous step (and, after a reset, the finished episode's).  Here
        # the columns ride behind the constant per-env columns of the self observation (`self_obs_extra`) and are refreshed from the AMP window in
        # front of every launch that writes observations (_refresh_hist_obs).
        self._hist_obs_cols = 0
        if self._enable_hist_obs:
            if self.self_obs_v == 2:
                raise NotImplementedError("enableHistObs with self_obs_v=2: get_self_obs_size (humanoid.py:513-514) multiplies the widened block by the "
                                          "number of past states, which the observation function does not produce")
            self._hist_obs_cols = int(env.get("numAMPObsSteps", 10)) * self._num_amp_obs_per_step
            hist = torch.zeros((self.num_envs, self._hist_obs_cols), dtype=torch.float32, device=self.device)
            self._self_obs_extra = hist if self._self_obs_extra is None else torch.cat([self._self_obs_extra, hist], dim=-1).contiguous()
            self._num_self_obs += self._hist_obs_cols
        # extended bodies of the full-body reward (humanoid_im.py:74-82)
        ext = list(robot.get("extend_config", [])) if self._is_robot else []
        self.num_extend_bodies = len(ext)
        self.extend_body_parent_ids = self._build_key_body_ids_tensor([e["parent_name"] for e in ext]) if ext else None
        self.extend_body_pos_in_parent = (torch.tensor([e["pos"] for e in ext], dtype=torch.float32, device=self.device).repeat(self.num_envs, 1, 1)
                                          if ext else None)
        self._ext_parent_i32 = self.extend_body_parent_ids.to(torch.int32).contiguous() if ext else None
        self._ext_offset_f32 = self.extend_body_pos_in_parent[0].contiguous() if ext else None
        self._track_bodies_id = self._build_key_body_ids_tensor(self._track_bodies)
        self._reset_bodies_id = self._build_key_body_ids_tensor(self._reset_bodies)
        self._full_track_bodies_id = self._build_key_body_ids_tensor(self._full_track_bodies)
        self._eval_track_bodies_id = self._build_key_body_ids_tensor(self._eval_bodies)
        self._key_body_ids = self._build_key_body_ids_tensor(self.key_bodies)
        self._contact_body_ids = self._build_key_body_ids_tensor(env["contact_bodies"])

        # ---- BaseTask buffers (base_task.py:62-117) ----
        self.control_freq_inv = cfg["control"].get("decimation", 2)
        sim_cfg = cfg["sim"]
        step_dt = sim_cfg["physx"]["step_dt"]
        self.sim_dt = float(eval(step_dt)) if isinstance(step_dt, str) else float(step_dt)  # run_hydra.py:79
        self.dt = self.control_freq_inv * self.sim_dt  # humanoid.py:122
        self.num_obs = self.get_obs_size()
        self.num_states = env.get("numStates", 0)
        self.num_actions = self.get_action_size()
        cfg["env"]["numObservations"] = self.num_obs
        cfg["env"]["numActions"] = self.num_actions
        N, dev = self.num_envs, self.device
        f32 = dict(device=dev, dtype=torch.float32)
        i64 = dict(device=dev, dtype=torch.long)
        self.obs_buf = torch.zeros((N, self.num_obs), **f32)
        self.states_buf = torch.zeros((N, self.num_states), **f32)
        self.rew_buf = torch.zeros(N, **f32)
        self.reset_buf = torch.ones(N, **i64)
        self.progress_buf = torch.zeros(N, **i64)
        self.randomize_buf = torch.zeros(N, **i64)
        self.extras = {}
        self.viewer = None
        self.paused = False

        # ---- _setup_tensors (humanoid.py:179-247): Isaac Gym layouts ----
        NB, D = self.num_bodies, self.num_dof
        self._root_states = torch.zeros((N, 13), **f32)
        self._root_states[:, 2] = 0.89  # char_h, humanoid.py:1065
        self._root_states[:, 6] = 1.0
        self._humanoid_root_states = self._root_states.view(N, 1, 13)[..., 0, :]
        self._initial_humanoid_root_states = self._humanoid_root_states.clone()
        self._initial_humanoid_root_states[:, 7:13] = 0
        self._humanoid_actor_ids = torch.arange(N, device=dev, dtype=torch.int32)
        self._dof_state = torch.zeros((N * D, 2), **f32)
        self._dof_pos = self._dof_state.view(N, D, 2)[..., 0]
        self._dof_vel = self._dof_state.view(N, D, 2)[..., 1]
        self._initial_dof_pos = torch.zeros((N, D), **f32)
        self._initial_dof_vel = torch.zeros((N, D), **f32)
        self._rigid_body_state = torch.zeros((N * NB, 13), **f32)
        self._rigid_body_state_reshaped = self._rigid_body_state.view(N, NB, 13)
        self._rigid_body_pos = self._rigid_body_state_reshaped[..., 0:3]
        self._rigid_body_rot = self._rigid_body_state_reshaped[..., 3:7]
        self._rigid_body_vel = self._rigid_body_state_reshaped[..., 7:10]
        self._rigid_body_ang_vel = self._rigid_body_state_reshaped[..., 10:13]
        self._contact_forces = torch.zeros((N, NB, 3), **f32)
        self.dof_force_tensor = torch.zeros((N, D), **f32)
        self._pd_target = torch.zeros((N, D), **f32)
        self._terminate_buf = torch.ones(N, **i64)
        # S6 (gym.acquire_force_sensor_tensor, humanoid.py:183-190): [N, S*6], force then torque per sensor in the sensor body's frame
        sensors_on = self.self_obs_v == 3 and not self._is_robot
        self.vec_sensor_tensor = torch.zeros((N, 6 * len(self.force_sensor_joints)), **f32) if sensors_on else None
        self._sim_struct = abi.sim_state_struct(N, self._root_states, self._dof_state, self._rigid_body_state, self._contact_forces,
                                                self.dof_force_tensor, self._pd_target, force_sensor=self.vec_sensor_tensor,
                                                env_shape=self._env_shape)
        physx = sim_cfg["physx"]
        plane = env.get("plane", {})
        solver = cfg.get("solver", {})  # phc_amd-specific knobs of the penalty contact model (not in the reference)
        self._sim_params = abi.sim_params_struct(
            sim_dt=self.sim_dt, substeps=int(sim_cfg.get("substeps", 2)), control_freq_inv=self.control_freq_inv, gravity_z=-9.81,
            contact_stiffness=float(solver.get("contact_stiffness", 1.0e5)), contact_damping=float(solver.get("contact_damping", 1.0e3)),
            friction=float(plane.get("dynamicFriction", 1.0)), friction_viscous=float(solver.get("friction_viscous", 2.0e3)),
            angular_damping=0.01, max_angular_velocity=100.0, contact_offset=float(physx.get("contact_offset", 0.02)),
            # `pd` (robot_control.yaml): explicit torque per simulate call; solver.pd_damping "continuous" (default, mode 2: only the
            # spring term is held) or "held" (mode 1: the reference's letter, unstable on unloaded light links -- DESIGN.md)
            control_mode=0 if self.control_mode == "isaac_pd" else (1 if solver.get("pd_damping", "continuous") == "held" else 2),
            limit_stiffness=float(solver.get("joint_limit_stiffness", 2000.0 if self._is_robot else 0.0)),
            limit_damping=float(solver.get("joint_limit_damping", 20.0 if self._is_robot else 0.0)),
            lane_mapping=int(solver.get("lane_mapping", 0)),
            # body-body contact between non-adjacent links (robot.has_self_collision, on in the shipped robot yamls)
            self_collision=int(bool(solver.get("self_collision", self._has_self_collision))),
            self_stiffness_scale=float(solver.get("self_stiffness_scale", 0.25)), self_damping_ratio=float(solver.get("self_damping_ratio", 0.5)),
            force_sensor_bodies=[self._body_names.index(b) for b in self.force_sensor_joints] if sensors_on else (),
            # ground-contact model: `+solver.contact=tgs` (alias rigid) = the velocity-level rigid contact with what parse_sim_params hands PhysX
            # (run_hydra.py:88-91, sim/default_sim.yaml: num_position_iterations, max_depenetration_velocity, bounce_threshold_velocity; plane
            # restitution of the env yaml); default `penalty` (include/phc_amd.h, ABI 34)
            contact_model=str(solver.get("contact", "penalty")), contact_iterations=int(solver.get("contact_iterations", physx.get("num_position_iterations", 4))),
            contact_impedance=float(solver.get("contact_impedance", 1.0e5)),
            max_depenetration_velocity=float(physx.get("max_depenetration_velocity", 10.0)),
            bounce_threshold_velocity=float(physx.get("bounce_threshold_velocity", 0.2)), restitution=float(plane.get("restitution", 0.0)),
            # `solver.inertia_lag` (ABI 35): the sub-steps behind the first one of a simulate() call keep its articulated inertias and only redo the bias-force recursion.
            # Round 6: the DEFAULT for the penalty contact model (stepper -5 %; H1 -10 %, G1 -11 %) -- pinned like the fresh scheme since
            # tests/test_stepper_options.py::test_stepper_equals_the_double_precision_recursion; `+solver.inertia_lag=0` = every sub-step fresh.  The rigid model
            # (`+solver.contact=tgs`) re-solves every sub-step with fresh impedances and runs fresh.
            # `+solver.force_average=1` publishes contact_force / dof_force as means over the env step's sub-steps instead of the last one's values
            inertia_lag=int(bool(solver.get("inertia_lag", str(solver.get("contact", "penalty")) == "penalty"))), force_average=int(bool(solver.get("force_average", 0))))
        if self._sim_params.contact_model == 1 and max((int(c) for c in np.bincount(self.model.contact_body, minlength=1)), default=0) > 32:
            # (the rigid model's per-point active / released sets are 32-bit masks: a point beyond bit 31 could never be released)
            raise ValueError("solver.contact=tgs supports at most 32 ground-contact points per body; this model has more (use the penalty model)")

        # ---- action scaling (A1) + freeze masks (humanoid.py:1331-1409,1549-1554) ----
        self.dof_limits_lower, self.dof_limits_upper = (torch.from_numpy(x).to(dev) for x in self.model.dof_limits())
        self.dof_limits = torch.stack([self.dof_limits_lower, self.dof_limits_upper], dim=-1)
        self.torque_limits = torch.from_numpy(self.model.dof_effort.astype(np.float32)).to(dev)
        self.motor_efforts = self.torque_limits.clone()
        off, scale = self.model.pd_action_offset_scale(self._bias_offset, self._has_smpl_pd_offset, self._has_upright_start)
        if self._is_robot:
            off = np.zeros_like(off)  # humanoid.py:1406-1407
        self._pd_action_offset = torch.from_numpy(off).to(dev)
        self._pd_action_scale = torch.from_numpy(scale).to(dev)
        if self.control_mode == "pd":
            # torques = p_gains * (actions * action_scale + default_dof_pos - dof_pos) - d_gains * dof_vel (humanoid.py:1585-1590):
            # the PD target the stepper sees is default_dof_pos + action_scale * clip(action, +-10)
            self._torque_target_offset = self.default_dof_pos[0].contiguous()
            self._torque_target_scale = torch.full((D,), float(cfg["control"].get("action_scale", 1.0)), **f32)
        freeze = np.zeros(D, dtype=np.int32)
        for names, on in ((("L_Hand", "R_Hand"), self._freeze_hand), (("L_Toe", "R_Toe"), self._freeze_toe)):
            if on and not self._is_robot:
                for n in names:
                    i = self._dof_names.index(n) * 3
                    freeze[i:i + 3] = 1
        self._freeze_mask = torch.from_numpy(freeze).to(dev)
        self.actions = torch.zeros((N, self.num_actions), **f32)

        # ---- termination (humanoid.py:708-724, humanoid_im.py:539-543) ----
        self._termination_heights = torch.full((NB,), float(env["terminationHeight"]), **f32)
        if "Head" in self._body_names:
            self._termination_heights[self._body_names.index("Head")] = max(0.3, float(env["terminationHeight"]))
        self._termination_distances_full = torch.full((abi.MAX_BODIES,), float(env.get("terminationDistance", 0.5)), **f32)  # PHC_MAX_BODIES slots
        self._termination_distances = self._termination_distances_full[:NB]  # learner edits this view in place (im_amp.py:174)

        # ---- HumanoidAMP / HumanoidIm state (humanoid_amp.py:109-136, humanoid_im.py:71-123) ----
        self._motion_start_times = torch.zeros(N, **f32)
        self._motion_start_times_offset = torch.zeros(N, **f32)
        self._sampled_motion_ids = torch.arange(N, **i64)  # humanoid_im.py:121
        self._global_offset = torch.zeros((N, 3), **f32)
        self._cycle_counter = torch.zeros(N, device=dev, dtype=torch.int)
        self._point_goal = torch.zeros(N, **f32)                 # humanoid_im.py:95
        self._reset_seed = int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF
        self._reset_counter = 0
        self._launch_events = None   # see step()
        # device-built list of the envs that finished in the last step (phc_im_buffers_t.reset_list): reset_done() works on it
        self._use_reset_list = getattr(self, "_use_reset_list", True)
        self._reset_list = torch.zeros(abi.RESET_SUBLISTS * abi.reset_sublist_cap(N), device=dev, dtype=torch.int32) if self._use_reset_list else None
        self._reset_count = torch.zeros((3, abi.RESET_SUBLISTS, abi.RESET_COUNT_STRIDE), device=dev, dtype=torch.int32) if self._use_reset_list else None
        self._reset_slot, self._reset_list_pending = 0, False
        self._reset_rng_dev = torch.zeros(1, device=dev, dtype=torch.int64)   # phc_im_buffers_t.reset_rng_counter (advanced by the post-physics launch)
        self._cycle_phase = torch.zeros(N, **f32) if self.cycle_motion else None
        # draws behind the random reference offsets of zero_out_far_train (reset, clip restart) / cycle_motion_xp (clip restart)
        self._far_start = bool(self.zero_out_far and self.zero_out_far_train)
        self._offset_rand = torch.zeros((N, 2), **f32) if (self._far_start or (self.cycle_motion and self.cycle_motion_xp)) else None
        if not hasattr(self, "_recovery_counter"):
            self._recovery_counter = None                        # HumanoidImGetup owns one
        if self.zero_out_far and self._track_bodies[0] != self._body_names[0]:
            raise NotImplementedError("zero_out_far needs the root as the first track body (humanoid_im.py:785)")
        S, A = self._num_amp_obs_steps, self._num_amp_obs_per_step
        # AMP history (humanoid_amp.py:125-131): every env owns a strip of 2 S frames; the history is the window of S frames starting at
        # row `_amp_head` (newest first, like `_amp_obs_buf` of the reference).  A step writes its frame in the row BEFORE the window and
        # moves the head down: the old frames are where the reference's shift would put them, without the 14 KB per env of copy traffic;
        # once the head reaches the top the window is moved back to the bottom half by one ordinary shifting launch (every S steps).
        self._amp_strip = torch.zeros((N, 2 * S, A), **f32)
        self._amp_head = S
        self._amp_obs_demo_buf = None
        self.self_obs_buf = self.obs_buf[:, :self.get_self_obs_size()]
        # self_obs_v 2: the past_track_steps previous rigid-body states per env (`_rigid_body_*_hist`, humanoid.py:229-232), kept by the kernels
        self._body_state_hist = torch.zeros((N, self.past_track_steps, NB, 13), **f32) if self.self_obs_v == 2 else None
        self.reward_raw = torch.zeros((N, 5 if self.power_reward else 4), **f32)
        self.ref_body_pos = torch.zeros((N, NB, 3), **f32)
        self.ref_body_vel = torch.zeros((N, NB, 3), **f32)
        self.ref_body_rot = torch.zeros((N, NB, 4), **f32)
        self.ref_dof_pos = torch.zeros((N, D), **f32)
        # env.occl_training (humanoid.py:324-325, humanoid_im.py:96-97): per-env occlusion of tracked bodies, read by the task kernels
        self._occl_training = bool(env.get("occl_training", False))
        self._occl_training_prob = float(env.get("occl_training_prob", 0.1))
        self._occl_mask = None
        if self._occl_training:
            if list(self._track_bodies) != list(self._body_names) or self._num_traj_samples > 1:
                raise NotImplementedError("occl_training: the reference indexes the mask by body id and by env row (humanoid_im.py:800,1181): "
                                          "full-body tracking, without fut_tracks")
            J = len(self._track_bodies)
            self.random_occlu_idx = torch.zeros((N, J), dtype=torch.bool, device=dev)
            self.random_occlu_count = torch.zeros((N, J), **i64)
            self._occl_mask = torch.zeros((N, J), dtype=torch.uint8, device=dev)
        # env.res_action (humanoid.py:327, humanoid_im.py:1094-1099): actions are residuals on the reference pose of the next frame
        self._res_action = bool(env.get("res_action", False))
        if self._res_action:
            self._sim_struct.pd_ref = abi.ptr(self.ref_dof_pos)
        self.ref_motion_cache = {}
        self._tab = [torch.from_numpy(t).to(dev) for t in (track_slot, reset_mask, key_ids, amp_slot)]
        self._n_amp_joints = n_amp_joints
        self._im_params = None
        self._rebuild_im_params()

        # ---- reference motions (humanoid_im.py:316-367) ----
        self._load_motion(env["motion_file"])
        return

    # ------------------------------------------------------------------ small helpers
    def _build_key_body_ids_tensor(self, names):
        return torch.tensor([self._body_names.index(n) for n in names], device=self.device, dtype=torch.long)

    def _rebuild_im_params(self):
        track_slot, reset_mask, key_ids, amp_slot = self._tab
        self._im_params = abi.im_params_struct(
            dt=self.dt, max_episode_length=self.max_episode_length, reward_specs=self.reward_specs, power_reward=self.power_reward,
            power_coefficient=self.power_coefficient, enable_early_termination=self._enable_early_termination,
            use_mean_termination=bool(flags.im_eval and (not self.strict_eval)), disable_collision_check=flags.no_collision_check,
            local_root_obs=self._local_root_obs, root_height_obs=self._root_height_obs, num_track_bodies=len(self._track_bodies),
            track_slot=track_slot, reset_mask=reset_mask, num_reset_bodies=len(self._reset_bodies),
            first_reset_body=self._body_names.index(self._reset_bodies[0]), termination_distances=self._termination_distances_full,
            num_key_bodies=len(self.key_bodies), key_body_ids=key_ids, num_amp_joints=self._n_amp_joints, amp_joint_slot=amp_slot,
            num_amp_obs_steps=self._num_amp_obs_steps, num_amp_obs_per_step=self._num_amp_obs_per_step,
            num_self_obs=self.get_self_obs_size(), num_task_obs=self.get_task_obs_size(), obs_v=6 if self.obs_v in (4, 5) else self.obs_v, cycle_motion=self.cycle_motion,
            zero_out_far=self.zero_out_far, close_distance=self.close_distance, far_distance=self.far_distance,
            dofs_per_joint=1 if self._is_robot else 3, ext_parent=self._ext_parent_i32, ext_offset=self._ext_offset_f32,
            self_obs_v=self.self_obs_v, num_force_sensors=len(self.force_sensor_joints) if self.self_obs_v == 3 else 0, amp_obs_v=self.amp_obs_v,
            num_self_obs_hist=self.past_track_steps if self.self_obs_v == 2 else 0, track_body_reward=not self._full_body_reward,
            num_traj_samples=self._num_traj_samples, traj_sample_timestep=self._traj_sample_timestep,
            remove_base_rot=not self._has_upright_start, self_obs_extra=self._self_obs_extra, amp_obs_extra=self._amp_obs_extra,
            zero_out_far_train=self._far_start, zero_out_far_steps=self._zero_out_far_steps, cycle_motion_xp=self.cycle_motion_xp)
        self._flag_state = (flags.im_eval, flags.no_collision_check)
        self._im_params_gen = next(_GENERATION)   # (launch_generation(): a captured launch holds this struct BY VALUE)

    def _buffers(self, amp_in, amp_out):
        """The phc_im_buffers_t of a launch.  Building the struct costs ~20 us of host time (thirty data_ptr() calls and ctypes stores) and a rollout
        step needs two; the combinations that occur (AMP window position x reset-list slot) are few: cached (the rollout is host-bound,
        profiles/r02_notes.md).  A cached struct holds ~25 raw pointers: the entry keeps the TENSOR OBJECTS they were taken from and is only
        served while every one of the task's attributes still is that very object (a subclass, test or tool that rebinds a buffer gets a
        fresh struct instead of launches through a stale pointer)."""
        key = (amp_in.data_ptr(), amp_out.data_ptr(), self._reset_slot, self._motion_ids_are_identity())
        cache = self.__dict__.setdefault("_buffers_cache", {})
        tensors = self._buffer_tensors()
        hit = cache.get(key)
        if hit is None or len(hit[1]) != len(tensors) or any(a is not b for a, b in zip(hit[1], tensors)):
            if len(cache) > 256:
                cache.clear()
            hit = cache[key] = (self._buffers_uncached(amp_in, amp_out), tensors)
        b = hit[0]
        b.reset_list = abi.ptr(self._reset_list)     # (callers clear it for the masked sweep: restored on every use)
        return b

    def _buffer_tensors(self):
        """Every tensor whose address `_buffers_uncached` stores (the AMP windows are views of `_amp_strip`)."""
        return (self.progress_buf, self.reset_buf, self._terminate_buf, self.rew_buf, self.reward_raw, self.obs_buf, self._sampled_motion_ids,
                self._motion_start_times, self._motion_start_times_offset, self._global_offset, self.ref_body_pos, self.ref_body_rot,
                self.ref_body_vel, self.ref_dof_pos, self._cycle_counter, self._recovery_counter, self._point_goal, self._cycle_phase,
                self._reset_list, self._reset_count, self._offset_rand, self._body_state_hist, self._occl_mask, self._amp_strip, self._reset_rng_dev)

    def _motion_ids_are_identity(self):
        """`_sampled_motion_ids` is arange(num_envs) unless somebody assigned it (humanoid_im.py:121 sets it once; tests and tools may): checked once per
        tensor object and in-place version, the kernels then skip the table (phc_im_buffers_t.sampled_motion_ids NULL).  The check is a host
        sync: `launch_generation()` refreshes it OUTSIDE any stream capture, and a capture that would still find it dirty fails loudly here
        instead of inside hipStreamEndCapture."""
        t = self._sampled_motion_ids
        c = self.__dict__.get("_ids_identity_cache")
        if c is None or c[0] is not t or c[1] != t._version:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("_sampled_motion_ids changed since launch_generation(): the identity check (a host sync) cannot run inside a stream capture")
            c = self.__dict__["_ids_identity_cache"] = (t, t._version, bool(torch.equal(t, torch.arange(self.num_envs, device=t.device, dtype=t.dtype))))
        return c[2]

    def _buffers_uncached(self, amp_in, amp_out):
        return abi.im_buffers_struct(self.progress_buf, self.reset_buf, self._terminate_buf, self.rew_buf, self.reward_raw, self.obs_buf,
                                     amp_in, amp_out, None if self._motion_ids_are_identity() else self._sampled_motion_ids, self._motion_start_times,
                                     self._motion_start_times_offset,
                                     self._global_offset, self.ref_body_pos, self.ref_body_rot, self.ref_body_vel, self.ref_dof_pos,
                                     cycle_counter=self._cycle_counter, recovery_counter=self._recovery_counter,
                                     point_goal=self._point_goal, cycle_phase=self._cycle_phase, reset_list=self._reset_list,
                                     reset_count=self._reset_count, reset_slot=self._reset_slot, offset_rand=self._offset_rand,
                                     body_state_hist=self._body_state_hist, occl_mask=self._occl_mask, amp_env_stride=self._amp_strip.stride(0),
                                     reset_rng_counter=self._reset_rng_dev)

    @property
    def _amp_obs_buf(self):
        return self._amp_window(self._amp_head)

    def _amp_window(self, head):
        return self._amp_strip[:, head:head + self._num_amp_obs_steps]

    @property
    def _curr_amp_obs_buf(self):
        return self._amp_obs_buf[:, 0]

    @property
    def _hist_amp_obs_buf(self):
        return self._amp_obs_buf[:, 1:]

    # ------------------------------------------------------------------ sizes (humanoid.py:504-526, humanoid_amp_task.py:47-55)
    def get_self_obs_size(self):
        if self.self_obs_v == 2:   # humanoid.py:513-514: the block of every past state and of the current one
            return self._num_self_obs * (self.past_track_steps + 1)
        return self._num_self_obs

    def get_task_obs_size(self):
        if not self._enable_task_obs:
            return 0
        J = len(self._track_bodies)   # humanoid_im.py:486-520 with num_traj_samples = 1
        if self.obs_v in (2, 9) and self._track_bodies[0] != self._body_names[0]:
            raise NotImplementedError("obs_v 2 / 9 index the root as the first tracked body (humanoid_im.py:775-776,822)")
        if self.obs_v == 5:   # v6 + the one-hot id of the env's clip, "+ 30  # Hard coded." (humanoid_im.py:503-504,812-815)
            return 24 * J + 30
        return {1: 15 * J, 2: 15 * J + 3 * (J - 1), 3: 9 * J, 4: 24 * J, 6: 24 * J, 7: 9 * J, 8: 30 * J, 9: 18 * J + 6}[self.obs_v] * self._num_traj_samples

    def get_obs_size(self):
        return self.get_self_obs_size() + self.get_task_obs_size()

    def get_running_mean_size(self):
        return (self.get_obs_size(),)

    def get_action_size(self):
        return self._num_actions

    def get_dof_action_size(self):
        return self._dof_size

    def get_num_actors_per_env(self):
        return 1

    def get_num_amp_obs(self):
        return self._num_amp_obs_steps * self._num_amp_obs_per_step

    def get_num_enc_amp_obs(self):
        return self._num_amp_obs_enc_steps * self._num_amp_obs_per_step

    def get_task_obs_size_detail(self):
        """humanoid_im.py:522-537."""
        d = OrderedDict()
        d["target"] = self.get_task_obs_size()
        d["fut_tracks"] = self._fut_tracks
        d["num_traj_samples"] = self._num_traj_samples
        d["obs_v"] = self.obs_v
        d["track_bodies"] = self._track_bodies
        d["models_path"] = self.models_path
        env = self.cfg["env"]
        d["num_prim"] = env.get("num_prim", 2)
        d["training_prim"] = env.get("training_prim", 1)
        d["actors_to_load"] = env.get("actors_to_load", 2)
        d["has_lateral"] = env.get("has_lateral", True)
        return d

    def get_states(self):
        return self.states_buf

    # ------------------------------------------------------------------ motions
    def _load_motion(self, motion_train_file, motion_test_file=[]):
        assert self._dof_offsets[-1] == self.num_dof
        mf = motion_train_file
        if isinstance(mf, str) and mf.startswith("stand") and not self._is_robot:
            # "stand[:seconds]" -- the rest pose standing still (a physically feasible clip for end-to-end sanity runs)
            from ...utils.synthetic_motion import make_stand_clip
            mf = {"stand_00000": make_stand_clip(self.model, float(mf.split(":")[1]) if ":" in mf else 10.0)}
        if isinstance(mf, str) and mf.split(":")[0] in ("stand", "armswing") and self._is_robot:
            # robots: "stand[:seconds]" / "armswing[:seconds]" -- the default joint pose standing still / with swinging shoulder-pitch joints (round 5)
            from ...utils.synthetic_motion import make_robot_stand_clip
            from ...robots import ROBOTS
            kind = mf.split(":")[0]
            mf = {f"{kind}_00000": make_robot_stand_clip(self.model, ROBOTS[self.humanoid_type]["default_dof_pos"], float(mf.split(":")[1]) if ":" in mf else 10.0,
                                                        num_extend=self.num_extend_bodies, arm_swing=0.5 if kind == "armswing" else 0.0)}
        if isinstance(mf, str) and mf.split(":")[0] in ("squat", "stepinplace", "walk") and not self._is_robot:
            # "squat | stepinplace | walk[:seconds]" -- locomotion-class sanity clips (leg IK on prescribed pelvis / foot trajectories: feet leave the
            # ground and come back without sliding, the walk translates the centre of mass at 0.7 m/s)
            from ...utils.synthetic_motion import make_gait_clip
            kind = mf.split(":")[0]
            mf = {f"{kind}_00000": make_gait_clip(self.model, kind, float(mf.split(":")[1]) if ":" in mf else 10.0)}
        if isinstance(mf, str) and mf.startswith("armswing") and not self._is_robot:
            # "armswing[:seconds]" -- standing with swinging arms (the second feasible sanity clip)
            from ...utils.synthetic_motion import make_armswing_clip
            mf = {"armswing_00000": make_armswing_clip(self.model, float(mf.split(":")[1]) if ":" in mf else 10.0)}
        if isinstance(mf, str) and mf.startswith("locomotion") and not self._is_robot:
            # "locomotion[:num_clips[:seed[:seconds]]]" -- a multi-clip set of feasible stand / arm-swing / step-in-place / walk (/ squat) clips (round 5)
            from ...utils.synthetic_motion import make_locomotion_library
            parts = mf.split(":")
            mf = make_locomotion_library(self.model, int(parts[1]) if len(parts) > 1 else 64, int(parts[2]) if len(parts) > 2 else 0,
                                         float(parts[3]) if len(parts) > 3 else 8.0)
        if isinstance(mf, str) and mf.startswith("synthetic"):
            # "synthetic[:num_clips[:seed[:mean_seconds]]]" -- AMASS-shaped smooth random clips
            parts = mf.split(":")
            nclips = int(parts[1]) if len(parts) > 1 else 1
            seed = int(parts[2]) if len(parts) > 2 else 0
            mean_s = float(parts[3]) if len(parts) > 3 else 8.0
            min_frames = max(30, int(self._min_motion_len) if self._min_motion_len > 0 else 30)
            if self._is_robot:
                mf = make_robot_motion_dict(self.model, nclips, seed=seed, mean_seconds=mean_s, num_extend=self.num_extend_bodies, min_frames=min_frames)
            else:
                from ...utils.synthetic_motion import BASE_ROT
                mf = make_motion_dict(self.model.parent, nclips, seed=seed, body_names=self._body_names, mean_seconds=mean_s, min_frames=min_frames,
                                      base_rot=None if self._has_upright_start else BASE_ROT)
        from ...config import EasyDict
        motion_lib_cfg = EasyDict({"motion_file": mf, "device": self.device, "fix_height": FixHeightMode.full_fix,
                                   "min_length": self._min_motion_len, "max_length": -1, "im_eval": flags.im_eval,
                                   "multi_thread": False, "smpl_type": self.humanoid_type, "randomrize_heading": True, "step_dt": self.dt,
                                   "heading_rng": self.cfg["env"].get("heading_rng", "persistent"),
                                   "rank": int(self.cfg.get("rank", os.environ.get("RANK", 0)))})   # per-rank heading / crop stream (run_hydra.py:121 seeds per rank)
        if self._is_robot:  # humanoid_im.py:342-359
            motion_lib_cfg["robot"] = self.cfg["robot"]
            motion_lib_cfg["robot_model"] = self.model
            self._motion_lib_cls = MotionLibReal
        else:
            self._motion_lib_cls = MotionLibSMPL
        self._motion_train_lib = self._motion_lib_cls(motion_lib_cfg)
        self._motion_eval_lib = None  # built lazily by get_eval_motion_lib()
        self._motion_lib = self._motion_train_lib
        self._motion_lib.load_motions(skeleton_trees=self.skeleton_trees, gender_betas=self.humanoid_shapes.cpu(),
                                      limb_weights=self.humanoid_limb_and_weights.cpu(),
                                      random_sample=(not flags.test) and (not self.seq_motions),
                                      max_len=-1 if flags.test else self.max_len, start_idx=self.start_idx)

    def get_eval_motion_lib(self):
        """The reference's `_motion_eval_lib` (humanoid_im.py:333-336): same data, `im_eval=True` -> clips sorted by length."""
        if self._motion_eval_lib is None:
            from ...config import EasyDict
            cfg = EasyDict(dict(self._motion_train_lib.m_cfg))
            cfg.im_eval = True
            cfg.motion_file = self._motion_train_lib._motion_data_load
            self._motion_eval_lib = self._motion_lib_cls(cfg)
        return self._motion_eval_lib

    def begin_seq_motion_samples(self):
        """humanoid_im.py:468-472."""
        self.start_idx = 0
        self._motion_lib.load_motions(skeleton_trees=self.skeleton_trees, gender_betas=self.humanoid_shapes.cpu(),
                                      limb_weights=self.humanoid_limb_and_weights.cpu(), random_sample=False, start_idx=self.start_idx)
        self.reset()

    def forward_motion_samples(self):
        """humanoid_im.py:474-477."""
        self.start_idx += self.num_envs
        self._motion_lib.load_motions(skeleton_trees=self.skeleton_trees, gender_betas=self.humanoid_shapes.cpu(),
                                      limb_weights=self.humanoid_limb_and_weights.cpu(), random_sample=False, start_idx=self.start_idx)
        self.reset()

    def resample_motions(self):
        """humanoid_im.py:369-396: re-sample one clip per env, then reset everything."""
        self._reload_motions()
        self.reset()

    def _reload_motions(self):
        self._motion_lib.load_motions(skeleton_trees=self.skeleton_trees, limb_weights=self.humanoid_limb_and_weights.cpu(),
                                      gender_betas=self.humanoid_shapes.cpu(), random_sample=(not flags.test) and (not self.seq_motions),
                                      max_len=-1 if flags.test else self.max_len)

    # ------------------------------------------------------------------ step (base_task.py:216-234)
    def step(self, actions):
        self.pre_physics_step(actions)
        ev = self._launch_events    # bench.py: (stepper pair, post-physics pair) of HIP events for THIS step, or None (consumed)
        if ev is None:
            self._physics_step()
            self.post_physics_step()
            return
        self._launch_events = None
        for pair, fn in zip(ev, (self._physics_step, self.post_physics_step)):
            if pair is not None:
                pair[0].record()
            fn()
            if pair is not None:
                pair[1].record()

    def pre_physics_step(self, actions):
        # humanoid.py:1522-1572: the action -> PD-target map itself runs inside phc_sim_step
        # (the reference clones; the stepper only reads the tensor during this call, so the caller's buffer is used as is)
        self.actions = actions.to(self.device)
        if self.actions.dim() == 1:
            self.actions = self.actions[None]
        if self.collect_dataset:   # humanoid.py:1530-1535
            self.clean_actions = self.actions.clone()
            if self._add_action_noise:
                self.actions = self.actions + torch.normal(mean=0.0, std=self._action_noise_std, size=self.actions.shape, device=self.device)
        if self.control_mode == "pd":
            self.actions = torch.clip(self.actions, -10, 10)   # humanoid.py:1568-1570
        if self._occl_training:   # humanoid_im.py:1112-1113
            self._update_occl_training()

    def _update_occl_training(self):
        """humanoid_im.py:1081-1092: occlusion spans of 30-59 steps start with probability occl_training_prob per tracked body and step (never the
        root) -- and then the reference overwrites the mask: bodies 0..8 occluded, 9..23 visible (:1091-1092).  Both kept, in that order."""
        start = torch.bernoulli(torch.full_like(self.random_occlu_count, self._occl_training_prob, dtype=torch.float32)).bool()
        start[:, 0] = False
        spans = torch.randint(30, 60, self.random_occlu_count.shape, device=self.device)
        self.random_occlu_count = torch.clamp_min(torch.where(start, spans, self.random_occlu_count) - 1, 0)
        self.random_occlu_idx = self.random_occlu_count > 0
        self.random_occlu_idx[:] = True
        self.random_occlu_idx[:, 9:24] = False
        self._occl_mask.copy_(self.random_occlu_idx)

    def _physics_step(self):
        a = self.actions.contiguous()
        if self.control_mode == "pd":  # _compute_torques every simulate call (humanoid.py:1602-1616) happens inside the stepper
            off, scale = self._torque_target_offset, self._torque_target_scale
        else:
            off, scale = self._pd_action_offset, self._pd_action_scale
        L.check(self._lib.phc_sim_step(self._model_struct, self._sim_params, self._sim_struct, a.data_ptr(), off.data_ptr(), scale.data_ptr(),
                                       self._freeze_mask.data_ptr(), self.control_freq_inv, _stream()), "phc_sim_step")

    def post_physics_step(self):
        if (flags.im_eval, flags.no_collision_check) != self._flag_state:
            self._rebuild_im_params()
        new_head = self._amp_head - 1 if self._amp_head > 0 else self._num_amp_obs_steps
        amp_in, amp_out = self._amp_window(self._amp_head), self._amp_window(new_head)
        if self.cycle_motion:
            # the draw behind `_sample_time` of the envs whose clip restarts this step (humanoid_im.py:1127); one value per
            # env is drawn (the reference draws only as many as restart, so the RNG streams differ in length, not in law)
            torch.rand(self.num_envs, out=self._cycle_phase)
            if self._offset_rand is not None:
                torch.rand(self._offset_rand.shape, out=self._offset_rand)
        if self._use_reset_list and self._reset_list_pending:   # the previous step's list was never consumed (reset(env_ids) idiom): start this one empty
            self._reset_count[self._reset_slot].zero_()
        self._refresh_hist_obs()
        buf = self._buffers(amp_in, amp_out)
        L.check(self._lib.phc_im_post_physics(self._model_struct, self._motion_lib.struct, self._im_params, self._sim_struct, buf,
                                              _stream()), "phc_im_post_physics")
        self._obs_noise()
        self._post_physics_host(new_head)
        if flags.im_eval:  # humanoid_im.py:674-680
            t = self.progress_buf * self.dt + self._motion_start_times + self._motion_start_times_offset
            res = self._motion_lib.get_motion_state(self._sampled_motion_ids, t, self._global_offset)
            self.extras["mpjpe"] = (self._rigid_body_pos - res["rg_pos"]).norm(dim=-1).mean(dim=-1)
            self.extras["body_pos"] = self._rigid_body_pos.cpu().numpy()
            self.extras["body_pos_gt"] = res["rg_pos"].cpu().numpy()

    def _post_physics_host(self, new_head):
        """The HOST side of a post-physics step (no launch): the AMP window moved, a reset list is pending, the info dict.  A replayed hipGraph of a
        whole rollout step (IMAmpAgent.play_steps) runs the launches; the learner calls this to keep the task's host state in step."""
        if self._use_reset_list:
            self._reset_list_pending = True
        self._amp_head = new_head
        self.extras["terminate"] = self._terminate_buf         # humanoid.py:1649-1650
        self.extras["reward_raw"] = self.reward_raw.detach()
        self.extras["amp_obs"] = self._amp_obs_buf.view(-1, self.get_num_amp_obs())  # humanoid_amp.py:208-209

    def _reset_done_host(self, use_list):
        """The host side of reset_done() (see _post_physics_host)."""
        self._reset_counter += 1
        if use_list:
            self._reset_slot = (self._reset_slot + 1) % 3   # the kernel zeroed that counter for the next post-physics launch
            self._reset_list_pending = False

    def get_env_rng_state(self):
        """The counters behind the device-side start-time draws of `reset_done()` (phc_im_buffers_t.reset_rng_counter, advanced by every
        post-physics launch, and the host call counter): saved with a checkpoint (IMAmpAgent.get_full_state_weights, key `env_state`) so that a
        resumed run continues the draw sequence instead of repeating it from the top."""
        return {"reset_rng_counter": int(self._reset_rng_dev.item()), "reset_counter": int(self._reset_counter)}

    def set_env_rng_state(self, state):
        self._reset_rng_dev.fill_(int(state.get("reset_rng_counter", 0)))
        self._reset_counter = int(state.get("reset_counter", 0))

    def whole_step_capturable(self):
        """True when reset_done() + step() consist of launches and host bookkeeping only (no host-side random draws or syncs between them), so that
        the learner may capture them, with its own policy / critic segments, into ONE hipGraph per rollout step."""
        return bool(self._use_reset_list and not (self.cycle_motion or self._far_start or self._occl_training or self.add_obs_noise or self._fut_tracks_dropout
                                                  or self.collect_dataset or self.obs_v == 5 or self._hist_obs_cols or flags.im_eval or flags.test)
                    and type(self).step is HumanoidIm.step and type(self).reset_done is HumanoidIm.reset_done
                    and type(self).post_physics_step is HumanoidIm.post_physics_step and type(self).pre_physics_step is HumanoidIm.pre_physics_step)

    def rollout_step_key(self):
        """What a captured rollout step depends on besides the step index: the reset-list slot and whether a list is pending (the AMP window position
        is normalised by align_amp_window()).  Everything else a captured env launch bakes in is covered by `launch_generation()`."""
        return (self._reset_slot, self._reset_list_pending, self._amp_head)

    def launch_generation(self):
        """A token that changes whenever anything changes that the env launches take BY VALUE or by raw pointer besides the per-step buffers:
        the motion library (`phc_motion_lib_t`: frames, motion_lengths, length_starts -- re-allocated by every `load_motions`, i.e. by
        `resample_motions()`, and swapped by `im_eval.evaluate`), the parameter struct (`phc_im_params_t`, rebuilt when the evaluation flags
        flip) with its AMP reference table (re-built per library), and the clip-id table mode.  A learner that replays captured env launches
        (IMAmpAgent.play_steps) compares it BEFORE every rollout and drops its graphs when it moved (a replay after
        `resample_motions()` / `evaluate()` read freed library memory).  Called outside stream capture: it also performs the host-side refreshes a
        replay skips (parameter rebuild, AMP table, identity check of the clip-id table -- a host sync)."""
        if (flags.im_eval, flags.no_collision_check) != self._flag_state:
            self._rebuild_im_params()
        self._ensure_amp_ref_table()
        lib = self._motion_lib
        if "_serial" not in lib.__dict__:
            lib._serial = next(_GENERATION)
        return (lib._serial, getattr(lib, "frames_epoch", 0), self.__dict__.get("_amp_ref_gen", 0), self._im_params_gen, self._motion_ids_are_identity(),
                self._sampled_motion_ids.data_ptr())

    def align_amp_window(self):
        """Move the AMP history window to the top of its strip (head = S), as after every S-th step: a rollout that starts from there visits the same
        window positions at the same step indices in every epoch."""
        S = self._num_amp_obs_steps
        if self._amp_head != S:
            self._amp_strip[:, S:2 * S] = self._amp_strip[:, self._amp_head:self._amp_head + S].clone()
            self._amp_head = S
            self.extras["amp_obs"] = self._amp_obs_buf.view(-1, self.get_num_amp_obs())

    def replay_step_host(self, reset=True):
        """Host bookkeeping of one reset_done() (`reset`) + step() whose launches a graph replay has just issued."""
        if reset:
            self._reset_done_host(self._use_reset_list and self._reset_list_pending)
        new_head = self._amp_head - 1 if self._amp_head > 0 else self._num_amp_obs_steps
        self._post_physics_host(new_head)

    # ------------------------------------------------------------------ reset (humanoid.py:537-621)
    def reset(self, env_ids=None):
        safe_reset = (env_ids is None) or len(env_ids) == self.num_envs
        if env_ids is None:
            env_ids = torch.arange(self.num_envs, device=self.device, dtype=torch.long)
        self._reset_envs(env_ids)
        if safe_reset:
            # "simulate one step, then reset again" (humanoid.py:544-550)
            L.check(self._lib.phc_sim_step(self._model_struct, self._sim_params, self._sim_struct, None, None, None, None, 1, _stream()),
                    "phc_sim_step")
            self._reset_envs(env_ids)
        return

    def _reset_envs(self, env_ids):
        n = len(env_ids)
        if n == 0:
            return
        env_ids = torch.as_tensor(env_ids, device=self.device).to(torch.long).contiguous()
        start_at_zero = (self._state_init == HumanoidIm.StateInit.Start) or flags.test  # humanoid_im.py:1003-1011
        phase = torch.rand(env_ids.shape, device=self.device) if self._state_init != HumanoidIm.StateInit.Start else None
        if self._far_start:
            torch.rand(self._offset_rand.shape, out=self._offset_rand)
        cur = self._amp_obs_buf
        self._refresh_hist_obs()
        self._ensure_amp_ref_table()
        buf = self._buffers(cur, cur)
        L.check(self._lib.phc_im_reset(self._model_struct, self._motion_lib.struct, self._im_params, self._sim_struct, buf, n,
                                       env_ids.data_ptr(), abi.ptr(phase), int(bool(start_at_zero)), _stream()), "phc_im_reset")
        self._obs_noise(env_ids)
        self._reset_ref_env_ids = env_ids
        self._reset_ref_motion_ids = self._sampled_motion_ids[env_ids]
        self._reset_ref_motion_times = self._motion_start_times[env_ids]

    def reset_done(self):
        """MI355X-first variant of the rollout idiom `env.reset(reset_buf.nonzero())` (amp_agent.py:318-319): ONE launch resets
        exactly the envs whose reset_buf is set -- no `.nonzero()` device->host sync, the start-time phase is drawn in the kernel
        (counter-based hash seeded from torch's seed; the reference draws torch.rand(len(env_ids))), and the flags are not
        zeroed afterwards: nothing reads reset_buf before the next post-physics launch rewrites every entry of it.  The envs come
        from the list the post-physics kernel built on the device (dense wavefronts; a masked sweep over all envs is the fallback)."""
        start_at_zero = (self._state_init == HumanoidIm.StateInit.Start) or flags.test
        cur = self._amp_obs_buf
        use_list = self._use_reset_list and self._reset_list_pending
        if self._far_start:
            torch.rand(self._offset_rand.shape, out=self._offset_rand)
        self._refresh_hist_obs()
        self._ensure_amp_ref_table()
        buf = self._buffers(cur, cur)
        if not use_list:   # nothing appended since the last consumption (e.g. right after reset()): masked sweep over reset_buf
            buf.reset_list = None
        L.check(self._lib.phc_im_reset_done(self._model_struct, self._motion_lib.struct, self._im_params, self._sim_struct, buf,
                                            self._reset_seed, self._reset_counter + 1, int(bool(start_at_zero)), _stream()), "phc_im_reset_done")
        self._reset_done_host(use_list)
        self._obs_noise(reset_rows=True)

    def _refresh_hist_obs(self):
        """env.enableHistObs: the current AMP history (the one the launch about to be issued has not touched yet) into the trailing columns of
        the self observation's per-env extra block."""
        if self._hist_obs_cols:
            self._self_obs_extra[:, -self._hist_obs_cols:].copy_(self._amp_obs_buf.reshape(self.num_envs, self._hist_obs_cols))

    def _ensure_amp_ref_table(self):
        """phc_im_params_t.amp_ref_table: the AMP observation of every frame of the motion library, so that a reset fills an env's AMP history from S
        consecutive rows instead of S lookups + observation builds (the history frames of a reset were ~3/4 of the reset launch's instructions).
        Only where the history times fall on frames: clips at 30 fps, env dt a multiple of 1/30 s (start times are multiples of 1/30 s,
        motion_lib_base.py:414-423); the kernel still checks every blend factor and builds off-frame lookups in full.  Rebuilt when the library's
        frames change (re-sampled motions), re-attached when the parameter struct was rebuilt."""
        lib = self._motion_lib
        frames = lib.frames
        #  the table is built from THIS task's model and parameter struct and the decision below reads the evaluation flags: all of them are part
        # of the key, so a library shared by two tasks, parameters rebuilt after a flag flip, or flags toggled without a reload never get a stale table
        key = (getattr(lib, "frames_epoch", 0), frames.data_ptr(), tuple(frames.shape), frames._version, id(self), self._im_params_gen,
               bool(flags.im_eval), bool(flags.test))
        c = lib.__dict__.get("_amp_ref_cache")    # kept ON the library object: the train library keeps its table across an evaluation sweep
        if c is None or c[0] != key:
            table = None
            steps = self.dt * 30.0
            width = self._num_amp_obs_per_step - (0 if self._amp_obs_extra is None else self._amp_obs_extra.shape[1])
            cap_gib = float(self.cfg["env"].get("amp_ref_table_max_gib", 16.0))
            nbytes = frames.shape[0] * width * 4
            # not in evaluation / test mode: the sweep re-loads the library for every batch of clips (a table build + stream sync each) and
            # resets every env once per batch -- the full builds are cheaper there
            ok = (self.cfg["env"].get("amp_ref_table", True) and not os.environ.get("PHC_NO_AMP_REF_TABLE") and abs(steps - round(steps)) < 1e-6 and round(steps) >= 1
                  and not (flags.im_eval or flags.test or lib.m_cfg.get("im_eval", False))
                  and bool(torch.all(torch.abs(lib._motion_dt - 1.0 / 30.0) < 1e-7)) and nbytes <= cap_gib * 2 ** 30)
            if ok:
                table = torch.empty((frames.shape[0], width), dtype=torch.float32, device=self.device)
                if nbytes >= 2 ** 30 and int(self.cfg.get("rank", 0)) == 0:
                    print(f"[phc_amd] AMP reference table: {frames.shape[0]} frames x {width} floats = {nbytes / 2 ** 30:.2f} GiB "
                          f"(env.amp_ref_table_max_gib={cap_gib:g}, env.amp_ref_table=False turns it off)", flush=True)
                nxt = torch.arange(1, frames.shape[0] + 1, dtype=torch.long, device=self.device)
                last = (lib.length_starts + lib._motion_num_frames.to(torch.long) - 1).to(torch.long)
                nxt[last] = last                       # the last frame of a clip pairs with itself (idx1 = min(idx0 + 1, nf - 1))
                self._im_params.amp_ref_table = None
                L.check(self._lib.phc_amp_ref_table(self._model_struct, lib.struct, self._im_params, frames.shape[0], nxt.data_ptr(), table.data_ptr(),
                                                    _stream()), "phc_amp_ref_table")
                torch.cuda.current_stream(self.device).synchronize()   # (`nxt` is released right after)
            c = lib._amp_ref_cache = (key, table, next(_GENERATION))
        self._amp_ref_gen = c[2]
        self._im_params.amp_ref_table = abi.ptr(c[1])

    def _one_hot_obs(self):
        """obs_v 5 (humanoid_im.py:812-815, motion_lib_base.py:214): the one-hot id of the env's clip among the library's unique motions behind the
        v6 block; the size is hard-coded to 30 columns (:503-504), i.e. the option runs on libraries of exactly 30 clips.  The kernels write the
        v6 columns only; these are refreshed here (the ids change when motions are re-sampled)."""
        if self.obs_v != 5:
            return
        U = self._motion_lib._num_unique_motions
        if U != 30:
            raise ValueError(f"obs_v=5: the observation reserves 30 one-hot columns (humanoid_im.py:504), the motion library has {U} clips")
        off = self.get_self_obs_size() + 24 * len(self._track_bodies)
        self.obs_buf[:, off:off + U] = torch.nn.functional.one_hot(self._motion_lib._curr_motion_ids.to(self.device), num_classes=U).to(self.obs_buf.dtype)

    def _obs_noise(self, env_ids=None, reset_rows=False):
        self._one_hot_obs()
        self._fut_dropout(env_ids, reset_rows)
        self._add_obs_noise(env_ids, reset_rows)

    def _fut_dropout(self, env_ids=None, reset_rows=False):
        """env.fut_tracks_dropout (humanoid_im.py:824-830): every reference sample of a freshly computed task observation is zeroed with
        probability 0.1, not in test mode; before the observation noise, as in the reference.  Rows as in `_add_obs_noise`."""
        if not self._fut_tracks_dropout or flags.test:
            return
        so, T = self.get_self_obs_size(), self._num_traj_samples
        rows = self.num_envs if env_ids is None else len(env_ids)
        keep = (torch.rand((rows, T), device=self.device) >= 0.1).to(self.obs_buf.dtype)
        if env_ids is not None:
            blocks = self.obs_buf[env_ids, so:].view(rows, T, -1) * keep[:, :, None]
            self.obs_buf[env_ids, so:] = blocks.view(rows, -1)
            return
        if reset_rows:   # only the rows whose reset flag is set are fresh
            keep = torch.where((self.reset_buf != 0)[:, None], keep, torch.ones_like(keep))
        self.obs_buf[:, so:].view(rows, T, -1).mul_(keep[:, :, None])

    def _add_obs_noise(self, env_ids=None, reset_rows=False):
        """env.add_obs_noise (humanoid_im.py:710-711): N(0, 0.1) on every freshly computed observation row, not in test mode.  Rows: all (the
        step), `env_ids` (reset(env_ids)) or the rows whose reset flag is set (reset_done: a masked add, no device -> host sync)."""
        if not self.add_obs_noise or flags.test:
            return
        if getattr(self, "_obs_noise_rng", None) is None:   # own stream: switching the noise on leaves every other draw of the run as it was
            self._obs_noise_rng = torch.Generator(device=self.device)
            self._obs_noise_rng.manual_seed((int(torch.initial_seed()) + 0x6E6F6973) & 0x7FFFFFFFFFFFFFFF)
        draw = lambda rows: torch.randn((rows, self.obs_buf.shape[1]), device=self.device, generator=self._obs_noise_rng)
        if env_ids is not None:
            self.obs_buf[env_ids] += draw(len(env_ids)) * 0.1
        elif reset_rows:
            self.obs_buf.addcmul_(draw(self.num_envs), (self.reset_buf != 0).to(self.obs_buf.dtype)[:, None], value=0.1)
        else:
            self.obs_buf.add_(draw(self.num_envs), alpha=0.1)

    # ------------------------------------------------------------------ AMP demo observations (humanoid_amp.py:215-284)
    def fetch_amp_obs_demo(self, num_samples):
        if self._amp_obs_demo_buf is None:
            self._amp_obs_demo_buf = torch.zeros((num_samples, self._num_amp_obs_steps, self._num_amp_obs_per_step),
                                                 device=self.device, dtype=torch.float32)
        else:
            assert self._amp_obs_demo_buf.shape[0] == num_samples
        motion_ids = self._motion_lib.sample_motions(num_samples)
        motion_times0 = self._motion_lib.sample_time_interval(motion_ids)
        self.build_amp_obs_demo(motion_ids, motion_times0, out=self._amp_obs_demo_buf)
        if self._add_amp_input_noise:   # humanoid_amp.py:281-282
            self._amp_obs_demo_buf.add_(torch.randn_like(self._amp_obs_demo_buf), alpha=0.01)
        return self._amp_obs_demo_buf.view(-1, self.get_num_amp_obs())

    def build_amp_obs_demo(self, motion_ids, motion_times0, out=None):
        n = motion_ids.shape[0]
        if out is None:
            out = torch.empty((n, self._num_amp_obs_steps, self._num_amp_obs_per_step), device=self.device, dtype=torch.float32)
        ids = motion_ids.to(torch.long).contiguous()
        t0 = motion_times0.to(torch.float32).contiguous()
        self._ensure_amp_ref_table()
        L.check(self._lib.phc_amp_obs_demo(self._model_struct, self._motion_lib.struct, self._im_params, n, ids.data_ptr(),
                                           t0.data_ptr(), out.data_ptr(), _stream()), "phc_amp_obs_demo")
        return out

    # ------------------------------------------------------------------ misc API kept for the learner
    def _get_state_from_motionlib_cache(self, motion_ids, motion_times, offset=None):
        return self._motion_lib.get_motion_state(motion_ids, motion_times, offset=offset)

    def _refresh_sim_tensors(self):
        """Recompute rigid-body state from root/dof state (gym.refresh_rigid_body_state_tensor)."""
        L.check(self._lib.phc_refresh_body_state(self._model_struct, self._sim_struct, _stream()), "phc_refresh_body_state")

    def render(self, sync_frame_time=False):
        return

    def close(self):
        return
