"""Built-in configuration groups for the hot path, keyed like the reference's hydra tree
(`phc/data/cfg/{config,env/*,robot/*,learning/*,sim/*,control/*,domain_rand/*}.yaml`).

Users of the reference keep using THEIR yaml tree unchanged: `compose(..., cfg_dir="<PHC>/phc/data/cfg")`
(or `PHC_CFG_DIR`).  These built-ins exist so that the package runs where that tree is absent (the
GPU box, CI); `tests/test_config.py` checks them key-by-key against the reference's yamls when the
reference checkout is present.  Only the groups of the imitation path are provided.
"""
import copy

_SMPL_RESET_BODIES = ['Pelvis', 'L_Hip', 'L_Knee', 'R_Hip', 'R_Knee', 'Torso', 'Spine', 'Chest', 'Neck', 'Head', 'L_Thorax',
                      'L_Shoulder', 'L_Elbow', 'L_Wrist', 'L_Hand', 'R_Thorax', 'R_Shoulder', 'R_Elbow', 'R_Wrist', 'R_Hand']

ROOT = {  # config.yaml
    "project_name": "PHC", "notes": "Default Notes", "exp_name": "humanoid_smpl", "headless": True, "seed": 0, "no_log": False,
    "resume_str": None, "num_threads": 64, "test": False, "dataset": False, "mlp": False,
    "output_path": "output/HumanoidIm/${exp_name}", "torch_deterministic": False, "epoch": 0, "im_eval": False, "horovod": False,
    "rl_device": "cuda:0", "device": "cuda", "device_id": 0, "metadata": False, "play": "${test}", "train": True,
    "collect_dataset": False, "disable_multiprocessing": True, "server_mode": False, "has_eval": True, "no_virtual_display": True,
    "render_o3d": False, "debug": False, "follow": False, "add_proj": False, "real_traj": False,
}
DEFAULTS = {"env": "env_im", "robot": "smpl_humanoid", "learning": "im", "sim": "default_sim", "control": "default_control",
            "domain_rand": "default_dr"}

_ENV_IM = {  # env/env_im.yaml
    "task": "HumanoidIm", "project_name": "PHC", "notes": "", "motion_file": "", "num_envs": 3072, "env_spacing": 5,
    "episode_length": 300, "is_flag_run": False, "enable_debug_vis": False, "fut_tracks": False, "self_obs_v": 1, "obs_v": 6,
    "auto_pmcp": False, "auto_pmcp_soft": True, "cycle_motion": False, "hard_negative": False, "min_length": 5, "kp_scale": 1,
    "power_reward": True, "shape_resampling_interval": 500, "control_mode": "isaac_pd", "power_scale": 1.0,
    "controlFrequencyInv": 2, "stateInit": "Random", "hybridInitProb": 0.5, "numAMPObsSteps": 10, "local_root_obs": True,
    "root_height_obs": True, "key_bodies": ["R_Ankle", "L_Ankle", "R_Wrist", "L_Wrist"],
    "contact_bodies": ["R_Ankle", "L_Ankle", "R_Toe", "L_Toe"], "reset_bodies": _SMPL_RESET_BODIES, "terminationHeight": 0.15,
    "enableEarlyTermination": True, "terminationDistance": 0.25, "numTrajSamples": 3, "trajSampleTimestepInv": 3,
    "enableTaskObs": True, "plane": {"staticFriction": 1.0, "dynamicFriction": 1.0, "restitution": 0.0},
}
_ENV_IM_PNN = dict(_ENV_IM, notes=" ", has_pnn=True, fitting=False, num_prim=3, training_prim=0, actors_to_load=0,
                   has_lateral=False, models=[], zero_out_far=False, zero_out_far_train=False, getup_udpate_epoch=78750)

_ENV_IM_GETUP_MCP = dict(  # env/env_im_getup_mcp.yaml
    _ENV_IM, task="HumanoidImMCPGetup", notes="Progressive MCP without softmax, zero out far", num_envs=1024, env_spacing=2,
    sym_loss_coef=1, add_obs_noise=False, add_action_noise=False, action_noise_std=0.05, mlp_bypass=False, mlp_model_path="",
    start_idx=0, collect_dataset=False, has_pnn=True, fitting=True, num_prim=4, training_prim=0, actors_to_load=4, has_lateral=False,
    models=[], zero_out_far=True, zero_out_far_train=False, cycle_motion=True, getup_udpate_epoch=1000, getup_schedule=True,
    recoverySteps=90, zero_out_far_steps=90, recoveryEpisodeProb=0.5, fallInitProb=0.3, hard_negative=False, z_activation="silu",
    power_coefficient=0.00005,
    # legacy keys of that yaml (the simulator settings actually used come from the sim group)
    asset={"assetRoot": "/", "assetFileName": "mjcf/smpl_humanoid.xml"}, sim=None, substeps=2,
    physx={"num_threads": 4, "solver_type": 1, "num_position_iterations": 4, "num_velocity_iterations": 0, "contact_offset": 0.02,
           "rest_offset": 0.0, "bounce_threshold_velocity": 0.2, "max_depenetration_velocity": 10.0, "default_buffer_size_multiplier": 10.0},
    flex={"num_inner_iterations": 10, "warm_start": 0.25})
_ENV_IM_GETUP_MCP.pop("min_length", None)

_H1_BODIES = ['pelvis', 'left_hip_yaw_link', 'left_hip_roll_link', 'left_hip_pitch_link', 'left_knee_link', 'left_ankle_link',
              'right_hip_yaw_link', 'right_hip_roll_link', 'right_hip_pitch_link', 'right_knee_link', 'right_ankle_link', 'torso_link',
              'left_shoulder_pitch_link', 'left_shoulder_roll_link', 'left_shoulder_yaw_link', 'left_elbow_link',
              'right_shoulder_pitch_link', 'right_shoulder_roll_link', 'right_shoulder_yaw_link', 'right_elbow_link']
_ENV_IM_H1 = {  # env/env_im_h1_phc.yaml
    "task": "HumanoidIm", "motion_file": "", "num_envs": 4096, "env_spacing": 5, "episode_length": 300, "is_flag_run": False,
    "enable_debug_vis": False, "sym_loss_coef": 1, "big_ankle": True, "fut_tracks": False, "obs_v": 6, "self_obs_v": 1, "auto_pmcp": False,
    "auto_pmcp_soft": True, "cycle_motion": False, "hard_negative": False, "masterfoot": False, "freeze_toe": False, "has_pnn": True,
    "num_prim": 3, "training_prim": 0, "has_lateral": False, "actors_to_load": 0, "fitting": False, "getup_schedule": False,
    "recoverySteps": 90, "recoveryEpisodeProb": 0.5, "fallInitProb": 0.3, "getup_udpate_epoch": 270171, "zero_out_far": False,
    "zero_out_far_train": False, "default_humanoid_mass": 51.436, "real_weight": True, "kp_scale": 1, "remove_toe_im": False,
    "power_reward": True, "power_coefficient": 0.0005, "powerScale": 1.0, "stateInit": "Random", "hybridInitProb": 0.5, "numAMPObsSteps": 10,
    "local_root_obs": True, "root_height_obs": True, "key_bodies": ["left_ankle_link", "right_ankle_link", "left_elbow_link", "right_elbow_link"],
    "contact_bodies": ["left_ankle_link", "right_ankle_link"], "reset_bodies": _H1_BODIES, "terminationHeight": 0.15,
    "enableEarlyTermination": True, "terminationDistance": 0.25, "numTrajSamples": 3, "trajSampleTimestepInv": 30, "enableTaskObs": True,
    "plane": {"staticFriction": 1.0, "dynamicFriction": 1.0, "restitution": 0.0},
}
_ROBOT_H1 = {  # robot/unitree_h1.yaml
    "humanoid_type": "h1", "bias_offset": False, "has_self_collision": True, "has_mesh": False, "has_jt_limit": False, "has_dof_subset": True,
    "has_upright_start": True, "has_smpl_pd_offset": False, "remove_toe": False, "motion_sym_loss": False, "sym_loss_coef": 1, "big_ankle": True,
    "has_shape_obs": False, "has_shape_obs_disc": False, "has_shape_variation": False, "masterfoot": False, "freeze_toe": False,
    "freeze_hand": False, "box_body": True, "real_weight": True, "real_weight_porpotion_capsules": True, "real_weight_porpotion_boxes": True,
    "body_names": _H1_BODIES,
    "limb_weight_group": [_H1_BODIES[1:6], _H1_BODIES[6:11], ['pelvis', 'torso_link'], ['right_shoulder_pitch_link', 'right_shoulder_roll_link'],
                          ['left_shoulder_yaw_link', 'left_elbow_link']],
    "dof_names": _H1_BODIES[1:], "right_foot_name": "right_ankle_link", "left_foot_name": "left_ankle_link", "sim_with_urdf": True,
    "asset": {"assetRoot": "./", "assetFileName": "phc/data/assets/robot/unitree_h1/h1.xml",
              "urdfFileName": "phc/data/assets/robot/unitree_h1/urdf/h1.urdf"},
    "extend_config": [{"joint_name": "left_hand_link", "parent_name": "left_elbow_link", "pos": [0.3, 0.0, 0.0], "rot": [1.0, 0.0, 0.0, 0.0]},
                      {"joint_name": "right_hand_link", "parent_name": "right_elbow_link", "pos": [0.3, 0.0, 0.0], "rot": [1.0, 0.0, 0.0, 0.0]},
                      {"joint_name": "head_link", "parent_name": "pelvis", "pos": [0.0, 0.0, 0.6], "rot": [1.0, 0.0, 0.0, 0.0]}],
    "base_link": "torso_link",
    # read by the reference's retargeting scripts only (scripts/data_process/fit_smpl_*.py); kept so that the yaml tree is complete
    "joint_matches": [["pelvis", "Pelvis"], ["left_hip_yaw_link", "L_Hip"], ["left_knee_link", "L_Knee"], ["left_ankle_link", "L_Ankle"],
                      ["right_hip_yaw_link", "R_Hip"], ["right_knee_link", "R_Knee"], ["right_ankle_link", "R_Ankle"],
                      ["left_shoulder_roll_link", "L_Shoulder"], ["left_elbow_link", "L_Elbow"], ["left_hand_link", "L_Hand"],
                      ["right_shoulder_roll_link", "R_Shoulder"], ["right_elbow_link", "R_Elbow"], ["right_hand_link", "R_Hand"], ["head_link", "Head"]],
    "smpl_pose_modifier": [{"Pelvis": "[np.pi/2, 0, np.pi/2]"}, {"L_Shoulder": "[0, 0, -np.pi/2]"}, {"R_Shoulder": "[0, 0, np.pi/2]"},
                           {"L_Elbow": "[0, -np.pi/2, 0]"}, {"R_Elbow": "[0, np.pi/2, 0]"}],
}

_ROBOT_H1_NOHEAD = dict(_ROBOT_H1, extend_config=_ROBOT_H1["extend_config"][:2],   # robot/unitree_h1_nohead.yaml: hands only
                        joint_matches=[m for m in _ROBOT_H1["joint_matches"] if m[0] not in ("left_hand_link", "right_hand_link")])

_G1_HAND = ["zero", "one", "two", "three", "four", "five", "six"]
_G1_LEG = ["hip_pitch", "hip_roll", "hip_yaw", "knee", "ankle_pitch", "ankle_roll"]
_G1_ARM = ["shoulder_pitch", "shoulder_roll", "shoulder_yaw", "elbow_pitch", "elbow_roll"] + _G1_HAND
_G1_BODIES = (["pelvis"] + [f"left_{n}_link" for n in _G1_LEG] + [f"right_{n}_link" for n in _G1_LEG] + ["torso_link"]
              + [f"left_{n}_link" for n in _G1_ARM] + [f"right_{n}_link" for n in _G1_ARM])
_ENV_IM_G1 = dict(_ENV_IM_H1, num_envs=3072, key_bodies=["left_ankle_roll_link", "right_ankle_roll_link", "left_zero_link", "right_zero_link"],   # env/env_im_g1_phc.yaml
                  contact_bodies=["left_ankle_roll_link", "right_ankle_roll_link"], reset_bodies=_G1_BODIES)
_ROBOT_G1 = dict(_ROBOT_H1, humanoid_type="g1", body_names=_G1_BODIES, dof_names=_G1_BODIES[1:],   # robot/unitree_g1.yaml
                 limb_weight_group=[_G1_BODIES[1:7], _G1_BODIES[7:13], ["pelvis", "torso_link"], _G1_BODIES[14:26], _G1_BODIES[26:38]],
                 right_foot_name="r_foot_roll", left_foot_name="l_foot_roll",
                 asset={"assetRoot": "./", "assetFileName": "phc/data/assets/robot/unitree_g1/g1.xml",
                        "urdfFileName": "phc/data/assets/robot/unitree_g1/g1.xml"},
                 extend_config=[{"joint_name": "head_link", "parent_name": "pelvis", "pos": [0.0, 0.0, 0.4], "rot": [1.0, 0.0, 0.0, 0.0]}],
                 joint_matches=[["pelvis", "Pelvis"], ["left_hip_pitch_link", "L_Hip"], ["left_knee_link", "L_Knee"], ["left_ankle_roll_link", "L_Ankle"],
                                ["right_hip_pitch_link", "R_Hip"], ["right_knee_link", "R_Knee"], ["right_ankle_roll_link", "R_Ankle"],
                                ["left_shoulder_roll_link", "L_Shoulder"], ["left_elbow_pitch_link", "L_Elbow"], ["left_zero_link", "L_Hand"],
                                ["right_shoulder_roll_link", "R_Shoulder"], ["right_elbow_pitch_link", "R_Elbow"], ["right_zero_link", "R_Hand"],
                                ["head_link", "Head"]])
_ROBOT_G1.pop("sim_with_urdf", None)

_ENV_VR = dict(_ENV_IM, notes="VR modell, three point tracking", reset_bodies=["Head", "L_Hand", "R_Hand"],   # env/env_vr.yaml
               trackBodies=["Head", "L_Hand", "R_Hand"])

_ROBOT_SMPL = {  # robot/smpl_humanoid.yaml
    "humanoid_type": "smpl", "bias_offset": False, "has_self_collision": True, "has_mesh": False, "has_jt_limit": False,
    "has_dof_subset": True, "has_upright_start": True, "has_smpl_pd_offset": False, "remove_toe": False, "motion_sym_loss": False,
    "sym_loss_coef": 1, "big_ankle": True, "has_shape_obs": False, "has_shape_obs_disc": False, "has_shape_variation": False,
    "masterfoot": False, "freeze_toe": False, "freeze_hand": False, "box_body": True, "real_weight": True,
    "real_weight_porpotion_capsules": True, "real_weight_porpotion_boxes": True,
    "asset": {"assetRoot": "/", "assetFileName": "mjcf/smpl_humanoid.xml"},
}

_SIM_DEFAULT = {  # sim/default_sim.yaml
    "sim_device": "cuda:0", "pipeline": "gpu", "graphics_device_id": 0, "subscenes": 0, "slices": 0, "use_flex": False, "substeps": 2,
    "physx": {"step_dt": "1/60", "num_threads": 4, "solver_type": 1, "num_position_iterations": 4, "num_velocity_iterations": 0,
              "contact_offset": 0.02, "rest_offset": 0.0, "bounce_threshold_velocity": 0.2, "max_depenetration_velocity": 10.0,
              "default_buffer_size_multiplier": 10.0},
    "flex": {"num_inner_iterations": 10, "warm_start": 0.25},
}
_SIM_ROBOT = copy.deepcopy(_SIM_DEFAULT)   # sim/robot_sim.yaml: 200 Hz, no explicit substeps key (gymapi default 2)
_SIM_ROBOT["physx"]["step_dt"] = "1/200"
_SIM_ROBOT.pop("substeps", None)
_CONTROL_DEFAULT = {"action_filter": False, "action_scale": 1, "decimation": 2, "action_cutfreq": 4.0, "control_mode": "isaac_pd"}
_DR_DEFAULT = {"has_domain_rand": False, "push_robots": False, "randomize_friction": False, "randomize_base_mass": False,
               "randomize_base_com": False, "randomize_link_mass": False, "randomize_pd_gain": False, "randomize_torque_rfi": False,
               "randomize_ctrl_delay": False, "add_noise": False}


def _learning(units, activation, net_name="amp", extra_cfg=None, extra_net=None):
    cfg = {  # learning/im.yaml `params.config`
        "name": "Humanoid", "env_name": "rlgpu", "multi_gpu": False, "ppo": True, "mixed_precision": False, "normalize_input": True,
        "normalize_value": True, "reward_shaper": {"scale_value": 1}, "normalize_advantage": True, "gamma": 0.99, "tau": 0.95,
        "learning_rate": 2e-5, "lr_schedule": "constant", "score_to_win": 20000, "max_epochs": 10000000, "save_best_after": 100,
        "save_frequency": 2500, "print_stats": False, "save_intermediate": True, "entropy_coef": 0.0, "truncate_grads": True,
        "grad_norm": 50.0, "e_clip": 0.2, "horizon_length": 32, "minibatch_size": 16384, "mini_epochs": 6, "critic_coef": 5,
        "clip_value": False, "bounds_loss_coef": 10, "amp_obs_demo_buffer_size": 200000, "amp_replay_buffer_size": 200000,
        "amp_replay_keep_prob": 0.01, "amp_batch_size": 512, "amp_minibatch_size": 4096, "disc_coef": 5, "disc_logit_reg": 0.01,
        "disc_grad_penalty": 5, "disc_reward_scale": 2, "disc_weight_decay": 0.0001, "normalize_amp_input": True,
        "task_reward_w": 0.5, "disc_reward_w": 0.5, "player": {"games_num": 50000000},
    }
    cfg.update(extra_cfg or {})
    return {"params": {
        "seed": 0, "algo": {"name": "im_amp"}, "model": {"name": "amp"},
        "network": {
            "name": net_name, "separate": True, "discrete": False,
            "space": {"continuous": {"mu_activation": "None", "sigma_activation": "None", "mu_init": {"name": "default"},
                                     "sigma_init": {"name": "const_initializer", "val": -2.9}, "fixed_sigma": True, "learn_sigma": False}},
            "mlp": {"units": list(units), "activation": activation, "d2rl": False, "initializer": {"name": "default"},
                    "regularizer": {"name": "None"}},
            "disc": {"units": [1024, 512], "activation": "relu", "initializer": {"name": "default"}},
        },
        "load_checkpoint": False, "config": cfg,
    }} if not extra_net else _with_net(units, activation, net_name, cfg, extra_net)


def _with_net(units, activation, net_name, cfg, extra_net):
    base = _learning(units, activation, net_name)
    base["params"]["config"] = cfg
    base["params"]["network"].update(extra_net)
    return base


_BIG = [2048, 1536, 1024, 1024, 512, 512]
GROUPS = {
    "env": {"env_im": _ENV_IM, "env_im_pnn": _ENV_IM_PNN, "env_im_getup_mcp": _ENV_IM_GETUP_MCP, "env_im_h1_phc": _ENV_IM_H1, "env_im_g1_phc": _ENV_IM_G1, "env_vr": _ENV_VR},
    "robot": {"smpl_humanoid": _ROBOT_SMPL,
              # robot/smpl_humanoid_shape.yaml: per-env body shapes + shape parameters in the policy and discriminator observations
              "smpl_humanoid_shape": dict(_ROBOT_SMPL, has_shape_obs=True, has_shape_obs_disc=True, has_shape_variation=True),
              "unitree_h1": _ROBOT_H1, "unitree_h1_nohead": _ROBOT_H1_NOHEAD, "unitree_g1": _ROBOT_G1},
    "learning": {"im": _learning([1024, 512], "relu"), "im_big": _learning(_BIG, "silu", extra_cfg={"save_frequency": 1500}),
                 "im_pnn": _learning([1024, 512], "relu", "amp_pnn"),
                 "im_pnn_big": _learning(_BIG, "silu", "amp_pnn", {"amp_dropout": False, "save_frequency": 1500}),
                 "im_mcp": _learning([1024, 512], "relu", "amp_mcp", {"player": {"games_num": 999999999999999999999999}},
                                     {"has_softmax": False, "ending_act": True}),
                 "im_mcp_big": _learning(_BIG, "silu", "amp_mcp", {"player": {"games_num": 999999999999999999999999},
                                                                  "save_frequency": 500, "amp_dropout": True},
                                         {"has_softmax": False, "ending_act": True})},
    "sim": {"default_sim": _SIM_DEFAULT, "robot_sim": _SIM_ROBOT},
    "control": {"default_control": _CONTROL_DEFAULT, "robot_control": dict(_CONTROL_DEFAULT, decimation=4, control_mode="pd")},
    "domain_rand": {"default_dr": _DR_DEFAULT},
}


def builtin_group(group, name):
    try:
        return copy.deepcopy(GROUPS[group][name])
    except KeyError:
        raise FileNotFoundError(f"no built-in config '{group}/{name}'; point PHC_CFG_DIR / cfg_dir at the reference's phc/data/cfg tree")
