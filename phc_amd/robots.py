"""Per-robot constants the reference keeps in task code rather than in the asset files.

H1: PD gains, default joint pose and torque limit of `Humanoid._create_envs` / `_process_dof_props`
(phc/env/tasks/humanoid.py:1112-1121, 1016-1022); the gains are listed there in DoF order
[l_hip_yaw, l_hip_roll, l_hip_pitch, l_knee, l_ankle, r_hip_yaw, ..., torso, l_shoulder_pitch, ..., r_elbow].
"""
import numpy as np

H1 = {
    "p_gains": {1: [200.0, 200.0, 300.0, 200.0, 200.0, 300.0, 120.0, 200.0, 200.0, 60.0, 60.0, 40.0, 40.0, 40.0, 20.0, 40.0, 40.0, 40.0, 20.0],
                2: [200, 200, 200, 300, 40, 200, 200, 200, 300, 40, 300, 100, 100, 100, 100, 100, 100, 100, 100]},
    "d_gains": {1: [5.0, 5.0, 7.5, 5.0, 5.0, 7.5, 3.0, 5.0, 5.0, 1.5, 1.5, 1.0, 1.0, 1.0, 0.5, 1.0, 1.0, 1.0, 0.5],
                2: [5, 5, 5, 6, 2, 5, 5, 5, 6, 2, 6, 2, 2, 2, 2, 2, 2, 2, 2]},
    "default_dof_pos": [0, 0, -0.4, 0.8, -0.4, 0, 0, -0.4, 0.8, -0.4, 0, 0, 0, 0, 0, 0, 0, 0, 0],
    "torque_limit": 350.0,   # "No torque limit, set to 350" (humanoid.py:1022)
}


def apply_robot_gains(model, robot, pd_v=1):
    """Install the task-code gains / limits on a compiled model (before `pack()`)."""
    model.dof_kp[:] = np.asarray(robot["p_gains"][pd_v], dtype=np.float64)
    model.dof_kd[:] = np.asarray(robot["d_gains"][pd_v], dtype=np.float64)
    model.dof_effort[:] = robot["torque_limit"]
    return model


# Isaac Gym rigid-shape collision filters the reference sets when robot.has_self_collision (humanoid.py:1205-1226): two shapes
# of one actor collide iff (filter_a & filter_b) == 0; body order of the respective asset.
COLLISION_FILTERS = {
    "smpl": [0, 0, 7, 16, 12, 0, 56, 2, 33, 128, 0, 192, 0, 64, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
    "h1": [0, 2, 0, 2, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
}


def apply_collision_filter(model, humanoid_type):
    model.collision_filter[:] = np.asarray(COLLISION_FILTERS[humanoid_type], dtype=np.int64)
    return model
