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

# G1 (humanoid.py:1123-1181): 37 DoFs in the order legs (pitch, roll, yaw, knee, ankle pitch, ankle roll) x 2, torso, arms
# (shoulder pitch / roll / yaw, elbow pitch / roll, hand zero..six) x 2; only pd_v 1 is defined, default pose all zero, torque
# limits hard-coded per DoF (`torque_limits_hard_coded`).
_G1_ARM_P, _G1_ARM_D, _G1_ARM_T = [40.0, 40.0, 40.0, 60.0, 40.0] + [20.0] * 7, [1.0, 1.0, 1.0, 1.5, 1.0] + [0.5] * 7, [20.0] * 5 + [0.7] * 7
_G1_LEG_P, _G1_LEG_D, _G1_LEG_T = [200.0, 200.0, 200.0, 300.0, 200.0, 200.0], [5.0, 5.0, 5.0, 7.5, 5.0, 5.0], [88.0, 88.0, 88.0, 139.0, 40.0, 40.0]
G1 = {
    "p_gains": {1: _G1_LEG_P * 2 + [120.0] + _G1_ARM_P * 2},
    "d_gains": {1: _G1_LEG_D * 2 + [3.0] + _G1_ARM_D * 2},
    "default_dof_pos": [0.0] * 37,
    "torque_limit": _G1_LEG_T * 2 + [88.0] + _G1_ARM_T * 2,
}
ROBOTS = {"h1": H1, "g1": G1}


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
# G1: the reference lists one filter per collision SHAPE (40: torso_link carries three -- torso, head, logo -- elbow_roll_link two
# -- forearm, palm -- zero_link none); the stepper has one capsule per body, so a body takes the OR of its shapes' bits (for
# this table no pair of bodies changes its verdict: every bit of a multi-shape body meets the same partners).
_G1_SHAPE_FILTERS = [0, 0, 2688, 0, 8192, 0, 8192, 0, 1344, 0, 4096, 0, 4096, 3072, 768, 192, 1, 0, 1, 0, 32, 8, 40, 0, 0, 0, 0, 0, 2, 0, 2, 0, 16,
                     4, 20, 0, 0, 0, 0, 0]
_G1_SHAPES_PER_BODY = [1] * 13 + [3] + ([1, 1, 1, 1, 2, 0] + [1] * 6) * 2


def _per_body_filters(shape_filters, shapes_per_body):
    out, k = [], 0
    for n in shapes_per_body:
        f = 0
        for v in shape_filters[k:k + n]:
            f |= v
        out.append(f)
        k += n
    assert k == len(shape_filters)
    return out


COLLISION_FILTERS["g1"] = _per_body_filters(_G1_SHAPE_FILTERS, _G1_SHAPES_PER_BODY)


def apply_collision_filter(model, humanoid_type):
    """Isaac Gym's collision filter words (humanoid.py:1205-1226).  G1's table is per collision SHAPE (40 shapes: torso_link carries three,
    elbow_roll_link two): since round 4 the model carries every shape, each with its own word."""
    if humanoid_type == "g1" and int(model.shapes_per_body.sum()) == len(_G1_SHAPE_FILTERS) and list(model.shapes_per_body) == _G1_SHAPES_PER_BODY:
        model.set_shape_filters(_G1_SHAPE_FILTERS)
    else:
        model.set_body_filters(COLLISION_FILTERS[humanoid_type])
    return model
