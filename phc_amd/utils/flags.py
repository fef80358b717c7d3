"""Process-global mutable flags object, same role and attribute names as the reference's
phc/utils/flags.py:8-13 (+ the attributes phc/run_hydra.py:278-295 adds at start-up)."""


class Flags(object):
    def __init__(self, items):
        for key, val in items.items():
            setattr(self, key, val)


flags = Flags({
    "test": False, "debug": False, "real_traj": False, "im_eval": False, "follow": False, "show_traj": False,
    "server_mode": False, "no_collision_check": False, "no_virtual_display": True, "render_o3d": False,
    "add_proj": False, "has_eval": True, "trigger_input": False, "fixed": False, "divide_group": False,
    "small_terrain": False,
})
