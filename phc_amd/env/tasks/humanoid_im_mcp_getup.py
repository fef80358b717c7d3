"""HumanoidImMCPGetup = primitive composer + fall recovery (reference: phc/env/tasks/humanoid_im_mcp_getup.py), the task of
env_im_getup_mcp.yaml (config 3's last stage)."""
from .humanoid_im_getup import HumanoidImGetup
from .humanoid_im_mcp import MCPMixin


class HumanoidImMCPGetup(MCPMixin, HumanoidImGetup):

    def __init__(self, cfg, sim_params=None, physics_engine=None, device_type="cuda", device_id=0, headless=True):
        self._mcp_config(cfg)
        super().__init__(cfg=cfg, sim_params=sim_params, physics_engine=physics_engine, device_type=device_type, device_id=device_id,
                         headless=headless)
        self._mcp_load()
