"""dev tool: variations around tests/test_env_gpu.py::test_ppo_train_epoch_on_device with the update graph forced on"""
import os, subprocess, sys
VARS = ["seed0_first", "seed0_oracle", "seed0_scipy", "seed0_pytestimport"]
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from phc_amd.config import compose
    from phc_amd.env.tasks.vec_task import parse_task
    from phc_amd.learning.amp_agent import IMAmpAgent
    v = sys.argv[1]
    if v.endswith("oracle"):
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
        import phc_oracle  # noqa: F401
    if v.endswith("scipy"):
        import scipy.ndimage, scipy.sparse.linalg, scipy.special  # noqa: F401,E401
    if v.endswith("pytestimport"):
        import pytest  # noqa: F401
    if v.startswith("seed0"):
        torch.manual_seed(0)
    over = [f"env.num_envs={512 if v.endswith('envs512') else 256}", "env.motion_file=synthetic:2:3", "learning.params.config.minibatch_size=2048",
            "learning.params.config.amp_obs_demo_buffer_size=4096", "learning.params.config.amp_replay_buffer_size=4096", "+learning.params.config.hip_graph=True"]
    if v.endswith("amp1024"):
        over.append("learning.params.config.amp_minibatch_size=1024")
    cfg = compose(over)
    task, env = parse_task(cfg)
    ag = IMAmpAgent(env, cfg)
    ag.init_train()
    for _ in range(2):
        ag.train_epoch()
    print("graph" if ag._graph is not None else "eager", "ok")
else:
    for v in VARS:
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True)
        print(f"{v:16s} rc={r.returncode} {r.stdout.strip()[-9:]}", flush=True)
