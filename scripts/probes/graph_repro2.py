import os, subprocess, sys
VARS = ["inline_copy", "inline_detach"]
if len(sys.argv) > 1:
    v = sys.argv[1]
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path[:0] = [root, os.path.join(root, "tests"), os.path.join(root, "oracle")]
    if v == "plain_env":
        os.environ["PHC_FORCE_GRAPH"] = "1"
        import test_env_gpu as t
        t.test_ppo_train_epoch_on_device()
    elif v == "plain_cfgkey":
        import test_env_gpu as t
        mk = t.make_task
        t.make_task = lambda n, **kw: mk(n, **dict(kw, **{"+learning.params.config.hip_graph": True}))
        t.test_ppo_train_epoch_on_device()
    else:
        import numpy as np, torch
        from phc_amd.config import compose
        from phc_amd.env.tasks.vec_task import parse_task
        from phc_amd.learning.amp_agent import IMAmpAgent
        torch.manual_seed(0)
        cfg = compose(["env.num_envs=256", "env.motion_file=synthetic:2:3", "learning.params.config.minibatch_size=2048",
                       "learning.params.config.amp_obs_demo_buffer_size=4096", "learning.params.config.amp_replay_buffer_size=4096",
                       "+learning.params.config.hip_graph=True"])
        task, env = parse_task(cfg)
        agent = IMAmpAgent(env, task.cfg)
        agent.init_train()
        w0 = agent.model.a2c_network.mu.weight.detach().clone() if v == "inline_detach" else agent.model.a2c_network.mu.weight.clone()
        for _ in range(2):
            info = agent.train_epoch()
            assert np.isfinite([info["actor_loss"], info["critic_loss"], info["disc_loss"], info["kl"], info["mean_task_reward"]]).all(), info
        assert not torch.equal(w0, agent.model.a2c_network.mu.weight)
    print("ok")
else:
    for v in VARS:
        r = subprocess.run([sys.executable, __file__, v], capture_output=True, text=True)
        print(f"{v:14s} rc={r.returncode} {r.stdout.strip()[-3:]}", flush=True)
