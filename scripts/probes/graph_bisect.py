"""dev tool: bisect which stage of the real optimizer step breaks hipGraph capture at a small batch (each stage in its own process)"""
import os, subprocess, sys
STAGES = ["A_preproc", "B_forward", "C_disc_loss", "D_backward", "E_full", "F_train_epoch", "G_as_test", "H_deep_stack"]
if len(sys.argv) > 1 and sys.argv[1] in STAGES:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from phc_amd.config import compose
    from phc_amd.env.tasks.vec_task import parse_task
    from phc_amd.learning.amp_agent import IMAmpAgent
    from phc_amd.learning.fast_ops import adam_clip_step, ppo_loss
    stage = sys.argv[1]
    if stage not in ("G_as_test", "H_deep_stack"):
        os.environ["PHC_NO_GRAPH"] = "1"
    n_envs, mb = int(sys.argv[2]), int(sys.argv[3])
    cfg = compose([f"env.num_envs={n_envs}", "env.motion_file=synthetic:2:3", f"learning.params.config.minibatch_size={mb}", "+learning.params.config.hip_graph=True",
                   "learning.params.config.amp_obs_demo_buffer_size=4096", "learning.params.config.amp_replay_buffer_size=4096"])
    task, env = parse_task(cfg)
    ag = IMAmpAgent(env, cfg)
    ag.init_train()
    ag.train_epoch()
    if stage == "H_deep_stack":   # is it the depth of the host stack at capture_end (pytest runs tests ~60 Python frames deep)?
        sys.setrecursionlimit(10000)
        def deep(k):
            if k == 0:
                ag.train_epoch()
                return
            deep(k - 1)
        deep(int(os.environ.get("DEPTH", "400")))
        print("graph" if ag._graph is not None else "eager", "ok")
        sys.exit(0)
    if stage == "G_as_test":
        ag.train_epoch()
        print("graph" if ag._graph is not None else "eager", "ok")
        sys.exit(0)
    if stage == "F_train_epoch":
        del os.environ["PHC_NO_GRAPH"]
        for _ in range(2):
            info = ag.train_epoch()
        print("graph" if ag._graph is not None else "eager", "ok")
        sys.exit(0)
    ag.set_train()
    ag._graph_static_dataset()
    ag._g_idx = ag._idx_buf[:ag.minibatch_size].clone()
    ag._g_step = torch.zeros((), dtype=torch.int64, device=ag.device)
    ag._g_info = torch.zeros(10, device=ag.device)
    d = ag._g_data
    idx, aidx = ag._g_idx, ag._g_idx[:ag._amp_minibatch_size]
    def body():
        obs = ag._preproc_obs(d["obs"], use_temp=ag.temp_running_mean, row_index=idx)
        a, r, dm = (ag._preproc_amp_obs(d[k], aidx) for k in ("amp_obs", "amp_obs_replay", "amp_obs_demo"))
        if stage == "A_preproc": return
        dm.requires_grad_(True)
        with ag._autocast():
            res = ag.model.forward_heads({"obs": obs, "amp_obs": a, "amp_obs_replay": r, "amp_obs_demo": dm})
        if stage == "B_forward": return
        di = ag._disc_loss(torch.cat([res["disc_agent_logit"], res["disc_agent_replay_logit"]], dim=0), res["disc_demo_logit"], dm)
        if stage == "C_disc_loss": return
        ppo, st = ppo_loss(res["mu"].contiguous(), res["value"].contiguous(), res["logstd"], d["actions"], d["old_logp_actions"], d["advantages"],
                           d["returns"], d["old_values"], d["mu"], d["sigma"], ag.e_clip, ag.critic_coef, ag.entropy_coef, ag.bounds_loss_coef, ag.clip_value,
                           unit_grad=True, row_index=idx)
        ag.grads.zero()
        (ppo + ag._disc_coef * di["disc_loss"]).backward()
    if stage == "E_full":
        body = ag._graph_step_body
    with ag.grads.shadow_scope():
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2): body()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        g.replay(); torch.cuda.synchronize()
    print("ok")
else:
    n_envs, mb = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("256", "2048")
    for st in STAGES:
        r = subprocess.run([sys.executable, __file__, st, n_envs, mb], capture_output=True, text=True)
        err = [l for l in r.stderr.strip().splitlines() if "Warn" not in l and "amdgpu.ids" not in l]
        print(f"envs={n_envs} mb={mb} {st:12s} rc={r.returncode} {r.stdout.strip()[-3:]} {err[-1][:140] if r.returncode and err else ''}", flush=True)
