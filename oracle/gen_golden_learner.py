"""TEST INFRASTRUCTURE -- goldens for the learner rows (A1, P2, P3, P5, P6, P8, B4) from the reference's OWN method bodies.

Run in the build container (needs /root/reference):  python oracle/gen_golden_learner.py
Writes tests/golden/learner_fns.npz, learner_step.npz, learner_step_policy_actions.npz, learner_step_wide.npz, learner_pnn.npz, pd_offset_scale.npz.  Deterministic: every draw comes from a
seeded generator, so re-running reproduces the committed files bit for bit.

How the reference is executed: `ref_shim.install()` makes `phc.learning.*` importable with EMPTY rl_games agent base classes
(oracle/rl_games_stub.py); agents are created with `__new__` (their constructors need a simulator) and receive exactly the attributes the
called method reads; the networks are built by the reference's own builders (`AMPBuilder`, `AMPPNNBuilder`, `AMPMCPBuilder` ->
`network_builder.A2CBuilder.Network`, `pnn.PNN`) from the reference's own `phc/data/cfg/learning/im*.yaml` with smaller layer widths.

  learner_fns   CommonAgent.discount_values (common_agent.py:493-505), _actor_loss / _critic_loss (:564-587), bound_loss (:512-520),
                _calc_advs (:589-599)
  learner_step  AMPAgent._calc_disc_rewards / _combine_rewards (amp_agent.py:848-878) and ONE full AMPAgent.calc_gradients (:554-688):
                forward through ModelAMPContinuous.Network (amp_models.py), _disc_loss (:732-789), backward, clip_grad_norm_(50), Adam --
                losses, every parameter's gradient, every parameter after the step, the running statistics after the step;
                learner_step_policy_actions / learner_step_wide: the same with actions drawn from the fixture policy itself (in-distribution
                neglogp: the bf16 device path is compared on the ACTOR terms too), the second at wider layers (192-96, minibatch 256)
  learner_pnn   the PNN / MCP networks' state-dict key sets + forward outputs, and the reference's checkpoint loaders
                network_loader.load_pnn (:54-74) / load_mcp_mlp (:11-52) on a checkpoint with the reference's key set
  pd_offset_scale  Humanoid._build_pd_action_offset_scale (humanoid.py:1331-1409) on the joint limits of the SMPL / H1 / G1 assets, all
                flag combinations"""
import copy
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

ref_shim.install()
REF = ref_shim.REFERENCE_ROOT
OUT = os.path.join(ROOT, "tests", "golden")

O, M, A = 40, 36, 9          # obs, amp obs (3 steps x 12), actions
T, N = 8, 16                  # horizon, envs
MB, AMB = 64, 32              # minibatch, amp minibatch


def net_params(name, units=(64, 32), disc_units=(48, 24)):
    p = yaml.safe_load(open(os.path.join(REF, "phc/data/cfg/learning", name)))["params"]
    p["network"]["mlp"]["units"] = list(units)
    p["network"]["disc"]["units"] = list(disc_units)
    return p


def np_state(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def fake_agent(cls, params, model, rms, vms, ams):
    c = params["config"]
    a = cls.__new__(cls)
    a.model, a.running_mean_std, a.value_mean_std, a._amp_input_mean_std = model, rms, vms, ams
    a.normalize_input, a.normalize_value, a._normalize_amp_input = True, True, True
    a._disc_reward_mean_std = None
    a.ppo_device = "cpu"
    a.horizon_length, a.gamma, a.tau = T, c["gamma"], c["tau"]
    a.e_clip, a.critic_coef, a.entropy_coef, a.bounds_loss_coef = c["e_clip"], c["critic_coef"], c["entropy_coef"], c["bounds_loss_coef"]
    a.clip_value, a.truncate_grads, a.grad_norm = c["clip_value"], c["truncate_grads"], c["grad_norm"]
    a.normalize_advantage = c["normalize_advantage"]
    a.last_lr = float(c["learning_rate"])
    a._disc_coef, a._disc_logit_reg, a._disc_grad_penalty = c["disc_coef"], c["disc_logit_reg"], c["disc_grad_penalty"]
    a._disc_weight_decay, a._disc_reward_scale = c["disc_weight_decay"], c["disc_reward_scale"]
    a._task_reward_w, a._disc_reward_w = c["task_reward_w"], c["disc_reward_w"]
    a._amp_minibatch_size = AMB
    a.temp_running_mean = True
    a.is_rnn, a.mixed_precision, a.multi_gpu = False, False, False
    a.vec_env = types.SimpleNamespace(env=types.SimpleNamespace(task=types.SimpleNamespace(_num_amp_obs_steps=3)))
    a.scaler = torch.cuda.amp.GradScaler(enabled=False)
    a.optimizer = torch.optim.Adam(model.parameters(), a.last_lr, eps=1e-08, weight_decay=c.get("weight_decay", 0.0))  # common_agent.py:67
    return a


def warm_stats(mod, dim, g, batches=3, rows=256, scale=2.0, shift=0.5):
    mod.train()
    for _ in range(batches):
        mod(torch.randn(rows, dim, generator=g) * scale + shift)
    return mod


def gen_fns():
    ca = ref_shim.ref_module("phc.learning.common_agent")
    g = torch.Generator().manual_seed(101)
    a = ca.CommonAgent.__new__(ca.CommonAgent)
    a.horizon_length, a.gamma, a.tau, a.bounds_loss_coef, a.normalize_advantage = T, 0.99, 0.95, 10, True
    fd = (torch.rand(T, N, generator=g) < 0.2).float()
    v, r, nv = (torch.randn(T, N, 1, generator=g) for _ in range(3))
    nv = nv * (1 - (torch.rand(T, N, 1, generator=g) < 0.1).float())      # next values zeroed on termination (amp_agent.py:354-356)
    out = {"gae_fdones": fd, "gae_values": v, "gae_rewards": r, "gae_next_values": nv, "gae_advs": a.discount_values(fd, v, r, nv),
           "gamma": torch.tensor(a.gamma), "tau": torch.tensor(a.tau)}
    B = 96
    old_lp, lp = torch.randn(B, generator=g) * 0.3 + 12, torch.randn(B, generator=g) * 0.3 + 12
    adv = torch.randn(B, generator=g)
    ai = a._actor_loss(old_lp, lp, adv, 0.2)
    out.update(al_old_logp=old_lp, al_logp=lp, al_adv=adv, al_loss=ai["actor_loss"], al_clipped=ai["actor_clipped"].float())
    vp, val, ret = (torch.randn(B, 1, generator=g) for _ in range(3))
    val = vp + 0.5 * torch.randn(B, 1, generator=g)
    out.update(cl_value_preds=vp, cl_values=val, cl_returns=ret, cl_loss_clip=a._critic_loss(vp, val, 0.2, ret, True)["critic_loss"],
               cl_loss_noclip=a._critic_loss(vp, val, 0.2, ret, False)["critic_loss"])
    mu = torch.randn(B, A, generator=g) * 1.2
    out.update(bl_mu=mu, bl_loss=a.bound_loss(mu))
    rets, vals = torch.randn(B, 1, generator=g), torch.randn(B, 1, generator=g)
    out.update(adv_returns=rets, adv_values=vals, adv_out=a._calc_advs({"returns": rets, "values": vals}))
    np.savez_compressed(os.path.join(OUT, "learner_fns.npz"), **{k: t.numpy() for k, t in out.items()})


def build_reference_model(params, builder_mod, builder_cls, model_name="amp", extra=None):
    rms_mod = ref_shim.ref_module("phc.utils.running_mean_std")
    am = ref_shim.ref_module("phc.learning.amp_models")
    b = getattr(ref_shim.ref_module(builder_mod), builder_cls)()
    b.load(params["network"])
    rms = rms_mod.RunningMeanStd((O,))
    kw = dict(actions_num=A, input_shape=(O,), amp_input_shape=(M,), value_size=1, num_seqs=N, mean_std=rms)
    kw.update(extra or {})
    net = b.build(model_name, **kw)
    return am.ModelAMPContinuous.Network(net), rms, rms_mod


def gen_step(fname="learner_step.npz", dims=None, units=(64, 32), disc_units=(48, 24), policy_actions=False, seed=7, gseed=202):
    """One whole AMPAgent.calc_gradients.  `policy_actions`: the minibatch's actions are draws of the fixture policy itself,
    a = mu(obs) + sigma * clip(N(0, 1), +-2), and the stored old mu is that mu plus a small perturbation -- neglogp stays O(10), so the
    actor loss, the KL and the actor's gradients can be compared after bf16 GEMMs too (VERDICT r2 weak #1a: with N(0, 0.7) actions, ~13 sigma
    from mu, neglogp ~ 760 and a 1e-2 relative error of mu moves it by O(1)).  `dims` = (O, M, A, MB, AMB) for a wider fixture."""
    global O, M, A, MB, AMB
    saved = (O, M, A, MB, AMB)
    if dims is not None:
        O, M, A, MB, AMB = dims
    try:
        _gen_step(fname, units, disc_units, policy_actions, seed, gseed)
    finally:
        O, M, A, MB, AMB = saved


def _gen_step(fname, units, disc_units, policy_actions, seed, gseed):
    aa = ref_shim.ref_module("phc.learning.amp_agent")
    torch.manual_seed(seed)
    params = net_params("im.yaml", units, disc_units)
    model, rms, rms_mod = build_reference_model(params, "phc.learning.amp_network_builder", "AMPBuilder")
    g = torch.Generator().manual_seed(gseed)
    with torch.no_grad():   # biases are zero-initialised (network_builder.py:277-284): give them values so that their gradients matter
        for n_, p in model.named_parameters():
            if n_.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    vms, ams = rms_mod.RunningMeanStd((1,)), rms_mod.RunningMeanStd((M,))
    warm_stats(rms, O, g)
    warm_stats(vms, 1, g, scale=3.0, shift=1.0)
    warm_stats(ams, M, g, scale=1.5, shift=-0.2)
    agent = fake_agent(aa.AMPAgent, params, model, rms, vms, ams)
    agent.running_mean_std_temp = copy.deepcopy(rms)      # pre_epoch (amp_agent.py:527-528)
    agent.running_mean_std_temp.freeze()
    warm_stats(rms, O, g, batches=1)                       # the live statistics have moved on since the snapshot
    out = {"model/" + k: v for k, v in np_state(model.state_dict()).items()}
    for nm, m in (("running_mean_std", rms), ("running_mean_std_temp", agent.running_mean_std_temp), ("reward_mean_std", vms), ("amp_input_mean_std", ams)):
        out.update({f"{nm}/" + k: v for k, v in np_state(m.state_dict()).items()})
    out["param_names"] = np.array([n_ for n_, _ in model.named_parameters()])
    out["param_requires_grad"] = np.array([p.requires_grad for _, p in model.named_parameters()])

    # P6: discriminator reward + combination on a [T, N, M] rollout tensor (evaluation mode, as play_steps does)
    agent.set_eval()
    amp_roll = torch.randn(T, N, M, generator=g) * 1.5 - 0.2
    task_r = torch.rand(T, N, 1, generator=g)
    disc_r = agent._calc_disc_rewards(amp_roll)
    out.update(p6_amp_obs=amp_roll.numpy(), p6_task_rewards=task_r.numpy(), p6_disc_rewards=disc_r.numpy(),
               p6_combined=agent._combine_rewards(task_r, {"disc_rewards": disc_r}).numpy())

    # P8: one calc_gradients on a fixed minibatch
    d = {"old_values": torch.randn(MB, 1, generator=g), "old_logp_actions": torch.randn(MB, generator=g) * 0.5 + 10,
         "advantages": torch.randn(MB, generator=g), "mu": torch.randn(MB, A, generator=g) * 0.7,
         "sigma": torch.full((MB, A), float(np.exp(-2.9))), "returns": torch.randn(MB, 1, generator=g),
         "actions": torch.randn(MB, A, generator=g) * 0.7, "obs": torch.randn(MB, O, generator=g) * 2 + 0.5,
         "amp_obs": torch.randn(MB, M, generator=g) * 1.5 - 0.2, "amp_obs_replay": torch.randn(MB, M, generator=g) * 1.5 - 0.2,
         "amp_obs_demo": torch.randn(MB, M, generator=g) * 1.2 + 0.3}
    if policy_actions:
        with torch.no_grad():
            agent.set_eval()
            mu_pol = model.a2c_network.eval_actor({"obs": agent.running_mean_std_temp(d["obs"])})[0]
            d["actions"] = mu_pol + d["sigma"] * torch.randn(MB, A, generator=g).clamp(-2.0, 2.0)
            d["mu"] = mu_pol + d["sigma"] * 0.3 * torch.randn(MB, A, generator=g)      # the "old" policy: a small KL
    # old_logp consistent with the current policy so that the ratio straddles the clip range
    with torch.no_grad():
        agent.set_eval()
        res = model({"is_train": True, "prev_actions": d["actions"], "obs": agent.running_mean_std_temp(d["obs"]), "amp_obs": ams(d["amp_obs"]),
                     "amp_obs_replay": ams(d["amp_obs_replay"]), "amp_obs_demo": ams(d["amp_obs_demo"])})
        d["old_logp_actions"] = res["prev_neglogp"] + torch.randn(MB, generator=g) * 0.25
    out.update({"in/" + k: v.numpy().copy() for k, v in d.items()})
    agent.calc_gradients({k: v.clone() for k, v in d.items()})
    tr = agent.train_result
    for k in ("actor_loss", "critic_loss", "b_loss", "entropy", "kl", "disc_loss", "disc_grad_penalty", "disc_logit_loss", "disc_agent_acc",
              "disc_demo_acc", "actor_clip_frac"):
        out["res/" + k] = np.asarray(tr[k].detach().numpy() if torch.is_tensor(tr[k]) else tr[k])
    out["res/disc_agent_logit"] = tr["disc_agent_logit"].numpy()
    out["res/disc_demo_logit"] = tr["disc_demo_logit"].numpy()
    for n_, p in model.named_parameters():
        if p.grad is not None:
            out["grad/" + n_] = p.grad.numpy().copy()          # after clip_grad_norm_(50)
    out.update({"model_after/" + k: v for k, v in np_state(model.state_dict()).items()})
    for nm, m in (("running_mean_std", rms), ("amp_input_mean_std", ams)):
        out.update({f"{nm}_after/" + k: v for k, v in np_state(m.state_dict()).items()})
    sd = agent.optimizer.state_dict()
    out["opt/param_ids"] = np.array(sd["param_groups"][0]["params"])
    out["opt/state_ids"] = np.array(sorted(sd["state"]))
    for i, st in sd["state"].items():
        out[f"opt/{i}/exp_avg"], out[f"opt/{i}/exp_avg_sq"], out[f"opt/{i}/step"] = st["exp_avg"].numpy(), st["exp_avg_sq"].numpy(), np.asarray(float(st["step"]))
    out["dims"] = np.array([O, M, A, T, N, MB, AMB])
    out["units"], out["disc_units"] = np.array(units), np.array(disc_units)
    np.savez_compressed(os.path.join(OUT, fname), **out)


def gen_pnn():
    """PNN / MCP networks from the reference's builders + the reference's checkpoint loaders."""
    torch.manual_seed(11)
    g = torch.Generator().manual_seed(303)
    out = {}
    num_prim = 3
    detail = {"fut_tracks": False, "obs_v": 6, "num_traj_samples": 1, "track_bodies": [], "num_prim": num_prim, "training_prim": 1,
              "actors_to_load": 0, "has_lateral": False, "models_path": []}
    p = net_params("im_pnn.yaml")
    model, rms, _ = build_reference_model(p, "phc.learning.amp_network_pnn_builder", "AMPPNNBuilder", "amp_pnn",
                                          extra=dict(self_obs_size=O - 10, task_obs_size=10, task_obs_size_detail=detail))
    out["pnn_param_names"] = np.array([n_ for n_, _ in model.named_parameters()])
    out["pnn_requires_grad"] = np.array([q.requires_grad for _, q in model.named_parameters()])
    out.update({"pnn_model/" + k: v for k, v in np_state(model.state_dict()).items()})
    x = torch.randn(24, O, generator=g)
    model.eval()
    with torch.no_grad():
        mu, sigma = model.a2c_network.eval_actor({"obs": x})
        out["pnn_x"], out["pnn_mu"], out["pnn_sigma"] = x.numpy(), mu.numpy(), sigma.numpy()
        out["pnn_value"] = model.a2c_network.eval_critic({"obs": x}).numpy()
    # env-side loader on a checkpoint dict with the reference's key set (network_loader.py:54-74)
    nl = ref_shim.ref_module("phc.learning.network_loader")
    ck = {"model": model.state_dict()}
    pnn = nl.load_pnn(ck, num_prim=num_prim, has_lateral=False, activation="relu", device="cpu")
    with torch.no_grad():
        _, acts = pnn(x)
        out["load_pnn_actions"] = torch.stack(acts, dim=1).numpy()
    # forward_pmcp.py:44-51 column copy, as the script does it
    ck2 = {"model": copy.deepcopy(model.state_dict())}
    pnn_keys = [k for k in ck2["model"] if "pnn" in k]
    src, dst = [k for k in pnn_keys if "actors.1" in k], [k for k in pnn_keys if "actors.2" in k]
    for s_, d_ in zip(src, dst):
        ck2["model"][d_].copy_(ck2["model"][s_])
    out.update({"pmcp_model/" + k: v for k, v in np_state(ck2["model"]).items()})

    # MCP composer network
    pm = net_params("im_mcp.yaml")
    detail_m = dict(detail)
    torch.manual_seed(12)
    rms_mod = ref_shim.ref_module("phc.utils.running_mean_std")
    am = ref_shim.ref_module("phc.learning.amp_models")
    b = ref_shim.ref_module("phc.learning.amp_network_mcp_builder").AMPMCPBuilder()
    b.load(pm["network"])
    netm = b.build("amp_mcp", actions_num=num_prim, input_shape=(O,), amp_input_shape=(M,), value_size=1, num_seqs=N, mean_std=rms_mod.RunningMeanStd((O,)),
                   self_obs_size=O - 10, task_obs_size=10, task_obs_size_detail=detail_m)
    mm = am.ModelAMPContinuous.Network(netm)
    out["mcp_param_names"] = np.array([n_ for n_, _ in mm.named_parameters()])
    out["mcp_requires_grad"] = np.array([q.requires_grad for _, q in mm.named_parameters()])
    out.update({"mcp_model/" + k: v for k, v in np_state(mm.state_dict()).items()})
    out["mcp_has_softmax"] = np.asarray(bool(pm["network"].get("has_softmax", True)))
    out["mcp_ending_act"] = np.asarray(bool(pm["network"].get("ending_act", True)))
    mm.eval()
    with torch.no_grad():
        mu, sigma = mm.a2c_network.eval_actor({"obs": x})
        out["mcp_mu"], out["mcp_sigma"] = mu.numpy(), sigma.numpy()
    # load_mcp_mlp on a plain `amp` checkpoint (actor_mlp + mu), network_loader.py:11-52
    p0 = net_params("im.yaml")
    torch.manual_seed(13)
    m0, _, _ = build_reference_model(p0, "phc.learning.amp_network_builder", "AMPBuilder")
    with torch.no_grad():
        for n_, q in m0.named_parameters():
            if n_.endswith("bias"):
                q.copy_(torch.randn(q.shape, generator=g) * 0.05)
    out.update({"amp_model/" + k: v for k, v in np_state(m0.state_dict()).items()})
    mlp = nl.load_mcp_mlp({"model": m0.state_dict()}, activation="relu", device="cpu", mlp_name="actor_mlp")
    with torch.no_grad():
        out["load_mcp_mlp_out"] = mlp(x).numpy()
        out["amp_mu"] = m0.a2c_network.eval_actor({"obs": x})[0].numpy()
    # lateral connections (pnn.py:25-38,84-126; off in the shipped yamls)
    torch.manual_seed(14)
    pnn_mod = ref_shim.ref_module("phc.learning.pnn")
    lat = pnn_mod.PNN({"input_size": O, "units": [64, 32], "activation": "relu", "dense_func": torch.nn.Linear}, output_size=A, numCols=3, has_lateral=True)
    out.update({"lat_model/" + k: v for k, v in np_state(lat.state_dict()).items()})
    lat.eval()
    with torch.no_grad():
        a_all, outs = lat(x, idx=-1)
        out["lat_out_all"] = torch.stack(outs, dim=1).numpy()
        a1, outs1 = lat(x, idx=1)
        out["lat_out_idx1"], out["lat_n_idx1"] = a1.numpy(), np.asarray(len(outs1))
        out["lat_out_idx0"] = lat(x, idx=0)[0].numpy()
    np.savez_compressed(os.path.join(OUT, "learner_pnn.npz"), **out)


def gen_pd():
    """A1 on the three shipped morphologies: the limits are those of the compiled assets (phc_amd.model; pinned to the reference's MJCF
    parse elsewhere), the arithmetic is the reference's `Humanoid._build_pd_action_offset_scale` run on a `__new__`-made task."""
    hm = ref_shim.ref_module("phc.env.tasks.humanoid")
    from phc_amd.model import load_model
    out = {}
    for asset, htype in (("smpl_humanoid", "smpl"), ("h1_humanoid", "h1"), ("g1_humanoid", "g1")):
        m = load_model(asset)
        lo, hi = m.dof_limits()
        offs = [0]
        for i in range(1, m.num_bodies):
            if m.dof_count[i]:
                offs.append(offs[-1] + int(m.dof_count[i]))
        names = [m.body_names[i] for i in range(1, m.num_bodies) if m.dof_count[i]]
        out[f"{htype}/lim_low"], out[f"{htype}/lim_high"] = lo, hi
        variants = [(False, False, True), (True, False, True)] + ([(False, True, True), (False, True, False)] if htype == "smpl" else [])
        for bias, pdoff, upright in variants:
            t = hm.Humanoid.__new__(hm.Humanoid)
            t._dof_offsets, t._dof_names = offs, names
            t.dof_limits_lower, t.dof_limits_upper = torch.from_numpy(lo.copy()), torch.from_numpy(hi.copy())
            t._bias_offset, t._has_smpl_pd_offset, t._has_upright_start, t.humanoid_type, t.device = bias, pdoff, upright, htype, "cpu"
            t._build_pd_action_offset_scale()
            tag = f"{htype}/bias{int(bias)}_pdoff{int(pdoff)}_upright{int(upright)}"
            out[tag + "/offset"], out[tag + "/scale"] = np.asarray(t._pd_action_offset, dtype=np.float32), np.asarray(t._pd_action_scale, dtype=np.float32)
    np.savez_compressed(os.path.join(OUT, "pd_offset_scale.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["fns", "step", "step_policy", "step_wide", "pnn", "pd"]
    for w in which:
        {"fns": gen_fns, "step": gen_step, "pnn": gen_pnn, "pd": gen_pd,
         "step_policy": lambda: gen_step("learner_step_policy_actions.npz", policy_actions=True, seed=21, gseed=203),
         "step_wide": lambda: gen_step("learner_step_wide.npz", dims=(128, 96, 23, 256, 64), units=(192, 96), disc_units=(96, 48),
                                       policy_actions=True, seed=22, gseed=204)}[w]()
        print("wrote", w)
