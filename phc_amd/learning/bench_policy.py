"""The env step in the regime a TRAINED policy produces (bench.py `other_workloads.trained_policy`).

The headline protocol (a fixed random-action tensor) is a reset storm: every humanoid falls within ~5 steps, so ~20 % of the envs are reset on every
step and the reset launch is priced at a rate no training run sees.  Here the same 4096-env task tracks the 64-clip locomotion library
(`env.motion_file=locomotion:64:0`) with a policy trained ON THE SPOT -- `IMAmpAgent.train_epoch` for at most `train_s` seconds (`mini_epochs=3`: 24 optimizer steps
per rollout at 4096 envs), stopping early once episodes last `target_episode_len` steps on average (episodes start at a random
clip time and end with the 8 s clip: a perfect tracker averages ~120 steps) -- and then K env steps are timed with the policy's actions
(mu + sigma * noise, as a rollout draws them): whole step and the three launches (reset of the finished envs, stepper, post-physics) by HIP events, resets per step
counted on the device.

    python -m phc_amd.learning.bench_policy [--envs 4096] [--train-s 90] [--steps 300]      -> one JSON line
"""
import argparse
import json
import sys
import time

import numpy as np
import torch


def run(envs=4096, train_s=90.0, steps=300, target_episode_len=90.0, solver=(), log=None, use_policy_graph=True):
    from ..config import compose
    from ..env.tasks.vec_task import parse_task
    from .amp_agent import IMAmpAgent
    torch.manual_seed(0)
    mini_epochs = max(1, 24 // max(1, envs * 32 // 16384))
    cfg = compose([f"env.num_envs={envs}", "env.motion_file=locomotion:64:0", f"learning.params.config.mini_epochs={mini_epochs}"] + [f"+solver.{kv}" for kv in solver])
    task, env = parse_task(cfg)
    agent = IMAmpAgent(env, cfg)
    agent.init_train()
    t0, ep_len, epochs = time.time(), 0.0, 0
    while time.time() - t0 < train_s:
        agent.train_epoch()
        epochs += 1
        if epochs % 50 == 0:
            ep_len = agent.batch_size / max(float(agent.exp["dones"].float().sum()), 1.0)
            if log:
                log(f"[bench_policy] epoch {epochs} ({time.time() - t0:.0f} s): mean episode length {ep_len:.1f} steps")
            if ep_len >= target_episode_len:
                break
    train_wall = time.time() - t0
    ep_len = agent.batch_size / max(float(agent.exp["dones"].float().sum()), 1.0)

    # ---- timed region: the rollout's own step sequence, eager launches, events around the env part of every step ----
    agent.set_eval()
    net = agent.model.a2c_network
    N, dev = task.num_envs, task.device
    actions = torch.zeros(N, task.num_actions, device=dev)
    resets = torch.zeros((), dtype=torch.long, device=dev)

    def policy_launches():
        with torch.no_grad(), agent._autocast():
            mu, logstd = net.eval_actor(agent._preproc_obs(task.obs_buf))
        torch.clamp(mu.float() + torch.exp(logstd.float()) * torch.randn_like(mu, dtype=torch.float32), -1.0, 1.0, out=actions)

    # The policy's inference (normaliser, three GEMMs, sampling: ~10 launches) is ONE captured hipGraph here -- task.obs_buf -> `actions`, both at fixed addresses --
    # so that the host issues four launches per step and the GPU, not the host, is the bottleneck of the loop: the HIP events then bracket device time
    # (with eager inference the host needs ~250 us per step and an event pair around `reset_done()` measures the host, 28 us, not the 6 us launch).
    g_pol = None
    if use_policy_graph:
        with agent.grads.shadow_scope():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    policy_launches()
            torch.cuda.current_stream().wait_stream(side)
            g_pol = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_pol):
                policy_launches()
    policy = g_pol.replay if g_pol is not None else policy_launches

    def one(ev_all=None, ev_reset=None, ev_launch=None):
        policy()
        if ev_all is not None:
            ev_all[0].record()
        if ev_reset is not None:
            ev_reset[0].record()
        task.reset_done()
        if ev_reset is not None:
            ev_reset[1].record()
        if ev_launch is not None:
            task._launch_events = ev_launch
        env.step(actions)
        if ev_all is not None:
            ev_all[1].record()
        resets.add_(task.reset_buf.sum())

    E = lambda: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    with agent.grads.shadow_scope():
        for _ in range(30):
            one()
        resets.zero_()
        ev_all = [E() if k % 4 == 3 else None for k in range(steps)]      # whole env step: on the steps that carry no inner event pair (a pair costs the stream ~1.5 us)
        ev_reset = [E() if k % 4 == 1 else None for k in range(steps)]
        ev_sim = [E() if k % 4 == 0 else None for k in range(steps)]
        ev_post = [E() if k % 4 == 2 else None for k in range(steps)]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(steps):
            one(ev_all[k], ev_reset[k], (ev_sim[k], ev_post[k]) if (ev_sim[k] is not None or ev_post[k] is not None) else None)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t1
    mean = lambda evs: float(np.mean([a.elapsed_time(b) for a, b in (e for e in evs if e is not None)])) * 1e3
    step_us = mean(ev_all)
    return {"workload": f"SMPL humanoid, {N} envs, 64-clip locomotion library, actions of a policy trained on the spot ({epochs} epochs, {train_wall:.0f} s, "
                        f"mini_epochs={mini_epochs}); mean episode length {ep_len:.0f} steps",
            "envs_per_gpu": N, "steps": steps, "policy_train_s": train_wall, "policy_train_epochs": epochs, "mean_episode_length_steps": ep_len,
            "resets_per_step_share": float(resets.item()) / (steps * N), "envs_within_5_steps_of_a_reset": float((task.progress_buf < 5).float().mean().item()),
            "env_step_us": step_us, "value": N / (step_us * 1e-6), "unit": "env-steps/s",
            "reset_launch_us": mean(ev_reset), "stepper_launch_us": mean(ev_sim), "post_physics_launch_us": mean(ev_post),
            "wall_us_per_step_incl_policy_inference": wall / steps * 1e6,
            "policy_inference_as_one_graph": g_pol is not None,
            "method": "HIP events around task.reset_done() + env.step(actions) on every 4th timed step and around each of the three launches on one of the other three; the policy's "
                      "inference between two steps is one captured hipGraph, so the loop is GPU-bound and the events bracket device time"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--train-s", type=float, default=90.0)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--target-episode-len", type=float, default=90.0)
    ap.add_argument("--solver", action="append", default=[])
    a = ap.parse_args()
    out = run(a.envs, a.train_s, a.steps, a.target_episode_len, a.solver, log=lambda s: print(s, file=sys.stderr, flush=True))
    print("POLICY_JSON" + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
