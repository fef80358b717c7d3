"""PPO samples/s for bench.py: the reference's `performance/total_fps = batch_size / (play_time + update_time)`
(phc/learning/common_agent.py:134-138) of full `train_epoch`s -- rollout of horizon_length steps with policy
inference, discriminator rewards, GAE, then mini_epochs x minibatches optimizer steps, each with ONE gradient
all-reduce when more than one rank runs."""
import time

import torch


def time_ppo_epochs(task, env, cfg, epochs, dist=None, warmup=1):
    from .amp_agent import IMAmpAgent
    # bench-sized replay buffers: the 200k x 1960 fp32 buffers of the shipped yaml are kept (1.57 GB each, HBM is 288 GB)
    agent = IMAmpAgent(env, cfg, dist=dist)
    agent.init_train()
    if agent._graph_enabled():
        # epoch 1 captures the update graph, epoch 2 the rollout segments, epoch 3 is the first one with a non-empty replay buffer (its
        # batch stops aliasing the agent batch: one-time buffer set-up, 92 vs 61 ms of update on a freshly started box -- scripts/ppo_epoch_trace.py);
        # from epoch 4 on every epoch costs the same
        warmup = max(warmup, 4)
    for _ in range(warmup):
        agent.train_epoch()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    agent.allreduce_timing = [] if dist is not None else None
    c0 = agent.num_collectives
    t0 = time.perf_counter()
    infos = [agent.train_epoch() for _ in range(epochs)]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    world = 1
    if dist is not None:
        t = torch.tensor([el], device=task.device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        world = dist.get_world_size()
    n_opt = agent.mini_epochs_num * agent.num_minibatches
    comm = {}
    if dist is not None:
        # evidence of the path's one collective: world size as every rank sees it, all-reduces actually issued, time of each
        ws = [None] * world
        dist.all_gather_object(ws, {"rank": dist.get_rank(), "world": dist.get_world_size(), "backend": dist.get_backend(),
                                    "device": torch.cuda.current_device(), "collectives": agent.num_collectives - c0})
        # the replicas stay in lock-step: every rank's parameter vector (fp64 sum and sum of squares) after the timed epochs
        flat = torch.cat([p.detach().double().flatten() for p in agent.model.parameters()])
        sig = torch.stack([flat.sum(), flat.square().sum()])
        sigs = [torch.zeros_like(sig) for _ in range(world)]
        dist.all_gather(sigs, sig)
        identical = all(bool(torch.equal(sigs[0], x)) for x in sigs)
        ar = sorted(a.elapsed_time(b) for a, b in agent.allreduce_timing)
        nbytes = int(agent.grads.flat.numel() * 4)
        med = ar[len(ar) // 2] if ar else None
        comm = {"ppo_comm": {"backend": dist.get_backend(), "ranks": ws, "replicas_identical": identical, "grad_allreduces_per_epoch": (agent.num_collectives - c0) / epochs,
                             "allreduce_bytes": nbytes, "allreduce_ms_median": med, "allreduce_ms_mean": (sum(ar) / len(ar)) if ar else None,
                             "allreduce_ms_max": ar[-1] if ar else None,
                             # ring all-reduce moves 2 (G-1)/G x bytes per rank
                             "allreduce_busbw_GBs": (2 * (world - 1) / world * nbytes / (med * 1e-3) / 1e9) if med else None,
                             "allreduce_ms_per_epoch": (sum(ar) / epochs) if ar else None}}
    return {**comm, "ppo_samples_per_s": agent.batch_size * world * epochs / el, "ppo_epoch_ms": el / epochs * 1e3,
            "ppo_play_ms": 1e3 * sum(i["play_time"] for i in infos) / epochs, "ppo_update_ms": 1e3 * sum(i["update_time"] for i in infos) / epochs,
            "ppo_config": {"horizon": agent.horizon_length, "batch_per_gpu": agent.batch_size, "minibatch": agent.minibatch_size,
                           "optimizer_steps_per_epoch": n_opt, "gemm_dtype": "bf16" if agent.bf16 else "f32",
                           "grad_allreduce_bytes": int(agent.grads.flat.numel() * 4), "collectives_per_epoch": int((agent.num_collectives - c0) / epochs),
                           "update_graph": agent._graph is not None}}
