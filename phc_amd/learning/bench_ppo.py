"""PPO samples/s for bench.py: the reference's `performance/total_fps = batch_size / (play_time + update_time)`
(phc/learning/common_agent.py:134-138) of full `train_epoch`s -- rollout of horizon_length steps with policy
inference, discriminator rewards, GAE, then mini_epochs x minibatches optimizer steps, each with ONE gradient
all-reduce when more than one rank runs."""
import time

import torch


MFMA_BF16_PEAK_TFLOPS = 2500.0    # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 MFMA peak ~2.5 PFLOP/s
MFMA_F32_PEAK_TFLOPS = 157.3       # fp32 vector / xf32-free matrix peak used for the fp32-GEMM fallback


def gemm_flops_per_epoch(agent):
    """Algorithmic GEMM flops of ONE train_epoch of `agent` (2 m k n per matrix product), from the network's Linear layers:
      rollout      per step: actor + critic forward on N envs, critic again for the next values; discriminator forward on the T x N AMP batch;
      update       per optimizer step, minibatch of m rows: actor / critic forward, input gradient (all layers but the first) and weight
                   gradient; discriminator on [agent; replay; demo] = 3 m_amp rows: forward, input gradient, weight gradient, plus the
                   gradient penalty on the demo rows (amp_agent.py:640-688): an input-gradient pass through every layer incl. the first, and
                   its own backward (double backward: one more forward-shaped and one more weight-gradient product per layer).
    Layers of width 1 (value / logit heads) are counted too (they run as dot-product kernels, not GEMMs: < 0.1 % of the total)."""
    from torch import nn
    net = agent.model.a2c_network

    def mk(mod):   # [(k, n)] of the Linear layers of a module, in order
        return [(m.in_features, m.out_features) for m in mod.modules() if isinstance(m, nn.Linear)]
    disc = mk(net._disc_mlp) + mk(net._disc_logits)
    critic = mk(net.critic_mlp) + mk(net.value)
    if hasattr(net, "pnn"):        # progressive network: the column that trains does forward + backward, frozen columns forward only
        cols = [mk(c) for c in net.pnn.actors]
        train_cols = [c for c, col in zip(cols, net.pnn.actors) if any(p.requires_grad for p in col.parameters())]
        actor_fwd = sum(cols, [])
        actor_bwd = sum(train_cols, [])
    else:
        actor_fwd = actor_bwd = mk(net.actor_mlp) + mk(net.mu)
    prod = lambda layers, rows: sum(2.0 * rows * k * n for k, n in layers)
    T, N = agent.horizon_length, agent.num_actors
    m = agent.minibatch_size
    m_amp = int(getattr(agent, "_amp_minibatch_size", 0) or agent.config.get("amp_minibatch_size", m))
    n_opt = agent.mini_epochs_num * agent.num_minibatches
    rollout = T * (prod(actor_fwd, N) + 2 * prod(critic, N)) + prod(disc, T * N)
    per_step = (prod(actor_fwd, m) + 2 * prod(actor_bwd, m) - prod(actor_bwd[:1], m)          # fwd + dW + dX (no dX into the observations)
                + 3 * prod(critic, m) - prod(critic[:1], m)
                + 3 * prod(disc, 3 * m_amp) - prod(disc[:1], 3 * m_amp)                       # BCE path on the three row blocks
                + 3 * prod(disc, m_amp))                                                      # gradient penalty on the demo rows: d logit / d input, then its backward
    return {"rollout": rollout, "update": n_opt * per_step, "per_optimizer_step": per_step}


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
    fl = gemm_flops_per_epoch(agent)
    upd_ms = 1e3 * sum(i["update_time"] for i in infos) / epochs
    peak = MFMA_BF16_PEAK_TFLOPS if agent.bf16 else MFMA_F32_PEAK_TFLOPS
    roof = {"ppo_roofline": {"bound": "mfma", "what": "GEMM flops of one PPO update (all optimizer steps of an epoch) / measured update time", "achieved": fl["update"] / (upd_ms * 1e-3) / 1e12,
                             "peak": peak, "unit": "TFLOP/s", "frac": fl["update"] / (upd_ms * 1e-3) / 1e12 / peak, "gemm_tflop_per_update": fl["update"] / 1e12,
                             "gemm_tflop_per_rollout": fl["rollout"] / 1e12, "gemm_gflop_per_optimizer_step": fl["per_optimizer_step"] / 1e9,
                             "note": "the update is a chain of ~25 mid-sized GEMMs (16384 x 1024 x 1024 and smaller) per optimizer step with HBM-bound normaliser / "
                                     "loss / reduction / Adam kernels between them (profiles/*_ppo_optimizer_step_kernels.txt): about half of the step is GEMM time"}}
    return {**comm, **roof, "ppo_samples_per_s": agent.batch_size * world * epochs / el, "ppo_epoch_ms": el / epochs * 1e3,
            "ppo_play_ms": 1e3 * sum(i["play_time"] for i in infos) / epochs, "ppo_update_ms": 1e3 * sum(i["update_time"] for i in infos) / epochs,
            "ppo_config": {"learning": str(cfg.get("learning_name", "")) or None, "horizon": agent.horizon_length, "batch_per_gpu": agent.batch_size, "minibatch": agent.minibatch_size,
                           "optimizer_steps_per_epoch": n_opt, "gemm_dtype": "bf16" if agent.bf16 else "f32", "actor_precision": "split_bf16" if getattr(agent, "_actor_split", False) else ("bf16" if agent.bf16 else "f32"),
                           "grad_allreduce_bytes": int(agent.grads.flat.numel() * 4), "collectives_per_epoch": int((agent.num_collectives - c0) / epochs),
                           "update_graph": agent._graph is not None,
                           "update_streams": 2 if agent._branches is not None else 1}}
