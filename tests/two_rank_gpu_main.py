"""Child of test_two_rank_update_on_one_gpu: two ranks (gloo, both on cuda:0) run the multi-rank code path of the learner on the device --
eager all-reduce between the captured forward / backward and the optimizer launches -- and report whether the replicas stayed identical."""
import json
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, world, port, graph, q, backend="gloo", split=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = rank if backend == "nccl" else 0          # RCCL: one GPU per rank; gloo: both ranks share cuda:0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{dev}"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from phc_amd.config import compose
    from phc_amd.env.tasks.vec_task import parse_task
    from phc_amd.learning.amp_agent import IMAmpAgent
    torch.manual_seed(100 + rank)
    cfg = compose(["env.num_envs=128", f"env.motion_file=synthetic:2:{rank}", "learning.params.config.minibatch_size=2048",
                   "learning.params.config.amp_minibatch_size=1024", "learning.params.config.amp_obs_demo_buffer_size=4096",
                   "learning.params.config.amp_replay_buffer_size=4096", f"+learning.params.config.hip_graph={graph}",
                   f"+learning.params.config.force_collectives={world == 1}", f"+learning.params.config.split_allreduce={split}", f"device_id={dev}", f"rl_device=cuda:{dev}"])
    task, env = parse_task(cfg, device_id=dev)
    agent = IMAmpAgent(env, cfg, dist=dist)
    agent.init_train()
    for _ in range(3):
        info = agent.train_epoch()
    flat = agent.grads.flat_param.detach()
    stats = agent.running_mean_std.running_mean.detach()
    if backend != "nccl":
        flat, stats = flat.cpu(), stats.cpu()
    gather = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gather, flat)
    sg = [torch.zeros_like(stats) for _ in range(world)]
    dist.all_gather(sg, stats)
    if rank == 0:
        q.put({"same_params": all(bool(torch.equal(gather[0], g)) for g in gather), "same_stats": all(bool(torch.allclose(sg[0], s)) for s in sg),
               "graph": agent._graph is not None, "finite": bool(torch.isfinite(flat).all()), "actor_loss": info["actor_loss"],
               "collectives": agent.num_collectives, "expected_collectives": (2 if split else 1) * 3 * agent.mini_epochs_num * agent.num_minibatches,
               "backend": dist.get_backend(), "world": dist.get_world_size()})
    dist.destroy_process_group()


if __name__ == "__main__":
    # usage: two_rank_gpu_main.py [graph] [nccl] [split] [world=N]
    graph = "graph" in sys.argv[1:]
    split = "split" in sys.argv[1:]
    backend = "nccl" if "nccl" in sys.argv[1:] else "gloo"
    world = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("world=")), 2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=worker, args=(r, world, port, graph, q, backend, split)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=500)
    out = q.get(timeout=10)
    out["exitcodes"] = [p.exitcode for p in procs]
    print(json.dumps(out))
