"""Entry point with the reference's command line (`python phc/run_hydra.py learning=im env=env_im ... key=val`,
phc/run_hydra.py:264-339):

    python -m phc_amd.run learning=im env=env_im env.num_envs=4096 env.motion_file=<amass.pkl | synthetic:N:seed> [max_epochs=100]
    torchrun --nproc-per-node 8 -m phc_amd.run ...          # one process per GPU, RCCL gradient all-reduce

Registers the env under the names the reference uses ('rlgpu') and runs the `im_amp` agent."""
import os
import sys

import torch


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    from .config import compose
    from .env.tasks.vec_task import RLGPUEnv, parse_task
    from .learning.amp_agent import IMAmpAgent
    from .utils.flags import flags
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    max_epochs = 10
    for a in list(argv):
        if a.startswith("max_epochs="):
            max_epochs = int(a.split("=", 1)[1])
            argv.remove(a)
    cfg = compose(argv + [f"device_id={local_rank}", f"rl_device=cuda:{local_rank}"])
    flags.test, flags.im_eval, flags.debug = bool(cfg.test), bool(cfg.im_eval), bool(cfg.debug)  # run_hydra.py:278-295
    seed = int(cfg.seed) + rank  # per-rank seed offset (run_hydra.py:121)
    torch.manual_seed(seed)
    task, env = parse_task(cfg, device_id=local_rank)
    vec_env = RLGPUEnv(env)
    agent = IMAmpAgent(env, cfg, dist=dist)
    ckpt = cfg.get("checkpoint", None)
    if ckpt:
        agent.restore(ckpt)
    info = agent.train(max_epochs)
    if rank == 0:
        out = os.path.join(cfg.output_path, "Humanoid.pth")
        os.makedirs(cfg.output_path, exist_ok=True)
        agent.save(out)
        print("saved", out)
    if dist is not None:
        dist.destroy_process_group()
    return info


if __name__ == "__main__":
    main()
