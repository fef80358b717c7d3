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
    if not ckpt and (cfg.test or cfg.get("epoch", 0) != 0):
        # run_hydra.py:312-318: epoch=-1 -> output_path/Humanoid.pth, epoch=N -> Humanoid_{N:08d}.pth
        ep = int(cfg.get("epoch", 0))
        cand = os.path.join(cfg.output_path, "Humanoid.pth" if ep <= 0 else f"Humanoid_{ep:08d}.pth")
        ckpt = cand if os.path.exists(cand) else None
    if ckpt:
        agent.restore(ckpt, load_optimizer=not cfg.test)
    if cfg.test:
        # player mode (phc/learning/im_amp_players.py:25-384): no learning; im_eval=True sweeps the whole motion set and reports the
        # success rate / MPJPE table, otherwise a deterministic-policy rollout of `games` env steps reports episode statistics
        if cfg.im_eval:
            info, failed = agent.eval(output_dir=cfg.output_path)
            if rank == 0:
                print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in info.items()}, "failed clips:", len(failed))
        else:
            info = play(agent, task, env, steps=int(cfg.get("games", 300)))
            if rank == 0:
                print(info)
        if dist is not None:
            dist.destroy_process_group()
        return info
    info = agent.train(max_epochs, output_dir=cfg.output_path)
    if rank == 0:
        out = os.path.join(cfg.output_path, "Humanoid.pth")
        os.makedirs(cfg.output_path, exist_ok=True)
        agent.save(out)
        print("saved", out)
    if dist is not None:
        dist.destroy_process_group()
    return info


def play(agent, task, env, steps=300):
    """Deterministic-policy rollout (players.PpoPlayerContinuous with is_determenistic=True): mean reward and episode length."""
    agent.set_eval()
    obs = env.reset()
    rew_sum = torch.zeros(task.num_envs, device=task.device)
    ep_len = torch.zeros(task.num_envs, device=task.device)
    lens, rews = [], []
    with torch.no_grad():
        for _ in range(steps):
            obs, r, done, info = env.step(agent.preprocess_actions(agent.get_action_values(obs)["mus"]))
            rew_sum += r
            ep_len += 1
            ids = done.nonzero(as_tuple=False).flatten()
            if len(ids):
                lens.append(ep_len[ids].clone())
                rews.append(rew_sum[ids].clone())
                rew_sum[ids] = 0
                ep_len[ids] = 0
                obs = env.reset(ids)
    lens = torch.cat(lens) if lens else ep_len
    rews = torch.cat(rews) if rews else rew_sum
    return {"episodes": int(lens.numel()), "mean_episode_length": float(lens.mean()), "mean_episode_reward": float(rews.mean()), "steps": steps}


if __name__ == "__main__":
    main()
