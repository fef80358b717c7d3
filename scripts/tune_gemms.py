"""Offline selection of the library GEMM solutions of the PPO path (PyTorch TunableOp over hipBLASLt / rocBLAS) on the GPU box:
runs eager train_epochs of the bench-sized learner with tuning on, so that every GEMM shape of the rollout and of the optimizer step is timed against
the library's candidate solutions once, and writes the winners to a TunableOp results file.  Measured in round 4 (profiles/r04_ppo/README.md): the
library's default heuristic already picks the fastest candidate for most of the 25 shapes and the update does not get faster with the file loaded, so
no loader ships -- this script stays as the probe that produced that result.

    python scripts/tune_gemms.py [--learning im] [--out gpurun_out/tuned/tuned_gemms_gfx950.csv] [--iters 10] [--ms 10]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--learning", default="im")
    ap.add_argument("--out", default="gpurun_out/tuned/tuned_gemms_gfx950.csv")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--ms", type=int, default=10)
    ap.add_argument("--num-envs", type=int, default=4096)
    a = ap.parse_args()
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    import torch.cuda.tunable as tun
    tun.enable(True)
    tun.tuning_enable(True)
    tun.set_max_tuning_iterations(a.iters)
    tun.set_max_tuning_duration(a.ms)
    tun.set_filename(os.path.abspath(a.out))
    from phc_amd.config import compose
    from phc_amd.env.tasks.vec_task import parse_task
    from phc_amd.learning.amp_agent import IMAmpAgent
    torch.manual_seed(0)
    cfg = compose([f"env.num_envs={a.num_envs}", "env.motion_file=synthetic:1:0", f"learning={a.learning}", "+learning.params.config.hip_graph=False"])
    task, env = parse_task(cfg)
    agent = IMAmpAgent(env, cfg)
    agent.init_train()
    t0 = time.perf_counter()
    for e in range(3):       # (epoch 3: the replay buffer is non-empty -- no new shapes, but the same code path as the timed epochs)
        agent.train_epoch()
        torch.cuda.synchronize()
        print(f"epoch {e}: {time.perf_counter() - t0:.1f} s, {len(tun.get_results())} tuned entries", flush=True)
    res = tun.get_results()
    for r in res:
        print(r)
    # (the runtime writes the file itself; keep a copy of what it holds in memory next to it in case it only writes at exit)
    with open(os.path.abspath(a.out) + ".results.txt", "w") as f:
        f.write("validators: %r\n" % (tun.get_validators(),))
        for r in res:
            f.write(",".join(str(x) for x in r) + "\n")


if __name__ == "__main__":
    main()
