"""One-rank RCCL latency of the path's collective at the two gradient-bucket sizes (bench.py `collective_us`).

No multi-GPU node has ever been available to this project, so what an all-reduce of the flat gradient costs over xGMI is UNMEASURED; what can be measured on one GPU is the
floor every rank pays per optimizer step whatever the world size: RCCL's launch + its one-rank "reduction" (a copy kernel over the buffer) on the real bucket sizes --
23.2 MB (`learning=im`: 5.8 M fp32 gradients) and 149 MB (`learning=im_pnn_big`).  The multi-rank cost on top of it is transport: a ring all-reduce moves
2 (G-1)/G x bytes per rank over the slowest link (7 x ~153 GB/s xGMI links per GPU, MI355X_MICROARCH.md).

    python -m phc_amd.learning.bench_collective         -> one line  COLLECTIVE_JSON{...}
"""
import json
import os

import torch


def run(sizes_bytes=(23_200_000, 149_000_000), reps=50):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29700 + os.getpid() % 200))
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
    out = {}
    for nbytes in sizes_bytes:
        x = torch.randn(nbytes // 4, device="cuda")
        for _ in range(5):
            dist.all_reduce(x)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        torch.cuda.synchronize()
        for a, b in ev:
            a.record()
            dist.all_reduce(x)
            b.record()
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        out[f"{nbytes / 1e6:.1f}MB"] = {"bytes": int(nbytes), "median_us": t[len(t) // 2], "min_us": t[0], "max_us": t[-1]}
    dist.destroy_process_group()
    return {"backend": "nccl (RCCL)", "world_size": 1, "allreduce": out,
            "note": "ONE rank: launch + RCCL's single-rank copy path on the real bucket sizes -- the per-optimizer-step floor of the collective; transport over xGMI is unmeasured "
                    "(no multi-GPU node available), ring estimate 2 (G-1)/G x bytes / per-link bandwidth"}


if __name__ == "__main__":
    print("COLLECTIVE_JSON" + json.dumps(run()), flush=True)
