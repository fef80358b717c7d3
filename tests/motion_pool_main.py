"""Child of tests/test_abi_and_host.py::test_motion_pool_* : rank `r` of a `world`-rank gloo group loads its motion-library shard twice
(load_motions + the re-sample of `resample_motions()`) through the forkserver clip pool, with the process group alive (its threads are exactly what
a `fork` pool would have copied), then takes part in an all_gather of what it drew.  Prints one JSON line per rank."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from phc_amd.config import EasyDict
    from phc_amd.env.tasks.humanoid_im import SkeletonTree
    from phc_amd.model import load_model
    from phc_amd import motion_lib as ML
    from phc_amd.utils.synthetic_motion import make_motion_dict
    m = load_model("smpl_humanoid")
    tree = SkeletonTree(m.body_names, m.parent, m.local_translation)
    clips = make_motion_dict(m.parent, 24, seed=0, body_names=m.body_names, mean_seconds=1.0, min_frames=30)
    torch.manual_seed(rank)                                           # per-rank seed, as bench.py / run.py do (run_hydra.py:121)
    cfg = EasyDict({"motion_file": clips, "device": "cpu", "min_length": -1, "im_eval": False, "step_dt": 1 / 30, "rank": rank,
                    "num_workers": 2, "pool_min_jobs": 1})
    lib = ML.MotionLibSMPL(cfg)
    out = {"rank": rank}
    draws = []
    for call in range(2):
        lib.load_motions(skeleton_trees=[tree] * 16, random_sample=True)
        pooled = lib.frames.clone()
        draws.append(lib._curr_motion_ids.clone())
        # the same draw computed in-process must give the same records: replay this call's job list without the pool
        consts = lib._clip_consts([tree])
        uniq = list(dict.fromkeys(int(u) for u in lib._curr_motion_ids.tolist()))
        recs = {u: ML._run_clip_job(lib._clip_payload(lib._motion_data_list[u], 0, -1, None), consts)[0] for u in uniq}
        nf = int(lib._motion_num_frames[0])
        first = torch.from_numpy(recs[int(lib._curr_motion_ids[0])])
        # (the pooled library additionally carries env 0's random heading on positions / rotations: compare a heading-free column, dof velocities)
        dv0 = lib.dvs[:nf].reshape(nf, -1)
        o = first.shape[1] - 0
        assert first.shape[0] == nf
        out[f"frames_{call}"] = int(pooled.shape[0])
        ref_dvs = first[:, 24 * 3 + 24 * 4 + 24 * 3 + 24 * 3 + 24 * 4: 24 * 3 + 24 * 4 + 24 * 3 + 24 * 3 + 24 * 4 + 23 * 3]
        out[f"dvs_equal_{call}"] = bool(torch.equal(dv0, ref_dvs))
    out["pool_pids"] = sorted(p.pid for k, pool in ML._POOL.items() for p in pool._pool)
    out["pool_context"] = next(iter(ML._POOL.values()))._ctx.get_start_method()
    out["resample_changed"] = bool(not torch.equal(draws[0], draws[1]))
    gathered = [torch.zeros(16, dtype=torch.long) for _ in range(world)]
    dist.all_gather(gathered, draws[1])
    out["distinct_shards"] = len({tuple(g.tolist()) for g in gathered})
    yaw = torch.tensor([float(lib.grs[0, 0, 2])])
    yaws = [torch.zeros(1) for _ in range(world)]
    dist.all_gather(yaws, yaw)
    out["distinct_headings"] = len({round(float(y), 6) for y in yaws})
    dist.barrier()
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
