"""TEST INFRASTRUCTURE ONLY -- golden for task observation version 7 (`env.obs_v=7`, the reference's keypoint models, README.MD:209-278):
`compute_imitation_observations_v7` (phc/env/tasks/humanoid_im.py:1362-1393) on the inputs already stored in tests/golden/task_fns.npz,
for all 24 bodies and for the three-point subset of env_vr.yaml.   python oracle/gen_golden_v7.py -> tests/golden/task_fns_v7.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_shim  # noqa: E402

ref_shim.install()
import torch  # noqa: E402

from gen_golden import OUT, t2n  # noqa: E402


def main():
    him = ref_shim.ref_module("phc.env.tasks.humanoid_im")
    g = np.load(os.path.join(OUT, "task_fns.npz"))
    names = list(np.load(os.path.join(OUT, "skeleton_smpl.npz"))["node_names"])
    T = lambda k: torch.from_numpy(g[k])
    bp, br, bv = T("body_pos"), T("body_rot"), T("body_vel")
    full = him.compute_imitation_observations_v7(bp[:, 0], br[:, 0], bp, bv, T("ref1_pos"), T("ref1_vel"), 1, True)
    ids = torch.tensor([names.index(b) for b in ("Head", "L_Hand", "R_Hand")])
    vr = him.compute_imitation_observations_v7(bp[:, 0], br[:, 0], bp[:, ids], bv[:, ids], T("ref1_pos")[:, ids], T("ref1_vel")[:, ids], 1, True)
    np.savez_compressed(os.path.join(OUT, "task_fns_v7.npz"), task_obs=t2n(full), task_obs_vr=t2n(vr), track_ids=t2n(ids))
    print("v7 golden:", tuple(full.shape), tuple(vr.shape))


if __name__ == "__main__":
    main()
