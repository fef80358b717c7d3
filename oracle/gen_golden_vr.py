"""TEST INFRASTRUCTURE ONLY -- golden for env_vr.yaml (three-point tracking: trackBodies = reset_bodies = Head, L_Hand, R_Hand):
the reference's task-obs v6 and reset functions on the body SUBSETS the task hands them (humanoid_im.py:761-772,1163-1171),
from the inputs already stored in tests/golden/task_fns.npz.   python oracle/gen_golden_vr.py -> tests/golden/task_fns_vr.npz"""
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

VR_BODIES = ["Head", "L_Hand", "R_Hand"]


def main():
    him = ref_shim.ref_module("phc.env.tasks.humanoid_im")
    g = np.load(os.path.join(OUT, "task_fns.npz"))
    names = list(np.load(os.path.join(OUT, "skeleton_smpl.npz"))["node_names"])
    ids = torch.tensor([names.index(b) for b in VR_BODIES])
    T = lambda k: torch.from_numpy(g[k])
    bp, br, bv, bw = T("body_pos"), T("body_rot"), T("body_vel"), T("body_ang_vel")
    task_obs = him.compute_imitation_observations_v6(bp[:, 0], br[:, 0], bp[:, ids], br[:, ids], bv[:, ids], bw[:, ids], T("ref1_pos")[:, ids],
                                                     T("ref1_rot")[:, ids], T("ref1_vel")[:, ids], T("ref1_ang_vel")[:, ids], 1, True)
    E = bp.shape[0]
    td = torch.full((E, 24), 0.25)
    reset, term = him.compute_humanoid_im_reset(torch.zeros(E, dtype=torch.long), T("progress"), torch.zeros(E, 24, 3), torch.zeros(4, dtype=torch.long),
                                                bp[:, ids], T("ref_pos")[:, ids], T("pass_time"), True, td[:, ids], False, False)
    np.savez_compressed(os.path.join(OUT, "task_fns_vr.npz"), track_ids=t2n(ids), task_obs=t2n(task_obs), reset=t2n(reset), terminate=t2n(term))
    print("vr golden:", tuple(task_obs.shape), int(term.sum()), "terminated of", E)


if __name__ == "__main__":
    main()
