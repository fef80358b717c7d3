"""TEST INFRASTRUCTURE -- golden for env.occl_training (humanoid_im.py:96-97,796-804,845-851,1081-1092,1180-1181): the reference replaces the
reference state of the occluded tracked bodies by the simulated one before the task observation (obs_v 6 / 8 / 9: all four fields; obs_v 7:
position only) and before the early-termination distance.  The substitution is applied here exactly as those lines write it, then the
reference's own jit functions run on the states of tests/golden/task_fns.npz.  Two masks: the one `_update_occl_training` always ends with
(bodies 0..8 occluded: its random draw is overwritten at :1091-1092) and a random one with the root visible.
python oracle/gen_golden_occl.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
him = ref_shim.ref_module("phc.env.tasks.humanoid_im")

g = np.load(os.path.join(ROOT, "tests", "golden", "task_fns.npz"))
t = lambda k: torch.from_numpy(g[k])
bp, br, bv, bav = t("body_pos"), t("body_rot"), t("body_vel"), t("body_ang_vel")
N, J = bp.shape[0], 24
rid = torch.from_numpy(g["reset_body_ids"].astype(np.int64))
progress = torch.from_numpy(g["progress"].astype(np.int64))
pass_time = torch.from_numpy(g["pass_time"].astype(bool))
masks = {}
TERM_DIST = {False: 0.08, True: 0.045}
m = torch.ones(N, J, dtype=torch.bool)
m[:, list(range(9, 24))] = False                      # :1091-1092
masks["fixed"] = m
gen = torch.Generator().manual_seed(11)
m = torch.rand(N, J, generator=gen) < 0.3
m[:, 0] = False                                       # :1084
masks["random"] = m
out = {}
for name, mask in masks.items():
    out[f"mask_{name}"] = mask.numpy().astype(np.uint8)
    r1 = [t(k).clone() for k in ("ref1_pos", "ref1_rot", "ref1_vel", "ref1_ang_vel")]
    for r, b in zip(r1, (bp, br, bv, bav)):           # :800-804
        r[mask] = b[mask]
    out[f"v6_{name}"] = him.compute_imitation_observations_v6(bp[:, 0], br[:, 0], bp, br, bv, bav, *r1, 1, True).numpy()
    out[f"v8_{name}"] = him.compute_imitation_observations_v8(bp[:, 0], br[:, 0], bp, br, bv, bav, *r1, 1, True).numpy()
    r7 = [t(k).clone() for k in ("ref1_pos", "ref1_vel")]
    r7[0][mask] = bp[mask]                            # :849-851 (position and the unused rotation only)
    out[f"v7_{name}"] = him.compute_imitation_observations_v7(bp[:, 0], br[:, 0], bp, bv, r7[0], r7[1], 1, True).numpy()
    body = bp[:, rid].clone()
    ref = t("ref_pos")[:, rid].clone()
    ref[mask[:, rid]] = body[mask[:, rid]]            # :1180-1181
    for use_mean in (False, True):
        td = TERM_DIST[use_mean]   # tighter than the 0.25 m of the yamls, so that the occluded bodies decide a good share of the envs
        args = (torch.zeros(N, dtype=torch.long), progress, torch.zeros(N, 24, 3), torch.zeros(4, dtype=torch.long))
        reset, term = him.compute_humanoid_im_reset(*args, body, ref, pass_time, True, torch.full((N, len(rid)), td), False, use_mean)
        _, term0 = him.compute_humanoid_im_reset(*args, body, t("ref_pos")[:, rid], pass_time, True, torch.full((N, len(rid)), td), False, use_mean)
        out[f"reset_{name}_mean{int(use_mean)}"] = reset.numpy()
        out[f"terminate_{name}_mean{int(use_mean)}"] = term.numpy()
        print(name, "mean" if use_mean else "max", "terminated", int(term.sum()), "of", int(term0.sum()), "without occlusion")
out["term_dist"] = np.array([TERM_DIST[False], TERM_DIST[True]], dtype=np.float32)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "task_occl.npz"), **out)
print("wrote task_occl.npz", {k: v.shape for k, v in out.items()})
