"""P10: full-dataset evaluation sweep of the imitation policy -- `IMAmpAgent.eval` + `_post_step_eval`
(phc/learning/im_amp.py:136-363) and the auto-PMCP bookkeeping (`update_training_data`, :126-132).

Protocol (as the reference): switch the task to evaluation mode (termination distance 0.5 m on the mean body
distance, start every clip at t = 0, clips taken sequentially `num_envs` at a time from the length-sorted eval
library), act with the mean action, and per clip record whether it terminated before its last frame and the
per-frame joint positions.  Reports success rate and MPJPE metrics (mm), returns the failed keys and re-weights the
training sampler (hard / soft auto-PMCP).  `compute_metrics_lite` of the un-vendored smpl_sim package is restated for
the fields the reference logs (global / root-relative / Procrustes-aligned MPJPE, velocity and acceleration error)."""
import os

import joblib
import numpy as np
import torch

from ..utils.flags import flags


def _procrustes(pred, gt):
    """Per-frame similarity alignment of pred [T,J,3] onto gt."""
    mu_p, mu_g = pred.mean(1, keepdims=True), gt.mean(1, keepdims=True)
    p, g = pred - mu_p, gt - mu_g
    H = np.einsum("tji,tjk->tik", p, g)
    U, S, Vt = np.linalg.svd(H)
    d = np.sign(np.linalg.det(np.einsum("tij,tjk->tik", Vt.transpose(0, 2, 1), U.transpose(0, 2, 1))))
    D = np.tile(np.eye(3), (len(pred), 1, 1))
    D[:, 2, 2] = d
    R = np.einsum("tij,tjk,tkl->til", Vt.transpose(0, 2, 1), D, U.transpose(0, 2, 1))
    scale = (S * np.stack([np.ones_like(d), np.ones_like(d), d], -1)).sum(-1) / np.maximum((p ** 2).sum((1, 2)), 1e-12)
    return scale[:, None, None] * np.einsum("tij,tkj->tki", R, p) + mu_g


METRICS = ("mpjpe_g", "mpjpe_l", "mpjpe_pa", "accel_dist", "vel_dist")


def compute_metrics_per_clip(pred_pos_all, gt_pos_all, root_idx=0):
    """-> {metric: float64 [len(pred_pos_all)]}, NaN where a clip is too short for the metric (no frame; fewer than three for the differences)."""
    m = {k: np.full(len(pred_pos_all), np.nan) for k in METRICS}
    for i, (pred, gt) in enumerate(zip(pred_pos_all, gt_pos_all)):
        if len(pred) == 0:
            continue
        m["mpjpe_g"][i] = np.linalg.norm(pred - gt, axis=-1).mean() * 1000
        pl, gl = pred - pred[:, root_idx:root_idx + 1], gt - gt[:, root_idx:root_idx + 1]
        m["mpjpe_l"][i] = np.linalg.norm(pl - gl, axis=-1).mean() * 1000
        m["mpjpe_pa"][i] = np.linalg.norm(_procrustes(pl, gl) - gl, axis=-1).mean() * 1000
        if len(pred) > 2:
            vp, vg = np.diff(pred, axis=0), np.diff(gt, axis=0)
            m["vel_dist"][i] = np.linalg.norm(vp - vg, axis=-1).mean() * 1000
            m["accel_dist"][i] = np.linalg.norm(np.diff(vp, axis=0) - np.diff(vg, axis=0), axis=-1).mean() * 1000
    return m


def compute_metrics_lite(pred_pos_all, gt_pos_all, root_idx=0):
    """The list form of the un-vendored `smpl_sim.smpllib.smpl_eval.compute_metrics_lite` (one entry per clip that has the metric)."""
    return {k: v[~np.isnan(v)].tolist() for k, v in compute_metrics_per_clip(pred_pos_all, gt_pos_all, root_idx).items()}


def evaluate(agent, output_dir=None, log=print):
    """Run the sweep with `agent`'s current policy.  Returns (eval_info dict, failed_keys).

    Several ranks: the batches of `num_envs` clips are dealt round-robin -- rank r evaluates batches r, r + world, ... of the
    length-sorted library --, every rank fills its clips' entries of dense per-clip arrays (failed flag, the five metrics), ONE all-reduce(sum)
    merges them, and every rank derives the same failed keys and re-weights its sampler identically.  (The reference evaluates on one process.)"""
    task, env = agent.task, agent.vec_env
    agent.set_eval()
    lib_train = task._motion_lib
    saved = dict(td=task._termination_distances.clone(), test=flags.test, im_eval=flags.im_eval, start_idx=task.start_idx)
    task._termination_distances[:] = 0.5  # im_amp.py:174 (UHC's termination distance)
    flags.test, flags.im_eval = True, True
    task._motion_lib = task.get_eval_motion_lib()
    lib = task._motion_lib
    U, N = lib._num_unique_motions, task.num_envs
    dist = getattr(agent, "dist", None)
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    num_batches = (U + N - 1) // N
    failed = np.zeros(U, dtype=np.float64)                      # 1 where the clip terminated before its last frame
    per_clip = {k: np.zeros(U) for k in METRICS}                # metric value of every clip this rank evaluated ...
    have = {k: np.zeros(U) for k in METRICS}                    # ... and whether the clip has that metric
    try:
        with torch.no_grad():
            for bi in range(rank, num_batches, world):
                if bi == 0:
                    task.begin_seq_motion_samples()
                else:   # seek: forward_motion_samples() advances start_idx by num_envs and loads (humanoid_im.py:474-477)
                    task.start_idx = (bi - 1) * N
                    task.forward_motion_samples()
                num_steps = lib.get_motion_num_steps().cpu().numpy()
                terminate_state = torch.zeros(N, device=task.device, dtype=torch.bool)
                preds, gts = [], []
                obs = env.reset()
                curr = 0
                while True:
                    res = agent.get_action_values(obs)
                    obs, r, done, info = env.step(agent.preprocess_actions(res["mus"]))  # deterministic policy (is_determenistic=True)
                    # a termination after the clip's last frame is not a failure (im_amp.py:248)
                    term = torch.logical_and(torch.as_tensor(curr <= num_steps - 1, device=task.device), info["terminate"].bool())
                    terminate_state |= term
                    # how long this batch runs (im_amp.py:251-268): until the longest clip still alive has ended -- in the LAST batch, whose
                    # envs past the library's final clip hold wrapped-around duplicates, only the envs up to that final clip count
                    alive = (~terminate_state).cpu().numpy()
                    if alive.any():
                        last = np.flatnonzero(lib._curr_motion_ids.cpu().numpy() == U - 1)
                        if len(last):
                            bound = int(last[0]) + 1
                            curr_max = num_steps[:bound][alive[:bound]].max() if alive[:bound].any() else curr - 1
                        else:
                            curr_max = num_steps[alive].max()
                        if curr >= curr_max:
                            curr_max = curr + 1
                    else:
                        curr_max = num_steps.max()
                    preds.append(info["body_pos"])
                    gts.append(info["body_pos_gt"])
                    curr += 1
                    if curr >= curr_max or not alive.any():
                        break
                P, G = np.stack(preds), np.stack(gts)
                own = min(N, U - bi * N)                          # the envs past the library's end hold duplicates of its first clips
                clips = slice(bi * N, bi * N + own)
                failed[clips] = terminate_state.cpu().numpy()[:own]
                frames = [max(min(int(num_steps[i]) - 1, P.shape[0]), 0) for i in range(own)]
                m = compute_metrics_per_clip([P[:n, i] for i, n in enumerate(frames)], [G[:n, i] for i, n in enumerate(frames)])
                for k in METRICS:
                    per_clip[k][clips] = np.nan_to_num(m[k])
                    have[k][clips] = ~np.isnan(m[k])
        if world > 1:   # ONE collective: [failed | 5 metrics | 5 masks] x U
            flat = torch.from_numpy(np.concatenate([failed] + [per_clip[k] for k in METRICS] + [have[k] for k in METRICS])).to(task.device)
            dist.all_reduce(flat)
            parts = flat.cpu().numpy().reshape(1 + 2 * len(METRICS), U)
            failed = parts[0]
            per_clip = {k: parts[1 + i] for i, k in enumerate(METRICS)}
            have = {k: parts[1 + len(METRICS) + i] for i, k in enumerate(METRICS)}
        term = failed > 0.5
        keys = lib._motion_data_keys
        failed_keys, success_keys = keys[term], keys[~term]
        mean_of = lambda k, sel: float(per_clip[k][sel & (have[k] > 0.5)].mean()) if (sel & (have[k] > 0.5)).any() else float("nan")
        m_all = {k: mean_of(k, np.ones(U, dtype=bool)) for k in METRICS}
        m_succ = {k: mean_of(k, ~term) for k in METRICS} if ((~term) & (have["mpjpe_g"] > 0.5)).any() else m_all   # "No success!!!" (im_amp.py:330-332)
        # per-clip table of the last sweep (scripts / the player's report: which clips fail, how closely each one is tracked)
        agent.last_eval_per_clip = {"keys": [str(k) for k in keys], "failed": term.copy(), **{k: np.where(have[k] > 0.5, per_clip[k], np.nan) for k in METRICS}}
        eval_info = {"eval/success_rate": float(1 - term.mean()), "eval/mpjpe_all": m_all["mpjpe_g"], "eval/mpjpe_succ": m_succ["mpjpe_g"],
                     "eval/accel_dist": m_succ["accel_dist"], "eval/vel_dist": m_succ["vel_dist"], "eval/mpjpel_all": m_all["mpjpe_l"],
                     "eval/mpjpel_succ": m_succ["mpjpe_l"], "eval/mpjpe_pa": m_succ["mpjpe_pa"]}
    finally:
        task._termination_distances[:] = saved["td"]
        flags.test, flags.im_eval = saved["test"], saved["im_eval"]
        task._motion_lib = lib_train
        task.start_idx = saved["start_idx"]
        task.reset()  # back to training mode: reset ALL environments (im_amp.py:229)
    if log is not None:
        log(f"eval: success rate {eval_info['eval/success_rate']:.4f}  G-MPJPE all {eval_info['eval/mpjpe_all']:.1f} mm  "
            f"succ {eval_info['eval/mpjpe_succ']:.1f} mm  failed {len(failed_keys)}/{U}" + (f"  ({world} ranks, {num_batches} batches)" if world > 1 else ""))
    # update_training_data (im_amp.py:126-132): every rank holds the merged result and re-weights its own sampler identically
    if task.auto_pmcp:
        lib_train.update_hard_sampling_weight(list(failed_keys))
    elif task.auto_pmcp_soft:
        lib_train.update_soft_sampling_weight(list(failed_keys))
    # rank 0 alone writes -- atomically, so that `restore()` can never pick up a torn file
    if output_dir is not None and getattr(agent, "rank", 0) == 0:
        os.makedirs(output_dir, exist_ok=True)
        final = os.path.join(output_dir, f"failed_{agent.epoch_num:010d}.pkl")
        tmp = os.path.join(output_dir, f".tmp_failed_{agent.epoch_num:010d}.{os.getpid()}")
        joblib.dump({"failed_keys": failed_keys, "termination_history": lib_train._termination_history.cpu()}, tmp)
        os.replace(tmp, final)
    if output_dir is not None and world > 1:
        dist.barrier()
    return eval_info, failed_keys
