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


def compute_metrics_lite(pred_pos_all, gt_pos_all, root_idx=0):
    m = {"mpjpe_g": [], "mpjpe_l": [], "mpjpe_pa": [], "accel_dist": [], "vel_dist": []}
    for pred, gt in zip(pred_pos_all, gt_pos_all):
        if len(pred) == 0:
            continue
        m["mpjpe_g"].append(np.linalg.norm(pred - gt, axis=-1).mean() * 1000)
        pl, gl = pred - pred[:, root_idx:root_idx + 1], gt - gt[:, root_idx:root_idx + 1]
        m["mpjpe_l"].append(np.linalg.norm(pl - gl, axis=-1).mean() * 1000)
        m["mpjpe_pa"].append(np.linalg.norm(_procrustes(pl, gl) - gl, axis=-1).mean() * 1000)
        if len(pred) > 2:
            vp, vg = np.diff(pred, axis=0), np.diff(gt, axis=0)
            m["vel_dist"].append(np.linalg.norm(vp - vg, axis=-1).mean() * 1000)
            m["accel_dist"].append(np.linalg.norm(np.diff(vp, axis=0) - np.diff(vg, axis=0), axis=-1).mean() * 1000)
    return m


def evaluate(agent, output_dir=None, log=print):
    """Run the sweep with `agent`'s current policy.  Returns (eval_info dict, failed_keys)."""
    task, env = agent.task, agent.vec_env
    agent.set_eval()
    lib_train = task._motion_lib
    saved = dict(td=task._termination_distances.clone(), test=flags.test, im_eval=flags.im_eval, start_idx=task.start_idx)
    task._termination_distances[:] = 0.5  # im_amp.py:174 (UHC's termination distance)
    flags.test, flags.im_eval = True, True
    task._motion_lib = task.get_eval_motion_lib()
    lib = task._motion_lib
    U, N = lib._num_unique_motions, task.num_envs
    terminate_memory, pred_all, gt_all = [], [], []
    try:
        task.begin_seq_motion_samples()
        with torch.no_grad():
            while True:
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
                    preds.append(info["body_pos"])
                    gts.append(info["body_pos_gt"])
                    curr += 1
                    alive = (~terminate_state).cpu().numpy()
                    curr_max = num_steps[alive].max() if alive.any() else 0
                    if curr >= curr_max or not alive.any():
                        break
                terminate_memory.append(terminate_state.cpu().numpy())
                P, G = np.stack(preds), np.stack(gts)
                for i in range(N):
                    n = max(min(int(num_steps[i]) - 1, P.shape[0]), 0)
                    pred_all.append(P[:n, i])
                    gt_all.append(G[:n, i])
                if task.start_idx + N >= U:
                    break
                task.forward_motion_samples()
        term = np.concatenate(terminate_memory)[:U]
        pred_all, gt_all = pred_all[:U], gt_all[:U]
        succ = np.flatnonzero(~term).tolist()
        keys = lib._motion_data_keys
        failed_keys, success_keys = keys[term], keys[~term]
        m_all = {k: float(np.mean(v)) if len(v) else float("nan") for k, v in compute_metrics_lite(pred_all, gt_all).items()}
        m_succ = compute_metrics_lite([pred_all[i] for i in succ], [gt_all[i] for i in succ])
        m_succ = {k: float(np.mean(v)) for k, v in m_succ.items()} if len(succ) and len(m_succ["mpjpe_g"]) else m_all
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
            f"succ {eval_info['eval/mpjpe_succ']:.1f} mm  failed {len(failed_keys)}/{U}")
    # update_training_data (im_amp.py:126-132)
    if task.auto_pmcp:
        lib_train.update_hard_sampling_weight(list(failed_keys))
    elif task.auto_pmcp_soft:
        lib_train.update_soft_sampling_weight(list(failed_keys))
    # every rank runs the sweep (the sampler weights must stay in step), rank 0 alone writes -- atomically, so that `restore()` can
    # never pick up a torn file (the reference evaluates and writes on rank 0 only, im_amp.py:136-242)
    if output_dir is not None and getattr(agent, "rank", 0) == 0:
        os.makedirs(output_dir, exist_ok=True)
        final = os.path.join(output_dir, f"failed_{agent.epoch_num:010d}.pkl")
        tmp = os.path.join(output_dir, f".tmp_failed_{agent.epoch_num:010d}.{os.getpid()}")
        joblib.dump({"failed_keys": failed_keys, "termination_history": lib_train._termination_history.cpu()}, tmp)
        os.replace(tmp, final)
    dist = getattr(agent, "dist", None)
    if output_dir is not None and dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    return eval_info, failed_keys
