"""TEST INFRASTRUCTURE -- golden for P10 / f-2: the evaluation sweep's bookkeeping from the reference's OWN method bodies.

Run in the build container (needs /root/reference):  python oracle/gen_golden_eval.py   -> tests/golden/eval_sweep.npz

`IMAmpAgent._post_step_eval` (phc/learning/im_amp.py:244-363) and `update_training_data` (:126-132) are called on a `__new__`-made agent whose
`vec_env.env.task` is a scripted stand-in: U = 10 clips of different lengths, N = 4 envs (three batches, the last one wrapping around), a
seeded script of per-step `info` dicts (terminate flags incl. terminations AFTER a clip's last frame -- not failures, :248 --, mpjpe, body
positions).  The module's `compute_metrics_lite` (un-vendored smpl_sim) is replaced by a recorder that stores what it is called with, so the
fixture pins WHICH frames of WHICH clips reach the metrics, the batch boundaries (steps per batch), terminate_memory, the success rate and
the failed / successful keys; `update_training_data` runs with the reference's own `MotionLibBase.update_soft_sampling_weight /
update_hard_sampling_weight` (motion_lib_base.py:351-387) on a `__new__`-made library.  tests/test_learner_parity.py drives
`phc_amd.learning.im_eval.evaluate` with the same script."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_shim  # noqa: E402

ref_shim.install()
OUT = os.path.join(ROOT, "tests", "golden")
U, N, J = 10, 4, 5


def make_script(seed=136):
    rng = np.random.default_rng(seed)
    num_steps = np.sort(rng.integers(6, 15, U))[::-1].copy()          # the eval library is sorted by length, longest first (motion_lib_base.py:145)
    fail_at = {1: 3, 4: 2, 6: 7, 9: 1}                                 # clip -> step index at which it terminates early (a failure)
    late = {2: int(num_steps[2]), 7: int(num_steps[7]) + 1}           # terminate flag raised at / after the last frame: NOT a failure
    T = 40
    body_pos = rng.standard_normal((3, T, N, J, 3)).astype(np.float32)
    body_gt = (body_pos + 0.05 * rng.standard_normal(body_pos.shape)).astype(np.float32)
    mpjpe = np.linalg.norm(body_pos - body_gt, axis=-1).mean(-1).astype(np.float32)
    return dict(num_steps=num_steps.astype(np.int32), fail_clips=np.array(sorted(fail_at)), fail_steps=np.array([fail_at[k] for k in sorted(fail_at)]),
                late_clips=np.array(sorted(late)), late_steps=np.array([late[k] for k in sorted(late)]), body_pos=body_pos, body_gt=body_gt, mpjpe=mpjpe)


def terminate_flags(script, ids, step):
    t = np.zeros(len(ids), dtype=bool)
    for c, s in zip(script["fail_clips"], script["fail_steps"]):
        t |= (ids == c) & (step == s)
    for c, s in zip(script["late_clips"], script["late_steps"]):
        t |= (ids == c) & (step == s)
    return t


class Lib:
    def __init__(self, script):
        self._num_unique_motions = U
        self._motion_data_keys = np.array([f"clip_{i:02d}" for i in range(U)])
        self.script = script
        self.load(0)

    def load(self, start_idx):
        self._curr_motion_ids = torch.remainder(torch.arange(N) + start_idx, U)      # motion_lib_base.py:210
        self._steps = torch.from_numpy(self.script["num_steps"][self._curr_motion_ids.numpy()].astype(np.int32))

    def get_motion_num_steps(self, motion_ids=None):
        return self._steps


def main():
    im = ref_shim.ref_module("phc.learning.im_amp")
    mlb = ref_shim.ref_module("phc.utils.motion_lib_base")
    script = make_script()
    calls = []

    def recorder(pred_all, gt_all):
        calls.append(([np.asarray(p) for p in pred_all], [np.asarray(g) for g in gt_all]))
        per = [float(np.linalg.norm(p - g, axis=-1).mean()) if len(p) else 0.0 for p, g in zip(pred_all, gt_all)]
        return {k: per for k in ("mpjpe_g", "mpjpe_l", "mpjpe_pa", "accel_dist", "vel_dist")}
    im.compute_metrics_lite = recorder

    lib = Lib(script)
    task = types.SimpleNamespace(_motion_lib=lib, num_envs=N, start_idx=0)

    def forward_motion_samples():
        task.start_idx += N
        lib.load(task.start_idx)
    task.forward_motion_samples = forward_motion_samples
    agent = im.IMAmpAgent.__new__(im.IMAmpAgent)
    agent.vec_env = types.SimpleNamespace(env=types.SimpleNamespace(task=task))
    agent.device = "cpu"
    agent.terminate_state = torch.zeros(N)
    agent.terminate_memory, agent.mpjpe, agent.mpjpe_all = [], [], []
    agent.gt_pos, agent.gt_pos_all, agent.pred_pos, agent.pred_pos_all = [], [], [], []
    agent.curr_stpes, agent.success_rate = 0, 0
    agent.pbar = types.SimpleNamespace(update=lambda *a: None, refresh=lambda: None, clear=lambda: None, set_description=lambda s: None)
    steps_per_batch, batch, step = [], 0, 0
    while True:
        ids = lib._curr_motion_ids.numpy()
        info = {"terminate": torch.from_numpy(terminate_flags(script, ids, step)), "mpjpe": torch.from_numpy(script["mpjpe"][batch, step]),
                "body_pos": script["body_pos"][batch, step], "body_pos_gt": script["body_gt"][batch, step]}
        done, res = agent._post_step_eval(info, torch.zeros(N, dtype=torch.long))
        step += 1
        if res["end"]:
            steps_per_batch.append(step)
            break
        if int(done.sum()) == N:
            steps_per_batch.append(step)
            batch, step = batch + 1, 0
    pred_all, gt_all = calls[0]
    out = {"script/" + k: v for k, v in script.items()}
    out.update(U=np.array(U), N=np.array(N), steps_per_batch=np.array(steps_per_batch), terminate_memory=np.concatenate(agent.terminate_memory),
               success_rate=np.array(agent.success_rate), failed_keys=np.array(res["failed_keys"]), success_keys=np.array(res["success_keys"]),
               metric_frames_all=np.array([len(p) for p in pred_all]), metric_sum_all=np.array([float(np.sum(p, dtype=np.float64)) for p in pred_all]),
               metric_gt_sum_all=np.array([float(np.sum(g, dtype=np.float64)) for g in gt_all]),
               metric_frames_succ=np.array([len(p) for p in calls[1][0]]), eval_success_rate=np.array(res["eval_info"]["eval/success_rate"]),
               eval_mpjpe_all=np.array(res["eval_info"]["eval/mpjpe_all"]), eval_mpjpe_succ=np.array(res["eval_info"]["eval/mpjpe_succ"]))
    # update_training_data (im_amp.py:126-132) with the reference's library methods
    for mode in ("soft", "hard"):
        ml = mlb.MotionLibBase.__new__(mlb.MotionLibBase)
        ml._motion_data_keys = lib._motion_data_keys
        ml._num_unique_motions = U
        ml._device = "cpu"
        ml._termination_history = torch.zeros(U)
        ml._success_rate = torch.zeros(U)
        ml._sampling_history = torch.zeros(U)
        ml._sampling_prob = torch.ones(U) / U
        task2 = types.SimpleNamespace(_motion_lib=ml, auto_pmcp=(mode == "hard"), auto_pmcp_soft=(mode == "soft"))
        agent.vec_env = types.SimpleNamespace(env=types.SimpleNamespace(task=task2))
        agent.network_path, agent.epoch_num = "/tmp", 7
        agent.update_training_data(res["failed_keys"])
        out[f"{mode}/sampling_prob"] = ml._sampling_prob.numpy()
        out[f"{mode}/termination_history"] = ml._termination_history.numpy()
        import joblib
        dumped = joblib.load("/tmp/failed_0000000007.pkl")
        assert sorted(dumped) == ["failed_keys", "termination_history"]
    np.savez_compressed(os.path.join(OUT, "eval_sweep.npz"), **out)
    print("steps per batch", steps_per_batch, "success rate", agent.success_rate, "failed", list(res["failed_keys"]))


if __name__ == "__main__":
    main()
