"""Stepper launch time vs number of sub-steps: t = a + b * nsub separates the fixed part (state load, initial FK, store / publish, launch)
from the per-sub-step cost.   python scripts/probes/sim_substep_scan.py [num_envs] [lane_mapping]"""
import os
import sys

import torch

sys.path.insert(0, ".")
from phc_amd import _lib as L  # noqa: E402
from phc_amd.config import compose  # noqa: E402
from phc_amd.env.tasks.humanoid_im import _stream  # noqa: E402
from phc_amd.env.tasks.vec_task import parse_task  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    mapping = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    torch.manual_seed(0)
    task, env = parse_task(compose([f"env.num_envs={n}", "env.motion_file=synthetic:1:0", f"+solver.lane_mapping={mapping}"] + sys.argv[3:]))
    env.reset()
    a = (torch.rand(n, task.num_actions, device=task.device) * 2 - 1) * 0.1
    lift = float(os.environ.get("PHC_SCAN_LIFT", "0"))     # raise every humanoid: no body reaches the ground -> the contact-point loops are skipped
    if lift:
        task._root_states[:, 2] += lift
    root0, dof0 = task._root_states.clone(), task._dof_state.clone()
    for calls in (0, 2, 4):
        ts = []
        for it in range(30):
            task._root_states.copy_(root0); task._dof_state.copy_(dof0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(task._lib.phc_sim_step(task._model_struct, task._sim_params, task._sim_struct, a.data_ptr(), task._pd_action_offset.data_ptr(),
                                           task._pd_action_scale.data_ptr(), task._freeze_mask.data_ptr(), calls, _stream()), "phc_sim_step")
            e1.record()
            torch.cuda.synchronize()
            if it >= 5:
                ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        print(f"num_sim_calls {calls} (sub-steps {calls * 2}): median {ts[len(ts) // 2]:7.1f} us   min {ts[0]:7.1f} us")


if __name__ == "__main__":
    main()
