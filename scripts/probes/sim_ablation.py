"""Where does the stepper's LAUNCH time go?  Phase ablation: the instrumented library (-DPHC_SIM_PROFILE, scripts/probes/sim_phase_profile.py build)
skips one phase per run (wave-uniform branch on a device flag; results meaningless, timing only) and the launch is timed by HIP events.  The
marginal time of a phase = full launch - launch without it.  (The s_memtime phase timer needs an s_waitcnt(0) at every phase boundary, which
distorts a latency-bound kernel; bit 15 switches it off here.)

    python scripts/probes/sim_ablation.py [num_envs] [config overrides]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PROF = os.path.join(ROOT, "phc_amd", "_obj", "libphc_amd_prof.so")
os.environ["PHC_AMD_LIB"] = PROF
import torch  # noqa: E402
from phc_amd import _lib as L  # noqa: E402
from phc_amd.config import compose  # noqa: E402
from phc_amd.env.tasks.humanoid_im import _stream  # noqa: E402
from phc_amd.env.tasks.vec_task import parse_task  # noqa: E402

PHASES = ["body-body contact", "velocity products + per-body init", "drive exchange", "backward sweep", "acceleration sweep", "joint integration",
          "kinematics (pointer jumping)", "epilogue (store + publish)", "initial kinematics sweep"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    torch.manual_seed(0)
    task, env = parse_task(compose([f"env.num_envs={n}", "env.motion_file=synthetic:1:0"] + sys.argv[2:]))
    raw = C.CDLL(PROF)
    env.reset()
    a = (torch.rand(n, task.num_actions, device=task.device) * 2 - 1) * 0.1
    for _ in range(10):   # the state of the bench protocol: a few steps after the resets
        task.reset_done(); env.step(a)
    root0, dof0 = task._root_states.clone(), task._dof_state.clone()

    def timed(mask, calls=None):
        raw.phc_debug_set_skip(mask | (1 << 15))
        ts = []
        for it in range(40):
            task._root_states.copy_(root0); task._dof_state.copy_(dof0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.check(task._lib.phc_sim_step(task._model_struct, task._sim_params, task._sim_struct, a.data_ptr(), task._pd_action_offset.data_ptr(),
                                           task._pd_action_scale.data_ptr(), task._freeze_mask.data_ptr(), task.control_freq_inv if calls is None else calls,
                                           _stream()), "phc_sim_step")
            e1.record()
            torch.cuda.synchronize()
            if it >= 8:
                ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        return ts[len(ts) // 2]
    full = timed(0)
    print(f"{n} envs: full launch {full:.1f} us (median of 32, HIP events); 0 sub-steps {timed(0, 0):.1f} us")
    for b, name in enumerate(PHASES):
        t = timed(1 << b)
        print(f"  without {name:36s} {t:7.1f} us   -> marginal {full - t:6.1f} us")
    t = timed((1 << 3) | (1 << 4))
    print(f"  without both sweeps {'':28s} {t:7.1f} us   -> marginal {full - t:6.1f} us")
    t = timed(0x17f)
    print(f"  without every phase {'':28s} {t:7.1f} us")
    raw.phc_debug_set_skip(0)


if __name__ == "__main__":
    main()
