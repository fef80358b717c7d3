"""TEST INFRASTRUCTURE -- golden for S9: the explicit `pd` torque of the robot control mode, from the reference's OWN code.

Run in the build container (needs /root/reference):  python oracle/gen_golden_torques.py   -> tests/golden/pd_torques.npz

  * gains / default pose / hard-coded torque limits: the `if self.humanoid_type in ['h1'] ... elif ... ['g1']` statement of the reference's
    `Humanoid._build_env` (phc/env/tasks/humanoid.py:1112-1181) is cut out of the method's AST and executed on a stand-in `self`
    (the method itself needs the simulator);
  * torque limits: `Humanoid._process_dof_props` (:989-1028) on a `__new__`-made task with the asset's DoF properties;
  * torques: `Humanoid._compute_torques` (:1575-1599) on that task with seeded joint states and actions (some far outside [-1, 1]: the clip).
The stepper's `control_mode` 1 must hold exactly this torque over a simulate call (tests/test_h1.py)."""
import ast
import inspect
import os
import sys
import textwrap
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


def gains_block(hum_mod):
    """The `if humanoid_type in ['h1'] ... elif in ['g1']` statement that assigns p_gains, as a code object."""
    for name, fn in inspect.getmembers(hum_mod.Humanoid, inspect.isfunction):
        src = textwrap.dedent(inspect.getsource(fn))
        if "self.p_gains = to_torch" not in src:
            continue
        tree = ast.parse(src)
        for node in ast.walk(tree):
            if isinstance(node, ast.If) and "humanoid_type" in ast.unparse(node.test) and "p_gains" in ast.unparse(node) and "'h1'" in ast.unparse(node.test):
                return compile(ast.Module(body=[node], type_ignores=[]), f"<{name}: gains>", "exec"), name
    raise RuntimeError("gains statement not found")


def main():
    hum = ref_shim.ref_module("phc.env.tasks.humanoid")
    from phc_amd.model import load_model
    code, where = gains_block(hum)
    to_torch = lambda x, device=None, **kw: torch.as_tensor(np.asarray(x, dtype=np.float32) if not torch.is_tensor(x) else x, dtype=torch.float32)
    out = {"source_method": np.array(where)}
    rng = np.random.default_rng(1575)
    for htype, pd_v, asset in (("h1", 1, "h1_humanoid"), ("h1", 2, "h1_humanoid"), ("g1", 1, "g1_humanoid")):
        m = load_model(asset)
        nd = m.num_dof
        stand = types.SimpleNamespace(humanoid_type=htype, device="cpu", num_dof=nd, cfg=types.SimpleNamespace(env={"pd_v": pd_v}))
        exec(code, {"to_torch": to_torch, "torch": torch, "np": np}, {"self": stand})
        task = hum.Humanoid.__new__(hum.Humanoid)
        task.num_dof, task.device = nd, "cpu"
        if hasattr(stand, "torque_limits_hard_coded"):
            task.torque_limits_hard_coded = stand.torque_limits_hard_coded
        lo, hi = m.dof_limits()
        props = np.zeros(nd, dtype=[("lower", "f4"), ("upper", "f4"), ("velocity", "f4"), ("effort", "f4")])   # (gym's DoF properties: a structured array)
        props["lower"], props["upper"], props["velocity"], props["effort"] = lo, hi, 100.0, m.dof_effort
        task._process_dof_props(props)
        task.p_gains, task.d_gains, task.default_dof_pos = stand.p_gains, stand.d_gains, stand.default_dof_pos
        task.cfg = types.SimpleNamespace(control=types.SimpleNamespace(action_scale=1.0))
        n = 6
        task._dof_pos = torch.from_numpy((rng.standard_normal((n, nd)) * 0.4).astype(np.float32)) + task.default_dof_pos
        task._dof_vel = torch.from_numpy((rng.standard_normal((n, nd)) * 3.0).astype(np.float32))
        actions = torch.from_numpy((rng.standard_normal((n, nd)) * np.array([0.3, 0.3, 1.5, 1.5, 6.0, 6.0])[:, None]).astype(np.float32))
        torques = task._compute_torques(actions)
        tag = f"{htype}_pdv{pd_v}/"
        out.update({tag + "p_gains": stand.p_gains.numpy(), tag + "d_gains": stand.d_gains.numpy(), tag + "default_dof_pos": stand.default_dof_pos.numpy(),
                    tag + "torque_limits": task.torque_limits.numpy(), tag + "dof_pos": task._dof_pos.numpy(), tag + "dof_vel": task._dof_vel.numpy(),
                    tag + "actions": actions.numpy(), tag + "torques": torques.numpy()})
        print(tag, "saturated entries:", int((torques.abs() >= task.torque_limits - 1e-6).sum()), "of", torques.numel())
    np.savez_compressed(os.path.join(OUT, "pd_torques.npz"), **out)
    print("wrote pd_torques.npz; gains taken from Humanoid." + where)


if __name__ == "__main__":
    main()
