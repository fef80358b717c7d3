"""AMASS -> PHC motion pkl (SURVEY f-4; reference scripts/data_process/convert_amass_isaac.py:26-140).

Input: the reference's intermediate AMASS dict (key -> {pose_aa [T,72] SMPL axis-angles in SMPL joint order, trans [T,3],
beta / betas, gender}); output: the motion schema the motion library reads (M1: pose_quat_global, pose_quat,
root_trans_offset, trans_orig, pose_aa, beta, gender, fps).

    python -m phc_amd.utils.convert_amass --in_file amass_take.pkl --out_file amass_isaac.pkl

The reference builds the skeleton from SMPL model files (`smpl_sim.SMPL_Robot`, licensed, third party) but forces the neutral
zero-beta body (:93-94), i.e. the same tree as the shipped `smpl_humanoid.xml`; only its parent table and the pelvis offset
`local_translation[0]` enter the conversion (:100-107).  Here both come from the compiled model (phc_amd/assets).
`SMPL_BONE_ORDER_NAMES` lives in smpl_sim (version unpinned in requirement.txt); it is the published SMPL joint order under
PHC's body names, restated below.
"""
import argparse

import joblib
import numpy as np

from ..model import load_model
from ..motion_lib import _q_conj, _q_mul, _q_pos_unit

# smpl_sim.smpllib.smpl_mujoco.SMPL_BONE_ORDER_NAMES: the SMPL model's joint order
SMPL_BONE_ORDER_NAMES = ["Pelvis", "L_Hip", "R_Hip", "Torso", "L_Knee", "R_Knee", "Spine", "L_Ankle", "R_Ankle", "Chest", "L_Toe", "R_Toe",
                         "Neck", "L_Thorax", "R_Thorax", "Head", "L_Shoulder", "R_Shoulder", "L_Elbow", "R_Elbow", "L_Wrist", "R_Wrist",
                         "L_Hand", "R_Hand"]
UPRIGHT_FIX = np.array([0.5, 0.5, 0.5, 0.5])   # sRot.from_quat([0.5, 0.5, 0.5, 0.5]) (:110), xyzw


def _rotvec_to_quat(rv):
    ang = np.linalg.norm(rv, axis=-1, keepdims=True)
    small = ang < 1e-12
    k = np.where(small, 0.5, np.sin(0.5 * ang) / np.where(small, 1.0, ang))
    return np.concatenate([rv * k, np.cos(0.5 * ang)], axis=-1)


def convert_entry(entry, body_names, parents, root_offset, fps=30.0):
    """One clip (convert_amass_isaac.py:55-137, upright_start branch, `double` off)."""
    pose_aa = np.asarray(entry["pose_aa"], dtype=np.float64).copy()
    root_trans = np.asarray(entry["trans"], dtype=np.float64).copy()
    T = pose_aa.shape[0]
    beta = np.asarray(entry["beta"] if "beta" in entry else entry["betas"], dtype=np.float64).copy()
    if beta.ndim == 2:
        beta = beta[0]
    smpl_2_mujoco = [SMPL_BONE_ORDER_NAMES.index(q) for q in body_names if q in SMPL_BONE_ORDER_NAMES]   # :85
    pose_aa = np.concatenate([pose_aa[:, :66], np.zeros((T, 6))], axis=1)                                  # hands zeroed :87
    pose_aa_mj = pose_aa.reshape(-1, 24, 3)[:, smpl_2_mujoco]
    local = _rotvec_to_quat(pose_aa_mj)                                                                    # :94
    beta[:] = 0                                                                                            # neutral model :96
    root_trans_offset = root_trans + np.asarray(root_offset, dtype=np.float64)                             # :103
    # SkeletonState.from_rotation_and_root_translation(is_local=True).global_rotation (skeleton3d.py:390-408)
    glob = np.zeros_like(local)
    for j in range(24):
        glob[:, j] = local[:, j] if parents[j] < 0 else _q_pos_unit(_q_mul(glob[:, parents[j]], local[:, j]))
    pose_quat_global = _q_mul(glob, np.broadcast_to(_q_conj(UPRIGHT_FIX), glob.shape))                     # :110
    # ... is_local=False -> local_rotation (skeleton3d.py:444-462)
    pose_quat = pose_quat_global.copy()
    for j in range(24):
        if parents[j] >= 0:
            pose_quat[:, j] = _q_pos_unit(_q_mul(_q_conj(pose_quat_global[:, parents[j]]), pose_quat_global[:, j]))
    return {"pose_quat_global": pose_quat_global, "pose_quat": pose_quat, "trans_orig": root_trans, "root_trans_offset": root_trans_offset,
            "beta": beta, "gender": "neutral", "pose_aa": pose_aa, "fps": fps}


def convert(amass_data, model=None):
    model = model or load_model("smpl_humanoid")
    return {k: convert_entry(v, model.body_names, model.parent, model.local_translation[0]) for k, v in amass_data.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--in_file", type=str, default="sample_data/amass_copycat_take6_train.pkl")
    ap.add_argument("--out_file", type=str, default="sample_data/amass_isaac_converted.pkl")
    a = ap.parse_args()
    out = convert(joblib.load(a.in_file))
    joblib.dump(out, a.out_file)
    print(f"{len(out)} clips -> {a.out_file}")


if __name__ == "__main__":
    main()
