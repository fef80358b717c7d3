"""TEST INFRASTRUCTURE ONLY -- known-answer vectors on the inputs of the reference's own rotation check
(poselib/poselib/core/tests/test_rotation.py:12-40: the one executable check the reference ships) and the skeleton the reference
parses from its only test fixture (poselib/poselib/skeleton/tests/ant.xml) -> tests/golden/poselib_kat.npz"""
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
    r3 = ref_shim.ref_module("poselib.poselib.core.rotation3d")
    from poselib.poselib.skeleton.skeleton3d import SkeletonTree
    q = torch.from_numpy(np.array([[0, 1, 2, 3], [-2, 3, -1, 5]], dtype=np.float32))          # test_rotation.py:12
    r = r3.quat_normalize(q)
    x = torch.from_numpy(np.array([[1, 0, 0], [0, -1, 0]], dtype=np.float32))                   # :15
    rng = np.random.default_rng(1)
    angle = torch.tensor(3.7)
    axis = torch.tensor([1.0, 4.2, 0.6])
    rot = r3.quat_from_angle_axis(angle, axis)                                                   # :25
    xs = torch.from_numpy(rng.random((5, 6, 3)))
    y = r3.quat_rotate(r3.quat_inverse(rot), r3.quat_rotate(rot, xs))                            # :27 (asserted == xs at :30)
    qa = torch.from_numpy(rng.normal(size=(16, 4)))
    qb = torch.from_numpy(rng.normal(size=(16, 4)))
    qa_n, qb_n = r3.quat_normalize(qa), r3.quat_normalize(qb)
    ang, ax = r3.quat_angle_axis(r3.quat_mul_norm(qa_n, r3.quat_inverse(qb_n)))
    tree = SkeletonTree.from_mjcf(os.path.join(ref_shim.REFERENCE_ROOT, "poselib/poselib/skeleton/tests/ant.xml"))
    np.savez_compressed(os.path.join(OUT, "poselib_kat.npz"), q=t2n(q), q_normalized=t2n(r), x=t2n(x), rotated=t2n(r3.quat_rotate(r, x)),
                        rot=t2n(rot), xs=t2n(xs), roundtrip=t2n(y), qa=t2n(qa), qb=t2n(qb), qa_n=t2n(qa_n), mul_norm=t2n(r3.quat_mul_norm(qa_n, qb_n)),
                        diff_angle=t2n(ang), diff_axis=t2n(ax), ant_names=np.array(tree.node_names), ant_parents=t2n(tree.parent_indices),
                        ant_local_translation=t2n(tree.local_translation))
    print("poselib KAT written;", len(tree.node_names), "ant nodes")


if __name__ == "__main__":
    main()
