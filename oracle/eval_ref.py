"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's per-image evaluation maths.

Follows /root/reference/valid.py:146-172 (translation / angle / 2D reprojection / 3D vertex errors per ground truth)
and utils.py:31-58 (calcAngularDistance, compute_projection, compute_transformation, calc_pts_diameter).  Pinned
against tests/golden/eval_metrics.npz, produced by the reference's own functions (oracle/gen_golden.py --eval).
Only tests/ may import this module; the product path is singleshotpose_amd.utils.pose_errors_batched (HIP).
"""
import math

import numpy as np


def pose_errors_ref(vertices, R_gt, t_gt, R_pr, t_pr, K):
    """vertices (4,N) homogeneous float64; R (3,3), t (3,1) -> (pixel_dist, vertex_dist, trans_dist, angle_dist)."""
    # valid.py:148 trans_dist
    trans_dist = np.sqrt(np.sum(np.square(t_gt - t_pr)))
    # utils.py:31-35
    rot_diff = np.dot(R_gt, np.transpose(R_pr))
    with np.errstate(invalid='ignore'):
        angle_dist = np.rad2deg(np.arccos((np.trace(rot_diff) - 1.0) / 2.0))
    Rt_gt = np.concatenate((R_gt, t_gt), axis=1)
    Rt_pr = np.concatenate((R_pr, t_pr), axis=1)

    def project(Rt):            # utils.py:40-45: float32 result array
        proj = np.zeros((2, vertices.shape[1]), dtype='float32')
        cam = (K.dot(Rt)).dot(vertices)
        proj[0, :] = cam[0, :] / cam[2, :]
        proj[1, :] = cam[1, :] / cam[2, :]
        return proj
    pixel_dist = np.mean(np.linalg.norm(project(Rt_gt) - project(Rt_pr), axis=0))       # valid.py:160-165
    vertex_dist = np.mean(np.linalg.norm(Rt_gt.dot(vertices) - Rt_pr.dot(vertices), axis=0))   # valid.py:168-172
    return float(pixel_dist), float(vertex_dist), float(trans_dist), float(angle_dist)


def pts_diameter_ref(pts):
    """utils.py:50-58, row by row: max over i of max_j>=i |p_i - p_j|."""
    diameter = -1.0
    for i in range(pts.shape[0]):
        diff = pts[i][None, :] - pts[i:, :]
        d = math.sqrt((diff * diff).sum(axis=1).max())
        diameter = max(diameter, d)
    return diameter


def synthetic_eval_case(seed, n_pose=6, n_vert=700):
    """Seeded ape-sized mesh, ground-truth poses and perturbed predictions."""
    rs = np.random.RandomState(seed)
    half = np.array([0.038, 0.039, 0.046])
    pts = rs.uniform(-1, 1, (n_vert, 3)) * half
    K = np.array([[572.4114, 0.0, 325.2611], [0.0, 573.5704, 242.0489], [0.0, 0.0, 1.0]])

    def rot(axis, ang):
        axis = axis / np.linalg.norm(axis)
        Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        return np.eye(3) + math.sin(ang) * Kx + (1 - math.cos(ang)) * Kx.dot(Kx)
    R_gt, t_gt, R_pr, t_pr = [], [], [], []
    for i in range(n_pose):
        R = rot(rs.standard_normal(3), rs.uniform(0, math.pi / 3))
        t = np.array([[rs.uniform(-.1, .1)], [rs.uniform(-.1, .1)], [rs.uniform(0.6, 1.2)]])
        if i == 0:
            dR, dt = np.eye(3), np.zeros((3, 1))             # identical pose: angle argument rounds to ~1
        else:
            dR = rot(rs.standard_normal(3), rs.uniform(0, 0.2))
            dt = rs.standard_normal((3, 1)) * 0.01
        R_gt.append(R); t_gt.append(t); R_pr.append(dR.dot(R)); t_pr.append(t + dt)
    return pts, K, np.stack(R_gt), np.stack(t_gt), np.stack(R_pr), np.stack(t_pr)
