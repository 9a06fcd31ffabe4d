"""Mirror of the pure-numpy metric code of the reference's eval.py:
geodesic_distance_rotmats_pairwise_np (:47-59), euclidean_distance_points_pairwise_np (:62-77),
euclidean_distance_joint_conf_pairwise_np (:80-95), magd_for_grasp_distribution (:98-137) — same names,
arguments and return values, with the python per-row loops replaced by vectorised numpy (identical results:
the loops only fill a matrix row by row)."""
import numpy as np
import torch


def geodesic_distance_rotmats_pairwise_np(r1s, r2s):
    """[N,3,3] x [M,3,3] -> [N,M] angular distances (eval.py:47-59)."""
    rot_rot_transpose = np.einsum('aij,bkj->abik', r1s, r2s, optimize=True)
    tr = np.trace(rot_rot_transpose, axis1=-2, axis2=-1)
    return np.arccos(np.clip((tr - 1.0) / 2.0, -1.0, 1.0))


def _pairwise_sq(a, b):
    # eval.py:62-95 compute `deltas = b - a[idx]` in the inputs' dtype and store einsum('ij,ij->i') into a
    # float64 matrix; the broadcast form below evaluates the same expression for all idx at once
    deltas = b[None, :, :] - a[:, None, :]
    return np.einsum('nij,nij->ni', deltas, deltas).astype(np.float64)


def euclidean_distance_points_pairwise_np(pt1, pt2):
    return _pairwise_sq(pt1, pt2)


def euclidean_distance_joint_conf_pairwise_np(joint1, joint2):
    return _pairwise_sq(joint1, joint2)


def magd_for_grasp_distribution(grasp1, grasp2):
    """Mean Absolute Grasp Deviation (eval.py:98-137): for every predicted grasp the ground-truth grasp nearest
    in squared translation distance; returns (sum transl dist, sum geodesic rot dist, sum joint dist, coverage)."""
    if torch.is_tensor(grasp1['rot_matrix']):
        grasp1['rot_matrix'] = grasp1['rot_matrix'].cpu().data.numpy()
        grasp1['transl'] = grasp1['transl'].cpu().data.numpy()
        grasp1['joint_conf'] = grasp1['joint_conf'].cpu().data.numpy()
    transl_dist_mat = euclidean_distance_points_pairwise_np(grasp1['transl'], grasp2['transl'])
    rot_dist_mat = geodesic_distance_rotmats_pairwise_np(grasp1['rot_matrix'], grasp2['rot_matrix'])
    joint_dist_mat = euclidean_distance_joint_conf_pairwise_np(grasp1['joint_conf'], grasp2['joint_conf'])
    transl_loss = np.min(transl_dist_mat, axis=1)
    cor = np.argmin(transl_dist_mat, axis=1)
    rows = np.arange(transl_loss.shape[0])
    rot_loss = rot_dist_mat[rows, cor].astype(transl_loss.dtype)
    joint_loss = joint_dist_mat[rows, cor].astype(transl_loss.dtype)
    coverage = len(set(cor.tolist())) / len(grasp2['transl'])
    return np.sum(transl_loss), np.sum(rot_loss), np.sum(joint_loss), coverage
