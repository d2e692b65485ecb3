"""Pose evaluation metrics on the tensors' device: MPJPE and the Procrustes-aligned
reconstruction error of /root/reference/eval.py:183-216 and /root/reference/utils/pose_utils.py:10-75, batched
(one torch.linalg.svd over the batch instead of a numpy loop after a device->host copy)."""
import torch


def mpjpe(pred, gt):
    """Mean per-joint position error per sample: [B,J,3] x [B,J,3] -> [B] (eval.py:211)."""
    return torch.sqrt(((pred - gt) ** 2).sum(dim=-1)).mean(dim=-1)


def similarity_transform(S1, S2):
    """S1 [B,J,3] aligned to S2 [B,J,3] by the optimal similarity transform (scale, rotation with det = +1,
    translation) -- the orthogonal Procrustes solution of pose_utils.py:10-58."""
    S1d, S2d = S1.to(torch.float64), S2.to(torch.float64)
    mu1, mu2 = S1d.mean(dim=1, keepdim=True), S2d.mean(dim=1, keepdim=True)
    X1, X2 = S1d - mu1, S2d - mu2
    var1 = (X1 ** 2).sum(dim=(1, 2))
    K = X1.transpose(1, 2) @ X2                                  # [B,3,3] = sum_j x1_j x2_j^T
    U, s, Vh = torch.linalg.svd(K)
    V = Vh.transpose(1, 2)
    Z = torch.eye(3, dtype=torch.float64, device=S1.device).repeat(S1.shape[0], 1, 1)
    Z[:, 2, 2] = torch.sign(torch.linalg.det(U @ V.transpose(1, 2)))
    R = V @ Z @ U.transpose(1, 2)
    scale = (R @ K).diagonal(dim1=1, dim2=2).sum(dim=1) / var1
    t = mu2.transpose(1, 2) - scale.view(-1, 1, 1) * (R @ mu1.transpose(1, 2))
    out = scale.view(-1, 1, 1) * (R @ S1d.transpose(1, 2)) + t
    return out.transpose(1, 2).to(S1.dtype)


def reconstruction_error(pred, gt, reduction=None):
    """Procrustes-aligned MPJPE (pose_utils.py:67-75): per sample, or its 'mean' / 'sum'."""
    re = torch.sqrt(((similarity_transform(pred, gt) - gt) ** 2).sum(dim=-1)).mean(dim=-1)
    return re.mean() if reduction == 'mean' else (re.sum() if reduction == 'sum' else re)


def regress_joints(vertices, J_regressor):
    """H36M joints from mesh vertices (eval.py:186,203): [B,V,3] x [17,V] -> [B,17,3]."""
    return torch.matmul(J_regressor.to(vertices.dtype).to(vertices.device)[None], vertices)


def pose_errors(pred_vertices, J_regressor, joint_mapper, gt_keypoints_3d=None, gt_vertices=None):
    """The 3D pose evaluation block of eval.py:183-216 on the device: the 14 common joints regressed from the predicted
    mesh and centred on its pelvis (regressed joint 0), against either given ground-truth joints (H36M / MPI-INF:
    `gt_keypoints_3d` [B,14,3], already mapped and pelvis-centred by the dataset) or joints regressed the same way
    from `gt_vertices` (3DPW).  Returns (mpjpe [B], reconstruction_error [B], pred_joints17 [B,17,3])."""
    if (gt_keypoints_3d is None) == (gt_vertices is None):
        raise ValueError('pose_errors: give exactly one of gt_keypoints_3d / gt_vertices')
    mapper = torch.as_tensor(joint_mapper, dtype=torch.long, device=pred_vertices.device)
    j17 = regress_joints(pred_vertices, J_regressor)
    pred = j17[:, mapper, :] - j17[:, [0], :]
    if gt_vertices is not None:
        g17 = regress_joints(gt_vertices, J_regressor)
        gt = g17[:, mapper, :] - g17[:, [0], :]
    else:
        gt = gt_keypoints_3d.to(pred.dtype)
    return mpjpe(pred, gt), reconstruction_error(pred, gt, reduction=None), j17
