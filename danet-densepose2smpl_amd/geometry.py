"""Geometry helpers with the reference's names (/root/reference/utils/geometry.py)."""
import torch

from . import ops


def batch_rodrigues(theta):
    """geometry.py:9-23 (HIP kernel; labels only, no gradient)."""
    return ops.batch_rodrigues(theta)


def rot6d_to_rotmat(x):
    """geometry.py:47-61 (HIP kernel, forward + backward)."""
    return ops.rot6d_to_rotmat(x)


def perspective_projection(points, rotation, translation, focal_length, camera_center):
    """geometry.py:63-91.  rotation may be None (identity, as at every call site of the hot path)."""
    if rotation is not None:
        points = torch.einsum('bij,bkj->bki', rotation, points)
    points = points + translation.unsqueeze(1)
    proj = points[:, :, :2] / points[:, :, 2:3]
    return proj * focal_length + camera_center.unsqueeze(1)


def softmax_integral_tensor(preds, num_joints, hm_width, hm_height):
    """/root/reference/utils/keypoints.py:372-394 (2-D): softmax over the map, expected (x, y) index."""
    B = preds.shape[0]
    heat = torch.softmax(preds.reshape(B, num_joints, -1).float(), dim=2).reshape(B, num_joints, hm_height, hm_width)
    xs = torch.arange(hm_width, dtype=torch.float32, device=preds.device)
    ys = torch.arange(hm_height, dtype=torch.float32, device=preds.device)
    x = (heat.sum(dim=2) * xs).sum(dim=2, keepdim=True)
    y = (heat.sum(dim=3) * ys).sum(dim=2, keepdim=True)
    return torch.cat((x, y), dim=2)
