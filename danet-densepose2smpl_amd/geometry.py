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


def estimate_translation(S, joints_2d, focal_length=5000., img_size=224.):
    """Camera translation that brings the 3-D joints closest to the 2-D key-points -- the weighted least squares of
    /root/reference/utils/geometry.py:94-157 (GT joints 25:49 only, weights = sqrt(confidence)), batched on the
    tensors' device: the 3x3 normal equations of all samples are built with two einsums and solved in one
    torch.linalg.solve (float64, like the reference's numpy) instead of a device->host copy and a Python loop.
    S [B,49,3], joints_2d [B,49,3] (pixels, confidence) -> [B,3]."""
    S = S[:, 25:, :].to(torch.float64)
    conf = joints_2d[:, 25:, 2].to(torch.float64)
    j2d = joints_2d[:, 25:, :2].to(torch.float64)
    B, J = S.shape[0], S.shape[1]
    f, c = float(focal_length), float(img_size) / 2.
    # per joint two rows (x, y):  [f 0 c-u] t = (u-c) Z - f X ;  [0 f c-v] t = (v-c) Z - f Y
    Q = torch.zeros(B, J, 2, 3, dtype=torch.float64, device=S.device)
    Q[:, :, 0, 0] = f
    Q[:, :, 1, 1] = f
    Q[:, :, :, 2] = c - j2d
    rhs = (j2d - c) * S[:, :, 2:3] - f * S[:, :, :2]
    w = conf.view(B, J, 1, 1)                                   # (sqrt(conf))^2 on both sides of the normal equations
    A = torch.einsum('bjki,bjkl->bil', Q * w, Q)
    b = torch.einsum('bjki,bjk->bi', Q * w, rhs)
    return torch.linalg.solve(A, b).to(torch.float32)


def label_prologue(opt_joints, smpl_joints, keypoints, has_iuv, has_dp=None, smpl_2dkps=None, focal_length=5000., img_res=224):
    """The geometry of the reference's step prologue (train/trainer.py:170-210) on the device: camera translation of
    the (pseudo-)label fit from the normalised 2-D key-points, its 24 SMPL joints projected to [-1,1] with a visibility
    flag, and the weak-perspective camera (s, tx, ty) the IUV renderer takes.  Masked writes (torch.where), no host
    round trip.  -> (target_smpl_kps [B,24,3], target_cam [B,3], cam_t [B,3])"""
    B = opt_joints.shape[0]
    kp = keypoints.clone()
    kp[:, :, :-1] = 0.5 * img_res * (kp[:, :, :-1] + 1)
    cam_t = estimate_translation(opt_joints, kp, focal_length, img_res)
    centre = torch.zeros(B, 2, device=opt_joints.device) + 0.5 * img_res
    xy = perspective_projection(smpl_joints.detach(), None, cam_t, focal_length, centre) / (0.5 * img_res) - 1
    vis = (has_iuv > 0).to(xy.dtype).view(B, 1, 1).expand(B, 24, 1)
    tk = torch.cat([xy, vis], dim=-1)
    if has_dp is not None and smpl_2dkps is not None:
        tk = torch.where((has_dp > 0).view(B, 1, 1), smpl_2dkps.to(tk.dtype), tk)
    cam = torch.stack([(2. * focal_length / img_res) / cam_t[:, 2], cam_t[:, 0], cam_t[:, 1]], dim=-1)
    return tk, cam, cam_t
