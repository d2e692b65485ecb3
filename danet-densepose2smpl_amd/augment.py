"""Input pipeline on the device: the crop / flip / rotation / scale augmentation of
/root/reference/datasets/base_dataset.py:115-214 and /root/reference/utils/imutils.py:11-153, batched.

The reference augments one sample at a time on the host (numpy + cv2 / scipy image ops inside the DataLoader workers).
Here a batch of decoded images that already sits in HBM is cropped, rotated, flipped and colour-jittered by ONE
resampling pass (`F.grid_sample` with the reference's own 3x3 transform), and the keypoint / pose label transforms are
batched tensor code that reproduces the reference's arithmetic -- including `transform`'s truncation to integers --
exactly (goldens g14).  Pixel values of `crop_images` are NOT bit-comparable with the reference: it crops an integer box
and resizes with scipy.misc.imresize / imrotate (PIL), neither of which exists in this image; the geometry (which source
point lands on which output pixel) is the same matrix.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import constants


def augm_params(n, is_train, noise_factor=0.4, rot_factor=30., scale_factor=0.25, rng=None):
    """base_dataset.py:115-142 for n samples: (flip [n] int, pn [n,3], rot [n] degrees, sc [n])."""
    rng = np.random.default_rng() if rng is None else rng
    flip, pn, rot, sc = np.zeros(n, np.int64), np.ones((n, 3)), np.zeros(n), np.ones(n)
    if is_train:
        flip = (rng.uniform(size=n) <= 0.5).astype(np.int64)
        pn = rng.uniform(1 - noise_factor, 1 + noise_factor, (n, 3))
        rot = np.clip(rng.standard_normal(n) * rot_factor, -2 * rot_factor, 2 * rot_factor)
        sc = np.clip(rng.standard_normal(n) * scale_factor + 1, 1 - scale_factor, 1 + scale_factor)
        rot[rng.uniform(size=n) <= 0.6] = 0
    return flip, pn, rot, sc


def _f64(x, device):
    return torch.as_tensor(x, dtype=torch.float64, device=device)


def get_transform(center, scale, res, rot=0.):
    """imutils.py:11-36, batched: center [B,2], scale [B], rot [B] degrees (or scalars) -> [B,3,3] float64 mapping
    (0-based) source pixels to the res = (rows, cols) crop."""
    dev = center.device if torch.is_tensor(center) else None
    center = _f64(center, dev).reshape(-1, 2)
    B = center.shape[0]
    scale = _f64(scale, dev).reshape(-1).expand(B)
    rot = _f64(rot, dev).reshape(-1).expand(B)
    h = 200 * scale
    t = torch.zeros(B, 3, 3, dtype=torch.float64, device=center.device)
    t[:, 0, 0] = float(res[1]) / h
    t[:, 1, 1] = float(res[0]) / h
    t[:, 0, 2] = res[1] * (-center[:, 0] / h + .5)
    t[:, 1, 2] = res[0] * (-center[:, 1] / h + .5)
    t[:, 2, 2] = 1
    rad = -rot * math.pi / 180                       # "to match direction of rotation from cropping"
    sn, cs = torch.sin(rad), torch.cos(rad)
    rm = torch.zeros_like(t)
    rm[:, 0, 0], rm[:, 0, 1], rm[:, 1, 0], rm[:, 1, 1], rm[:, 2, 2] = cs, -sn, sn, cs, 1
    tm = torch.eye(3, dtype=torch.float64, device=center.device).repeat(B, 1, 1)
    tm[:, 0, 2], tm[:, 1, 2] = -res[1] / 2, -res[0] / 2
    ti = tm.clone()
    ti[:, :2, 2] *= -1
    rotated = ti @ rm @ tm @ t
    return torch.where((rot == 0).view(B, 1, 1), t, rotated)     # (the reference skips the products when rot == 0)


def transform(pts, center, scale, res, invert=0, rot=0.):
    """imutils.py:38-46 for pts [B,N,2] (1-based pixel coordinates, as the reference passes them): the transformed
    points TRUNCATED to integers, + 1 -- returned as float64 holding integers."""
    t = get_transform(center, scale, res, rot)
    if invert:
        t = torch.linalg.inv(t)
    p = _f64(pts, t.device)
    hom = torch.cat([p - 1, torch.ones_like(p[..., :1])], dim=-1)              # [B,N,3]
    new = torch.einsum('bij,bnj->bni', t, hom)
    return torch.trunc(new[..., :2]) + 1


def flip_kp(kp):
    """imutils.py:135-143 for kp [B,N,C] with N = 24 or 49: left/right permutation, x negated."""
    perm = constants.J24_FLIP_PERM if kp.shape[1] == 24 else constants.J49_FLIP_PERM
    kp = kp[:, perm].clone()
    kp[..., 0] = -kp[..., 0]
    return kp


def flip_pose(pose):
    """imutils.py:145-153 for pose [B,72] axis-angle."""
    pose = pose[:, constants.SMPL_POSE_FLIP_PERM].clone()
    pose[:, 1::3] = -pose[:, 1::3]
    pose[:, 2::3] = -pose[:, 2::3]
    return pose


def _where_rows(cond, a, b):
    return torch.where(torch.as_tensor(cond, device=a.device).bool().view(-1, *([1] * (a.dim() - 1))), a, b)


def j2d_processing(kp, center, scale, rot, flip, res=constants.IMG_RES):
    """base_dataset.py:158-171 for kp [B,N,3] (x, y, confidence): crop transform (truncating), normalisation to
    [-1, 1], flip where flip[b]."""
    kp = _f64(kp, kp.device if torch.is_tensor(kp) else None).clone()
    kp[..., :2] = transform(kp[..., :2] + 1, center, scale, [res, res], rot=rot)
    kp[..., :-1] = 2. * kp[..., :-1] / res - 1.
    return _where_rows(flip, flip_kp(kp), kp).to(torch.float32)


def j3d_processing(S, rot, flip):
    """base_dataset.py:173-187 for S [B,N,4] (xyz, confidence): in-plane rotation, flip."""
    S = _f64(S, S.device if torch.is_tensor(S) else None).clone()
    B = S.shape[0]
    rad = -_f64(rot, S.device).reshape(-1).expand(B) * math.pi / 180
    sn, cs = torch.sin(rad), torch.cos(rad)
    rm = torch.eye(3, dtype=torch.float64, device=S.device).repeat(B, 1, 1)
    rm[:, 0, 0], rm[:, 0, 1], rm[:, 1, 0], rm[:, 1, 1] = cs, -sn, sn, cs
    S[..., :-1] = torch.einsum('bij,bkj->bki', rm, S[..., :-1])
    return _where_rows(flip, flip_kp(S), S).to(torch.float32)


def _aa_to_rotmat(aa):
    ang = aa.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    ax = aa / ang
    K = torch.zeros(aa.shape[0], 3, 3, dtype=aa.dtype, device=aa.device)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -ax[:, 2], ax[:, 1], ax[:, 2], -ax[:, 0], -ax[:, 1], ax[:, 0]
    s, c = torch.sin(ang).unsqueeze(-1), torch.cos(ang).unsqueeze(-1)
    return torch.eye(3, dtype=aa.dtype, device=aa.device) + s * K + (1 - c) * (K @ K)


def _rotmat_to_aa(R):
    """Rotation vector of R (the convention of cv2.Rodrigues: angle in [0, pi])."""
    cos = ((R.diagonal(dim1=1, dim2=2).sum(-1) - 1) / 2).clamp(-1, 1)
    ang = torch.acos(cos)
    v = torch.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], dim=-1)
    small = ang < 1e-6
    near_pi = (math.pi - ang) < 1e-4
    scale = torch.where(small, torch.full_like(ang, 0.5), ang / (2 * torch.sin(ang).clamp_min(1e-12)))
    out = v * scale.unsqueeze(-1)
    if near_pi.any():                                 # sin(angle) -> 0: axis from the symmetric part, sign from v
        S = (R + torch.eye(3, dtype=R.dtype, device=R.device)) / 2
        ax = torch.sqrt(S.diagonal(dim1=1, dim2=2).clamp_min(0))
        k = ax.argmax(dim=-1)
        col = S[torch.arange(R.shape[0]), :, k]
        ax = col / col.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        sgn = torch.where((v * ax).sum(-1, keepdim=True) < 0, -torch.ones_like(ang).unsqueeze(-1), torch.ones_like(ang).unsqueeze(-1))
        out = torch.where(near_pi.unsqueeze(-1), ax * sgn * ang.unsqueeze(-1), out)
    return out


def rot_aa(aa, rot):
    """imutils.py:115-127 for aa [B,3]: the global orientation rotated by -rot degrees about the camera axis."""
    aa = _f64(aa, aa.device if torch.is_tensor(aa) else None)
    rad = -_f64(rot, aa.device).reshape(-1).expand(aa.shape[0]) * math.pi / 180
    Rz = torch.eye(3, dtype=torch.float64, device=aa.device).repeat(aa.shape[0], 1, 1)
    Rz[:, 0, 0], Rz[:, 0, 1], Rz[:, 1, 0], Rz[:, 1, 1] = torch.cos(rad), -torch.sin(rad), torch.sin(rad), torch.cos(rad)
    return _rotmat_to_aa(Rz @ _aa_to_rotmat(aa))


def pose_processing(pose, rot, flip):
    """base_dataset.py:189-198 for pose [B,72]."""
    pose = _f64(pose, pose.device if torch.is_tensor(pose) else None).clone()
    pose[:, :3] = rot_aa(pose[:, :3], rot)
    return _where_rows(flip, flip_pose(pose), pose).to(torch.float32)


def crop_images(imgs, center, scale, rot=0., res=constants.IMG_RES):
    """The crop of imutils.py:55-89 as one resampling pass: imgs [B,C,H,W] float -> [B,C,res,res]; output pixel (v, u)
    shows source point T^-1 (u, v, 1), bilinear, zeros outside the image."""
    B, C, H, W = imgs.shape
    t = get_transform(_f64(center, imgs.device), _f64(scale, imgs.device), [res, res], _f64(rot, imgs.device))
    tinv = torch.linalg.inv(t)
    v, u = torch.meshgrid(torch.arange(res, dtype=torch.float64, device=imgs.device),
                          torch.arange(res, dtype=torch.float64, device=imgs.device), indexing='ij')
    hom = torch.stack([u, v, torch.ones_like(u)], dim=-1).view(1, -1, 3)
    src = torch.einsum('bij,bnj->bni', tinv, hom.expand(B, -1, -1))[..., :2].view(B, res, res, 2)
    grid = torch.stack([2 * src[..., 0] / max(W - 1, 1) - 1, 2 * src[..., 1] / max(H - 1, 1) - 1], dim=-1)
    return F.grid_sample(imgs, grid.to(imgs.dtype), mode='bilinear', padding_mode='zeros', align_corners=True)


def rgb_processing(imgs, center, scale, rot, flip, pn, res=constants.IMG_RES, normalize=True):
    """base_dataset.py:144-156 (+ normalize_img): imgs [B,3,H,W] in 0..255 -> [B,3,res,res]: crop, flip, per-channel
    pixel noise clamped to [0, 255], / 255, ImageNet normalisation."""
    out = crop_images(imgs, center, scale, rot, res)
    out = _where_rows(flip, out.flip(-1), out)
    out = (out * torch.as_tensor(pn, dtype=out.dtype, device=out.device).view(-1, 3, 1, 1)).clamp(0., 255.) / 255.
    if normalize:
        mean = torch.tensor(constants.IMG_NORM_MEAN, dtype=out.dtype, device=out.device).view(1, 3, 1, 1)
        std = torch.tensor(constants.IMG_NORM_STD, dtype=out.dtype, device=out.device).view(1, 3, 1, 1)
        out = (out - mean) / std
    return out


def generate_heatmap(joints, heatmap_size, sigma=1, joints_vis=None):
    """imutils.py:156-220 for a batch: joints [B,J,>=2] in [0, 1] image coordinates -> (target [B,J,H,W],
    target_weight [B,J,1]).  A joint whose 3-sigma patch lies completely outside gets weight 0 and an empty map."""
    if not hasattr(heatmap_size, '__len__'):
        heatmap_size = [heatmap_size, heatmap_size]
    Wd, Hd = int(heatmap_size[0]), int(heatmap_size[1])
    dev = joints.device
    B, J = joints.shape[:2]
    w = torch.ones(B, J, 1, dtype=torch.float32, device=dev) if joints_vis is None else joints_vis[..., :1].to(torch.float32).clone()
    tmp = sigma * 3
    mu_x = torch.trunc(joints[..., 0].to(torch.float64) * Wd + 0.5)          # int(x * size + 0.5)
    mu_y = torch.trunc(joints[..., 1].to(torch.float64) * Hd + 0.5)
    ulx, uly = torch.trunc(mu_x - tmp), torch.trunc(mu_y - tmp)
    brx, bry = torch.trunc(mu_x + tmp + 1), torch.trunc(mu_y + tmp + 1)
    outside = (ulx >= Wd) | (uly >= Hd) | (brx < 0) | (bry < 0)
    w = torch.where(outside.unsqueeze(-1), torch.zeros_like(w), w)
    xs = torch.arange(Wd, dtype=torch.float64, device=dev).view(1, 1, 1, Wd)
    ys = torch.arange(Hd, dtype=torch.float64, device=dev).view(1, 1, Hd, 1)
    # the patch is g[(y - ul_y), (x - ul_x)] with its centre at size // 2 = tmp: offsets from (ul + tmp), not from mu
    dx = xs - (ulx + tmp).view(B, J, 1, 1)
    dy = ys - (uly + tmp).view(B, J, 1, 1)
    g = torch.exp(-(dx ** 2 + dy ** 2) / (2 * sigma ** 2)).to(torch.float32)
    inside = (xs >= ulx.view(B, J, 1, 1)) & (xs < brx.view(B, J, 1, 1)) & (ys >= uly.view(B, J, 1, 1)) & (ys < bry.view(B, J, 1, 1))
    on = inside & (w > 0.5).view(B, J, 1, 1)
    return torch.where(on, g, torch.zeros_like(g)), w
