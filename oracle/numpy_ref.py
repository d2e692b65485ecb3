"""TEST INFRASTRUCTURE ONLY -- numpy restatements of the reference's torch-only helpers.

Each function cites the reference lines it follows; all are pinned by golden vectors
generated from the reference itself (tests/golden/make_golden.py -> tests/golden/*.npz).
"""
import numpy as np


def quat_to_rotmat(quat):
    """/root/reference/utils/geometry.py:25-45 (w,x,y,z) -> [B,3,3]."""
    q = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    R = np.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                  2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                  2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], axis=1)
    return R.reshape(-1, 3, 3)


def batch_rodrigues(theta):
    """/root/reference/utils/geometry.py:9-23: axis-angle [B,3] -> R via quaternion;
    angle = ||theta + 1e-8||, axis = theta / angle."""
    theta = np.asarray(theta)
    angle = np.linalg.norm(theta + 1e-8, axis=1, keepdims=True)
    normalized = theta / angle
    half = angle * 0.5
    quat = np.concatenate([np.cos(half), np.sin(half) * normalized], axis=1)
    return quat_to_rotmat(quat)


def rot6d_to_rotmat(x):
    """/root/reference/utils/geometry.py:47-61: view [-1,3,2]; a1 = x[:,:,0], a2 = x[:,:,1];
    Gram-Schmidt; columns (b1,b2,b3)."""
    x = np.asarray(x).reshape(-1, 3, 2)
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = a1 / np.maximum(np.linalg.norm(a1, axis=1, keepdims=True), 1e-12)
    u = a2 - (b1 * a2).sum(1, keepdims=True) * b1
    b2 = u / np.maximum(np.linalg.norm(u, axis=1, keepdims=True), 1e-12)
    b3 = np.cross(b1, b2)
    return np.stack([b1, b2, b3], axis=-1)


def perspective_projection(points, rotation, translation, focal_length, camera_center):
    """/root/reference/utils/geometry.py:63-91."""
    p = np.einsum('bij,bkj->bki', rotation, points) + translation[:, None, :]
    p = p / p[:, :, -1:]
    B = points.shape[0]
    K = np.zeros((B, 3, 3), points.dtype)
    K[:, 0, 0] = focal_length
    K[:, 1, 1] = focal_length
    K[:, 2, 2] = 1.0
    K[:, :-1, -1] = camera_center
    return np.einsum('bij,bkj->bki', K, p)[:, :, :-1]


def iuvmap_clean(U, V, Index, Ann=None):
    """/root/reference/utils/iuvmap.py:6-38: argmax -> exact one-hot, mask U,V."""
    def onehot(x):
        am = np.argmax(x, axis=1)
        return (am[:, None] == np.arange(x.shape[1])[None, :, None, None]).astype(x.dtype)
    I = onehot(Index)
    A = None if Ann is None else onehot(Ann)
    return I * U, I * V, I, A


INDEX2MASK = [[0], [1, 2], [3], [4], [5], [6], [7, 9], [8, 10], [11, 13], [12, 14], [15, 17], [16, 18],
              [19, 21], [20, 22], [23, 24]]


def iuv_img2map(img):
    """/root/reference/utils/iuvmap.py:103-147 (uv_rois=None): 3-ch IUV image ->
    U,V,Index [B,25,H,W] and Ann [B,15,H,W]; part = round(ch0*24) (half-to-even like torch)."""
    part = np.rint(img[:, 0] * np.float32(24))
    I = (part[:, None] == np.arange(25, dtype=img.dtype)[None, :, None, None]).astype(img.dtype)
    U = I * img[:, 1:2]
    V = I * img[:, 2:3]
    A = np.stack([sum(I[:, p] for p in grp) for grp in INDEX2MASK], axis=1)
    return U, V, I, A


def softmax_integral(hm):
    """/root/reference/utils/keypoints.py:372-394 (2-D branch :354-365): softmax over H*W,
    expectation of the x / y pixel index -> [B,J,2] in (x, y) order."""
    B, J, H, W = hm.shape
    f = hm.reshape(B, J, -1).astype(np.float64)
    f = np.exp(f - f.max(-1, keepdims=True))
    f = (f / f.sum(-1, keepdims=True)).reshape(B, J, H, W)
    x = (f.sum(2) * np.arange(W)).sum(-1)
    y = (f.sum(3) * np.arange(H)).sum(-1)
    return np.stack([x, y], -1)


def normalize_undigraph(A):
    """/root/reference/utils/graph.py:232-261: D^-1/2 A D^-1/2, D = column sums."""
    Dl = A.sum(0)
    d = np.where(Dl > 0, np.power(np.where(Dl > 0, Dl, 1.0), -0.5), 0.0)
    return (d[:, None] * A) * d[None, :]


def normalize_digraph(A, AD_mode=True):
    """/root/reference/utils/graph.py:176-229."""
    if AD_mode:
        Dl = A.sum(0)
        d = np.where(Dl > 0, 1.0 / np.where(Dl > 0, Dl, 1.0), 0.0)
        return A * d[None, :]
    Dl = A.sum(1)
    d = np.where(Dl > 0, 1.0 / np.where(Dl > 0, Dl, 1.0), 0.0)
    return d[:, None] * A
