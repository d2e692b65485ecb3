"""Model assets for the SMPL layer and the IUV renderer.

The reference loads two licence-gated files that are absent from its repo:
``data/smpl/SMPL_*.pkl`` (via smplx; /root/reference/models/smpl.py:18-25,
path_config.py:65,71) and ``data/UV_data/UV_Processed.mat``
(/root/reference/utils/densepose_methods.py:16-23).  This module provides

* seeded *synthetic* stand-ins with exactly the public tensor shapes / dtypes
  (6890 vertices, 13776 faces, 24 joints, 10 betas, 207 pose-basis rows,
  7829 DensePose vertices, 13774 DensePose faces, part ids 1..24), used by the
  tests and by ``bench.py`` (there is no network for the real files), and
* loaders for the real files when a user supplies them (``load_smpl_npz``,
  ``load_densepose_mat``).

Only numpy is used here; tensors are created by the callers.
"""
import os
import numpy as np

NUM_VERTS = 6890
NUM_FACES = 13776
NUM_JOINTS = 24
NUM_BETAS = 10
NUM_POSE_BASIS = 207
NUM_DP_VERTS = 7829
NUM_DP_FACES = 13774

# Kinematic tree (parents[0] = -1): same table as row 0 of
# /root/reference/utils/smpl_utlis.py:13 (whose root entry is 0).
SMPL_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14,
                         16, 17, 18, 19, 20, 21], dtype=np.int32)

# 21 landmark vertices appended after the 24 posed joints by smplx's
# VertexJointSelector (nose, eyes, ears, 6 feet, 10 finger tips) -- public
# smplx vertex ids, order as in SURVEY.md Appendix C.1.
SMPL_LANDMARK_VERTS = np.array([
    332, 6260, 2800, 4071, 583,            # nose, reye, leye, rear, lear
    3216, 3226, 3387, 6617, 6624, 6787,    # LBigToe LSmallToe LHeel RBigToe RSmallToe RHeel
    2746, 2319, 2445, 2556, 2673,          # left thumb..pinky tips
    6191, 5782, 5905, 6016, 6133,          # right thumb..pinky tips
], dtype=np.int32)

# Approximate rest-pose joint locations (metres, x right, y up before the flip)
_REST_JOINTS = np.array([
    [0.00, -0.24, 0.03], [0.06, -0.33, 0.02], [-0.06, -0.33, 0.02], [0.00, -0.12, 0.00],
    [0.10, -0.71, 0.02], [-0.10, -0.71, 0.02], [0.00, 0.02, 0.02], [0.09, -1.10, -0.02],
    [-0.09, -1.10, -0.02], [0.00, 0.07, 0.04], [0.11, -1.16, 0.08], [-0.11, -1.16, 0.08],
    [0.00, 0.28, 0.00], [0.08, 0.19, 0.00], [-0.08, 0.19, 0.00], [0.00, 0.37, 0.04],
    [0.18, 0.22, -0.01], [-0.18, 0.22, -0.01], [0.43, 0.21, -0.03], [-0.43, 0.21, -0.03],
    [0.68, 0.22, -0.03], [-0.68, 0.22, -0.03], [0.76, 0.21, -0.04], [-0.76, 0.21, -0.04],
], dtype=np.float64)


def _sphere_topology(rings=84, segs=82):
    """Closed genus-0 triangulation with 2 + rings*segs = 6890 vertices and
    2*rings*segs = 13776 outward-CCW faces (same counts as SMPL)."""
    assert 2 + rings * segs == NUM_VERTS
    faces = []

    def vid(r, s):
        return 1 + r * segs + (s % segs)
    top, bot = 0, 1 + rings * segs
    for s in range(segs):
        faces.append((top, vid(0, s + 1), vid(0, s)))
    for r in range(rings - 1):
        for s in range(segs):
            a, b, c, d = vid(r, s), vid(r, s + 1), vid(r + 1, s), vid(r + 1, s + 1)
            faces.append((a, b, c))
            faces.append((b, d, c))
    for s in range(segs):
        faces.append((bot, vid(rings - 1, s), vid(rings - 1, s + 1)))
    faces = np.asarray(faces, dtype=np.int32)
    assert faces.shape == (NUM_FACES, 3)
    theta = (np.arange(rings) + 1.0) * np.pi / (rings + 1.0)
    phi = np.arange(segs) * 2.0 * np.pi / segs
    pts = [(0.0, 1.0, 0.0)]
    for r in range(rings):
        for s in range(segs):
            pts.append((np.sin(theta[r]) * np.cos(phi[s]), np.cos(theta[r]), np.sin(theta[r]) * np.sin(phi[s])))
    pts.append((0.0, -1.0, 0.0))
    return np.asarray(pts, dtype=np.float64), faces


def make_synthetic_smpl(seed=0):
    """Seeded synthetic SMPL-like model with the public SMPL tensor shapes.

    Returns a dict of numpy arrays: v_template[6890,3] f32, faces[13776,3] i32,
    shapedirs[6890,3,10] f32, posedirs[207,20670] f32, J_regressor[24,6890] f32
    (row-stochastic), lbs_weights[6890,24] f32 (<=4 nnz, rows sum to 1),
    parents[24] i32, J_regressor_extra[9,6890] f32, landmark_verts[21] i32.
    The body is expressed in the SPIN camera convention (y down), like the real
    model as used by the reference.
    """
    rng = np.random.default_rng(seed)
    unit, faces = _sphere_topology()
    # body-like blob: tall ellipsoid with arms-width bulge, y range approx [-1.25, 0.5]
    radii = np.array([0.30, 0.875, 0.17])
    centre = np.array([0.0, -0.375, 0.0])
    v = unit * radii + centre
    # widen around shoulder height so the arm joints are inside the surface
    bulge = np.exp(-((v[:, 1] - 0.21) / 0.10) ** 2)
    v[:, 0] *= 1.0 + 1.7 * bulge
    v += rng.normal(0.0, 0.002, v.shape)
    joints = _REST_JOINTS.copy()
    # SPIN convention: y (and z) flipped so that +y points down in the image.
    flip = np.array([1.0, -1.0, -1.0])
    v = v * flip
    joints = joints * flip
    # mirrored coordinates invert orientation: swap two indices to stay outward-CCW
    # in the right-handed (x right, y down, z forward) frame.  (Two axis flips keep
    # orientation, so no swap is needed -- keep the faces as built.)

    # shape blend-shapes: low-frequency displacement fields + small noise
    shapedirs = np.zeros((NUM_VERTS, 3, NUM_BETAS))
    for l in range(NUM_BETAS):
        M = rng.normal(0.0, 0.02, (3, 3))
        ph = rng.uniform(0, 2 * np.pi, 3)
        fr = rng.uniform(1.0, 4.0, 3)
        shapedirs[:, :, l] = v @ M.T + 0.01 * np.sin(v * fr + ph) + rng.normal(0, 0.001, v.shape)
    # pose blend-shapes
    posedirs = rng.normal(0.0, 0.003, (NUM_POSE_BASIS, NUM_VERTS * 3))
    # joint regressor: row-stochastic, supported on the vertices nearest each joint
    d2 = ((v[None, :, :] - joints[:, None, :]) ** 2).sum(-1)          # [24, V]
    J_regressor = np.zeros((NUM_JOINTS, NUM_VERTS))
    for j in range(NUM_JOINTS):
        idx = np.argsort(d2[j])[:48]
        w = rng.uniform(0.2, 1.0, idx.shape[0])
        J_regressor[j, idx] = w / w.sum()
    # skinning weights: inverse-square distance to the 4 nearest joints
    lbs_weights = np.zeros((NUM_VERTS, NUM_JOINTS))
    near = np.argsort(d2.T, axis=1)[:, :4]
    for k in range(4):
        lbs_weights[np.arange(NUM_VERTS), near[:, k]] = 1.0 / (d2.T[np.arange(NUM_VERTS), near[:, k]] + 1e-3)
    lbs_weights /= lbs_weights.sum(1, keepdims=True)
    # 9 extra regressed joints (J_regressor_extra of the reference, models/smpl.py:21)
    J_extra = np.zeros((9, NUM_VERTS))
    for e in range(9):
        c = v[rng.integers(0, NUM_VERTS)]
        idx = np.argsort(((v - c) ** 2).sum(-1))[:32]
        w = rng.uniform(0.2, 1.0, idx.shape[0])
        J_extra[e, idx] = w / w.sum()
    return {
        'v_template': v.astype(np.float32),
        'faces': faces,
        'shapedirs': shapedirs.astype(np.float32),
        'posedirs': posedirs.astype(np.float32),
        'J_regressor': J_regressor.astype(np.float32),
        'lbs_weights': lbs_weights.astype(np.float32),
        'parents': SMPL_PARENTS.copy(),
        'J_regressor_extra': J_extra.astype(np.float32),
        'landmark_verts': SMPL_LANDMARK_VERTS.copy(),
    }


def make_synthetic_densepose(smpl=None, seed=0):
    """Seeded DensePose-like topology with the public ``UV_Processed.mat`` field
    names, shapes and index bases (/root/reference/utils/densepose_methods.py:18-23):
    All_vertices[7829] (1-based SMPL ids), All_Faces[13774,3] (1-based),
    All_FaceIndices[13774] in 1..24, All_U_norm / All_V_norm[7829] in [0,1]."""
    if smpl is None:
        smpl = make_synthetic_smpl(seed)
    rng = np.random.default_rng(seed + 1)
    v = smpl['v_template'].astype(np.float64)
    faces = smpl['faces'].astype(np.int64)
    joints = smpl['J_regressor'].astype(np.float64) @ v
    cent = v[faces].mean(1)
    part = np.argmin(((cent[:, None, :] - joints[None, :, :]) ** 2).sum(-1), axis=1) + 1   # 1..24
    # drop the two smallest-area faces to reach 13774
    e1, e2 = v[faces[:, 1]] - v[faces[:, 0]], v[faces[:, 2]] - v[faces[:, 0]]
    area = np.linalg.norm(np.cross(e1, e2), axis=1)
    keep = np.ones(len(faces), bool)
    keep[np.argsort(area)[:NUM_FACES - NUM_DP_FACES]] = False
    faces, part = faces[keep], part[keep]
    # seam vertices: touched by faces of more than one part -> duplicated
    vmin = np.full(NUM_VERTS, 99)
    vmax = np.zeros(NUM_VERTS, int)
    for k in range(3):
        np.minimum.at(vmin, faces[:, k], part)
        np.maximum.at(vmax, faces[:, k], part)
    seam = np.nonzero(vmax > vmin)[0]
    n_dup = NUM_DP_VERTS - NUM_VERTS
    assert len(seam) >= n_dup, len(seam)
    seam = rng.permutation(seam)[:n_dup]
    dup_of = -np.ones(NUM_VERTS, int)
    dup_of[seam] = NUM_VERTS + np.arange(n_dup)
    all_vertices = np.concatenate([np.arange(NUM_VERTS), seam]) + 1               # 1-based
    dp_faces = faces.copy()
    for k in range(3):
        use_dup = (dup_of[faces[:, k]] >= 0) & (part > vmin[faces[:, k]])
        dp_faces[use_dup, k] = dup_of[faces[use_dup, k]]
    # per-vertex U,V: normalised position inside the bounding box of the vertex's part
    vpart = np.zeros(NUM_DP_VERTS, int)
    for k in range(3):
        vpart[dp_faces[:, k]] = part
    pos = v[all_vertices - 1]
    U = np.zeros(NUM_DP_VERTS)
    V = np.zeros(NUM_DP_VERTS)
    for p in range(1, 25):
        m = vpart == p
        if not m.any():
            continue
        lo, hi = pos[m].min(0), pos[m].max(0)
        span = np.maximum(hi - lo, 1e-6)
        U[m] = (pos[m, 0] - lo[0]) / span[0]
        V[m] = (pos[m, 1] - lo[1]) / span[1]
    return {
        'All_vertices': all_vertices.astype(np.uint32),
        'All_Faces': (dp_faces + 1).astype(np.uint32),
        'All_FaceIndices': part.astype(np.uint8),
        'All_U_norm': np.clip(U, 0, 1).astype(np.float64),
        'All_V_norm': np.clip(V, 0, 1).astype(np.float64),
    }


def densepose_render_tables(dp):
    """Constant tables the IUV renderer builds from the DensePose topology, exactly as
    /root/reference/utils/renderer.py:236-249 does: vert_mapping = All_vertices-1,
    faces = All_Faces-1, per-face texture (FaceIndex/24, mean U, mean V)."""
    vert_mapping = dp['All_vertices'].astype(np.int64).reshape(-1) - 1
    faces = dp['All_Faces'].astype(np.int64) - 1
    fi = dp['All_FaceIndices'].reshape(-1).astype(np.float64)
    num_part = float(fi.max())
    U = dp['All_U_norm'].reshape(-1).astype(np.float64)
    V = dp['All_V_norm'].reshape(-1).astype(np.float64)
    tex = np.stack([fi / num_part, U[faces].mean(1), V[faces].mean(1)], axis=1)
    return vert_mapping.astype(np.int32), faces.astype(np.int32), tex.astype(np.float32)


def load_smpl_npz(path):
    """Load a real SMPL model converted to .npz with the key names above
    (a one-off conversion of SMPL_NEUTRAL.pkl; the .pkl itself needs chumpy)."""
    d = np.load(path, allow_pickle=False)
    out = {k: d[k] for k in d.files}
    out.setdefault('parents', SMPL_PARENTS.copy())
    out.setdefault('landmark_verts', SMPL_LANDMARK_VERTS.copy())
    return out


class _PickledArray(object):
    """Stand-in for a chumpy array inside an SMPL .pkl (chumpy is not installed here and not needed): keeps the pickled state,
    whose 'x' entry is the ndarray of a leaf `chumpy.ch.Ch` (what the official SMPL files store for v_template, shapedirs,
    posedirs, weights and J)."""
    def __init__(self, *args, **kwargs):
        self.state = {}

    def __setstate__(self, state):
        self.state = state if isinstance(state, dict) else {'x': state}

    def __array__(self, dtype=None, copy=None):
        for k in ('x', 'r', 'a'):
            if k in self.state:
                return np.asarray(self.state[k], dtype=dtype)
        raise ValueError('SMPL .pkl: a chumpy object without a stored array (keys %s)' % sorted(self.state))


def load_smpl_pkl(path, extra_regressor=None, num_betas=NUM_BETAS):
    """The official SMPL model file (SMPL_NEUTRAL.pkl, what `smplx.SMPL(model_path)` reads in
    /root/reference/models/smpl.py:15-19 with path_config.SMPL_MODEL_DIR) -> the dict of arrays smpl.SMPL takes, following
    smplx.body_models.SMPL.__init__: latin-1 unpickling, shapedirs cut to `num_betas`, posedirs [V, 3, 207] -> [207, V * 3],
    J_regressor densified, parents = kintree_table[0] with the root set to -1.  chumpy objects and the old scipy.sparse module
    paths inside the file are resolved without either package's pickling support.  extra_regressor: path of the reference's
    J_regressor_extra.npy (path_config.JOINT_REGRESSOR_TRAIN_EXTRA) or the [9, V] array; zeros when absent."""
    import pickle
    import scipy.sparse as sp

    class _Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if module.split('.')[0] == 'chumpy':
                return _PickledArray
            if module.startswith('scipy.sparse') and hasattr(sp, name):
                return getattr(sp, name)
            return super().find_class(module, name)

    with open(path, 'rb') as f:
        d = _Unpickler(f, encoding='latin1').load()
    if not isinstance(d, dict):
        d = dict(vars(d))
    dense = lambda a: np.asarray(a.todense()) if sp.issparse(a) else np.asarray(a)
    v_template = np.asarray(dense(d['v_template']), np.float32)
    V = v_template.shape[0]
    shapedirs = np.asarray(dense(d['shapedirs']), np.float32)[:, :, :num_betas]
    posedirs = np.asarray(dense(d['posedirs']), np.float32).reshape(V * 3, -1).T
    parents = np.asarray(d['kintree_table'])[0].astype(np.int64)
    parents[0] = -1
    if extra_regressor is None:
        extra = np.zeros((9, V), np.float32)
    else:
        extra = np.asarray(np.load(extra_regressor) if isinstance(extra_regressor, str) else extra_regressor, np.float32)
    return {'v_template': v_template, 'faces': np.asarray(d['f']).astype(np.int32), 'shapedirs': shapedirs, 'posedirs': np.ascontiguousarray(posedirs),
            'J_regressor': np.asarray(dense(d['J_regressor']), np.float32), 'lbs_weights': np.asarray(dense(d['weights']), np.float32),
            'parents': parents.astype(np.int32), 'J_regressor_extra': extra, 'landmark_verts': SMPL_LANDMARK_VERTS.copy()}


def load_densepose_mat(path):
    """Load the real UV_Processed.mat (user-supplied)."""
    from scipy.io import loadmat
    m = loadmat(path)
    return {
        'All_vertices': np.asarray(m['All_vertices']).reshape(-1),
        'All_Faces': np.asarray(m['All_Faces']),
        'All_FaceIndices': np.asarray(m['All_FaceIndices']).reshape(-1),
        'All_U_norm': np.asarray(m['All_U_norm']).reshape(-1),
        'All_V_norm': np.asarray(m['All_V_norm']).reshape(-1),
    }


def default_mean_params(seed=0):
    """Stand-in for data/smpl_mean_params.npz (keys cam, shape, pose; see
    /root/reference/models/danet/smpl_regressor.py:52-62): identity-ish 6-D pose."""
    pose6 = np.tile(np.array([1.0, 0.0, 0.0, 1.0, 0.0, 0.0], np.float32), 24)
    return {'pose': pose6, 'shape': np.zeros(10, np.float32),
            'cam': np.array([0.9, 0.0, 0.0], np.float32)}
