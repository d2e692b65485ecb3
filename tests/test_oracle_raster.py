"""Known-answer tests that pin the IUV-raster oracle (oracle/raster_ref.c).  neural_renderer
is absent and the reference has no tests, so the rule is pinned analytically."""
import numpy as np

import oracle
from oracle import numpy_ref as R
from conftest import rand_pose_shape

F0, ORIG, S = 5000.0, 224.0, 56


def _cam_tz(s):
    return 2 * F0 / (ORIG * s)


def _tri_scene(tris, tex=None, s=1.0, orig=ORIG, S_=S):
    """tris: [F,3,3] camera-space triangles BEFORE translation (t is added by the renderer)."""
    tris = np.asarray(tris, np.float32)
    F = tris.shape[0]
    verts = tris.reshape(1, F * 3, 3)
    vm = np.arange(F * 3, dtype=np.int32)
    faces = np.arange(F * 3, dtype=np.int32).reshape(F, 3)
    if tex is None:
        tex = np.stack([(np.arange(F) + 1) / 24.0, np.full(F, 0.25), np.full(F, 0.75)], 1).astype(np.float32)
    cam = np.array([[s, 0.0, 0.0]], np.float32)
    return oracle.raster_forward(verts, cam, vm, faces, tex, F0, orig, S_)


def _front_tri(x0, y0, x1, y1, x2, y2, z=0.0):
    """Triangle given in image pixel coords of the orig-size image, front-facing order is
    the caller's responsibility."""
    tz = _cam_tz(1.0)
    def back(u, v):
        return ((u - ORIG / 2) * (tz + z) / F0, (v - ORIG / 2) * (tz + z) / F0, z)
    return [back(x0, y0), back(x1, y1), back(x2, y2)]


def test_single_triangle_pixel_set():
    # image-space (y down) clockwise-on-screen = CCW in y-up NDC -> front-facing
    tri = _front_tri(40.0, 30.0, 40.0, 190.0, 200.0, 110.0)
    img, fidx, depth = _tri_scene([tri])
    # analytic coverage: pixel centres (c+.5, r+.5)*4 inside the triangle
    cov = np.zeros((S, S), bool)
    P = np.array([[40.0, 30.0], [40.0, 190.0], [200.0, 110.0]])
    for r in range(S):
        for c in range(S):
            p = np.array([(c + 0.5) * ORIG / S, (r + 0.5) * ORIG / S])
            d = [(P[(i + 1) % 3][0] - P[i][0]) * (p[1] - P[i][1]) - (P[(i + 1) % 3][1] - P[i][1]) * (p[0] - P[i][0]) for i in range(3)]
            cov[r, c] = all(x < -1e-3 for x in d) or all(x > 1e-3 for x in d)
    border = np.zeros((S, S), bool)   # pixels whose centre is within 1e-3 of an edge are undecided
    got = fidx[0] >= 0
    assert cov.sum() > 300
    assert (got[cov]).all()
    assert got.sum() - cov.sum() <= 4          # only centres exactly on an edge may differ
    np.testing.assert_allclose(img[0, 0][got], 1 / 24.0)
    np.testing.assert_allclose(img[0, 1][got], 0.25)
    np.testing.assert_allclose(img[0, 2][got], 0.75)
    assert (img[0][:, ~got] == 0).all()
    np.testing.assert_allclose(depth[0][got], _cam_tz(1.0), rtol=1e-5)
    assert np.isinf(depth[0][~got]).all()


def test_backface_culled_and_winding():
    tri = _front_tri(40.0, 30.0, 40.0, 190.0, 200.0, 110.0)
    back = [tri[0], tri[2], tri[1]]
    img, fidx, _ = _tri_scene([back])
    assert (fidx < 0).all() and (img == 0).all()


def test_nearer_triangle_wins_and_tie_goes_to_lower_index():
    a = _front_tri(20.0, 20.0, 20.0, 200.0, 200.0, 110.0, z=0.5)
    b = _front_tri(60.0, 20.0, 60.0, 200.0, 220.0, 110.0, z=-0.5)     # nearer
    _, fidx, depth = _tri_scene([a, b])
    both_a = _tri_scene([a])[1][0] >= 0
    both_b = _tri_scene([b])[1][0] >= 0
    overlap = both_a & both_b
    assert overlap.sum() > 100
    assert (fidx[0][overlap] == 1).all()
    assert (fidx[0][both_a & ~both_b] == 0).all()
    # identical triangles: the lower face index wins (strict <)
    _, fidx2, _ = _tri_scene([a, a])
    assert set(np.unique(fidx2)) == {-1, 0}


def test_behind_camera_and_outside_frustum_not_drawn():
    tz = _cam_tz(1.0)
    behind = [(-.1, -.1, -tz - 1.0), (-.1, .1, -tz - 1.0), (.1, 0, -tz - 1.0)]
    too_near = [(-.001, -.001, -tz + 0.05), (-.001, .001, -tz + 0.05), (.001, 0, -tz + 0.05)]
    outside = _front_tri(400.0, 30.0, 400.0, 190.0, 500.0, 110.0)
    img, fidx, _ = _tri_scene([behind, too_near, outside])
    assert (fidx < 0).all()


def test_camera_relations():
    # a point on the optical axis lands at the principal point; s <-> t_z as renderer.py:289
    for s in (0.6, 1.0, 1.1):
        tz = 2 * F0 / (ORIG * s)
        half = tz / F0 * 8                # +-8 orig-pixels around the axis
        tri = [(-half, -half, 0), (-half, half, 0), (half, 0.0, 0)]
        _, fidx, depth = _tri_scene([tri], s=s)
        rr, cc = np.nonzero(fidx[0] >= 0)
        assert len(rr) > 0 and abs(rr.mean() - (S - 1) / 2) <= 1.0 and abs(cc.mean() - (S - 1) / 2) <= 1.0
        np.testing.assert_allclose(depth[0][fidx[0] >= 0], tz, rtol=1e-5)
    # orig != 224 keeps the reference's scaled principal point (renderer.py:219-224)
    orig = 256.0
    tz = 2 * F0 / orig
    half = tz / (F0 * orig / 224) * 10
    tri = [(-half, -half, 0), (-half, half, 0), (half, 0.0, 0)]
    _, fidx, _ = _tri_scene([tri], s=1.0, orig=orig, S_=64)
    rr, cc = np.nonzero(fidx[0] >= 0)
    c_expected = (orig / 2 * orig / 224) * 64 / orig - 0.5
    assert abs(cc.mean() - c_expected) <= 1.5 and abs(rr.mean() - c_expected) <= 1.5


def test_mesh_render_decodes_to_valid_parts(smpl_model, dp_tables):
    vm, faces, tex = dp_tables
    betas, pose = rand_pose_shape(4)
    verts, _ = oracle.lbs_forward(smpl_model, betas, pose, False, np.float32)
    rng = np.random.default_rng(0)
    cam = np.stack([rng.uniform(0.6, 1.1, 4), rng.uniform(-.1, .1, 4), rng.uniform(-.1, .1, 4)], 1)
    img, fidx, depth = oracle.raster_forward(verts, cam, vm, faces, tex, F0, 256.0, 64)
    part = np.rint(img[:, 0] * 24)
    assert set(np.unique(part)).issubset(set(range(25)))
    fg = fidx >= 0
    assert 0.05 < fg.mean() < 0.9                    # the body covers a sensible share of the image
    assert len(np.unique(part[fg])) >= 12            # many parts visible
    assert (part[~fg] == 0).all() and (part[fg] >= 1).all()
    U, V, I, A = R.iuv_img2map(img)
    np.testing.assert_array_equal(I.sum(1), np.ones_like(I[:, 0]))
    np.testing.assert_array_equal(A.sum(1), np.ones_like(I[:, 0]))
    # every drawn pixel is a front-facing face: its texel is the face's
    np.testing.assert_array_equal(img[:, 0][fg], tex[fidx[fg], 0])
