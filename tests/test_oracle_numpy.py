"""Pins oracle/numpy_ref.py against golden vectors produced by the reference itself
(tests/golden/make_golden.py; reference files utils/geometry.py, utils/iuvmap.py,
utils/graph.py, utils/keypoints.py)."""
import numpy as np

from conftest import golden
from oracle import numpy_ref as R


def test_geometry_vs_reference():
    g = golden('g1_geometry')
    np.testing.assert_allclose(R.batch_rodrigues(g['theta'].astype(np.float64)), g['R'], atol=2e-6)
    np.testing.assert_allclose(R.quat_to_rotmat(g['quat'].astype(np.float64)), g['Rq'], atol=2e-6)
    np.testing.assert_allclose(R.rot6d_to_rotmat(g['x6'].astype(np.float64)), g['R6'], atol=2e-6)
    proj = R.perspective_projection(g['pts'].astype(np.float64), g['rot'].astype(np.float64),
                                    g['t'].astype(np.float64), 5000., g['cc'].astype(np.float64))
    np.testing.assert_allclose(proj, g['proj'], rtol=2e-5, atol=1e-3)


def test_rodrigues_near_zero_is_finite():
    R0 = R.batch_rodrigues(np.zeros((2, 3)))
    assert np.isfinite(R0).all()
    np.testing.assert_allclose(R0, np.tile(np.eye(3), (2, 1, 1)), atol=1e-7)


def test_iuvmap_vs_reference():
    g = golden('g2_iuvmap')
    cU, cV, cI, cA = R.iuvmap_clean(g['U'], g['V'], g['I'], g['A'])
    for a, b in ((cU, g['cU']), (cV, g['cV']), (cI, g['cI']), (cA, g['cA'])):
        np.testing.assert_array_equal(a, b)
    mU, mV, mI, mA = R.iuv_img2map(g['img'])
    np.testing.assert_array_equal(mI, g['mI'])       # integer-valued planes: bit exact
    np.testing.assert_array_equal(mA, g['mA'])
    np.testing.assert_allclose(mU, g['mU'], atol=0)
    np.testing.assert_allclose(mV, g['mV'], atol=0)
    assert (np.argmax(mI, 1) == g['part']).all()


def test_graph_and_softmax_integral_vs_reference():
    g = golden('g3_graph')
    np.testing.assert_allclose(R.normalize_undigraph(g['Ar'][0].astype(np.float64)), g['und'][0], atol=1e-6)
    np.testing.assert_allclose(R.normalize_digraph(g['Ar'][0].astype(np.float64), AD_mode=False), g['dig_da'], atol=1e-12)
    np.testing.assert_allclose(R.normalize_digraph(g['Ar'][0].astype(np.float64), AD_mode=True), g['dig_ad'], atol=1e-12)
    np.testing.assert_allclose(R.softmax_integral(10 * g['hm']), g['softint'], atol=2e-4)
