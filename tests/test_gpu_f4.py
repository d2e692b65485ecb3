"""SURVEY 8 row f4 on the device: the evaluation metrics (eval.py:183-216) and the batched input pipeline
(datasets/base_dataset.py:115-214, utils/imutils.py:11-220) run on cuda tensors against the same reference goldens (g13, g14)
the CPU tests in test_host_logic.py use -- the code is device-agnostic tensor arithmetic, this pins that it stays so (fp64
transforms, integer truncation, scatter-free heat maps) on the GPU the evaluation loop and the data loader would use."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_eval_metrics_on_device_vs_reference():
    from danet_densepose2smpl_amd import metrics
    g = golden('g13_eval_metrics')
    pred, gt = _t(g['pred']), _t(g['gt'])
    np.testing.assert_allclose(metrics.mpjpe(pred, gt).cpu().numpy(), g['mpjpe'], rtol=1e-5)
    np.testing.assert_allclose(metrics.similarity_transform(pred, gt).cpu().numpy(), g['aligned'], atol=5e-5)
    np.testing.assert_allclose(metrics.reconstruction_error(pred, gt).cpu().numpy(), g['recon'], rtol=1e-4, atol=1e-6)
    assert metrics.mpjpe(pred, gt).is_cuda


def test_pose_evaluation_block_on_device():
    from danet_densepose2smpl_amd import metrics
    rng = np.random.default_rng(5)
    B, V = 5, 300
    Jr = rng.random((17, V)).astype(np.float32); Jr /= Jr.sum(1, keepdims=True)
    mapper = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10]
    pv, gv = rng.normal(size=(B, V, 3)).astype(np.float32), rng.normal(size=(B, V, 3)).astype(np.float32)
    j = np.einsum('jv,bvk->bjk', Jr, pv); pj = j[:, mapper] - j[:, [0]]
    g = np.einsum('jv,bvk->bjk', Jr, gv); gj = g[:, mapper] - g[:, [0]]
    e, r, j17 = metrics.pose_errors(_t(pv), _t(Jr), mapper, gt_vertices=_t(gv))
    assert e.is_cuda and r.is_cuda
    np.testing.assert_allclose(j17.cpu().numpy(), j, atol=1e-5)
    np.testing.assert_allclose(e.cpu().numpy(), np.sqrt(((pj - gj) ** 2).sum(-1)).mean(-1), rtol=1e-5)
    rc = metrics.reconstruction_error(torch.from_numpy(pj), torch.from_numpy(gj)).numpy()          # the CPU result of the same primitive
    np.testing.assert_allclose(r.cpu().numpy(), rc, rtol=1e-4, atol=1e-6)


def test_augmentation_arithmetic_on_device_vs_reference():
    from danet_densepose2smpl_amd import augment
    g = golden('g14_augment')
    c, sc, rot, flip = (_t(g[k]) for k in ('center', 'scale', 'rot', 'flip'))
    np.testing.assert_allclose(augment.get_transform(c, sc, [224, 224], rot).cpu().numpy(), g['T'], rtol=1e-12, atol=1e-10)
    np.testing.assert_array_equal(augment.transform(_t(g['pts']), c, sc, [224, 224], rot=rot).cpu().numpy(), g['fwd'])
    np.testing.assert_array_equal(augment.transform(_t(g['opts']), c, sc, [224, 224], invert=1, rot=rot).cpu().numpy(), g['inv'])
    for N in (24, 49):
        got = augment.j2d_processing(_t(g['kp%d' % N]), c, sc, rot, flip)
        np.testing.assert_allclose(got.cpu().numpy(), g['j2d%d' % N], rtol=0, atol=1e-6)
    np.testing.assert_allclose(augment.j3d_processing(_t(g['S']), rot, flip).cpu().numpy(), g['j3d'], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(augment.flip_pose(_t(g['pose'])).cpu().numpy(), g['pose_flipped'])
    B = g['hm_joints'].shape[0]
    for sigma in (1, 2):
        sel = [b for b in range(B) if 1 + b % 2 == sigma]
        hm, w = augment.generate_heatmap(_t(g['hm_joints'][sel]), 56, sigma=sigma, joints_vis=_t(g['hm_vis'][sel]))
        assert hm.is_cuda
        np.testing.assert_array_equal(w.cpu().numpy(), g['hm_w'][sel])
        np.testing.assert_allclose(hm.cpu().numpy(), g['hm'][sel].astype(np.float32), atol=1e-3)
    # the image crop: a bright source pixel lands where transform() says, on the device as on the host
    img = torch.zeros(1, 3, 300, 400, device=DEV); img[0, :, 150, 210] = 255.
    cc, ss, rr = torch.tensor([[200., 160.]], device=DEV), torch.tensor([1.1], device=DEV), torch.tensor([25.], device=DEV)
    out = augment.crop_images(img, cc, ss, rr, 224)
    assert out.is_cuda
    v, u = np.unravel_index(int(out[0, 0].argmax()), (224, 224))
    p = (augment.get_transform(cc, ss, [224, 224], rr)[0].cpu() @ torch.tensor([210., 150., 1.], dtype=torch.float64)).numpy()
    assert abs(u - p[0]) <= 1.0 and abs(v - p[1]) <= 1.0
