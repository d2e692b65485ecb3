"""CPU tests of the host-side logic (pure torch / numpy parts of the product) against golden
vectors from the reference: IUV map glue, skeleton graphs, loss functions with per-sample
weights instead of boolean-mask indexing, config, state-dict naming."""
import numpy as np
import torch

import os

from conftest import golden, ROOT


def test_iuvmap_glue_vs_reference():
    from danet_densepose2smpl_amd.iuvmap import iuvmap_clean, iuv_img2map
    g = golden('g2_iuvmap')
    t = lambda k: torch.from_numpy(g[k])
    cU, cV, cI, cA = iuvmap_clean(t('U'), t('V'), t('I'), t('A'))
    for a, k in ((cU, 'cU'), (cV, 'cV'), (cI, 'cI'), (cA, 'cA')):
        np.testing.assert_array_equal(a.numpy(), g[k])
    mU, mV, mI, mA = iuv_img2map(t('img'))
    for a, k in ((mU, 'mU'), (mV, 'mV'), (mI, 'mI'), (mA, 'mA')):
        np.testing.assert_array_equal(a.numpy(), g[k])


def test_graphs_and_softmax_integral_vs_reference():
    from danet_densepose2smpl_amd import gcn
    from danet_densepose2smpl_amd.geometry import softmax_integral_tensor
    g = golden('g3_graph')
    np.testing.assert_array_equal(gcn.adjacency('smpl'), g['A_smpl'])
    np.testing.assert_array_equal(gcn.adjacency('smpl_2neigh'), g['A_smpl2'])
    np.testing.assert_allclose(gcn.normalize_undigraph(torch.from_numpy(g['Ar'])).numpy(), g['und'], atol=1e-6)
    np.testing.assert_allclose(gcn.normalize_digraph(g['Ar'][0].astype(np.float64), AD_mode=False), g['dig_da'], atol=1e-12)
    hm = torch.from_numpy(g['hm'])
    np.testing.assert_allclose(softmax_integral_tensor(10 * hm, 24, 16, 16).numpy(), g['softint'], atol=2e-4)


def test_predictor_graph_buffers_vs_reference():
    from danet_densepose2smpl_amd.config import reset_cfg
    from danet_densepose2smpl_amd.smpl_regressor import DecomposedPredictor
    reset_cfg()
    g = golden('g9_predictor')
    pose6 = torch.tensor([1., 0., 0., 1., 0., 0.]).repeat(24).unsqueeze(0)
    net = DecomposedPredictor(None, (torch.tensor([[0.9, 0., 0.]]), torch.zeros(1, 10), pose6), pretrained=False)
    for k in ('I_n', 'A_link', 'A_mask', 'A', 'r2p_A', 'p2r_A'):
        np.testing.assert_allclose(getattr(net, k).numpy(), g[k], atol=1e-6, err_msg=k)


def test_masked_losses_vs_reference_boolean_indexing():
    from danet_densepose2smpl_amd.smpl_regressor import SMPL_Regressor as R
    from danet_densepose2smpl_amd.iuv_estimator import IUV_Estimator as E
    from danet_densepose2smpl_amd.iuvmap import iuv_img2map
    from danet_densepose2smpl_amd.config import reset_cfg
    reset_cfg()
    g = golden('g10_losses')
    t = lambda k: torch.from_numpy(g[k])
    lp, lb = R.smpl_losses(t('pred_rot'), t('pb'), t('gt_rot'), t('gb'), t('has_smpl'))
    np.testing.assert_allclose(lp.item(), g['loss_pose'], rtol=1e-5)
    np.testing.assert_allclose(lb.item(), g['loss_betas'], rtol=1e-5)
    np.testing.assert_allclose(R.keypoint_loss(t('kp2'), t('gk2'), 0.25, 1.0).item(), g['loss_kp2d'], rtol=1e-5)
    np.testing.assert_allclose(R.keypoint_3d_loss(t('pj'), t('g3'), t('has_kp3d')).item(), g['loss_kp3d'], rtol=1e-5)
    np.testing.assert_allclose(R.shape_loss(t('pv'), t('gv'), t('has_smpl')).item(), g['loss_verts'], rtol=1e-5)
    np.testing.assert_allclose(R.l1_losses(t('a'), t('b'), t('has_smpl')).item(), g['loss_l1'], rtol=1e-5)
    uvia = iuv_img2map(t('iuv_gt'))
    lU, lV, lI, lA = E.body_uv_losses(t('u'), t('v'), t('idx'), t('ann'), uvia, t('has_iuv'))
    for a, k in ((lU, 'loss_U'), (lV, 'loss_V'), (lI, 'loss_I'), (lA, 'loss_A')):
        np.testing.assert_allclose(a.item(), g[k], rtol=2e-5, err_msg=k)
    # no sample selected -> zeros, like the reference's early-outs
    z = torch.zeros(6)
    assert R.l1_losses(t('a'), t('b'), z).item() == 0.0
    assert all(x.item() == 0.0 for x in E.body_uv_losses(t('u'), t('v'), t('idx'), t('ann'), uvia, z.bool()))


def test_state_dict_names_match_reference_tree():
    """Spot-check of SURVEY.md Appendix F key names (full equality is checked against the reference
    modules in the build container by tools/check_state_dict.py)."""
    from danet_densepose2smpl_amd.config import reset_cfg
    from danet_densepose2smpl_amd.hrnet import PoseHighResolutionNet
    reset_cfg()
    sd = PoseHighResolutionNet(part_out_dim=7).state_dict()
    assert len(sd) == 1818
    for k, shp in (('conv1.weight', (64, 3, 3, 3)), ('transition1.1.0.0.weight', (96, 256, 3, 3)),
                   ('stage2.0.fuse_layers.0.1.0.weight', (48, 96, 1, 1)), ('stage2.0.fuse_layers.1.0.0.0.weight', (96, 48, 3, 3)),
                   ('stage4.2.fuse_layers.0.3.1.running_var', (48,)), ('final_pred.predict_partial_iuv.weight', (504, 48, 3, 3)),
                   ('final_pred.predict_hm.0.2.conv3.weight', (48, 12, 1, 1)), ('final_pred.predict_hm.1.bias', (24,))):
        assert tuple(sd[k].shape) == shp, k
    assert sum(v.numel() for k, v in sd.items() if 'running' not in k and 'num_batches' not in k) == 63870282


def test_config_overrides():
    from danet_densepose2smpl_amd.config import cfg, cfg_from_dict, reset_cfg
    reset_cfg()
    assert cfg.DANET.INIMG_SIZE == 224 and cfg.DANET.HEATMAP_SIZE == 56 and cfg.DANET.REFINE_STRATEGY == 'gcn'
    cfg_from_dict({'DANET.INIMG_SIZE': 256, 'DANET.HEATMAP_SIZE': 64})
    assert cfg.DANET.INIMG_SIZE == 256 and cfg.HR_MODEL.EXTRA.STAGE4.NUM_CHANNELS == [48, 96, 192, 384]
    reset_cfg()


import pytest


@pytest.mark.parametrize('align', [0, 1])
def test_dp_point_losses_vs_reference(align):
    """DensePose point supervision in masked-weight form == the reference evaluated on the has_dp subset
    (golden g11: losses and the gradients w.r.t. all four prediction maps)."""
    from danet_densepose2smpl_amd.config import reset_cfg, cfg_from_dict
    from danet_densepose2smpl_amd.iuv_estimator import IUV_Estimator
    reset_cfg()
    cfg_from_dict({'DANET.HEATMAP_SIZE': 16})
    g = golden('g11_dp_losses_align%d' % align)
    u, v, idx, ann = (torch.from_numpy(g[k]).requires_grad_(True) for k in ('u', 'v', 'idx', 'ann'))
    dp = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('dp__')}
    has_dp = torch.from_numpy(g['has_dp'])
    lU, lV, lI, lA = IUV_Estimator.dp_uvia_losses(u, v, idx, ann, dp, has_dp, bool(align))
    for ours, k in ((lU, 'loss_Udp'), (lV, 'loss_Vdp'), (lI, 'loss_IndexUVdp'), (lA, 'loss_segAnndp')):
        assert abs(float(ours) - float(g[k])) <= 1e-5 * max(1.0, abs(float(g[k]))), k
    (lU * 1.0 + lV * 2.0 + lI * 3.0 + lA * 4.0).backward()
    for t, k in ((u, 'gu'), (v, 'gv'), (idx, 'gidx'), (ann, 'gann')):
        np.testing.assert_allclose(t.grad.numpy(), g[k], atol=1e-6 + 1e-5 * np.abs(g[k]).max())
    # no labelled sample at all: zeros, as iuv_estimator.py:118-121
    z = IUV_Estimator.dp_uvia_losses(u.detach(), v.detach(), idx.detach(), ann.detach(), dp, torch.zeros(4), bool(align))
    assert all(float(t) == 0.0 for t in z)


def test_checkpoint_files_in_the_reference_format(tmp_path):
    """saver.py-style training checkpoints and demo.py-style pretrained files round-trip."""
    from danet_densepose2smpl_amd import checkpoint
    from danet_densepose2smpl_amd.config import reset_cfg, cfg_from_dict
    from danet_densepose2smpl_amd.danet import DaNet
    from danet_densepose2smpl_amd.trainer import default_options
    reset_cfg()
    cfg_from_dict({'DANET.INIMG_SIZE': 64, 'DANET.HEATMAP_SIZE': 16})
    torch.manual_seed(0)
    a = DaNet(default_options(2), None, pretrained=False)
    torch.manual_seed(1)
    b = DaNet(default_options(2), None, pretrained=False)
    with torch.no_grad():
        for p in a.parameters():
            p.add_(0.01)
    opt = torch.optim.Adam([p for p in a.parameters() if p.requires_grad], lr=1e-4)
    path = checkpoint.save_checkpoint(str(tmp_path / 'ck' / '00000010.pt'), {'model': a}, {'optimizer': opt}, epoch=3, batch_idx=7,
                                      batch_size=2, dataset_perm=[1, 0], total_step_count=10)
    raw = torch.load(path, weights_only=False)
    assert set(raw) >= {'model', 'optimizer', 'epoch', 'batch_idx', 'batch_size', 'dataset_perm', 'total_step_count'}
    assert not any(k.startswith('iuv2smpl.smpl.') for k in raw['model']) and len(raw['model']) > 2000
    book = checkpoint.load_checkpoint(path, {'model': b})
    assert book == {'epoch': 3, 'batch_idx': 7, 'batch_size': 2, 'dataset_perm': [1, 0], 'total_step_count': 10}
    sa, sb = a.state_dict(), b.state_dict()
    assert all(torch.equal(sa[k], sb[k]) for k in raw['model'])
    # a released-weights file: {'model': DataParallel-prefixed state dict}, loaded non-strictly
    torch.save({'model': {'module.' + k: v for k, v in raw['model'].items() if 'predict_hm' not in k}}, str(tmp_path / 'rel.pt'))
    torch.manual_seed(2)
    c = DaNet(default_options(2), None, pretrained=False)
    missing, unexpected = checkpoint.load_pretrained(c, str(tmp_path / 'rel.pt'))
    assert unexpected == [] and all('predict_hm' in k or k.startswith('iuv2smpl.smpl.') for k in missing) and missing
    k0 = 'img2iuv.iuv_est.conv1.weight'
    assert torch.equal(c.state_dict()[k0], sa[k0])
    with pytest.raises(ValueError):
        checkpoint.load_pretrained(c, str(tmp_path / 'nope.pt'))


def test_label_prologue_geometry_vs_reference():
    """batched estimate_translation + projected key-points + renderer camera == the reference's
    per-sample numpy loop and masked assignments (golden g12)."""
    from danet_densepose2smpl_amd.geometry import estimate_translation, label_prologue
    g = golden('g12_label_prologue')
    t = lambda k: torch.from_numpy(g[k])
    kp = t('keypoints').clone()
    kp_px = kp.clone()
    kp_px[:, :, :-1] = 0.5 * 224 * (kp_px[:, :, :-1] + 1)
    cam_t = estimate_translation(t('joints'), kp_px, 5000., 224)
    np.testing.assert_allclose(cam_t.numpy(), g['cam_t'], rtol=2e-5, atol=2e-5)
    tk, cam, ct = label_prologue(t('joints'), t('smpl_joints'), kp, t('has_iuv'), t('has_dp'), t('smpl_2dkps'), 5000., 224)
    np.testing.assert_allclose(ct.numpy(), g['cam_t'], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(tk.numpy(), g['target_smpl_kps'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(cam.numpy(), g['target_cam'], rtol=2e-5, atol=2e-5)


def test_eval_metrics_vs_reference():
    """batched MPJPE / Procrustes reconstruction error == the reference's numpy loop (golden g13,
    including a reflected sample that exercises the det(R) = -1 correction)."""
    from danet_densepose2smpl_amd import metrics
    g = golden('g13_eval_metrics')
    pred, gt = torch.from_numpy(g['pred']), torch.from_numpy(g['gt'])
    np.testing.assert_allclose(metrics.mpjpe(pred, gt).numpy(), g['mpjpe'], rtol=1e-5)
    np.testing.assert_allclose(metrics.similarity_transform(pred, gt).numpy(), g['aligned'], atol=2e-5)
    np.testing.assert_allclose(metrics.reconstruction_error(pred, gt).numpy(), g['recon'], rtol=1e-4, atol=1e-6)
    assert abs(float(metrics.reconstruction_error(pred, gt, 'mean')) - float(g['recon'].mean())) < 1e-6


def test_pose_evaluation_block():
    """eval.py:183-216 as one call -- joints regressed from the mesh, pelvis-centred, mapped to the 14 common
    joints, MPJPE + Procrustes error -- against the same steps written out with the pinned primitives."""
    from danet_densepose2smpl_amd import metrics
    rng = np.random.default_rng(5)
    B, V = 5, 300
    Jr = rng.random((17, V)).astype(np.float32); Jr /= Jr.sum(1, keepdims=True)
    mapper = [6, 5, 4, 1, 2, 3, 16, 15, 14, 11, 12, 13, 8, 10]              # constants.H36M_TO_J14 of the reference
    pv, gv = rng.normal(size=(B, V, 3)).astype(np.float32), rng.normal(size=(B, V, 3)).astype(np.float32)
    j = np.einsum('jv,bvk->bjk', Jr, pv); pj = j[:, mapper] - j[:, [0]]
    g = np.einsum('jv,bvk->bjk', Jr, gv); gj = g[:, mapper] - g[:, [0]]
    e, r, j17 = metrics.pose_errors(torch.from_numpy(pv), torch.from_numpy(Jr), mapper, gt_vertices=torch.from_numpy(gv))
    np.testing.assert_allclose(j17.numpy(), j, atol=1e-5)
    np.testing.assert_allclose(e.numpy(), np.sqrt(((pj - gj) ** 2).sum(-1)).mean(-1), rtol=1e-5)
    np.testing.assert_allclose(r.numpy(), metrics.reconstruction_error(torch.from_numpy(pj), torch.from_numpy(gj)).numpy(), rtol=1e-5)
    e2, r2, _ = metrics.pose_errors(torch.from_numpy(pv), torch.from_numpy(Jr), mapper, gt_keypoints_3d=torch.from_numpy(gj))
    np.testing.assert_allclose(e2.numpy(), e.numpy(), rtol=1e-6)
    with pytest.raises(ValueError):
        metrics.pose_errors(torch.from_numpy(pv), torch.from_numpy(Jr), mapper)


def test_augmentation_arithmetic_vs_reference():
    """SURVEY 8 row f4 (input pipeline): the batched crop transform (with the reference's truncation to integers),
    keypoint / pose flips, 2D / 3D keypoint processing and Gaussian heat-map targets == the reference's per-sample numpy
    code (golden g14); rot_aa against scipy's rotation algebra (cv2.Rodrigues is not available here)."""
    from danet_densepose2smpl_amd import augment
    from scipy.spatial.transform import Rotation
    g = golden('g14_augment')
    c, sc, rot, flip = (torch.from_numpy(g[k]) for k in ('center', 'scale', 'rot', 'flip'))
    np.testing.assert_allclose(augment.get_transform(c, sc, [224, 224], rot).numpy(), g['T'], rtol=1e-12, atol=1e-10)
    np.testing.assert_array_equal(augment.transform(torch.from_numpy(g['pts']), c, sc, [224, 224], rot=rot).numpy(), g['fwd'])
    np.testing.assert_array_equal(augment.transform(torch.from_numpy(g['opts']), c, sc, [224, 224], invert=1, rot=rot).numpy(), g['inv'])
    for N in (24, 49):
        got = augment.j2d_processing(torch.from_numpy(g['kp%d' % N]), c, sc, rot, flip)
        np.testing.assert_allclose(got.numpy(), g['j2d%d' % N], rtol=0, atol=1e-6)
    np.testing.assert_allclose(augment.j3d_processing(torch.from_numpy(g['S']), rot, flip).numpy(), g['j3d'], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(augment.flip_pose(torch.from_numpy(g['pose'])).numpy(), g['pose_flipped'])
    B = g['hm_joints'].shape[0]
    for sigma in (1, 2):
        sel = [b for b in range(B) if 1 + b % 2 == sigma]
        hm, w = augment.generate_heatmap(torch.from_numpy(g['hm_joints'][sel]), 56, sigma=sigma, joints_vis=torch.from_numpy(g['hm_vis'][sel]))
        np.testing.assert_array_equal(w.numpy(), g['hm_w'][sel])
        np.testing.assert_allclose(hm.numpy(), g['hm'][sel].astype(np.float32), atol=1e-3)
        assert ((hm.numpy() > 0) == (g['hm'][sel] > 0)).mean() > 0.9999            # same support (fp16 storage flushes the far tail)
    # rot_aa / pose_processing: R_z(-rot) composed with the global orientation
    rng = np.random.default_rng(0)
    aa = rng.normal(0, 1.2, (64, 3)); aa[0] = [0, 0, 0]; aa[1] = [np.pi - 1e-6, 0, 0]
    rots = rng.uniform(-60, 60, 64); rots[::4] = 0
    want = (Rotation.from_euler('z', -rots, degrees=True) * Rotation.from_rotvec(aa)).as_matrix()
    got = Rotation.from_rotvec(augment.rot_aa(torch.from_numpy(aa), torch.from_numpy(rots)).numpy()).as_matrix()
    np.testing.assert_allclose(got, want, atol=1e-6)
    pose = torch.from_numpy(rng.normal(0, 0.4, (8, 72)))
    out = augment.pose_processing(pose, torch.zeros(8), torch.tensor([0, 1] * 4))
    np.testing.assert_allclose(out[0].numpy(), pose[0].numpy(), atol=1e-6)
    np.testing.assert_allclose(out[1].numpy(), augment.flip_pose(pose[1:2])[0].numpy(), atol=1e-6)
    # the image crop uses the same matrix: a bright source pixel lands where transform() says
    img = torch.zeros(1, 3, 300, 400); img[0, :, 150, 210] = 255.
    cc, ss, rr = torch.tensor([[200., 160.]]), torch.tensor([1.1]), torch.tensor([25.])
    out = augment.crop_images(img, cc, ss, rr, 224)
    v, u = np.unravel_index(int(out[0, 0].argmax()), (224, 224))
    p = (augment.get_transform(cc, ss, [224, 224], rr)[0] @ torch.tensor([210., 150., 1.], dtype=torch.float64)).numpy()
    assert abs(u - p[0]) <= 1.0 and abs(v - p[1]) <= 1.0
    full = augment.rgb_processing(img, cc, ss, rr, torch.tensor([1]), torch.ones(1, 3))
    assert full.shape == (1, 3, 224, 224) and abs(int(full[0, 0].argmax()) % 224 - (223 - u)) <= 1


def test_dp_dict_producer_construction():
    """dp_utils.dp_annot_process (parity unpinned: the reference needs cv2 / pycocotools).  Analytic case: the
    crop window equals the annotated box, so the label image is sampled on its own 256-grid and points keep their
    relative position; points outside the crop are dropped; the per-part weight blocks select the points of each part."""
    from danet_densepose2smpl_amd import dp_utils, augment
    M, res = 56, 224
    center, scale = [150., 130.], 1.0                      # crop window = 200-pixel square around the centre
    ul = augment.transform(torch.tensor([[[1., 1.]]]), torch.tensor([center]), torch.tensor([scale]), [res, res], invert=1)[0, 0].numpy() - 1
    br = augment.transform(torch.tensor([[[res + 1., res + 1.]]]), torch.tensor([center]), torch.tensor([scale]), [res, res], invert=1)[0, 0].numpy() - 1
    bbox = [ul[0], ul[1], br[0] - ul[0], br[1] - ul[1]]
    lab = (np.arange(256)[:, None] // 19 + 1) * np.ones((1, 256), np.int64)        # horizontal bands 1..14
    lab = lab.astype(np.uint8)
    ann = {'bbox': bbox, 'dp_Ilabel': lab, 'dp_I': [1, 5, 24, 7, 3], 'dp_U': [.1, .2, .3, .4, .5], 'dp_V': [.9, .8, .7, .6, .5],
           'dp_x': [0., 127.5, 250., 300., 64.], 'dp_y': [0., 127.5, 200., 10., -20.]}
    d = dp_utils.dp_annot_process(ann, M, res, center, scale, 0)
    ref = dp_utils.empty_dp_dict(M)
    assert set(d) == set(ref) and all(d[k].shape == ref[k].shape and d[k].dtype == ref[k].dtype for k in ref)
    L = d['body_uv_ann_labels'].reshape(M, M)
    ys = np.rint(np.arange(M) * 255. / M).astype(int)
    np.testing.assert_array_equal(L, lab[ys][:, np.rint(np.arange(M) * 255. / M).astype(int)])
    assert (d['body_uv_ann_weights'] == 1).all()
    n = 3                                                  # points 4 (x beyond the box) and 5 (y above it) are dropped (as is x = 255: > M - 1)
    np.testing.assert_allclose(d['body_uv_X_points'][:n], np.array([0., 127.5, 250.]) / 255. * M, atol=1e-4)
    np.testing.assert_allclose(d['body_uv_Y_points'][:n], np.array([0., 127.5, 200.]) / 255. * M, atol=1e-4)
    np.testing.assert_array_equal(d['body_uv_I_points'][:n + 2], [1, 5, 24, 0, 0])
    assert d['body_uv_U_points'].shape == (196 * 25,) and np.allclose(d['body_uv_U_points'][196:199], [.1, .2, .3])
    w = d['body_uv_point_weights'].reshape(25, 196)
    assert w[0].sum() == 0 and w[1, 0] == 1 and w[5, 1] == 1 and w[24, 2] == 1 and w.sum() == 3
    with pytest.raises(ValueError):
        dp_utils.dp_annot_process(ann, M, res, center, scale, 1)
    sym = lambda I, U, V, x, y, Il: (I, U, V, 255. - x, y, Il[:, ::-1])      # a stand-in for the licensed symmetry tables
    df = dp_utils.dp_annot_process(ann, M, res, center, scale, 1, symmetric=sym)
    assert df['body_uv_I_points'][:3].tolist() != [] and df['body_uv_ann_labels'].shape == (M * M,)


def test_dp_dict_producer_matches_reference_golden():
    """Golden g15 (tests/golden/make_golden.py g15_dp_producer): the reference's dp_annot_process and get_symmetric_densepose run
    on synthetic symmetry tables and RLE annotations (with this repo's RLE decode / nearest remap standing in for pycocotools /
    cv2 inside the reference) -- every entry of the dp_dict, flipped and unflipped samples, absent parts, points falling
    outside the crop; plus known answers of the RLE string format."""
    from danet_densepose2smpl_amd import dp_utils
    g = golden('g15_dp_producer')
    sym = dp_utils.DensePoseSymmetry(dict(zip(('U_transforms', 'V_transforms'), dp_utils.synthetic_symmetry_tables())))
    for case in range(4):
        polys = [{'size': [256, 256], 'counts': bytes(c)} if len(c) else [] for c in g['c%d_counts' % case]]
        ann = {'bbox': g['c%d_bbox' % case].tolist(), 'dp_masks': polys}
        for k in ('dp_I', 'dp_U', 'dp_V', 'dp_x', 'dp_y'):
            ann[k] = g['c%d_%s' % (case, k)].tolist()
        d = dp_utils.dp_annot_process(ann, 56, 224, g['c%d_center' % case].tolist(), float(g['c%d_scale' % case]), int(g['c%d_flipped' % case]), symmetric=sym)
        for k, v in d.items():
            want = g['c%d_out_%s' % (case, k)]
            assert v.shape == want.shape and v.dtype == want.dtype, (case, k, v.dtype, want.dtype)
            np.testing.assert_allclose(v, want, rtol=0, atol=1e-5, err_msg='case %d %s' % (case, k))
        assert int(g['c%d_flipped' % case]) == case % 2 and (d['body_uv_I_points'] > 0).sum() > 5
    I, U, V = g['c0_dp_I'], g['c0_dp_U'], g['c0_dp_V']
    res = sym.get_symmetric_densepose(I, U, V, g['c0_dp_x'], g['c0_dp_y'], g['sym_in_mask'])
    for name, r in zip(('I', 'U', 'V', 'x', 'y', 'mask'), res):
        np.testing.assert_allclose(np.asarray(r), g['sym_' + name], rtol=0, atol=1e-6, err_msg=name)
    # the tables as scipy.io.loadmat returns them (object array [1, 24]) give the same object
    u, v = dp_utils.synthetic_symmetry_tables()
    obj = lambda t: np.array([[t[i] for i in range(24)]], dtype=object)
    sym2 = dp_utils.DensePoseSymmetry({'U_transforms': obj(u), 'V_transforms': obj(v)})
    np.testing.assert_array_equal(sym2.get_symmetric_densepose(I, U, V, g['c0_dp_x'], g['c0_dp_y'], g['sym_in_mask'])[1], res[1])
    # COCO RLE strings (maskApi.c): 5-bit groups + 48, 0x20 = continuation, differences to the count two back from the fourth on
    assert dp_utils.rle_encode(np.ones((3, 2), np.uint8))['counts'] == b'06'
    assert dp_utils.rle_decode({'size': [10, 10], 'counts': b'T3'}).sum() == 0            # one run of 100 zeros: 100 = 4 + 3 * 32
    m = dp_utils.rle_decode({'size': [4, 3], 'counts': [2, 3, 4, 3]})                     # uncompressed list form, column-major runs
    np.testing.assert_array_equal(m, np.array([[0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 0, 1]], np.uint8))
    rng = np.random.default_rng(3)
    for t in range(50):
        h, w = rng.integers(1, 70, 2)
        mk = (rng.uniform(size=(h, w)) < rng.uniform()).astype(np.uint8)
        np.testing.assert_array_equal(dp_utils.rle_decode(dp_utils.rle_encode(mk)), mk)
    with pytest.raises(ValueError):
        dp_utils.rle_decode({'size': [4, 4], 'counts': [3, 3]})


def test_smpl_pkl_loader_reads_the_official_file_layout(tmp_path):
    """assets.load_smpl_pkl (/root/reference/models/smpl.py:15-19 -> smplx.SMPL(model_path)): a file with the official layout --
    chumpy leaves for v_template / shapedirs / posedirs / weights, a scipy.sparse J_regressor, uint32 kintree_table with the
    2^32 - 1 root, 300 shape components, python-2 style protocol-2 pickle -- loads without chumpy into the arrays the synthetic
    model was built from.  (The file is written here with a throw-away `chumpy` module in sys.modules: parity with a real
    SMPL_NEUTRAL.pkl is unpinned, the model is licence-gated.)"""
    import pickle
    import sys
    import types
    import scipy.sparse as sp
    from danet_densepose2smpl_amd import assets
    m = assets.make_synthetic_smpl(0)
    V = m['v_template'].shape[0]
    ch_mod, pkg = types.ModuleType('chumpy.ch'), types.ModuleType('chumpy')

    class Ch(object):
        def __init__(self, x):
            self.x = np.asarray(x, np.float64)
    Ch.__module__, Ch.__qualname__ = 'chumpy.ch', 'Ch'
    ch_mod.Ch, pkg.ch = Ch, ch_mod
    sys.modules.update({'chumpy': pkg, 'chumpy.ch': ch_mod})
    try:
        shapedirs300 = np.concatenate([m['shapedirs'], np.zeros((V, 3, 290), np.float32)], -1)
        kt = np.stack([m['parents'].astype(np.int64) % (1 << 32), np.arange(24)]).astype(np.uint32)
        d = {'v_template': Ch(m['v_template']), 'f': m['faces'].astype(np.uint32), 'shapedirs': Ch(shapedirs300),
             'posedirs': Ch(m['posedirs'].T.reshape(V, 3, 207)), 'J_regressor': sp.csc_matrix(m['J_regressor'].astype(np.float64)),
             'weights': Ch(m['lbs_weights']), 'kintree_table': kt, 'J': Ch(np.zeros((24, 3))), 'bs_style': 'lbs', 'bs_type': 'lrotmin'}
        path = tmp_path / 'SMPL_NEUTRAL.pkl'
        with open(path, 'wb') as f:
            pickle.dump(d, f, protocol=2)
    finally:
        del sys.modules['chumpy'], sys.modules['chumpy.ch']
    got = assets.load_smpl_pkl(str(path), extra_regressor=m['J_regressor_extra'])
    for k in ('v_template', 'faces', 'shapedirs', 'posedirs', 'J_regressor', 'lbs_weights', 'parents', 'J_regressor_extra', 'landmark_verts'):
        assert got[k].shape == m[k].shape and got[k].dtype == m[k].dtype, (k, got[k].shape, got[k].dtype, m[k].dtype)
        np.testing.assert_allclose(got[k], m[k], rtol=0, atol=1e-7, err_msg=k)
    assert np.abs(assets.load_smpl_pkl(str(path))['J_regressor_extra']).sum() == 0
    from danet_densepose2smpl_amd.smpl import SMPL                 # smplx's model_path convention: a directory holding SMPL_<GENDER>.pkl
    layer = SMPL(str(tmp_path), joint_regressor_extra=m['J_regressor_extra'])
    ref = SMPL(m)
    for (ka, a), (kb, b) in zip(layer.named_buffers(), ref.named_buffers()):
        assert ka == kb and torch.equal(a, b), ka


def test_trainer_save_and_resume(tmp_path):
    """Trainer.save / Trainer.resume: parameters, optimizer state and the step count that drives the LR decay survive a
    round trip through a reference-format checkpoint (utils/saver.py, base_trainer.py:37-51)."""
    from danet_densepose2smpl_amd.config import reset_cfg, cfg_from_dict
    from danet_densepose2smpl_amd.trainer import Trainer, default_options
    reset_cfg()
    cfg_from_dict({'DANET.INIMG_SIZE': 64, 'DANET.HEATMAP_SIZE': 16})
    torch.manual_seed(1)
    a = Trainer(default_options(2), device=torch.device('cpu'), distributed=False, lr=3e-4)
    a.step_count = 1234
    for g in a.optimizer.param_groups:
        g['lr'] = 1.5e-4 if not torch.is_tensor(g['lr']) else g['lr'].fill_(1.5e-4)
    path = a.save(str(tmp_path / 'ck' / '00001234.pt'), epoch=2, batch_idx=17)
    torch.manual_seed(2)
    b = Trainer(default_options(2), device=torch.device('cpu'), distributed=False, lr=3e-4)
    pa, pb = dict(a.model.named_parameters()), dict(b.model.named_parameters())
    k0 = next(k for k in pa if pa[k].dim() == 4)
    assert not torch.equal(pa[k0], pb[k0])
    book = b.resume(path)
    assert b.step_count == 1234 and book['epoch'] == 2 and book['batch_idx'] == 17
    assert all(torch.equal(pa[k], pb[k]) for k in pa)
    assert abs(float(b.optimizer.param_groups[0]['lr']) - 1.5e-4) < 1e-12


def test_bench_cpu_baseline_leg_runs_on_host_cores():
    """bench.py's `cpu_baseline` (the oracle timed on the host; the only place outside tests / smoke that may use it)
    produces the fields the bench line carries."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = bench.cpu_baseline(64, 2)                  # (B >= 2: train-mode BatchNorm on the regressor's 1x1 feature maps)
    assert out['kind'] == 'port' and out['unit'] == 'images/sec' and out['value'] > 0 and out['cores'] >= 1 and 'sample' in out
    assert 'StepNets' in out['sample'] and '99.9' in out['sample']
    assert bench.pmc_traffic('no_such_kernel')[0] is None
