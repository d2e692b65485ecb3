"""SURVEY 8 row f2 ON THE DEVICE: a checkpoint FILE in the reference's layout -> the HIP DaNet -> the reference's outputs.

The file is what /root/reference/utils/saver.py:24-70 writes and /root/reference/demo.py:92-97 reads: {'model': state dict,
'optimizer', 'epoch', 'batch_idx', 'batch_size', 'dataset_perm', 'total_step_count'} with the keys of SURVEY Appendix F
(`img2iuv.*`, `iuv2smpl.smpl_para_Outs.*`), DataParallel's `module.` prefix, and WITHOUT the `iuv2smpl.smpl.*` tables (saver.py:32-34).
Its parameters are the closed-form `formula_tensor` of each key -- the same values the reference itself was run with when
tests/golden/make_golden.py produced g7 (IUV_Estimator, train mode) and g17 (the whole inference path danet.py:61-131 in eval mode),
so what comes out of `checkpoint.load_pretrained` + `DaNet.infer_net` is compared with the REFERENCE's own results.  Tolerances are
the fp32 mode's (conv.precision('fp32'), the reference's arithmetic type), set from the measured errors (gpurun_out/parity_measured.jsonl
-> profiles/r06_parity_measured.jsonl: maps 1.1e-6 .. 1.5e-6 of scale, STN centres 1.8e-7, para 1.2e-7, no arg-max flip): 2e-5 of scale on
maps, 2e-6 on the centres, 5e-6 abs on para; the bf16 production path from the same file: para within 5e-3 (measured 2.5e-4)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import golden, GOLDEN, record
sys.path.insert(0, GOLDEN)
from make_golden import formula_tensor, formula_input    # noqa: E402

pytestmark = pytest.mark.gpu

GRAPH_BUFFERS = ('I_n', 'A_link', 'A_mask', 'A', 'r2p_A', 'p2r_A')


def _cfg(**kw):
    from danet_densepose2smpl_amd.config import reset_cfg, cfg_from_dict
    reset_cfg()
    cfg_from_dict(kw)


def _rel(a, ref):
    ref = np.asarray(ref, np.float32)
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    return float(np.abs(a - ref).max() / (np.abs(ref).max() + 1e-6))


def _reference_layout_checkpoint(path, shapes, g):
    """{'model': {...}} as utils/saver.py writes it: formula parameters keyed relative to IUV_Estimator / DecomposedPredictor (how the
    reference modules were filled in make_golden.py), stored under their DaNet paths with the DataParallel prefix."""
    sd = {}
    for k, (shape, dtype) in shapes.items():
        if k.startswith('iuv2smpl.smpl.'):
            continue                                                          # saver.py:32-34 strips the SMPL tables
        if k.startswith('img2iuv.'):
            rel = k[len('img2iuv.'):]
            if rel in ('learned_ratio', 'learned_offset'):
                t = torch.from_numpy(g[rel])
            else:
                t = formula_tensor(rel, shape, dtype) if dtype.is_floating_point else torch.zeros(shape, dtype=dtype)
        elif k.startswith('iuv2smpl.smpl_para_Outs.'):
            rel = k[len('iuv2smpl.smpl_para_Outs.'):]
            if rel in GRAPH_BUFFERS:
                continue                                                      # constants of the architecture (golden g3 pins them)
            if rel == 'mean_cam_shape':
                t = torch.tensor([[0.9, 0., 0.] + [0.] * 10])
            elif rel == 'mean_pose':
                t = torch.tensor([1., 0., 0., 1., 0., 0.]).repeat(24).unsqueeze(0)
            else:
                t = formula_tensor(rel, shape, dtype) if dtype.is_floating_point else torch.zeros(shape, dtype=dtype)
        else:
            raise AssertionError('state-dict key outside SURVEY Appendix F: ' + k)
        assert tuple(t.shape) == tuple(shape), k
        sd['module.' + k] = t
    ckpt = {'model': sd, 'optimizer': {'state': {}, 'param_groups': []}, 'epoch': 3, 'batch_idx': 11, 'batch_size': 2,
            'dataset_perm': None, 'total_step_count': 1234}
    torch.save(ckpt, path)
    return len(sd)


@pytest.fixture(scope='module')
def loaded(tmp_path_factory):
    _cfg(**{'DANET.INIMG_SIZE': 128, 'DANET.HEATMAP_SIZE': 32, 'DANET.STN_CENTER_JITTER': 0., 'DANET.STN_SCALE_JITTER': 0.,
            'DANET.PARTDROP_RATE': 0.})
    from danet_densepose2smpl_amd import checkpoint
    from danet_densepose2smpl_amd.danet import DaNet
    from danet_densepose2smpl_amd.trainer import default_options
    g = golden('g17_infer')
    torch.manual_seed(123)                                                    # a DIFFERENT random initialisation: everything must come from the file
    model = DaNet(default_options(2), None, pretrained=False)
    shapes = {k: (tuple(v.shape), v.dtype) for k, v in model.state_dict().items()}
    path = str(tmp_path_factory.mktemp('f2') / 'danet_model_formula.pt')
    n = _reference_layout_checkpoint(path, shapes, g)
    missing, unexpected = checkpoint.load_pretrained(model, path)
    assert unexpected == []
    assert all(k.startswith('iuv2smpl.smpl.') or k.split('.')[-1] in GRAPH_BUFFERS or k.startswith('_') for k in missing), missing[:5]
    assert n > 2400
    return model.cuda(), g, path


def test_reference_layout_checkpoint_drives_infer_net_to_the_reference_result(loaded):
    """demo.py:92-151: load_state_dict(checkpoint['model'], strict=False) -> model.eval() -> infer_net(image) -> para."""
    from danet_densepose2smpl_amd import conv
    model, g, _ = loaded
    model.eval()
    img = formula_input('g17.img', (2, 3, 128, 128), -2.0, 2.0).cuda()
    with conv.precision('fp32'):
        rd = model.infer_net(img)
        uv = model.img2iuv(img)
    errs = {}
    for a, k in zip(uv['uvia_pred'], ('u_raw', None, 'index_raw', 'ann_raw')):
        if k:
            errs[k] = _rel(a, g[k])
            assert errs[k] < 2e-5, (k, errs[k])
    errs['stn_kps_pred'] = float(np.abs(uv['stn_kps_pred'].cpu().numpy() - g['stn_kps_pred']).max())
    assert errs['stn_kps_pred'] < 2e-6
    errs['part_iuv_pred'] = _rel(uv['part_iuv_pred'][:, ::6], g['part_iuv_pred'])
    assert errs['part_iuv_pred'] < 2e-5
    # the cleaned part-index plane (integer work, danet.py:84 -> utils/iuvmap.py:6-38): equal wherever the reference's own top-2
    # logits are further apart than the raw maps' measured error (elsewhere the arg-max is not defined at fp32 resolution)
    idx = rd['visualization']['iuv_pred'][2].argmax(1).cpu().numpy().astype(np.uint8)
    ref_raw = torch.from_numpy(g['index_raw'])
    top = ref_raw.topk(2, dim=1).values
    decided = ((top[:, 0] - top[:, 1]) > 4 * errs['index_raw'] * float(ref_raw.abs().max())).numpy()
    assert decided.mean() > 0.98
    assert (idx == g['index_clean'])[decided].all()
    errs['index_flips'] = int((idx != g['index_clean']).sum())
    para = rd['para']
    assert para.shape == (2, 229)
    errs['para'] = float(np.abs(para.cpu().numpy() - g['para']).max())
    assert errs['para'] < 5e-6, errs
    record('f2_infer_net_fp32_vs_reference', errs)
    # the bf16 production path from the same file: the regressor's output stays close (random-weight-net noise bound of test_gpu_models)
    rb = model.infer_net(img)['para']
    eb = float(np.abs(rb.cpu().numpy() - g['para']).max())
    record('f2_infer_net_bf16_vs_reference', {'para': eb})
    assert torch.isfinite(rb).all() and eb < 5e-3, eb
    rot = rb[:, 13:].reshape(-1, 3, 3)
    assert (torch.bmm(rot, rot.transpose(1, 2)) - torch.eye(3, device='cuda')).abs().max() < 1e-4


@pytest.mark.parametrize('align', [1])
def test_reference_layout_checkpoint_drives_the_estimator_to_golden_g7(loaded, align):
    """The same file, `img2iuv` in train mode on the g7 inputs (64 x 64): weights, BatchNorm affine parameters and the learned STN
    ratios all arrived where the reference keeps them."""
    from danet_densepose2smpl_amd import conv
    model, _, path = loaded
    g = golden('g7_estimator_align%d' % align)
    est = model.img2iuv
    with torch.no_grad():                                                     # g7 was produced with the yaml's learned ratios
        keep = est.learned_ratio.clone(), est.learned_offset.clone()
        est.learned_ratio.copy_(torch.from_numpy(g['learned_ratio']))
        est.learned_offset.copy_(torch.from_numpy(g['learned_offset']))
    t = lambda k: torch.from_numpy(g[k]).cuda()
    est.train()
    try:
        with conv.precision('fp32'), torch.no_grad():
            rd = est(t('img'), t('iuv_gt'), t('kps'), has_iuv=torch.ones(2, device='cuda'))
    finally:
        model.eval()
        with torch.no_grad():
            est.learned_ratio.copy_(keep[0]); est.learned_offset.copy_(keep[1])
        # train mode moved the running statistics: restore them from the file so the module-scoped model stays what the file says
        from danet_densepose2smpl_amd import checkpoint
        checkpoint.load_pretrained(model, path)
    meas = {k: _rel(a, g[k]) for a, k in zip(rd['uvia_pred'], ('u', 'v', 'index', 'ann'))}
    meas['part_iuv_pred'] = _rel(rd['part_iuv_pred'], g['part_iuv_pred'])
    record('f2_estimator_from_file_fp32_vs_g7', meas)
    assert max(meas.values()) < 1e-3, meas
    for k in g.files:
        if k.startswith('loss__'):
            ours, ref = float(rd['losses'][k[6:]].detach().sum()), float(g[k].sum())
            assert abs(ours - ref) <= 1e-4 * abs(ref) + 1e-6, (k, ours, ref)


def test_training_checkpoint_round_trip_through_the_device(loaded, tmp_path):
    """saver.py:24-70 in the other direction: Trainer-side save from the DEVICE model, reload into a fresh model, identical
    infer_net output (bit for bit) and bookkeeping."""
    from danet_densepose2smpl_amd import checkpoint
    from danet_densepose2smpl_amd.danet import DaNet
    from danet_densepose2smpl_amd.trainer import default_options
    model, _, _ = loaded
    model.eval()
    p = checkpoint.save_checkpoint(str(tmp_path / 'ck' / '00001234.pt'), {'model': model}, epoch=5, batch_idx=2, batch_size=2,
                                   dataset_perm=[0, 1], total_step_count=1234)
    raw = torch.load(p, weights_only=False)
    assert not any(k.startswith('iuv2smpl.smpl.') for k in raw['model'])
    torch.manual_seed(7)
    other = DaNet(default_options(2), None, pretrained=False)
    book = checkpoint.load_checkpoint(p, {'model': other})
    assert book['epoch'] == 5 and book['total_step_count'] == 1234
    other = other.cuda().eval()
    img = formula_input('g17.img', (2, 3, 128, 128), -2.0, 2.0).cuda()
    assert torch.equal(model.infer_net(img)['para'], other.infer_net(img)['para'])
