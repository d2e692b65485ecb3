"""GPU parity of the HIP rasteriser (csrc/iuv_raster.hip) against the CPU oracle: the integer
part-id plane must be BIT-EXACT (BASELINE.json north_star); we require the whole image,
face-index map and depth map to be identical."""
import numpy as np
import pytest
import torch

import oracle
from conftest import rand_pose_shape

pytestmark = pytest.mark.gpu


def _scene(smpl_model, B, seed, orig=256.0, S=64):
    betas, pose = rand_pose_shape(B, seed, pose_sigma=0.35)
    verts, _ = oracle.lbs_forward(smpl_model, betas, pose, False, np.float32)
    rng = np.random.default_rng(seed)
    cam = np.stack([rng.uniform(0.6, 1.1, B), rng.uniform(-.1, .1, B), rng.uniform(-.1, .1, B)], 1).astype(np.float32)
    return verts.astype(np.float32), cam


@pytest.mark.parametrize('orig,S', [(256.0, 64), (224.0, 56), (224.0, 224)])
def test_raster_bit_exact_vs_oracle(smpl_model, dp_tables, orig, S):
    from danet_densepose2smpl_amd.renderer import IUV_Renderer
    vm, faces, tex = dp_tables
    rend = IUV_Renderer(orig_size=int(orig), out_size=S, smpl_model=smpl_model)
    B = 32 if S <= 64 else 4
    n_img = 0
    for seed in range(4 if S <= 64 else 1):
        verts, cam = _scene(smpl_model, B, 1000 + seed, orig, S)
        img, fidx, depth = rend.verts2uvimg(torch.from_numpy(verts).cuda(), torch.from_numpy(cam).cuda(), return_aux=True)
        ref_img, ref_f, ref_d = oracle.raster_forward(verts, cam, vm, faces, tex, 5000.0, orig, S)
        np.testing.assert_array_equal(fidx.cpu().numpy(), ref_f)
        np.testing.assert_array_equal(np.rint(img[:, 0].cpu().numpy() * 24), np.rint(ref_img[:, 0] * 24))
        np.testing.assert_array_equal(img.cpu().numpy(), ref_img)
        np.testing.assert_array_equal(depth.cpu().numpy(), ref_d)
        assert (ref_f >= 0).mean() > 0.03
        n_img += B
    assert n_img >= 4


def test_raster_many_random_cameras_exact(smpl_model, dp_tables):
    """>= 1k random mesh/camera pairs, part-id plane bit exact (SURVEY.md 8c vii)."""
    from danet_densepose2smpl_amd.renderer import IUV_Renderer
    vm, faces, tex = dp_tables
    rend = IUV_Renderer(orig_size=256, out_size=64, smpl_model=smpl_model)
    total = 0
    for seed in range(8):
        verts, cam = _scene(smpl_model, 128, 5000 + seed)
        rng = np.random.default_rng(seed)
        cam[:, 0] = rng.uniform(0.3, 2.5, 128)          # includes bodies larger than the frame
        cam[:, 1:] = rng.uniform(-0.6, 0.6, (128, 2))
        img = rend.verts2uvimg(torch.from_numpy(verts).cuda(), torch.from_numpy(cam).cuda())
        ref_img, _, _ = oracle.raster_forward(verts, cam, vm, faces, tex, 5000.0, 256.0, 64)
        np.testing.assert_array_equal(np.rint(img[:, 0].cpu().numpy() * 24), np.rint(ref_img[:, 0] * 24))
        total += 128
    assert total >= 1000


def test_raster_degenerate_inputs(smpl_model, dp_tables):
    from danet_densepose2smpl_amd.renderer import IUV_Renderer
    vm, faces, tex = dp_tables
    rend = IUV_Renderer(orig_size=256, out_size=64, smpl_model=smpl_model)
    verts, cam = _scene(smpl_model, 3, 1)
    cam[0, 0] = 0.0            # s = 0 -> t_z = 1e13: everything beyond far -> empty
    cam[1, 0] = -0.9           # camera behind
    verts[2] = 0.0             # all vertices coincide: zero-area faces
    img, fidx, _ = rend.verts2uvimg(torch.from_numpy(verts).cuda(), torch.from_numpy(cam).cuda(), return_aux=True)
    ref_img, ref_f, _ = oracle.raster_forward(verts, cam, vm, faces, tex, 5000.0, 256.0, 64)
    np.testing.assert_array_equal(fidx.cpu().numpy(), ref_f)
    np.testing.assert_array_equal(img.cpu().numpy(), ref_img)
    assert (ref_f[0] < 0).all() and (ref_f[2] < 0).all()


def test_part_renderer_mask_and_parts(smpl_model):
    """PartRenderer (part_utils.py:8-53) on the HIP rasteriser.  Both windings of the SMPL faces are drawn
    (fill_back), so every pixel the oracle covers with either winding is covered; the part index of a pixel is the
    cube_parts entry of its face's colour, and the mask is 1 exactly on covered pixels."""
    from danet_densepose2smpl_amd.renderer import PartRenderer
    rng = np.random.default_rng(3)
    faces = np.asarray(smpl_model['faces']).astype(np.int32)
    F = faces.shape[0]
    tex = (rng.integers(0, 100, (F, 3)).astype(np.float32) + 0.5) / 100.0          # colours in the middle of a cube cell
    cube = rng.integers(0, 7, (100, 100, 100)).astype(np.float32)
    rend = PartRenderer(faces, tex[None, :, None, None, None, :].repeat(2, 2).repeat(2, 3).repeat(2, 4), cube, render_res=224)
    B = 6
    verts, cam = _scene(smpl_model, B, 77, 224.0, 224)
    mask, parts = rend(torch.from_numpy(verts).cuda(), torch.from_numpy(cam).cuda())
    assert mask.shape == (B, 224, 224) and parts.shape == (B, 224, 224) and parts.dtype == torch.long
    f2 = np.concatenate([faces, faces[:, ::-1]], 0).astype(np.int32)
    t2 = np.concatenate([tex, tex], 0)
    vm = np.arange(verts.shape[1], dtype=np.int32)
    ref_img, ref_f, _ = oracle.raster_forward(verts, cam, vm, f2, t2, 5000.0, 224.0, 224)
    np.testing.assert_array_equal(mask.cpu().numpy() > 0, ref_f >= 0)
    idx = np.floor(100 * np.transpose(ref_img, (0, 2, 3, 1))).astype(np.int64)
    want = cube[idx[..., 0], idx[..., 1], idx[..., 2]] * (ref_f >= 0)
    np.testing.assert_array_equal(parts.cpu().numpy(), want.astype(np.int64))
    front, _, _ = [a for a in oracle.raster_forward(verts, cam, vm, faces, tex, 5000.0, 224.0, 224)]
    assert ((ref_f >= 0).sum() >= (np.abs(front).sum(1) > 0).sum()) and (ref_f >= 0).mean() > 0.03
