"""IUV_Renderer with the reference's call signature (/root/reference/utils/renderer.py:202-298).
The rasterisation (neural_renderer in the reference) is the HIP kernel of csrc/iuv_raster.hip."""
import numpy as np
import torch

from . import assets, ops


class IUV_Renderer(object):
    def __init__(self, orig_size=224, out_size=56, focal_length=5000., densepose=None, smpl_model=None):
        """densepose: dict with the UV_Processed.mat fields (assets.load_densepose_mat), or None for
        the seeded synthetic topology."""
        self.orig_size = orig_size
        self.out_size = out_size
        self.focal_length = focal_length
        K = np.array([[focal_length, 0., orig_size / 2.], [0., focal_length, orig_size / 2.], [0., 0., 1.]])
        if orig_size != 224:                     # renderer.py:219-224: all four entries are scaled
            sc = orig_size / float(224)
            K[0, 0] *= sc; K[1, 1] *= sc; K[0, 2] *= sc; K[1, 2] *= sc
        self.K = torch.FloatTensor(K[None, :, :])
        self.R = torch.FloatTensor(np.eye(3)[None, :, :])
        self.t = torch.FloatTensor(np.array([0, 0, 5])[None, None, :])
        if densepose is None:
            densepose = assets.make_synthetic_densepose(smpl_model, 0)
        vm, faces, tex = assets.densepose_render_tables(densepose)
        self.vert_mapping = torch.from_numpy(vm.astype(np.int64))
        self.faces = torch.from_numpy(faces[None, :, :])
        self.textures = torch.from_numpy(tex[None, :, None, None, None, :])
        self._tables = {}
        self._np = (vm, faces, tex)

    def _dev(self, device):
        key = str(device)
        if key not in self._tables:
            vm, faces, tex = self._np
            self._tables[key] = (torch.from_numpy(vm).to(device), torch.from_numpy(faces).contiguous().to(device),
                                 torch.from_numpy(tex).contiguous().to(device))
        return self._tables[key]

    def verts2uvimg(self, verts, cam, return_aux=False):
        """verts [B,6890,3], cam [B,3] (s,x,y) -> IUV image [B,3,out,out]."""
        vm, faces, tex = self._dev(verts.device)
        return ops.iuv_raster(verts, cam, vm, faces, tex, self.focal_length, self.orig_size, self.out_size,
                              return_aux=return_aux)

    def camera_matrix(self, cam):
        """renderer.py:280-298."""
        B = cam.size(0)
        K = self.K.repeat(B, 1, 1).to(cam.device)
        R = self.R.repeat(B, 1, 1).to(cam.device)
        t = torch.stack([cam[:, 1], cam[:, 2], 2 * self.focal_length / (self.orig_size * cam[:, 0] + 1e-9)], dim=-1)
        return K, R, t.unsqueeze(1)


class PartRenderer(object):
    """Silhouette mask and body-part segmentation of a posed SMPL mesh, the call signature of
    /root/reference/utils/part_utils.py:8-53 (used by the LSP mask / part evaluation, eval.py:218-246).

    The reference renders per-face colours with neural_renderer (fill_back on, no anti-aliasing, ambient light only) and
    looks the part id of a pixel up as cube_parts[floor(100 * rgb)].  Here the HIP rasteriser (csrc/iuv_raster.hip, the
    kernel behind IUV_Renderer) draws the same flat per-face colours -- both windings of every face, which is what
    fill_back means -- and the lookup is one gather.  `faces` [F,3] int (the SMPL topology), `textures` the
    VERTEX_TEXTURE_FILE array ([1,F,t,t,t,3] or [F,3]; flat per face), `cube_parts` the CUBE_PARTS_FILE table."""

    def __init__(self, faces, textures, cube_parts, focal_length=5000., render_res=224):
        self.focal_length = focal_length
        self.render_res = render_res
        faces = np.asarray(faces).astype(np.int32).reshape(-1, 3)
        tex = np.asarray(textures, dtype=np.float32)
        tex = tex.reshape(tex.shape[-5], -1, 3)[:, 0, :] if tex.ndim >= 5 else tex.reshape(-1, 3)
        if tex.shape[0] != faces.shape[0]:
            raise ValueError('textures: %d faces, topology: %d' % (tex.shape[0], faces.shape[0]))
        self.faces = torch.from_numpy(faces)
        self.textures = torch.from_numpy(tex)
        self.cube_parts = torch.as_tensor(np.asarray(cube_parts), dtype=torch.float32)
        # fill_back: every face also with the opposite winding (the rasteriser culls faces of non-positive signed area)
        self._np = (np.concatenate([faces, faces[:, ::-1]], 0).astype(np.int32).copy(), np.concatenate([tex, tex], 0).copy())
        self._tables = {}

    def _dev(self, device, nv):
        key = (str(device), nv)
        if key not in self._tables:
            f2, t2 = self._np
            self._tables[key] = (torch.arange(nv, dtype=torch.int32, device=device), torch.from_numpy(f2).contiguous().to(device),
                                 torch.from_numpy(t2).contiguous().to(device), self.cube_parts.to(device))
        return self._tables[key]

    def get_parts(self, parts, mask):
        """part_utils.py:27-35: rendered colours [B,3,H,W] + mask [B,H,W] -> part indices [B,H,W] (long)."""
        bn, c, h, w = parts.shape
        cube = self.cube_parts.to(parts.device)
        idx = torch.floor(100 * parts.permute(0, 2, 3, 1).contiguous().view(-1, 3)).long()
        out = cube[idx[:, 0], idx[:, 1], idx[:, 2], None] * mask.reshape(-1, 1)
        return out.view(bn, h, w).long()

    def __call__(self, vertices, camera):
        """vertices [B,V,3], camera [B,3] (s, tx, ty) -> (mask [B,H,W] float, parts [B,H,W] long)."""
        vm, faces, tex, _ = self._dev(vertices.device, vertices.shape[1])
        rgb, fidx, _ = ops.iuv_raster(vertices, camera, vm, faces, tex, self.focal_length, self.render_res, self.render_res,
                                      return_aux=True)
        mask = (fidx >= 0).to(torch.float32)
        return mask, self.get_parts(rgb, mask)
