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
