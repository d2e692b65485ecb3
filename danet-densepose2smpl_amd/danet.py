"""DaNet orchestration with the reference's interface (/root/reference/models/danet/danet.py):
``DaNet(options, smpl_mean_params, pretrained)``, ``forward(in_dict)`` (training) and
``infer_net(image)`` (inference); sub-modules ``img2iuv``, ``iuv2smpl`` (+ ``iuv2smpl.smpl``) and the
attribute ``iuv_renderer``.  Default configuration only (INPUT_MODE 'iuv', DECOMPOSED, 'gcn')."""
import torch
import torch.nn as nn

from .config import cfg
from .geometry import batch_rodrigues
from .iuv_estimator import IUV_Estimator, DP2SMPL_MAPPING
from .iuvmap import iuvmap_clean
from . import part_ops
from . import segments

FUSED_PART_OPS = True       # part drop + per-part iuvmap_clean in one HIP kernel (False: the tensor-op formulation)
from .renderer import IUV_Renderer
from .smpl_regressor import SMPL_Regressor


class DaNet(nn.Module):
    def __init__(self, options, smpl_mean_params=None, pretrained=True, smpl_model=None, densepose=None):
        super().__init__()
        self.options = options
        self.img2iuv = IUV_Estimator(pretrained)
        final_feat_dim = getattr(self.img2iuv.iuv_est, 'final_feat_dim', None)
        self.iuv2smpl = SMPL_Regressor(options, cfg.DANET.INIMG_SIZE, final_feat_dim, smpl_mean_params, pretrained,
                                       smpl_model=smpl_model)
        self.iuv_renderer = IUV_Renderer(cfg.DANET.INIMG_SIZE, cfg.DANET.HEATMAP_SIZE, densepose=densepose,
                                         smpl_model=smpl_model)
        # part-drop bookkeeping (danet.py:251-274): partial channel k (1..6) of joint i shows DensePose part
        # DP2SMPL_MAPPING[i][k-1]; channel 0 (background) is never dropped
        sel = torch.zeros(24, 7, dtype=torch.long)
        sel[:, 1:] = torch.tensor(DP2SMPL_MAPPING, dtype=torch.long)
        self.register_buffer('_partial_src', sel, persistent=False)

    # ------------------------------------------------------------------------------------------
    def _clean_partial(self, part_iuv_pred):
        """iuvmap_clean on each of the 24 partial maps (danet.py:276-283), vectorised."""
        B, J, _, K, H, W = part_iuv_pred.shape
        flat = part_iuv_pred.reshape(B * J, 3, K, H, W)
        u, v, i, _ = iuvmap_clean(flat[:, 0], flat[:, 1], flat[:, 2])
        return torch.stack([u, v, i], dim=1).reshape(B, J, 3, K, H, W)

    def infer_net(self, image):
        """danet.py:61-131: image [B,3,H,W] -> {'para': [B,229], 'visualization': {...}}."""
        if self.training:
            raise ValueError('You should call this function only on inference.'
                             'Set the network in inference mode by net.eval().')
        with torch.no_grad():
            rd = {'visualization': {}}
            uv = self.img2iuv(image)
            u, v, idx, ann = iuvmap_clean(*uv['uvia_pred'])
            rd['visualization']['iuv_pred'] = [u, v, idx, ann]
            part_iuv_map = None
            if 'part_iuv_pred' in uv:
                rd['visualization']['part_iuv_pred'] = uv['part_iuv_pred']
                part_iuv_map = self._clean_partial(uv['part_iuv_pred'])
            iuv_map = torch.cat([u, v, idx], dim=1)
            out = self.iuv2smpl.smpl_infer_net({'iuv_map': iuv_map, 'part_iuv_map': part_iuv_map})
            rd['para'] = out['para']
            rd['visualization'].update(out['visualization'])
            return rd

    def forward(self, in_dict):
        with torch.set_grad_enabled(self.training):
            return self._forward(in_dict)

    def _forward(self, in_dict):
        if not isinstance(in_dict, dict) or 'opt_pose' not in in_dict:
            # danet.py:142-167,191 cannot run without labels either (uv_image_gt is undefined there)
            raise ValueError('DaNet.forward needs the training in_dict; use infer_net(image) for inference')
        image = in_dict['img']
        gt_pose, gt_betas = in_dict['opt_pose'], in_dict['opt_betas']
        target_kps, target_kps3d = in_dict.get('keypoints'), in_dict.get('pose_3d')
        has_iuv = in_dict['has_iuv'].reshape(-1) if 'has_iuv' in in_dict else None
        has_dp = in_dict.get('has_dp')
        has_kp3d = in_dict.get('has_pose_3d')
        target_smpl_kps = in_dict.get('target_smpl_kps')
        target_verts = in_dict.get('target_verts')
        valid_fit = in_dict.get('valid_fit')
        B = image.shape[0]
        D = cfg.DANET

        gt_rotmat = batch_rodrigues(gt_pose.reshape(-1, 3)).reshape(-1, 24 * 9)
        target_cam = in_dict['target_cam']
        target = torch.cat([target_cam, gt_betas, gt_rotmat], dim=1)
        # render all B label meshes and mask (static shapes; danet.py:163-165 renders the has_iuv subset)
        uv_image_gt = self.iuv_renderer.verts2uvimg(target_verts.detach(), target_cam.detach())
        if has_iuv is not None:
            uv_image_gt = uv_image_gt * (has_iuv > 0).to(uv_image_gt.dtype).view(B, 1, 1, 1)

        rd = {'losses': {}, 'metrics': {}, 'visualization': {}, 'prediction': {}}
        keep = keep25 = None
        if self.training and D.PARTDROP_RATE > 0:                            # danet.py:194-203
            keep = (torch.rand(B, 24, device=image.device) >= D.PARTDROP_RATE).to(torch.float32)
            keep25 = torch.cat([torch.ones(B, 1, device=image.device), keep], dim=1)
        from . import conv as _conv
        pk = None
        if keep is not None:                                                  # danet.py:264-274
            pk = keep25[:, self._partial_src]                                # [B,24,7]
        # (the cleaned partial maps are made together with the partial losses when the regressor will run: part_ops.part_joint)
        want_x24 = FUSED_PART_OPS and image.is_cuda and _conv.PRECISION != 'fp32' and not in_dict.get('pretrain_mode', False)
        uv = self.img2iuv(image, uv_image_gt, target_smpl_kps, uvia_dp_gt=in_dict.get('dp_dict'), has_iuv=has_iuv, has_dp=has_dp,
                          keep25=keep25, part_clean=(pk,) if want_x24 else None)
        u_pred, v_pred, index_pred, ann_pred = uv['uvia_pred']
        segments.note_losses(uv.get('losses', {}).keys())            # the estimator's losses hang off the segment open now

        if 'iuv_map' in uv:
            # the fused global-IUV op (csrc/iuv_ops.hip) already dropped, cleaned and concatenated: [U | V | I | 5 zeros] bf16
            iuv_map = uv['iuv_map']
            vis = [iuv_map[:, :25].detach(), iuv_map[:, 25:50].detach(), iuv_map[:, 50:75].detach(), None]
            if in_dict.get('vis_on', False):
                vis[3] = iuvmap_clean(u_pred, v_pred, index_pred, ann_pred)[3].detach()
            rd['visualization']['iuv_pred'] = vis
        else:
            if keep25 is not None:
                k4 = keep25.view(B, 25, 1, 1)
                u_pred, v_pred, index_pred = u_pred * k4, v_pred * k4, index_pred * k4
            u_cl, v_cl, i_cl, a_cl = iuvmap_clean(u_pred, v_pred, index_pred, ann_pred)
            rd['visualization']['iuv_pred'] = [u_cl.detach(), v_cl.detach(), i_cl.detach(), a_cl.detach()]
            iuv_map = torch.cat([u_cl, v_cl, i_cl], dim=1)
        if in_dict.get('vis_on', False):
            rd['visualization']['gt_uv'] = uv_image_gt
            if 'stn_kps_pred' in uv:
                rd['visualization']['stn_kps_pred'] = uv['stn_kps_pred']

        smpl_rd = None
        if not in_dict.get('pretrain_mode', False):
            part_pred = uv['part_iuv_pred']
            # the IUV -> SMPL regressor is a backward-pass segment of its own (segments.py: its gradient buckets are on the wire
            # while the estimator's and the backbone's backward run); identity unless a data-parallel trainer asked for cuts
            if FUSED_PART_OPS and part_pred.is_cuda and _conv.PRECISION != 'fp32':
                x24 = uv.get('part_x24')                                      # made beside the partial losses (one autograd node)
                if x24 is None:
                    _, x24 = part_ops.part_clean(part_pred, pk)               # one kernel; bf16, channels 21..23 zero
                iuv_map, x24 = segments.cut([iuv_map, x24])
                part_iuv_map = part_ops.padded_part_view(x24)                 # the [B,24,3,7,H,W] view of the padded operand
                part_iuv_map._nhwc_padded = x24
            else:
                if pk is not None:
                    part_pred = part_pred * pk.view(B, 24, 1, 7, 1, 1)
                iuv_map, part_iuv_map = segments.cut([iuv_map, self._clean_partial(part_pred)])
            rd['visualization']['part_iuv_pred'] = part_iuv_map
            smpl_rd = self.iuv2smpl({'iuv_map': iuv_map, 'part_iuv_map': part_iuv_map, 'target': target,
                                     'target_kps': target_kps, 'target_verts': target_verts, 'target_kps3d': target_kps3d,
                                     'has_kp3d': has_kp3d, 'has_smpl': valid_fit})
        for key in ('losses', 'metrics', 'visualization', 'prediction'):
            if key in uv:
                rd[key].update(uv[key])
            if smpl_rd is not None:
                rd[key].update(smpl_rd[key])
        for key in ('losses', 'metrics'):
            for k, v in rd[key].items():
                if v.dim() == 0:
                    rd[key][k] = v.unsqueeze(0)
        return rd
