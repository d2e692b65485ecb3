"""IUV estimator: backbone + global IUV heads + joint-centric part decomposition + IUV losses.

Mirrors /root/reference/models/danet/iuv_estimator.py (IUV_Estimator :17-301, body_uv_losses
:304-341, part_iuv_simp :422-445) for the default path (INPUT_MODE 'iuv', DECOMPOSED).  The 24
affine resamplings of the feature map are one HIP launch (nn.stn_gather); all batch filtering by
has_iuv is expressed as per-sample weights, so there is no boolean-mask indexing, no
torch.unique and no host synchronisation inside the step (hipGraph-capturable).
"""
import os
import pickle

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .config import cfg
from . import part_ops
from .geometry import softmax_integral_tensor
from .hrnet import PoseHighResolutionNet
from .iuvmap import iuv_img2map, iuvmap_clean
from .nn import stn_gather
from .resnet import PoseResNet

# kinematic tables of /root/reference/utils/smpl_utlis.py:13-17,30-79
SMPL_PARENTS = [0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
SMPL_CHILDREN = [3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 10, 11, 15, 16, 17, 15, 18, 19, 20, 21, 22, 23, 22, 23]
SMPL2DP_PART = [[1, 2], [8, 10], [7, 9], [1, 2], [8, 10, 12, 14], [7, 9, 11, 13], [1, 2], [12, 14, 5], [11, 13, 6],
                [1, 2], [12, 14, 5], [11, 13, 6], [1, 2, 23, 24], [15, 17], [16, 18], [23, 24], [15, 17], [16, 18],
                [15, 17, 19, 21], [16, 18, 20, 22], [19, 21, 4], [20, 22, 3], [19, 21, 4], [20, 22, 3]]
FUSED_PART_LOSSES = True    # partial-IUV losses through csrc/part_ops.hip (False: the tensor-op formulation)
FUSED_GLOBAL_IUV = True     # global IUV losses / clean / soft-argmax through csrc/iuv_ops.hip (False: the tensor-op formulation)

DP2SMPL_MAPPING = [[7, 8, 9, 10, 1, 2], [1, 2, 8, 10, 12, 14], [1, 2, 7, 9, 11, 13], [7, 8, 9, 10, 1, 2],
                   [1, 2, 8, 10, 12, 14], [1, 2, 7, 9, 11, 13], [7, 8, 9, 10, 1, 2], [8, 10, 12, 14, 5, 5],
                   [7, 9, 11, 13, 6, 6], [7, 8, 9, 10, 1, 2], [8, 10, 12, 14, 5, 5], [7, 9, 11, 13, 6, 6],
                   [1, 2, 23, 24, 23, 24], [1, 2, 15, 17, 19, 21], [1, 2, 16, 18, 20, 22], [1, 2, 23, 24, 23, 24],
                   [1, 2, 15, 17, 19, 21], [1, 2, 16, 18, 20, 22], [1, 2, 15, 17, 19, 21], [1, 2, 16, 18, 20, 22],
                   [15, 17, 19, 21, 4, 4], [16, 18, 20, 22, 3, 3], [15, 17, 19, 21, 4, 4], [16, 18, 20, 22, 3, 3]]

_LEARNED_RATIO_PATHS = ('data/pretrained_model/learned_ratio.pkl',)


def load_learned_ratio(path=None):
    """The 24 per-joint ratio/offset constants (iuv_estimator.py:21-31).  The reference ships them as
    data/pretrained_model/learned_ratio.pkl; without the file a neutral (1.0, 0.1) pair is used."""
    for p in ([path] if path else []) + list(_LEARNED_RATIO_PATHS):
        if p and os.path.isfile(p):
            with open(p, 'rb') as f:
                d = pickle.load(f, encoding='iso-8859-1')
            return np.asarray(d['ratio'], np.float32), np.asarray(d['offset'], np.float32)
    return np.ones(24, np.float32), 0.1 * np.ones(24, np.float32)


FUSED_STN_THETA = True       # visibility score + affine_para as one HIP launch (False: the tensor-op formulation)
PART_JOINT = bool(int(os.environ.get('DANET_PART_JOINT', '1')))            # the partial losses and the regressor's cleaned operand as one autograd node (A-B knob)
LOSS_FINALIZE = bool(int(os.environ.get('DANET_LOSS_FINALIZE', '1')))     # the estimator's losses leave their fused ops finished (glue.loss_finalize); 0: tensor-op scaling (A-B)


def _sample_points(maps, pts, align):
    """Bilinear sample of maps [B,J,H,W] at one point per (b,j): pts [B,J,2] in [-1,1] (x,y); zero
    padding -- the single-point grid_sample of iuv_estimator.py:180."""
    B, J, H, W = maps.shape
    x, y = pts[..., 0], pts[..., 1]
    if align:
        ix, iy = (x + 1) * 0.5 * (W - 1), (y + 1) * 0.5 * (H - 1)
    else:
        ix, iy = ((x + 1) * W - 1) * 0.5, ((y + 1) * H - 1) * 0.5
    x0, y0 = torch.floor(ix), torch.floor(iy)
    out = torch.zeros(B, J, device=maps.device, dtype=torch.float32)
    flat = maps.reshape(B, J, H * W).float()
    for dy in (0, 1):
        for dx in (0, 1):
            xx, yy = x0 + dx, y0 + dy
            w = (1 - (ix - xx).abs()) * (1 - (iy - yy).abs())
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).long()
            out = out + torch.where(ok, w, torch.zeros_like(w)) * flat.gather(2, idx.unsqueeze(-1)).squeeze(-1)
    return out


class IUV_Estimator(nn.Module):
    def __init__(self, pretrained=True, learned_ratio_path=None):
        super().__init__()
        if cfg.DANET.INPUT_MODE != 'iuv':
            raise NotImplementedError("only DANET.INPUT_MODE == 'iuv' (the default) is on the hot path")
        ratio, offset = load_learned_ratio(learned_ratio_path) if cfg.DANET.USE_LEARNED_RATIO else (None, None)
        if ratio is not None:
            self.register_buffer('learned_ratio', torch.from_numpy(ratio))
            self.register_buffer('learned_offset', torch.from_numpy(offset))
        else:
            self.learned_ratio = nn.Parameter(cfg.DANET.PART_UVI_SCALE * torch.ones(24))
            self.learned_offset = nn.Parameter(cfg.DANET.PART_UVI_LR_OFFSET * torch.ones(24))
        self.smpl_parents = [SMPL_PARENTS, None]
        self.smpl_children = [None, SMPL_CHILDREN]
        self.smpl2dp_part = SMPL2DP_PART
        self.dp2smpl_mapping = DP2SMPL_MAPPING
        part_out_dim = 1 + len(DP2SMPL_MAPPING[0])
        if cfg.DANET.IUV_REGRESSOR == 'resnet':
            self.iuv_est = PoseResNet(part_out_dim=part_out_dim)
            if pretrained:
                self.iuv_est.init_weights(cfg.MSRES_MODEL.get('PRETRAINED', ''))
        elif cfg.DANET.IUV_REGRESSOR == 'hrnet':
            self.iuv_est = PoseHighResolutionNet(part_out_dim=part_out_dim)
            if pretrained:
                key = {'imagenet': 'PRETRAINED_IM', 'coco': 'PRETRAINED_COCO'}.get(cfg.HR_MODEL.PRETR_SET)
                self.iuv_est.init_weights(cfg.HR_MODEL.get(key, '') if key else '')
        else:
            raise ValueError('unknown DANET.IUV_REGRESSOR %r' % cfg.DANET.IUV_REGRESSOR)
        vis = torch.zeros(24, 25)
        for i, parts in enumerate(SMPL2DP_PART):
            vis[i, parts] = 1
        self.register_buffer('_vis_membership', vis, persistent=False)
        self.register_buffer('_dp_sel', torch.tensor(DP2SMPL_MAPPING, dtype=torch.long), persistent=False)
        self.register_buffer('_parent_idx', torch.tensor(SMPL_PARENTS, dtype=torch.long), persistent=False)
        self.register_buffer('_child_idx', torch.tensor(SMPL_CHILDREN, dtype=torch.long), persistent=False)
        self.bodyfeat_channels = 1024
        self.part_channels = 256

    # --------------------------------------------------------------------------------------
    def affine_para(self, stn_centers, part_hidden=None):
        """iuv_estimator.py:262-301, vectorised over the 24 joints.
        Returns thetas [B,24,2,3] ([[s,0,cx],[0,s,cy]]) and scales [B,24]."""
        c = stn_centers
        box = c.max(dim=1)[0] - c.min(dim=1)[0]
        scale_box = box.max(dim=1)[0] / 2.
        scale_c = torch.norm(c[:, self._child_idx] - c, dim=2) / 2.
        scale_p = torch.norm(c[:, self._parent_idx] - c, dim=2) / 2.
        scale = 2 * torch.maximum(scale_c, scale_p)
        scale = torch.cat([scale_box.unsqueeze(1), scale[:, 1:]], dim=1).detach()
        scale = scale * F.relu(self.learned_ratio) + F.relu(self.learned_offset)
        jit = cfg.DANET.STN_SCALE_JITTER if self.training else 0
        if jit > 0:
            scale = scale * (1 + jit * (torch.rand_like(scale) - 0.5))
        if part_hidden is not None:
            hidden = part_hidden.clone()
            hidden[:, 0] = False
            scale = torch.where(hidden, 0.8 * scale_box.unsqueeze(1).expand_as(scale), scale)
        if jit > 0:
            scale = scale * (1 + jit * (torch.rand_like(scale) - 0.5))
        B = c.shape[0]
        theta = torch.zeros(B, 24, 2, 3, device=c.device, dtype=torch.float32)
        theta[:, :, 0, 0] = scale
        theta[:, :, 1, 1] = scale
        theta[:, :, :, 2] = c.detach()
        return theta, scale

    def stn_theta(self, stn_centers, am_raw, align):
        """affine_para (iuv_estimator.py:262-301) and the visibility test in front of it (:176-186) as ONE launch
        (csrc/glue.hip stn_theta_kernel): centres [B,24,2], arg-max index plane uint8 [B,H,W] -> thetas [B,24,2,3].
        No gradient, as in the reference (theta is detached before affine_grid, :197)."""
        from . import _lib
        from ._lib import ptr, check, stream
        c = stn_centers.detach().to(torch.float32).contiguous()
        B = c.shape[0]
        jit = float(cfg.DANET.STN_SCALE_JITTER) if self.training else 0.
        rnd = torch.rand(2, B, 24, device=c.device) if jit > 0 else None
        vis = float(cfg.DANET.STN_PART_VIS_SCORE)
        theta = torch.empty(B, 24, 2, 3, device=c.device, dtype=torch.float32)
        H, W = am_raw.shape[-2], am_raw.shape[-1]
        check(_lib.lib().danet_stn_theta_forward(
            ptr(c), ptr(am_raw) if vis > 0 else None, ptr(self._vis_membership), ptr(self.learned_ratio.detach()),
            ptr(self.learned_offset.detach()), ptr(rnd), ptr(self._child_idx), ptr(self._parent_idx), B, H, W, int(bool(align)),
            jit, vis, ptr(theta), stream()), 'danet_stn_theta_forward')
        return theta

    def part_iuv_simp(self, U, V, I):
        """Per-joint 7-channel (background + 6 DensePose parts) U,V,I maps (iuv_estimator.py:422-445):
        [B,25,H,W] x3 -> [B,24,3,7,H,W]."""
        sel = self._dp_sel                                   # [24,6]
        Us, Vs, Is = U[:, sel], V[:, sel], I[:, sel]         # [B,24,6,H,W]
        zeros = torch.zeros_like(Us[:, :, :1])
        bg = (Is.sum(dim=2, keepdim=True) < 0.5).to(Is.dtype)
        return torch.stack([torch.cat([zeros, Us], 2), torch.cat([zeros, Vs], 2), torch.cat([bg, Is], 2)], dim=2)

    @staticmethod
    def body_uv_losses(u_pred, v_pred, index_pred, ann_pred, uvia_list, has_iuv=None, class_dim=1):
        """iuv_estimator.py:304-341 with has_iuv as per-sample weights.  Tensors may carry extra
        leading 'joint' dims after the batch dim; `class_dim` is the channel (class) axis."""
        Umap, Vmap, Imap, Annmap = uvia_list
        B = u_pred.shape[0]
        dev = u_pred.device
        if has_iuv is None:
            w = torch.ones(B, device=dev)
        else:
            w = has_iuv.to(torch.float32)
        wshape = [B] + [1] * (u_pred.dim() - 1)
        wb = w.view(wshape)
        fg = (Imap > 0).to(torch.float32) * wb
        loss_U = (F.smooth_l1_loss(u_pred.float(), Umap, reduction='none') * fg).sum() / B
        loss_V = (F.smooth_l1_loss(v_pred.float(), Vmap, reduction='none') * fg).sum() / B
        loss_U = loss_U * cfg.DANET.POINT_REGRESSION_WEIGHTS
        loss_V = loss_V * cfg.DANET.POINT_REGRESSION_WEIGHTS

        def ce(pred, target_map):
            logp = F.log_softmax(pred.float(), dim=class_dim)
            tgt = torch.argmax(target_map, dim=class_dim, keepdim=True)
            nll = -logp.gather(class_dim, tgt)                       # [B,...,1,H,W]
            per_sample = nll.numel() / B
            return (nll * wb).sum() / (w.sum().clamp(min=1.0) * per_sample)
        loss_IndexUV = ce(index_pred, Imap)
        loss_segAnn = None if ann_pred is None else ce(ann_pred, Annmap)
        return loss_U, loss_V, loss_IndexUV, loss_segAnn

    @staticmethod
    def dp_uvia_losses(u_pred, v_pred, index_pred, ann_pred, dp, has_dp=None, align=True):
        """DensePose-COCO point supervision (iuv_estimator.py:343-419, called on the has_dp subset at :106-117), in
        masked-weight form: every sample is evaluated and weighted by has_dp, so shapes stay static (hipGraph-capturable)
        and an all-zero has_dp gives the zeros the reference returns at :118-121.
        dp: the 9 blobs of datasets/base_dataset.py:228-232 -- X/Y/Ind/I points [B,196], U/V points and weights [B,25*196],
        ann labels/weights [B,S*S].  Returns (loss_Udp, loss_Vdp, loss_IndexUVdp, loss_segAnndp)."""
        B, K, S = u_pred.shape[0], cfg.DANET.NUM_PATCHES + 1, u_pred.shape[-1]
        dev = u_pred.device
        w = torch.ones(B, device=dev) if has_dp is None else (has_dp > 0).to(torch.float32)
        n_on = w.sum().clamp(min=1.0)
        # bilinear pooling of the predictions at the annotated points: pixel coordinates -> [-1, 1] as the reference does
        xy = torch.stack([dp['body_uv_X_points'], dp['body_uv_Y_points']], dim=2).to(torch.float32)
        grid = ((xy - S / 2.) * (2. / S)).unsqueeze(1)                                   # [B,1,196,2]

        def pool(t):
            return F.grid_sample(t.float(), grid, mode='bilinear', padding_mode='zeros', align_corners=align).squeeze(2).transpose(1, 2)
        iu, iv, ii = pool(u_pred), pool(v_pred), pool(index_pred)                         # [B,196,K]
        tu = dp['body_uv_U_points'].view(B, K, 196).transpose(1, 2).to(torch.float32)
        tv = dp['body_uv_V_points'].view(B, K, 196).transpose(1, 2).to(torch.float32)
        pw = dp['body_uv_point_weights'].view(B, K, 196).transpose(1, 2).to(torch.float32)
        wb = w.view(B, 1, 1)
        # utils/net.py:18-35 with inside = outside = point weights, N = 1: a plain sum
        loss_U = (pw * F.smooth_l1_loss(pw * iu, pw * tu, reduction='none') * wb).sum() * cfg.DANET.POINT_REGRESSION_WEIGHTS
        loss_V = (pw * F.smooth_l1_loss(pw * iv, pw * tv, reduction='none') * wb).sum() * cfg.DANET.POINT_REGRESSION_WEIGHTS
        # patch-index cross-entropy over ALL 196 point slots of the labelled samples (mean), empty slots carry label 0
        ce_i = F.cross_entropy(ii.reshape(B * 196, K), dp['body_uv_I_points'].reshape(-1).to(torch.int64), reduction='none')
        loss_I = (ce_i.view(B, 196) * w.view(B, 1)).sum() / (n_on * 196) * cfg.DANET.PART_WEIGHTS
        # dense 15-way body-part segmentation
        na = ann_pred.shape[1]
        ce_a = F.cross_entropy(ann_pred.float().reshape(B, na, S * S).transpose(1, 2).reshape(B * S * S, na),
                               dp['body_uv_ann_labels'].reshape(-1).to(torch.int64), reduction='none')
        loss_A = (ce_a.view(B, S * S) * w.view(B, 1)).sum() / (n_on * S * S) * cfg.DANET.INDEX_WEIGHTS
        return loss_U, loss_V, loss_I, loss_A

    # --------------------------------------------------------------------------------------
    def forward(self, data, iuv_image_gt=None, smpl_kps_gt=None, kps3d_gt=None, uvia_dp_gt=None, has_iuv=None, has_dp=None,
                keep25=None, part_clean=None):
        """keep25 [B,25] (optional): DaNet's part-drop mask; with it the fused global-IUV op also produces the cleaned,
        concatenated regressor input (rd['iuv_map'], rd['iuv_argmax']) in the same launch.  part_clean = (keep [B,24,7] or None,)
        (optional): the caller will feed the cleaned partial maps to the regressor -- they come back as rd['part_x24'], made by the same
        autograd node as the partial losses (part_ops.part_joint)."""
        rd = {'losses': {}, 'metrics': {}, 'visualization': {}}
        align = bool(cfg.DANET.get('ALIGN_CORNERS', True))
        est = self.iuv_est(data)
        u_pred, v_pred = est['predict_u'], est['predict_v']
        index_pred, ann_pred = est['predict_uv_index'], est['predict_ann_index']
        from . import conv as _conv
        fp32 = _conv.PRECISION == 'fp32'                     # verification mode: the tensor-op formulations throughout
        fused = FUSED_GLOBAL_IUV and u_pred.is_cuda and not fp32

        uvia_list = None
        am_raw = None
        if fused:
            from . import iuv_ops
            want = self.training and iuv_image_gt is not None
            w = None if has_iuv is None else has_iuv.to(torch.float32)
            B, S2 = u_pred.shape[0], u_pred.shape[-1] * u_pred.shape[-2]
            # the four losses leave the op finished (iuv_estimator.py:325-339: U / V sums x weight / batch; index / ann sums / (labelled samples
            # x pixels)): scales = (a, b) per loss, loss = sum * a / (max(sum w, 1) * b)
            scales = ((cfg.DANET.POINT_REGRESSION_WEIGHTS / B, 0.), (cfg.DANET.POINT_REGRESSION_WEIGHTS / B, 0.), (1., S2), (1., S2)) if (want and LOSS_FINALIZE) else None
            sums, rd['iuv_map'], am_raw = iuv_ops.iuv_global(u_pred, v_pred, index_pred, ann_pred, iuv_image_gt if want else None, w, keep25, scales=scales)
            rd['iuv_argmax'] = am_raw
            if want and scales is not None:
                rd['losses'].update({'loss_U': sums[0], 'loss_V': sums[1], 'loss_IndexUV': sums[2], 'loss_segAnn': sums[3]})
            elif want:
                wsum = float(B) if w is None else w.sum().clamp(min=1.0)
                rd['losses'].update({'loss_U': sums[0] * (cfg.DANET.POINT_REGRESSION_WEIGHTS / B), 'loss_V': sums[1] * (cfg.DANET.POINT_REGRESSION_WEIGHTS / B),
                                     'loss_IndexUV': sums[2] / (wsum * S2), 'loss_segAnn': sums[3] / (wsum * S2)})
        elif self.training and iuv_image_gt is not None:
            uvia_list = iuv_img2map(iuv_image_gt)
            lU, lV, lI, lA = self.body_uv_losses(u_pred, v_pred, index_pred, ann_pred, uvia_list, has_iuv)
            rd['losses'].update({'loss_U': lU, 'loss_V': lV, 'loss_IndexUV': lI, 'loss_segAnn': lA})
        if self.training and uvia_dp_gt is not None:
            # DensePose-COCO point supervision (iuv_estimator.py:106-121).  `dp_active` (host-side flag a data loader /
            # trainer sets per batch) skips the work for batches without DensePose labels, as the reference's
            # `torch.sum(has_dp) > 0` does, without a device sync inside the step.
            if uvia_dp_gt.get('dp_active', True):
                dp = {k: v for k, v in uvia_dp_gt.items() if torch.is_tensor(v)}
                lU, lV, lI, lA = self.dp_uvia_losses(u_pred, v_pred, index_pred, ann_pred, dp, has_dp, align)
                rd['losses'].update({'loss_Udp': lU, 'loss_Vdp': lV, 'loss_IndexUVdp': lI, 'loss_segAnndp': lA})
            else:
                z = torch.zeros(1, device=data.device)
                rd['losses'].update({'loss_Udp': z, 'loss_Vdp': z.clone(), 'loss_IndexUVdp': z.clone(), 'loss_segAnndp': z.clone()})
        rd['uvia_pred'] = [u_pred, v_pred, index_pred, ann_pred]
        if not cfg.DANET.DECOMPOSED:
            return rd

        feat = est['xd']
        hm = est['predict_hm']
        S = hm.size(-1)
        rd['skps_hm_pred'] = hm.detach()
        if fused:
            from . import iuv_ops
            centers = iuv_ops.softargmax(hm, 10.0)
        else:
            centers = softmax_integral_tensor(10 * hm, hm.size(1), hm.size(-2), hm.size(-1))
        centers = centers / (0.5 * S) - 1

        if self.training and smpl_kps_gt is not None:
            if cfg.DANET.STN_KPS_WEIGHTS > 0 and smpl_kps_gt.shape[-1] == 3:
                wk = smpl_kps_gt[:, :, 2:3]
                l = (F.smooth_l1_loss(centers, smpl_kps_gt[:, :, :2], reduction='none') * wk).sum() / smpl_kps_gt.size(0)
                rd['losses']['loss_roi'] = l * cfg.DANET.STN_KPS_WEIGHTS
            if cfg.DANET.STN_CENTER_JITTER > 0:
                centers = centers + cfg.DANET.STN_CENTER_JITTER * (torch.rand_like(centers) - 0.5)

        if FUSED_STN_THETA and fused and am_raw is not None and centers.is_cuda:
            # visibility score + affine_para in one launch (csrc/glue.hip; ~160 tensor-op launches in the form below)
            thetas = self.stn_theta(centers, am_raw, align)
        else:
            hidden = None
            if cfg.DANET.STN_PART_VIS_SCORE > 0:
                if am_raw is not None:        # membership of the winning part, looked up per pixel
                    score_maps = self._vis_membership.t()[am_raw.long()].permute(0, 3, 1, 2)
                else:
                    _, _, index_cl, _ = iuvmap_clean(u_pred, v_pred, index_pred, ann_pred)
                    score_maps = torch.einsum('jc,bchw->bjhw', self._vis_membership, index_cl.detach())
                score = _sample_points(score_maps, centers.detach(), align)
                hidden = score < cfg.DANET.STN_PART_VIS_SCORE
            thetas, _ = self.affine_para(centers, hidden)
        rd['stn_kps_pred'] = centers.detach()
        part_maps = stn_gather(feat, thetas, align_corners=align)                       # [B,24*C,H,W]
        ppi = self.iuv_est.final_pred.predict_partial_iuv
        if FUSED_PART_LOSSES and not fp32 and self.training and part_maps.is_cuda and ppi.out_channels == 24 * 21 and ppi.groups == 24:
            # keep the grouped conv's zero-padded output (24 channels per joint): the fused part ops read it as it is,
            # the [B,24,3,7,H,W] tensor of the reference is a strided view of it
            from .conv import conv2d
            # (bf16 output: the part kernels read bf16 -- the same rounding the fp32 head output went through before)
            pp = conv2d(part_maps, ppi.weight, ppi.bias, ppi.stride[0], ppi.padding[0], ppi.dilation[0], ppi.groups,
                        False, keep_group_padding=True)
            part_pred = part_ops.padded_view6(pp)
            Sp = part_pred.size(-1)
        else:
            part_pred = ppi(part_maps)
            Sp = part_pred.size(-1)
            part_pred = part_pred.reshape(part_pred.size(0), 24, 3, -1, Sp, Sp)          # [B,24,3,7,H,W]

        if self.training and iuv_image_gt is not None and FUSED_PART_LOSSES and not fp32 and part_pred.is_cuda and \
                tuple(iuv_image_gt.shape[-2:]) == (Sp, Sp):
            # one kernel: ground-truth resampling + the three losses (no [B,24,3,7,H,W] fp32 intermediates);
            # rd['part_iuv_gt'] (visualisation only in the reference) is not materialised on this path
            B = part_pred.shape[0]
            w = None if has_iuv is None else has_iuv.to(torch.float32)
            if LOSS_FINALIZE:
                pr = cfg.DANET.POINT_REGRESSION_WEIGHTS / (24. * B)          # (sum / B * weight) / 24 joints
                scales = ((pr, 0.), (pr, 0.), (1., 24. * Sp * Sp))
                if PART_JOINT and part_clean is not None:
                    # the caller (DaNet) will clean this prediction for the regressor: both consumers in one autograd node, so that the
                    # backward is one launch (part_ops.PartJointFunction); part_clean = (keep [B,24,7] or None,)
                    x24, lU, lV, lI = part_ops.part_joint(part_pred, part_clean[0], iuv_image_gt, thetas, w, self._dp_sel, align, scales)
                    rd['part_x24'] = x24
                else:
                    lU, lV, lI = part_ops.part_losses(part_pred, iuv_image_gt, thetas, w, self._dp_sel, align, scales=scales)
                rd['losses'].update({'loss_pU': lU, 'loss_pV': lV, 'loss_pIndexUV': lI})
            else:
                sums = part_ops.part_losses(part_pred, iuv_image_gt, thetas, w, self._dp_sel, align)
                wsum = torch.tensor(float(B), device=sums.device) if w is None else w.sum().clamp(min=1.0)
                lU = sums[0] / B * cfg.DANET.POINT_REGRESSION_WEIGHTS
                lV = sums[1] / B * cfg.DANET.POINT_REGRESSION_WEIGHTS
                lI = sums[2] / (wsum * (24 * Sp * Sp))
                rd['losses'].update({'loss_pU': lU / 24., 'loss_pV': lV / 24., 'loss_pIndexUV': lI})
        elif self.training and iuv_image_gt is not None:
            if uvia_list is None:
                uvia_list = iuv_img2map(iuv_image_gt)
            simp = self.part_iuv_simp(*uvia_list[:3])                                    # [B,24,3,7,H,W]
            B = simp.shape[0]
            flat = simp.reshape(B * 24, 21, Sp, Sp)
            grid = F.affine_grid(thetas.detach().reshape(B * 24, 2, 3), list(flat.shape), align_corners=align)
            part_gt = F.grid_sample(flat, grid, mode='bilinear', padding_mode='zeros', align_corners=align)
            part_gt = part_gt.reshape(B, 24, 3, 7, Sp, Sp)
            rd['part_iuv_gt'] = part_gt
            lU, lV, lI, _ = self.body_uv_losses(part_pred[:, :, 0], part_pred[:, :, 1], part_pred[:, :, 2], None,
                                                [part_gt[:, :, 0], part_gt[:, :, 1], part_gt[:, :, 2], None], has_iuv,
                                                class_dim=2)
            # the reference sums 24 per-joint losses and divides by 24: U/V are sums over pixels (/B),
            # the index loss a mean over pixels -- the joint axis is already inside the mean here
            rd['losses'].update({'loss_pU': lU / 24., 'loss_pV': lV / 24., 'loss_pIndexUV': lI})
        rd['part_iuv_pred'] = part_pred
        return rd
