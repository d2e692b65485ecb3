"""IUV -> SMPL regressor (default decomposed predictor with GCN refinement) and its losses.

Mirrors /root/reference/models/danet/smpl_regressor.py: SMPL_Regressor (:38-319) and
DecomposedPredictor (:397-942, 'gcn' branch :844-895).  Module / parameter names are the
reference's, including the `rot2pos` / `pos2rot` stacks that the 'gcn'
forward allocates but never uses.  All sample selection by has_smpl / has_kp3d is done with
per-sample weights (no boolean-mask indexing, no host synchronisation).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import assets
from . import gcn_tail as _gcn_tail
from . import glue as _glue
from .config import cfg
from .gcn import GCN, adjacency, normalize_digraph, normalize_undigraph
from .geometry import perspective_projection, rot6d_to_rotmat
from .iuv_estimator import SMPL_PARENTS, SMPL_CHILDREN, DP2SMPL_MAPPING
from .nn import Conv2d, BatchNorm2d
from .resnet import SmplResNet, LimbResLayers
from .smpl import SMPL


class _StemNet(nn.Module):
    """nn.Sequential(Conv2d 1x1, BatchNorm2d, ReLU, SmplResNet) with the same child indices 0,1,3."""

    def __init__(self, in_channels, resnet):
        super().__init__()
        self.add_module('0', Conv2d(in_channels, 64, 1, bias=False))
        self.add_module('1', BatchNorm2d(64))
        self.add_module('3', resnet)

    def __getitem__(self, i):
        return self._modules[str(i)]

    def forward(self, x):
        return self._modules['3'](self._modules['1'](self._modules['0'](x), relu=True))


def _pool_conv1x1_grouped(cin, cout, groups):
    """nn.Sequential(AdaptiveAvgPool2d(1), Conv2d(..., groups)) -- child '1' holds the parameters.
    Inputs are already [B, C, 1, 1]; the grouped 1x1 conv is a batched matmul (tiny, torch op)."""
    seq = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(cin, cout, kernel_size=1, groups=groups))
    return seq


class _SideWindowOpen(torch.autograd.Function):
    """Identity on the joined result of the two branches: its backward is the FIRST node of the regressor's backward pass and opens the
    side-stream window (nn.SIDE_LIVE) -- from here on body_net's backward kernels run beside limb_net's."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        from . import nn as _nn
        _nn.SIDE_LIVE += 1
        return g


class _SideWindowClose(torch.autograd.Function):
    """Identity on the two branches' inputs: its backward runs when BOTH branches have delivered their input gradients and closes the
    window."""

    @staticmethod
    def forward(ctx, a, b):
        return a.view_as(a), b.view_as(b)

    @staticmethod
    def backward(ctx, ga, gb):
        from . import nn as _nn
        _nn.SIDE_LIVE = max(0, _nn.SIDE_LIVE - 1)
        return ga, gb


FUSED_SMPL_LOSSES = True    # SMPL-side losses through csrc/loss_ops.hip (False: the tensor-op formulation below)
REGROUP_PARTS = bool(int(__import__('os').environ.get('DANET_REGROUP_PARTS', '1')))   # the crops -> channel groups view as one launch each way (A-B knob)
BODY_STREAM = bool(int(__import__('os').environ.get('DANET_BODY_STREAM', '1')))     # body_net on a side stream beside limb_net (A-B knob)


def _masked_mean(per_sample_sum, mask, per_sample_count):
    """sum_b m_b * s_b / (sum_b m_b * count)  (0 when no sample is selected)."""
    m = mask.to(torch.float32)
    return (per_sample_sum * m).sum() / (m.sum().clamp(min=1.0) * per_sample_count)


class SMPL_Regressor(nn.Module):
    def __init__(self, options, orig_size=224, feat_in_dim=None, smpl_mean_params=None, pretrained=True, smpl_model=None):
        super().__init__()
        self.focal_length = 5000.
        self.options = options
        self.orig_size = orig_size
        if smpl_mean_params is None:
            mean_params = assets.default_mean_params()
        elif isinstance(smpl_mean_params, str):
            mean_params = np.load(smpl_mean_params)
        else:
            mean_params = smpl_mean_params
        init_pose_6d = torch.from_numpy(np.asarray(mean_params['pose'], np.float32)).unsqueeze(0)
        if not cfg.DANET.USE_6D_ROT:
            raise NotImplementedError('USE_6D_ROT=False is outside the default path')
        init_shape = torch.from_numpy(np.asarray(mean_params['shape'], np.float32)).unsqueeze(0)
        init_cam = torch.from_numpy(np.asarray(mean_params['cam'], np.float32)).unsqueeze(0)
        self.smpl = SMPL(smpl_model, batch_size=getattr(options, 'batch_size', 1), create_transl=False)
        if not cfg.DANET.DECOMPOSED:
            raise NotImplementedError('only the decomposed predictor (default) is on the hot path')
        self.smpl_para_Outs = DecomposedPredictor(feat_in_dim, (init_cam, init_shape, init_pose_6d), pretrained)

    def smpl_infer_net(self, in_dict):
        if self.training:
            raise ValueError('You should call this function only on inference.'
                             'Set the network in inference mode by net.eval().')
        with torch.no_grad():
            d = dict(in_dict)
            d['infer_mode'] = True
            return self._forward(d)

    def forward(self, in_dict):
        with torch.set_grad_enabled(self.training):
            return self._forward(in_dict)

    def _forward(self, in_dict):
        iuv_map = in_dict['iuv_map']
        part_iuv_map = in_dict.get('part_iuv_map')
        rd = {'losses': {}, 'metrics': {}, 'visualization': {}, 'prediction': {}}
        out = self.smpl_para_Outs(iuv_map, part_iuv_map)
        if in_dict.get('infer_mode', False):
            return out
        para = out['para']
        rd['visualization'].update(out['visualization'])
        rd['prediction']['cam'] = para[:, :3]
        rd['prediction']['shape'] = para[:, 3:13]
        rd['prediction']['pose'] = para[:, 13:].reshape(-1, 24, 3, 3).contiguous()
        if not self.training:
            return rd

        B = para.shape[0]
        dev = para.device
        target = in_dict['target']
        target_kps, target_kps3d = in_dict['target_kps'], in_dict['target_kps3d']
        target_vertices = in_dict['target_verts']
        has_kp3d = in_dict['has_kp3d'].reshape(B).to(torch.float32)
        has_smpl = in_dict['has_smpl'].reshape(B).to(torch.float32)
        D = cfg.DANET

        if D.ORTHOGONAL_WEIGHTS > 0:
            Rs = para[:, 13:].reshape(-1, 3, 3)
            orth = F.mse_loss(torch.bmm(Rs, Rs.transpose(1, 2)), torch.eye(3, device=dev).expand_as(Rs))
            rd['losses']['Rs_orth'] = orth * D.ORTHOGONAL_WEIGHTS
            rd['metrics']['orth'] = rd['losses']['Rs_orth'].detach()

        gt_rotmat = target[:, 13:].reshape(B, 24, 3, 3)
        pred_camera, pred_betas = para[:, :3], para[:, 3:13]
        pred_rotmat = para[:, 13:].reshape(B, 24, 3, 3)
        fused = FUSED_SMPL_LOSSES and para.is_cuda and len(out['joint_rotation']) <= 2 and \
            len(out.get('joint_position', ())) <= 2 and D.JOINT_POSITION_WEIGHTS > 0
        if fused:
            # every loss below in one op (csrc/loss_ops.hip): two launches forward, one backward
            from . import loss_ops
            with torch.no_grad():
                gt_pts = self.smpl(betas=target[:, 3:13].contiguous(), body_pose=gt_rotmat[:, 1:].contiguous(),
                                   global_orient=gt_rotmat[:, :1].contiguous(), pose2rot=False).smpl_joints
            pred = self.smpl(betas=pred_betas, body_pose=pred_rotmat[:, 1:], global_orient=pred_rotmat[:, :1], pose2rot=False,
                             rotmats=pred_rotmat if pred_rotmat.is_contiguous() else None)
            pred_vertices, pred_joints = pred.vertices, pred.joints
            w = {'SMPL_POSE': D.SMPL_POSE_WEIGHTS, 'JOINT_POSITION': D.JOINT_POSITION_WEIGHTS, 'PROJ_KPS': D.PROJ_KPS_WEIGHTS,
                 'KPS3D': D.KPS3D_WEIGHTS, 'SMPL_BETAS': D.SMPL_BETAS_WEIGHTS, 'VERTS': D.VERTS_WEIGHTS}
            rd['losses'].update(loss_ops.smpl_losses(
                para, out['joint_rotation'], out.get('joint_position', []), pred_joints, pred_vertices, target, gt_pts, target_vertices,
                target_kps, target_kps3d, has_smpl, has_kp3d, self.focal_length, D.INIMG_SIZE,
                self.options.openpose_train_weight, self.options.gt_train_weight, w))
            with torch.no_grad():
                pred_cam_t = torch.stack([pred_camera[:, 1], pred_camera[:, 2],
                                          2 * self.focal_length / (D.INIMG_SIZE * pred_camera[:, 0] + 1e-9)], dim=-1)
            rd['prediction']['vertices'] = pred_vertices
            rd['prediction']['cam_t'] = pred_cam_t
            for key in ('losses', 'metrics'):
                for k, v in rd[key].items():
                    if v.dim() == 0:
                        rd[key][k] = v.unsqueeze(0)
            return rd

        for i, rot in enumerate(out['joint_rotation']):                                 # smpl_regressor.py:147-155
            s = ((rot - target[:, 13:]) ** 2).sum(dim=1)
            rd['losses']['joint_rotation%d' % i] = _masked_mean(s, has_smpl, 216) * D.SMPL_POSE_WEIGHTS

        if 'joint_position' in out and D.JOINT_POSITION_WEIGHTS > 0:                    # :157-166
            with torch.no_grad():
                gt_pts = self.smpl(betas=target[:, 3:13].contiguous(), body_pose=gt_rotmat[:, 1:].contiguous(),
                                   global_orient=gt_rotmat[:, :1].contiguous(), pose2rot=False).smpl_joints
            for i, pos in enumerate(out['joint_position']):
                rd['losses']['joint_position%d' % i] = self.l1_losses(pos, gt_pts, has_smpl) * D.JOINT_POSITION_WEIGHTS

        pred = self.smpl(betas=pred_betas, body_pose=pred_rotmat[:, 1:], global_orient=pred_rotmat[:, :1], pose2rot=False,
                             rotmats=pred_rotmat if pred_rotmat.is_contiguous() else None)
        pred_vertices, pred_joints = pred.vertices, pred.joints
        # weak perspective (s,tx,ty) -> translation (:182-193)
        pred_cam_t = torch.stack([pred_camera[:, 1], pred_camera[:, 2],
                                  2 * self.focal_length / (D.INIMG_SIZE * pred_camera[:, 0] + 1e-9)], dim=-1)
        kp2d = perspective_projection(pred_joints, None, pred_cam_t, self.focal_length, torch.zeros(B, 2, device=dev))
        kp2d = kp2d / (D.INIMG_SIZE / 2.)

        loss_pose, loss_betas = self.smpl_losses(pred_rotmat, pred_betas, gt_rotmat, target[:, 3:13], has_smpl)
        loss_kp2d = self.keypoint_loss(kp2d, target_kps, self.options.openpose_train_weight, self.options.gt_train_weight)
        loss_kp3d = self.keypoint_3d_loss(pred_joints, target_kps3d, has_kp3d)
        loss_verts = self.shape_loss(pred_vertices, target_vertices, has_smpl)

        rd['losses'].update({'keypoints_2d': loss_kp2d * D.PROJ_KPS_WEIGHTS,
                             'keypoints_3d': loss_kp3d * D.KPS3D_WEIGHTS,
                             'smpl_pose': loss_pose * D.SMPL_POSE_WEIGHTS,
                             'smpl_betas': loss_betas * D.SMPL_BETAS_WEIGHTS,
                             'smpl_verts': loss_verts * D.VERTS_WEIGHTS,
                             'cam': (torch.exp(-pred_camera[:, 0] * 10) ** 2).mean()})
        rd['prediction']['vertices'] = pred_vertices
        rd['prediction']['cam_t'] = pred_cam_t
        for key in ('losses', 'metrics'):
            for k, v in rd[key].items():
                if v.dim() == 0:
                    rd[key][k] = v.unsqueeze(0)
        return rd


    # ---- loss helpers: reference names and semantics (smpl_regressor.py:233-298), with row selection
    # expressed as per-sample weights (mask in {0,1}) instead of boolean indexing ----
    @staticmethod
    def l1_losses(pred, target, mask):
        """:233-238  sum |pred-target| over selected rows / number of selected rows."""
        s = (pred - target).abs().reshape(pred.shape[0], -1).sum(dim=1)
        return _masked_mean(s, mask, 1)

    @staticmethod
    def smpl_losses(pred_rotmat, pred_betas, gt_rotmat, gt_betas, has_smpl):
        """:287-298  MSE (mean over the selected rows' elements) on rotation matrices and betas."""
        B = pred_betas.shape[0]
        s_pose = ((pred_rotmat.reshape(B, -1) - gt_rotmat.reshape(B, -1)) ** 2).sum(dim=1)
        s_beta = ((pred_betas - gt_betas) ** 2).sum(dim=1)
        return _masked_mean(s_pose, has_smpl, 216), _masked_mean(s_beta, has_smpl, pred_betas.shape[1])

    @staticmethod
    def keypoint_loss(pred_keypoints_2d, gt_keypoints_2d, openpose_weight, gt_weight):
        """:248-257  confidence-weighted squared error, mean over all [B,49,2] elements."""
        conf = gt_keypoints_2d[:, :, -1:]
        conf = torch.cat([conf[:, :25] * openpose_weight, conf[:, 25:] * gt_weight], dim=1)
        return (conf * (pred_keypoints_2d - gt_keypoints_2d[:, :, :-1]) ** 2).mean()

    @staticmethod
    def keypoint_3d_loss(pred_keypoints_3d, gt_keypoints_3d, has_pose_3d):
        """:259-276  last 24 joints, both pelvis-centred (mid of joints 2,3), rows with has_pose_3d."""
        p3 = pred_keypoints_3d[:, 25:, :]
        g3 = gt_keypoints_3d[:, :, :-1]
        c3 = gt_keypoints_3d[:, :, -1:]
        g3 = g3 - ((g3[:, 2] + g3[:, 3]) / 2)[:, None, :]
        p3 = p3 - ((p3[:, 2] + p3[:, 3]) / 2)[:, None, :]
        s3 = (c3 * (p3 - g3) ** 2).sum(dim=(1, 2))
        return _masked_mean(s3, has_pose_3d, 72)

    @staticmethod
    def shape_loss(pred_vertices, gt_vertices, has_smpl):
        """:278-285  per-vertex L1, mean over the selected rows' elements."""
        sv = (pred_vertices - gt_vertices).abs().sum(dim=(1, 2))
        return _masked_mean(sv, has_smpl, pred_vertices.shape[1] * 3)


class DecomposedPredictor(nn.Module):
    def __init__(self, feat_in_dim=None, mean_params=None, pretrained=True):
        super().__init__()
        if cfg.DANET.INPUT_MODE not in ('iuv', 'iuv_gt'):
            raise NotImplementedError("only DANET.INPUT_MODE == 'iuv' is on the hot path")
        if cfg.DANET.REFINE_STRATEGY != 'gcn':
            raise NotImplementedError("only DANET.REFINE_STRATEGY == 'gcn' (the default) is on the hot path")
        self.in_channels = 3 * (1 + 24)
        self.register_buffer('mean_cam_shape', torch.cat(mean_params[:2], dim=1))
        self.register_buffer('mean_pose', mean_params[2])
        num_layers = cfg.DANET.GLO_NUM_LAYERS
        self.body_net = _StemNet(self.in_channels, SmplResNet(resnet_nums=num_layers, in_channels=64, num_classes=13))
        limb_num_layers = 18
        self.limb_net = _StemNet((1 + len(DP2SMPL_MAPPING[0])) * 3,
                                 SmplResNet(resnet_nums=limb_num_layers, in_channels=64, num_classes=0, truncate=1))
        if pretrained:
            self.body_net[3].init_weights(cfg.MSRES_MODEL.get('PRETRAINED_%d' % num_layers, ''))
            self.limb_net[3].init_weights(cfg.MSRES_MODEL.get('PRETRAINED_%d' % limb_num_layers, ''))
        fd = cfg.DANET.REFINEMENT.FEAT_DIM
        self.rot_feat_len = self.pos_feat_len = fd
        self.limb_reslayer = LimbResLayers(limb_num_layers, inplanes=256, outplanes=fd, groups=24)

        # allocated by the reference for every 'gcn' model but unused by its forward (:583-600)
        self.rot2pos = nn.ModuleList([nn.Sequential(nn.Conv2d(2 * fd, 512, 1), nn.BatchNorm2d(512), nn.ReLU(True),
                                                    nn.Conv2d(512, fd, 1), nn.BatchNorm2d(fd), nn.ReLU(True)) for _ in range(24)])
        self.pos2rot = nn.Sequential(nn.Conv2d(fd * 3, 1024, 1), nn.BatchNorm2d(1024), nn.ReLU(True),
                                     nn.Conv2d(1024, fd, 1), nn.BatchNorm2d(fd), nn.ReLU(True))
        if cfg.DANET.REFINEMENT.POS_INTERSUPV:
            self.coord_regressors = nn.ModuleList([_pool_conv1x1_grouped(fd * 24, 3 * 24, 24) for _ in range(2)])
        rot_dim = 6
        self.pose_regressors = nn.ModuleList([_pool_conv1x1_grouped(fd * 24, rot_dim * 24, 24) for _ in range(2)])
        nn.init.xavier_uniform_(self.pose_regressors[0][1].weight, gain=0.01)
        nn.init.xavier_uniform_(self.pose_regressors[1][1].weight, gain=0.01)

        eye = torch.eye(24).unsqueeze(0)
        self.register_buffer('I_n', eye)
        self.register_buffer('A_link', torch.tensor(adjacency('smpl'), dtype=torch.float32) - eye)
        A_mask = torch.tensor(adjacency('smpl_2neigh'), dtype=torch.float32)
        for a, b in [(1, 2), (1, 3), (2, 3), (13, 14), (12, 13), (12, 14)]:
            A_mask[:, a, b] = 1
            A_mask[:, b, a] = 1
        self.register_buffer('A', normalize_undigraph(A_mask))
        self.register_buffer('A_mask', A_mask - eye)
        self.edge_importance = nn.Parameter(torch.ones(1, 24, 24))
        self.refine_gcn = GCN(128, 256, 128, num_layers=int(cfg.DANET.REFINEMENT.GCN_NUM_LAYER), num_nodes=24)

        chains = []                                     # joint -> its ancestors up to the root (:440-452)
        for i in range(24):
            c, p = [i], i
            while p != 0:
                p = SMPL_PARENTS[p]
                c.append(p)
            chains.append(c)
        r2p = np.zeros((24, 24))
        for i in range(24):
            r2p[i, chains[i]] = 1
            r2p[i, i] = 0
        self.register_buffer('r2p_A', torch.from_numpy(normalize_digraph(r2p, AD_mode=False)).float().unsqueeze(0))
        children = [[j for j, p in enumerate(SMPL_PARENTS) if p == i] for i in range(24)]
        p2r = np.zeros((24, 24))
        for i in range(24):
            p2r[i, children[i]] = 1
            p2r[i, SMPL_PARENTS[i]] = 1
            p2r[i, i] = 1
        self.r2p_gcn = GCN(128, 128, 128, num_layers=1, num_nodes=24)
        self.register_buffer('p2r_A', torch.from_numpy(normalize_digraph(p2r, AD_mode=False)).float().unsqueeze(0))
        self.p2r_gcn = GCN(128, 128, 128, num_layers=1, num_nodes=24)

    @staticmethod
    def _grouped_head(seq, feats):
        """feats [B,24,C] -> grouped 1x1 conv (one [C,out] matrix per joint) -> [B,24,out]."""
        conv = seq[1]
        out_g = conv.out_channels // 24
        W = conv.weight.view(24, out_g, -1)                          # [24,out,C]
        y = torch.einsum('bjc,joc->bjo', feats, W)
        return y + conv.bias.view(1, 24, out_g)

    def forward(self, body_iuv, limb_iuv):
        rd = {'visualization': {}, 'losses': {}}
        # body_net (32 images, tensors of <= 16 MB, most launches a handful of workgroups) is independent of limb_net (the 768 part
        # crops: chip-filling launches) until `para` is assembled: on a side stream its latency-bound launches run beside the limb
        # net's (round 6; autograd replays its backward on the same stream; its BatchNorms take the two-kernel backward there, the
        # one-pass kernel being confined to the step's own stream, nn.ONEPASS_STREAM)
        side = None
        window = False
        pad = getattr(limb_iuv, '_nhwc_padded', None)
        # (only when both inputs carry gradients -- as in a train step --: the backward window below is bracketed by their gradient nodes)
        if BODY_STREAM and body_iuv.is_cuda and self.training and torch.is_grad_enabled() and body_iuv.requires_grad and \
                (pad if pad is not None else limb_iuv).requires_grad:
            from .hrnet import _side_streams
            from . import nn as _nn
            cur = torch.cuda.current_stream(body_iuv.device)
            side = _side_streams(body_iuv.device, 1)[0]
            # between this fork and the join below (and between their mirror images in the backward pass) kernels of two streams share
            # the compute units: barrier kernels are held to one workgroup per compute unit meanwhile (nn.SIDE_LIVE)
            if pad is not None:
                body_iuv, pad = _SideWindowClose.apply(body_iuv, pad)          # (the padded operand is what limb_net reads, see below)
            else:
                body_iuv, limb_iuv = _SideWindowClose.apply(body_iuv, limb_iuv)
            window = True
            _nn.SIDE_LIVE += 1
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                global_para, _ = self.body_net(body_iuv)
                global_para = global_para + self.mean_cam_shape
        else:
            global_para, _ = self.body_net(body_iuv)
            global_para = global_para + self.mean_cam_shape
        nbs, S = limb_iuv.size(0), limb_iuv.size(-1)
        # the fused part_clean op hands over the zero-padded 24-channel NHWC bf16 operand of the stem conv
        stacked = pad
        if stacked is None:
            stacked = limb_iuv.reshape(nbs * 24, -1, S, S)
        _, lf = self.limb_net(stacked)
        lf = lf['x4']
        lf = self.limb_reslayer(_glue.regroup_parts(lf, nbs) if REGROUP_PARTS else lf.reshape(nbs, -1, lf.size(-2), lf.size(-1)))      # [B,24*128,1,1]
        rot_feats = lf.reshape(nbs, 24, -1).float()                                   # [B,24,128]

        rd['joint_position'] = []
        rd['joint_rotation'] = []
        # the whole graph tail below as ONE launch per direction when the configuration is the trained default (csrc/gcn_tail.hip, round 6)
        fused = _gcn_tail.fused_tail(self, rot_feats)
        if fused is not None:
            jr0, jp0, jp1, smpl_pose = fused
            rd['joint_rotation'].append(jr0)
            rd['joint_position'] += [jp0, jp1]
            if side is not None:
                cur.wait_stream(side)
                global_para.record_stream(cur)
                _nn.SIDE_LIVE = max(0, _nn.SIDE_LIVE - 1)
            rd['para'] = torch.cat([global_para, smpl_pose], dim=1)
            if window:
                rd['para'] = _SideWindowOpen.apply(rd['para'])
            return rd
        if self.training:
            p0 = self._grouped_head(self.pose_regressors[0], rot_feats).reshape(nbs, -1) + self.mean_pose
            rd['joint_rotation'].append(rot6d_to_rotmat(p0).reshape(nbs, -1))
        pos_init = self.r2p_gcn(rot_feats, self.r2p_A[0])
        sup = self.training and cfg.DANET.JOINT_POSITION_WEIGHTS > 0 and cfg.DANET.REFINEMENT.POS_INTERSUPV
        if sup:
            rd['joint_position'].append(self._grouped_head(self.coord_regressors[0], pos_init))
        if cfg.DANET.REFINEMENT.REFINE_ON:
            graph_A = self.A_mask * F.relu(self.edge_importance)
            norm_A = normalize_undigraph(self.I_n[0] + graph_A)[0]
            pos_ref = pos_init + self.refine_gcn(pos_init, norm_A)
            if sup:
                rd['joint_position'].append(self._grouped_head(self.coord_regressors[1], pos_ref))
        else:
            pos_ref = pos_init
        rot_ref = self.p2r_gcn(pos_ref, self.p2r_A[0])
        pose6 = self._grouped_head(self.pose_regressors[-1], rot_ref).reshape(nbs, -1) + self.mean_pose
        smpl_pose = rot6d_to_rotmat(pose6).reshape(nbs, -1)
        if side is not None:
            cur.wait_stream(side)
            global_para.record_stream(cur)
            _nn.SIDE_LIVE = max(0, _nn.SIDE_LIVE - 1)
        rd['para'] = torch.cat([global_para, smpl_pose], dim=1)
        if window:
            rd['para'] = _SideWindowOpen.apply(rd['para'])
        return rd
