"""Residual blocks, IUV heads and the regressor CNNs on the HIP conv / BatchNorm kernels.

Mirrors the module tree (and therefore the state-dict keys) of
/root/reference/models/module/res_module.py: BasicBlock (:27-56), Bottleneck (:59-97),
PoseResNet (:107-278), IUV_predict_layer (:281-390), SmplResNet (:393-497),
LimbResLayers (:500-535).  Conv -> BatchNorm -> (+residual) -> ReLU runs as one MFMA conv launch
plus a fused BatchNorm/add/ReLU pass; activations are bf16 NHWC.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .config import cfg
from .nn import Conv2d, BatchNorm2d, relu as relu_op
from . import conv as _conv
from .conv import ResLink, conv2d
from .deconv import ConvTranspose2d

BN_MOMENTUM = 0.1
import os as _os
PAD_NARROW_BLOCKS = bool(int(_os.environ.get('DANET_PAD_NARROW_BLOCKS', '1')))
BOTTLENECK_LINK = bool(int(_os.environ.get('DANET_BOTTLENECK_LINK', '1')))          # A-B knob: identity-shortcut gradient of a Bottleneck through conv1's dgrad epilogue
HEAD_FAN_OUT = bool(int(_os.environ.get('DANET_HEAD_FAN_OUT', '1')))                # A-B knob: nn.fan_out over the six consumers of the final feature map      # A-B knob, see Bottleneck._forward_padded
HEAD_STREAM = bool(int(_os.environ.get('DANET_HEAD_STREAM', '1')))      # the four global IUV heads on a side stream beside the heat-map head (A-B knob)


class ConvBN(nn.Module):
    """Children '0' (conv) and '1' (bn): the same keys as the reference's nn.Sequential(conv, bn[, relu])."""

    def __init__(self, cin, cout, k, stride=1, pad=0, groups=1, momentum=BN_MOMENTUM, relu=False):
        super().__init__()
        self.add_module('0', Conv2d(cin, cout, k, stride, pad, bias=False, groups=groups))
        self.add_module('1', BatchNorm2d(cout, momentum=momentum))
        self.relu = relu

    def forward(self, x, res=None, relu=None):
        return self._modules['1'](self._modules['0'](x), res=res, relu=self.relu if relu is None else relu)


def conv3x3(in_planes, out_planes, stride=1, bias=False, groups=1):
    return Conv2d(in_planes * groups, out_planes * groups, 3, stride, 1, bias=bias, groups=groups)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride, groups=groups)
        self.bn1 = BatchNorm2d(planes * groups, momentum=BN_MOMENTUM)
        self.conv2 = conv3x3(planes, planes, groups=groups)
        self.bn2 = BatchNorm2d(planes * groups, momentum=BN_MOMENTUM)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        residual = x if self.downsample is None else self.downsample(x)
        # identity shortcut: its gradient rides on a ResLink into conv1's data-gradient epilogue (no separate add pass)
        link = ResLink() if (self.downsample is None and x.is_cuda and x.requires_grad and torch.is_grad_enabled()) else None
        out = self.bn1(self.conv1(x, link=link), relu=True)
        return self.bn2(self.conv2(out), res=residual, relu=True, link=link)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1):
        super().__init__()
        self.conv1 = Conv2d(inplanes * groups, planes * groups, 1, bias=False, groups=groups)
        self.bn1 = BatchNorm2d(planes * groups, momentum=BN_MOMENTUM)
        self.conv2 = Conv2d(planes * groups, planes * groups, 3, stride, 1, bias=False, groups=groups)
        self.bn2 = BatchNorm2d(planes * groups, momentum=BN_MOMENTUM)
        self.conv3 = Conv2d(planes * groups, planes * self.expansion * groups, 1, bias=False, groups=groups)
        self.bn3 = BatchNorm2d(planes * self.expansion * groups, momentum=BN_MOMENTUM)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        residual = x if self.downsample is None else self.downsample(x)
        if PAD_NARROW_BLOCKS and self.conv1.groups == 1 and self.conv1.out_channels % 8 != 0 and x.is_cuda and not _conv.fp32_mode():
            return self._forward_padded(x, residual)
        # identity shortcut: its gradient rides on a ResLink into conv1's (pointwise) data-gradient epilogue, as in BasicBlock
        link = ResLink() if (BOTTLENECK_LINK and self.downsample is None and x.is_cuda and x.requires_grad and torch.is_grad_enabled()) else None
        out = self.bn1(self.conv1(x, link=link), relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        return self.bn3(self.conv3(out), res=residual, relu=True, link=link)

    def _forward_padded(self, x, residual):
        """A block whose inner width is no multiple of 8 (the heat-map head's Bottleneck(48, 12), res_module.py:364) at the
        next multiple of 8 THROUGHOUT: the weights are zero-padded (tiny tensors), the extra channels stay exactly zero through
        conv -> BatchNorm (gamma = beta = 0) -> ReLU -> conv, and the 4 MB activations are no longer padded before and sliced
        after every convolution (~40 copy / fill launches per step).  Parameters and buffers keep the reference's shapes."""
        P = self.conv1.out_channels
        d = (-P) % 8
        tr = self.training
        c1, c2, c3 = self.conv1.weight, self.conv2.weight, self.conv3.weight
        from .glue import pad_multi
        # the convolutions' weights are packed at the padded widths straight from the parameters (conv.pack_weight pad_to); the four
        # padded BatchNorm vectors of the block are ONE launch (and one for their gradients)
        g1, b1, g2, b2 = pad_multi([(self.bn1.weight, (P + d,)), (self.bn1.bias, (P + d,)), (self.bn2.weight, (P + d,)), (self.bn2.bias, (P + d,))])
        out = conv2d(x, c1, None, 1, 0, 1, 1, want_stats=tr, weight_pad=(P + d, c1.shape[1]))
        out = self.bn1.forward_padded(out, d, relu=True, padded=(g1, b1))
        out = conv2d(out, c2, None, self.conv2.stride[0], 1, 1, 1, want_stats=tr, weight_pad=(P + d, P + d))
        out = self.bn2.forward_padded(out, d, relu=True, padded=(g2, b2))
        out = conv2d(out, c3, None, 1, 0, 1, 1, want_stats=tr, weight_pad=(c3.shape[0], P + d))
        return self.bn3(out, res=residual, relu=True)


resnet_spec = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
               101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3])}


def make_res_layer(owner, block, planes, blocks, stride=1, groups=1, momentum=BN_MOMENTUM):
    """Stack of residual blocks; `owner.inplanes` tracks the running width (per group)."""
    downsample = None
    if stride != 1 or owner.inplanes != planes * block.expansion:
        downsample = ConvBN(owner.inplanes * groups, planes * block.expansion * groups, 1, stride, 0, groups=groups,
                            momentum=momentum)
    layers = [block(owner.inplanes, planes, stride, downsample, groups=groups) if groups != 1
              else block(owner.inplanes, planes, stride, downsample)]
    owner.inplanes = planes * block.expansion
    for _ in range(1, blocks):
        layers.append(block(owner.inplanes, planes, groups=groups) if groups != 1 else block(owner.inplanes, planes))
    return nn.Sequential(*layers)


class IUV_predict_layer(nn.Module):
    """Global heads U,V,Index (25) and Ann (15), the SMPL-joint heat-map head, and the grouped
    partial-IUV head (res_module.py:281-390).  Head outputs are fp32 (they feed the losses)."""

    def __init__(self, feat_dim=256, final_cov_k=3, part_out_dim=25):
        super().__init__()
        pad = 1 if final_cov_k == 3 else 0
        self.predict_u = Conv2d(feat_dim, 25, final_cov_k, 1, pad, out_fp32=True)
        self.predict_v = Conv2d(feat_dim, 25, final_cov_k, 1, pad, out_fp32=True)
        self.predict_ann_index = Conv2d(feat_dim, 15, final_cov_k, 1, pad, out_fp32=True)
        self.predict_uv_index = Conv2d(feat_dim, 25, final_cov_k, 1, pad, out_fp32=True)
        self.inplanes = feat_dim
        # BatchNorm in this head uses torch's default momentum (res_module.py:364)
        self.predict_hm = nn.Sequential(make_res_layer(self, Bottleneck, int(feat_dim / 4), 3, momentum=0.1),
                                        Conv2d(feat_dim, 24, 3, 1, 1, bias=True, out_fp32=True))
        if cfg.DANET.DECOMPOSED:
            self.predict_partial_iuv = Conv2d(feat_dim * 24, part_out_dim * 3 * 24, final_cov_k, 1, pad, groups=24,
                                              out_fp32=True)

    def forward(self, x):
        # six consumers of the final feature map (five heads + the STN gather of the part crops, 'xd'): their gradients are summed by
        # the fuse-sum kernel (nn.fan_out: two launches) instead of five pairwise adds of autograd
        from .nn import fan_out
        xs = fan_out(x, 6) if HEAD_FAN_OUT else [x] * 6
        if HEAD_STREAM and x.is_cuda and self.training and torch.is_grad_enabled():
            # the four global heads are independent of the heat-map head's chain of ten narrow (12 / 16-channel) layers, whose launches
            # are bound by their latency: side by side on two streams (round 6; autograd replays each on its own stream)
            from .hrnet import _side_streams
            cur = torch.cuda.current_stream(x.device)
            side = _side_streams(x.device, 2)[1]
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                heads = [self.predict_u(xs[0]), self.predict_v(xs[1]), self.predict_uv_index(xs[2]), self.predict_ann_index(xs[3])]
            hm = self.predict_hm(xs[4])
            cur.wait_stream(side)
            for h in heads:
                h.record_stream(cur)
            return {'predict_u': heads[0], 'predict_v': heads[1], 'predict_uv_index': heads[2], 'predict_ann_index': heads[3],
                    'predict_hm': hm, 'xd': xs[5]}
        return {'predict_u': self.predict_u(xs[0]), 'predict_v': self.predict_v(xs[1]),
                'predict_uv_index': self.predict_uv_index(xs[2]), 'predict_ann_index': self.predict_ann_index(xs[3]),
                'predict_hm': self.predict_hm(xs[4]), 'xd': xs[5]}


def _maxpool3x3s2(x):
    from .nn import maxpool3x3s2
    return maxpool3x3s2(x)


class PoseResNet(nn.Module):
    """ResNet trunk + 3 transposed convs (k4 s2) + IUV heads (res_module.py:107-278)."""

    def __init__(self, part_out_dim=25):
        super().__init__()
        self.inplanes = 64
        extra = cfg.MSRES_MODEL.EXTRA
        self.deconv_with_bias = extra['DECONV_WITH_BIAS']
        block, layers = resnet_spec[extra['NUM_LAYERS']]
        self.conv1 = Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.layer1 = make_res_layer(self, block, 64, layers[0])
        self.layer2 = make_res_layer(self, block, 128, layers[1], stride=2)
        self.layer3 = make_res_layer(self, block, 256, layers[2], stride=2)
        self.layer4 = make_res_layer(self, block, 512, layers[3], stride=2)
        mods = []
        for planes, kernel in zip(extra['NUM_DECONV_FILTERS'], extra['NUM_DECONV_KERNELS']):
            pad, outpad = {4: (1, 0), 3: (1, 1), 2: (0, 0)}[kernel]
            mods += [ConvTranspose2d(self.inplanes, planes, kernel, 2, pad, outpad, bias=self.deconv_with_bias),
                     BatchNorm2d(planes, momentum=BN_MOMENTUM), nn.ReLU(inplace=True)]
            self.inplanes = planes
        self.deconv_layers = nn.Sequential(*mods)
        self.final_feat_dim = extra['NUM_DECONV_FILTERS'][-1]
        self.final_pred = IUV_predict_layer(feat_dim=self.final_feat_dim, part_out_dim=part_out_dim)

    def forward(self, x):
        x = _maxpool3x3s2(self.bn1(self.conv1(x), relu=True))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        mods = list(self.deconv_layers)
        for i in range(0, len(mods), 3):
            x = mods[i + 1](mods[i](x), relu=True)
        return self.final_pred(x)                 # (with 'xd' = the feature map itself)

    def init_weights(self, pretrained=''):
        for m in self.deconv_layers.modules():
            if isinstance(m, ConvTranspose2d):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        for m in self.final_pred.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        if pretrained:
            _load_backbone(self, pretrained)


class SmplResNet(nn.Module):
    """ResNet on IUV maps (res_module.py:393-497): 7x7 s2 stem on `in_channels`, optional truncation,
    average pool + linear."""

    def __init__(self, resnet_nums, in_channels=3, num_classes=229, last_stride=2, n_extra_feat=0, truncate=0, **kwargs):
        super().__init__()
        self.inplanes = 64
        self.truncate = truncate
        block, layers = resnet_spec[resnet_nums]
        self.conv1 = Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.layer1 = make_res_layer(self, block, 64, layers[0])
        self.layer2 = make_res_layer(self, block, 128, layers[1], stride=2)
        self.layer3 = make_res_layer(self, block, 256, layers[2], stride=2) if truncate < 2 else None
        self.layer4 = make_res_layer(self, block, 512, layers[3], stride=last_stride) if truncate < 1 else None
        self.num_classes = num_classes
        if num_classes > 0:
            self.final_layer = nn.Linear(512 * block.expansion, num_classes)
            nn.init.xavier_uniform_(self.final_layer.weight, gain=0.01)
        self.n_extra_feat = n_extra_feat
        if n_extra_feat > 0:
            raise NotImplementedError('n_extra_feat > 0 is only used by INPUT_MODE variants outside the hot path')

    def forward(self, x, infeat=None):
        x = _maxpool3x3s2(self.bn1(self.conv1(x), relu=True))
        x2 = self.layer2(self.layer1(x))
        x3 = self.layer3(x2) if self.truncate < 2 else x2
        x4 = self.layer4(x3) if self.truncate < 1 else x3
        cls = None
        if self.num_classes > 0:
            xp = x4.float().mean(dim=(2, 3))
            cls = self.final_layer(xp)
        return cls, {'x4': x4}

    def init_weights(self, pretrained=''):
        if pretrained:
            _load_backbone(self, pretrained, drop_mismatched=True)


class LimbResLayers(nn.Module):
    """Grouped (x24) layer4 + average pool (res_module.py:500-535)."""

    def __init__(self, resnet_nums, inplanes, outplanes=None, groups=1, **kwargs):
        super().__init__()
        self.inplanes = inplanes
        block, layers = resnet_spec[resnet_nums]
        self.outplanes = 512 if outplanes is None else outplanes
        self.layer4 = make_res_layer(self, block, self.outplanes, layers[3], stride=2, groups=groups)

    def forward(self, x):
        x = self.layer4(x)
        return x.float().mean(dim=(2, 3), keepdim=True)


def _load_backbone(module, path, drop_mismatched=False):
    """Checkpoint loading with the reference's conventions (res_module.py:253-278, 466-497): plain
    state dict or {'state_dict': ...} with optional 'module.' prefixes; strict=False."""
    import os
    from collections import OrderedDict
    if not os.path.isfile(path):
        raise ValueError('pretrained model does not exist: %s' % path)
    ckpt = torch.load(path, map_location='cpu')
    if isinstance(ckpt, dict) and 'state_dict' in ckpt:
        ckpt = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in ckpt['state_dict'].items())
    if drop_mismatched:
        own = module.state_dict()
        ckpt = OrderedDict((k, v) for k, v in ckpt.items() if k not in own or own[k].shape == v.shape)
    module.load_state_dict(ckpt, strict=False)
