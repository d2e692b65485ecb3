"""HRNet-W48 backbone (+IUV heads) on the HIP kernels.

Mirrors the module tree / state-dict keys of /root/reference/models/module/hr_module.py
(HighResolutionModule :15-179, PoseHighResolutionNet :188-410).  Each cross-resolution fuse
(1x1 conv + BN + nearest upsample, strided 3x3 chains, sum, ReLU; hr_module.py:111-177) is one
sum_relu launch over at most four terms that reads the low-resolution terms in place.
"""
import os

import torch
import torch.nn as nn

from . import segments
from .config import cfg
from .nn import fan_out, fan_out_multi, sum_relu, sum_relu_multi, multi_batch_norm, multi_conv_bn
from .resnet import BasicBlock, Bottleneck, ConvBN, IUV_predict_layer, make_res_layer, BN_MOMENTUM
from .nn import Conv2d, BatchNorm2d
from .conv import multi_conv, ResLink

blocks_dict = {'BASIC': BasicBlock, 'BOTTLENECK': Bottleneck}


class _Chain(nn.Module):
    """Sequence of ConvBN stages indexed '0','1',... (same keys as the reference's nested Sequentials)."""

    def __init__(self, stages):
        super().__init__()
        for i, s in enumerate(stages):
            self.add_module(str(i), s)

    def forward(self, x):
        for m in self._modules.values():
            x = m(x)
        return x


LOCKSTEP_BRANCHES = bool(int(os.environ.get('DANET_LOCKSTEP', '1')))    # one multi-tensor BatchNorm launch per block level
LOCKSTEP_CONVS = bool(int(os.environ.get('DANET_LOCKSTEP_CONVS', '1')))     # ... and one multi-problem conv launch
FUSE_GROUP = int(os.environ.get('DANET_FUSE_GROUP', '12'))    # exchange paths per multi-problem launch (the kernels take up to 12: a four-branch module's first stage)
FUSE_SPLIT_RELU = bool(int(os.environ.get('DANET_FUSE_SPLIT_RELU', '0')))       # A-B knob: 1 = ReLU and non-ReLU exchange stages in separate launches (rounds 2-4)
SUM_MULTI = bool(int(os.environ.get('DANET_SUM_MULTI', '1')))       # a module's fuse sums (and their gradients) in one launch each
BRANCH_STREAMS = False      # run the low-resolution branches on side streams (set by the trainer's hipGraph capture)
_SIDE = {}


def _side_streams(device, n):
    key = (device.index, torch.cuda.current_stream(device).stream_id)
    pool = _SIDE.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool[:n]


class HighResolutionModule(nn.Module):
    def __init__(self, num_branches, block, num_blocks, num_inchannels, num_channels, fuse_method, multi_scale_output=True):
        super().__init__()
        if not (num_branches == len(num_blocks) == len(num_channels) == len(num_inchannels)):
            raise ValueError('NUM_BRANCHES(%d) does not match NUM_BLOCKS/NUM_CHANNELS/NUM_INCHANNELS' % num_branches)
        self.num_inchannels = num_inchannels
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        branches = []
        for i in range(num_branches):
            self.inplanes = num_inchannels[i]
            branches.append(make_res_layer(self, block, num_channels[i], num_blocks[i]))
            num_inchannels[i] = num_channels[i] * block.expansion
        del self.inplanes
        self.branches = nn.ModuleList(branches)
        self.fuse_layers = self._make_fuse_layers()

    def _make_fuse_layers(self):
        if self.num_branches == 1:
            return None
        nb, ch = self.num_branches, self.num_inchannels
        rows = []
        for i in range(nb if self.multi_scale_output else 1):
            row = []
            for j in range(nb):
                if j > i:        # low -> high resolution: 1x1 conv + BN, upsampled inside sum_relu
                    row.append(ConvBN(ch[j], ch[i], 1, 1, 0, momentum=0.1))
                elif j == i:
                    row.append(None)
                else:            # high -> low: (i-j) strided 3x3 convs, ReLU on all but the last
                    stages = []
                    for k in range(i - j):
                        last = k == i - j - 1
                        stages.append(ConvBN(ch[j], ch[i] if last else ch[j], 3, 2, 1, momentum=0.1, relu=not last))
                    row.append(_Chain(stages))
            rows.append(nn.ModuleList(row))
        return nn.ModuleList(rows)

    def get_num_inchannels(self):
        return self.num_inchannels

    def _lockstep_ok(self):
        n = len(self.branches[0])
        return all(len(br) == n and all(isinstance(b, BasicBlock) and b.downsample is None for b in br) for br in self.branches)

    def _branches_in_lockstep(self, x):
        """All branches advance one BasicBlock at a time so that the four BatchNorms of a level share ONE launch
        (nn.multi_batch_norm): the low-resolution branches' BatchNorm launches are dominated by the per-launch floor."""
        xs = list(x[:self.num_branches])
        for k in range(len(self.branches[0])):
            blocks = [br[k] for br in self.branches]
            # identity shortcuts: their gradients ride on ResLinks into conv1's data-gradient epilogue (resnet.BasicBlock)
            links = [ResLink() if (v.requires_grad and torch.is_grad_enabled()) else None for v in xs]
            if LOCKSTEP_CONVS:
                # conv -> bn -> relu and conv -> bn -> + identity -> relu: one launch each when the streamed 3x3 kernel takes the level's
                # convolutions (nn.multi_conv_bn: the BatchNorm is the tail of the convolutions' launch), two otherwise
                h = multi_conv_bn([b.conv1 for b in blocks], xs, [b.bn1 for b in blocks], None, relu=True, conv_links=links)
                xs = multi_conv_bn([b.conv2 for b in blocks], h, [b.bn2 for b in blocks], xs, relu=True, bn_links=links)
            else:
                h = [b.conv1(v, link=lk) for b, v, lk in zip(blocks, xs, links)]
                h = multi_batch_norm([b.bn1 for b in blocks], h, None, relu=True)
                h = [b.conv2(v) for b, v in zip(blocks, h)]
                xs = multi_batch_norm([b.bn2 for b in blocks], h, xs, relu=True, links=links)
        return xs

    def _branches_on_streams(self, x):
        """The resolution branches of a module are independent until the fuse layers: branch 0 (the
        chip-filling one) stays on the current stream, the low-resolution branches -- whose kernels
        launch far fewer workgroups than there are CUs -- run beside it on side streams.  Autograd
        replays each branch's backward on the stream its forward ran on."""
        cur = torch.cuda.current_stream(x[0].device)
        side = _side_streams(x[0].device, self.num_branches - 1)
        out = [None] * self.num_branches
        for i in range(1, self.num_branches):
            side[i - 1].wait_stream(cur)
            with torch.cuda.stream(side[i - 1]):
                out[i] = self.branches[i](x[i])
        out[0] = self.branches[0](x[0])
        for i in range(1, self.num_branches):
            cur.wait_stream(side[i - 1])
        return out

    def forward(self, x):
        if self.num_branches == 1:
            return [self.branches[0](x[0])]
        if LOCKSTEP_BRANCHES and x[0].is_cuda and self.training and self._lockstep_ok():
            x = self._branches_in_lockstep(x)
        elif BRANCH_STREAMS and x[0].is_cuda:
            x = self._branches_on_streams(x)
        else:
            x = [self.branches[i](x[i]) for i in range(self.num_branches)]
        # branch j feeds the exchange path to every other output and its own fuse sum: its gradient is the sum of that many
        # contributions -- one kernel (nn.fan_out) instead of autograd's pairwise adds
        nout = len(self.fuse_layers)
        # (nout - 1 paths + own sum, or nout paths; the branches' gradient sums share one launch: nn.fan_out_multi)
        fan = fan_out_multi([x[j] for j in range(self.num_branches)], nout) if SUM_MULTI else [fan_out(x[j], nout) for j in range(self.num_branches)]
        take = [0] * self.num_branches

        def use(j):
            take[j] += 1
            return fan[j][take[j] - 1]
        xin = {(i, j): use(j) for i in range(nout) for j in range(self.num_branches) if j != i}
        if LOCKSTEP_BRANCHES and x[0].is_cuda and self.training:
            fused = self._fuse_paths_in_lockstep(xin)
        else:
            fused = {(i, j): self.fuse_layers[i][j](xin[(i, j)]) for i in range(nout)
                     for j in range(self.num_branches) if j != i}
        groups = []
        for i in range(len(self.fuse_layers)):
            terms = [use(j) if j == i else fused[(i, j)] for j in range(self.num_branches)]
            shifts = [j - i if j > i else 0 for j in range(self.num_branches)]
            groups.append((terms, shifts))
        # the module's fuse sums are independent of each other: one launch per pass (nn.sum_relu_multi) instead of one per output
        return sum_relu_multi(groups, relu=True) if SUM_MULTI else [sum_relu(t, s, relu=True) for t, s in groups]

    def _fuse_paths_in_lockstep(self, x):
        """Every (output i, input j) exchange path is a chain of 1..3 conv+BN stages; the paths are independent, so
        stage k of all of them runs as a few multi-problem conv launches and multi-tensor BatchNorm launches
        (groups of four) instead of one launch per stage."""
        paths = {}
        for i in range(len(self.fuse_layers)):
            for j in range(self.num_branches):
                if j != i:
                    m = self.fuse_layers[i][j]
                    paths[(i, j)] = [m] if isinstance(m, ConvBN) else list(m._modules.values())
        cur = {key: x[key] for key in paths}              # x: {(output i, input j): branch j's output for that path}
        depth = max(len(st) for st in paths.values())
        for k in range(depth):
            # stage k of every path that has one: paths that end here (BatchNorm without ReLU) and paths that go on (with ReLU) share
            # launches -- the BatchNorm kernels take the ReLU flag per job (round 5; rounds 2-4 launched the two kinds separately)
            for relu in ((False, True) if FUSE_SPLIT_RELU else (None,)):
                keys = [key for key, st in paths.items() if len(st) > k and (relu is None or bool(st[k].relu) == relu)]
                # same output-tile count first, so that a group qualifies for the multi-problem conv launch
                keys.sort(key=lambda key: (paths[key][k]._modules['0'].out_channels % 48 == 0, paths[key][k]._modules['0'].out_channels))
                for g0 in range(0, len(keys), FUSE_GROUP):
                    grp = keys[g0:g0 + FUSE_GROUP]
                    convs = [paths[key][k]._modules['0'] for key in grp]
                    bns = [paths[key][k]._modules['1'] for key in grp]
                    h = multi_conv(convs, [cur[key] for key in grp]) if LOCKSTEP_CONVS else [c(cur[key]) for c, key in zip(convs, grp)]
                    h = multi_batch_norm(bns, h, None, relu=[bool(paths[key][k].relu) for key in grp])
                    for key, v in zip(grp, h):
                        cur[key] = v
        return cur


class PoseHighResolutionNet(nn.Module):
    def __init__(self, part_out_dim=25):
        super().__init__()
        extra = cfg.HR_MODEL.EXTRA
        self.inplanes = 64
        self.conv1 = Conv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.conv2 = Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.layer1 = make_res_layer(self, Bottleneck, 64, 4)

        self.stage2_cfg = extra['STAGE2']
        ch = self._widths(self.stage2_cfg)
        self.transition1 = self._make_transition_layer([256], ch)
        self.stage2, pre = self._make_stage(self.stage2_cfg, ch)
        self.stage3_cfg = extra['STAGE3']
        ch = self._widths(self.stage3_cfg)
        self.transition2 = self._make_transition_layer(pre, ch)
        self.stage3, pre = self._make_stage(self.stage3_cfg, ch)
        self.stage4_cfg = extra['STAGE4']
        ch = self._widths(self.stage4_cfg)
        self.transition3 = self._make_transition_layer(pre, ch)
        self.stage4, pre = self._make_stage(self.stage4_cfg, ch, multi_scale_output=False)
        self.final_feat_dim = pre[0]
        self.final_pred = IUV_predict_layer(feat_dim=self.final_feat_dim, part_out_dim=part_out_dim)
        self.pretrained_layers = extra.get('PRETRAINED_LAYERS', ['*'])

    @staticmethod
    def _widths(stage_cfg):
        block = blocks_dict[stage_cfg['BLOCK']]
        return [c * block.expansion for c in stage_cfg['NUM_CHANNELS']]

    def _make_transition_layer(self, pre, cur):
        layers = []
        for i in range(len(cur)):
            if i < len(pre):
                layers.append(ConvBN(pre[i], cur[i], 3, 1, 1, momentum=0.1, relu=True) if cur[i] != pre[i] else None)
            else:
                stages = []
                for j in range(i + 1 - len(pre)):
                    outc = cur[i] if j == i - len(pre) else pre[-1]
                    stages.append(ConvBN(pre[-1], outc, 3, 2, 1, momentum=0.1, relu=True))
                layers.append(_Chain(stages))
        return nn.ModuleList(layers)

    def _make_stage(self, layer_config, num_inchannels, multi_scale_output=True):
        block = blocks_dict[layer_config['BLOCK']]
        modules = []
        for i in range(layer_config['NUM_MODULES']):
            mso = multi_scale_output or i != layer_config['NUM_MODULES'] - 1
            modules.append(HighResolutionModule(layer_config['NUM_BRANCHES'], block, layer_config['NUM_BLOCKS'],
                                                num_inchannels, layer_config['NUM_CHANNELS'], layer_config['FUSE_METHOD'], mso))
            num_inchannels = modules[-1].get_num_inchannels()
        return nn.Sequential(*modules), num_inchannels

    def _run_stage(self, stage, xs):
        for m in stage:
            # a backward-pass segment per module (segments.py): identity unless a data-parallel trainer asked for cuts
            xs = m(segments.cut(xs))
        return xs

    def forward(self, x):
        x = self.bn1(self.conv1(x), relu=True)
        x = self.bn2(self.conv2(x), relu=True)
        x = self.layer1(x)
        xs = [t(x) if t is not None else x for t in self.transition1]
        ys = self._run_stage(self.stage2, xs)
        xs = [self.transition2[i](ys[-1]) if self.transition2[i] is not None else ys[i]
              for i in range(self.stage3_cfg['NUM_BRANCHES'])]
        ys = self._run_stage(self.stage3, xs)
        xs = [self.transition3[i](ys[-1]) if self.transition3[i] is not None else ys[i]
              for i in range(self.stage4_cfg['NUM_BRANCHES'])]
        ys = self._run_stage(self.stage4, xs)
        feat = ys[0]
        return self.final_pred(feat)              # (with 'xd' = the feature map itself)

    def init_weights(self, pretrained=''):
        """hr_module.py:380-410: conv weights ~ N(0, 0.001), BN (1, 0), optional checkpoint."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, std=0.001)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if pretrained and os.path.isfile(pretrained):
            sd = torch.load(pretrained, map_location='cpu')
            keep = {k: v for k, v in sd.items() if k.split('.')[0] in self.pretrained_layers or self.pretrained_layers[0] == '*'}
            self.load_state_dict(keep, strict=False)
        elif pretrained:
            raise ValueError('{} is not exist!'.format(pretrained))
