"""TEST INFRASTRUCTURE ONLY -- plain PyTorch fp32 restatement of the reference's networks.

Runs on CPU (or any torch device) with stock torch ops only; state-dict keys equal the
reference's, so one set of weights drives the reference (here, in the build container), this
oracle (everywhere) and the HIP product.  Pinned against golden vectors produced by the
reference itself (tests/golden/make_golden.py -> g5*/g6*/g9* fixtures, tests/test_oracle_nets.py).
Each class cites the reference lines it follows.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

BN_MOMENTUM = 0.1

SMPL_PARENTS = [0, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
W48 = {'STAGE2': (1, [4, 4], [48, 96]), 'STAGE3': (4, [4, 4, 4], [48, 96, 192]), 'STAGE4': (3, [4, 4, 4, 4], [48, 96, 192, 384])}


def conv_bn(cin, cout, k, stride, pad, relu, groups=1, momentum=BN_MOMENTUM):
    mods = [nn.Conv2d(cin, cout, k, stride, pad, bias=False, groups=groups), nn.BatchNorm2d(cout, momentum=momentum)]
    if relu:
        mods.append(nn.ReLU(inplace=False))
    return nn.Sequential(*mods)


class BasicBlock(nn.Module):
    """/root/reference/models/module/res_module.py:27-56"""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes * groups, planes * groups, 3, stride, 1, bias=False, groups=groups)
        self.bn1 = nn.BatchNorm2d(planes * groups, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(planes * groups, planes * groups, 3, 1, 1, bias=False, groups=groups)
        self.bn2 = nn.BatchNorm2d(planes * groups, momentum=BN_MOMENTUM)
        self.downsample = downsample

    def forward(self, x):
        r = x if self.downsample is None else self.downsample(x)
        o = F.relu(self.bn1(self.conv1(x)))
        return F.relu(self.bn2(self.conv2(o)) + r)


class Bottleneck(nn.Module):
    """res_module.py:59-97"""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1):
        super().__init__()
        g = groups
        self.conv1 = nn.Conv2d(inplanes * g, planes * g, 1, bias=False, groups=g)
        self.bn1 = nn.BatchNorm2d(planes * g, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(planes * g, planes * g, 3, stride, 1, bias=False, groups=g)
        self.bn2 = nn.BatchNorm2d(planes * g, momentum=BN_MOMENTUM)
        self.conv3 = nn.Conv2d(planes * g, planes * 4 * g, 1, bias=False, groups=g)
        self.bn3 = nn.BatchNorm2d(planes * 4 * g, momentum=BN_MOMENTUM)
        self.downsample = downsample

    def forward(self, x):
        r = x if self.downsample is None else self.downsample(x)
        o = F.relu(self.bn1(self.conv1(x)))
        o = F.relu(self.bn2(self.conv2(o)))
        return F.relu(self.bn3(self.conv3(o)) + r)


SPEC = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]), 101: (Bottleneck, [3, 4, 23, 3])}


def make_layer(state, block, planes, blocks, stride=1, groups=1):
    """res_module.py:138-153 / 511-528; state = {'inplanes': n}"""
    ds = None
    if stride != 1 or state['inplanes'] != planes * block.expansion:
        ds = conv_bn(state['inplanes'] * groups, planes * block.expansion * groups, 1, stride, 0, False, groups)
    layers = [block(state['inplanes'], planes, stride, ds, groups)]
    state['inplanes'] = planes * block.expansion
    layers += [block(state['inplanes'], planes, groups=groups) for _ in range(1, blocks)]
    return nn.Sequential(*layers)


class HRModule(nn.Module):
    """/root/reference/models/module/hr_module.py:15-179"""

    def __init__(self, nb, num_blocks, inch, ch, multi_scale_output=True):
        super().__init__()
        self.nb = nb
        branches = []
        for i in range(nb):
            st = {'inplanes': inch[i]}
            branches.append(make_layer(st, BasicBlock, ch[i], num_blocks[i]))
            inch[i] = ch[i]
        self.branches = nn.ModuleList(branches)
        self.inch = inch
        rows = []
        for i in range(nb if multi_scale_output else 1):
            row = []
            for j in range(nb):
                if j > i:
                    row.append(nn.Sequential(nn.Conv2d(inch[j], inch[i], 1, 1, 0, bias=False), nn.BatchNorm2d(inch[i]),
                                             nn.Upsample(scale_factor=2 ** (j - i), mode='nearest')))
                elif j == i:
                    row.append(None)
                else:
                    st = []
                    for k in range(i - j):
                        last = k == i - j - 1
                        st.append(conv_bn(inch[j], inch[i] if last else inch[j], 3, 2, 1, not last, momentum=0.1))
                    row.append(nn.Sequential(*st))
            rows.append(nn.ModuleList(row))
        self.fuse_layers = nn.ModuleList(rows) if nb > 1 else None

    def forward(self, x):
        if self.nb == 1:
            return [self.branches[0](x[0])]
        x = [self.branches[i](x[i]) for i in range(self.nb)]
        out = []
        for i in range(len(self.fuse_layers)):
            y = x[0] if i == 0 else self.fuse_layers[i][0](x[0])
            for j in range(1, self.nb):
                y = y + (x[j] if i == j else self.fuse_layers[i][j](x[j]))
            out.append(F.relu(y))
        return out


class IUVHead(nn.Module):
    """res_module.py:281-390"""

    def __init__(self, feat_dim, part_out_dim=7):
        super().__init__()
        self.predict_u = nn.Conv2d(feat_dim, 25, 3, 1, 1)
        self.predict_v = nn.Conv2d(feat_dim, 25, 3, 1, 1)
        self.predict_ann_index = nn.Conv2d(feat_dim, 15, 3, 1, 1)
        self.predict_uv_index = nn.Conv2d(feat_dim, 25, 3, 1, 1)
        st = {'inplanes': feat_dim}
        self.predict_hm = nn.Sequential(make_layer(st, Bottleneck, feat_dim // 4, 3), nn.Conv2d(feat_dim, 24, 3, 1, 1))
        self.predict_partial_iuv = nn.Conv2d(feat_dim * 24, part_out_dim * 3 * 24, 3, 1, 1, groups=24)

    def forward(self, x):
        return {'predict_u': self.predict_u(x), 'predict_v': self.predict_v(x), 'predict_uv_index': self.predict_uv_index(x),
                'predict_ann_index': self.predict_ann_index(x), 'predict_hm': self.predict_hm(x)}


class HRNet(nn.Module):
    """hr_module.py:188-378 (widths given by `stages`, default W48)"""

    def __init__(self, part_out_dim=7, stages=None):
        super().__init__()
        stages = stages or W48
        self.conv1 = nn.Conv2d(3, 64, 3, 2, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        self.conv2 = nn.Conv2d(64, 64, 3, 2, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        st = {'inplanes': 64}
        self.layer1 = make_layer(st, Bottleneck, 64, 4)
        pre = [256]
        self.nbr = []
        for si, name in enumerate(('STAGE2', 'STAGE3', 'STAGE4')):
            nmod, nblk, ch = stages[name]
            setattr(self, 'transition%d' % (si + 1), self._transition(pre, ch))
            mods = []
            inch = list(ch)
            for m in range(nmod):
                mso = not (name == 'STAGE4' and m == nmod - 1)
                mods.append(HRModule(len(ch), nblk, inch, ch, mso))
                inch = mods[-1].inch
            setattr(self, 'stage%d' % (si + 2), nn.Sequential(*mods))
            pre = inch
            self.nbr.append(len(ch))
        self.final_pred = IUVHead(pre[0], part_out_dim)

    @staticmethod
    def _transition(pre, cur):
        layers = []
        for i in range(len(cur)):
            if i < len(pre):
                layers.append(conv_bn(pre[i], cur[i], 3, 1, 1, True, momentum=0.1) if cur[i] != pre[i] else None)
            else:
                st = []
                for j in range(i + 1 - len(pre)):
                    st.append(conv_bn(pre[-1], cur[i] if j == i - len(pre) else pre[-1], 3, 2, 1, True, momentum=0.1))
                layers.append(nn.Sequential(*st))
        return nn.ModuleList(layers)

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x = self.layer1(x)
        xs = [t(x) if t is not None else x for t in self.transition1]
        ys = xs
        for m in self.stage2:
            ys = m(ys)
        xs = [self.transition2[i](ys[-1]) if self.transition2[i] is not None else ys[i] for i in range(self.nbr[1])]
        ys = xs
        for m in self.stage3:
            ys = m(ys)
        xs = [self.transition3[i](ys[-1]) if self.transition3[i] is not None else ys[i] for i in range(self.nbr[2])]
        ys = xs
        for m in self.stage4:
            ys = m(ys)
        out = self.final_pred(ys[0])
        out['xd'] = ys[0]
        return out


class PoseResNet(nn.Module):
    """res_module.py:107-223 (ResNet-50 + 3 deconvs)"""

    def __init__(self, part_out_dim=7, num_layers=50):
        super().__init__()
        block, layers = SPEC[num_layers]
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        st = {'inplanes': 64}
        self.layer1 = make_layer(st, block, 64, layers[0])
        self.layer2 = make_layer(st, block, 128, layers[1], 2)
        self.layer3 = make_layer(st, block, 256, layers[2], 2)
        self.layer4 = make_layer(st, block, 512, layers[3], 2)
        mods, inp = [], st['inplanes']
        for _ in range(3):
            mods += [nn.ConvTranspose2d(inp, 256, 4, 2, 1, 0, bias=False), nn.BatchNorm2d(256, momentum=BN_MOMENTUM), nn.ReLU()]
            inp = 256
        self.deconv_layers = nn.Sequential(*mods)
        self.final_pred = IUVHead(256, part_out_dim)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        x = self.deconv_layers(self.layer4(self.layer3(self.layer2(self.layer1(x)))))
        out = self.final_pred(x)
        out['xd'] = x
        return out


class SmplResNet(nn.Module):
    """res_module.py:393-464"""

    def __init__(self, num, in_channels, num_classes, truncate=0):
        super().__init__()
        block, layers = SPEC[num]
        self.truncate = truncate
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64, momentum=BN_MOMENTUM)
        st = {'inplanes': 64}
        self.layer1 = make_layer(st, block, 64, layers[0])
        self.layer2 = make_layer(st, block, 128, layers[1], 2)
        self.layer3 = make_layer(st, block, 256, layers[2], 2) if truncate < 2 else None
        self.layer4 = make_layer(st, block, 512, layers[3], 2) if truncate < 1 else None
        self.num_classes = num_classes
        if num_classes > 0:
            self.final_layer = nn.Linear(512 * block.expansion, num_classes)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        x = self.layer2(self.layer1(x))
        x = self.layer3(x) if self.truncate < 2 else x
        x = self.layer4(x) if self.truncate < 1 else x
        cls = self.final_layer(x.mean(dim=(2, 3))) if self.num_classes > 0 else None
        return cls, x


class LimbResLayers(nn.Module):
    """res_module.py:500-535"""

    def __init__(self, inplanes=256, outplanes=128, groups=24):
        super().__init__()
        st = {'inplanes': inplanes}
        self.layer4 = make_layer(st, BasicBlock, outplanes, 2, 2, groups)

    def forward(self, x):
        return self.layer4(x).mean(dim=(2, 3), keepdim=True)


def stn_part_maps(feat, thetas, align_corners):
    """/root/reference/models/danet/iuv_estimator.py:193-204; thetas [B,24,2,3]."""
    outs = []
    for i in range(thetas.shape[1]):
        grid = F.affine_grid(thetas[:, i], list(feat.size()), align_corners=align_corners)
        outs.append(F.grid_sample(feat, grid, mode='bilinear', padding_mode='zeros', align_corners=align_corners))
    return torch.cat(outs, dim=1)


def hrnet_step_cpu(net, img, reps=1):
    """fwd+bwd of backbone+global heads on CPU (cpu_baseline leg of bench.py)."""
    import time
    net.train()
    t0 = time.time()
    for _ in range(reps):
        out = net(img)
        loss = sum(out[k].float().mean() for k in ('predict_u', 'predict_v', 'predict_uv_index', 'predict_ann_index', 'predict_hm'))
        loss.backward()
    return (time.time() - t0) / reps


class StepNets(nn.Module):
    """Every convolutional net of one DaNet optimisation step (models/danet/danet.py:133-366) in plain torch: HRNet-W48 +
    global heads + grouped partial-IUV head (iuv_estimator.py:193-211) + body_net / limb_net / grouped limb layer4
    (smpl_regressor.py:470-520).  Used by bench.py's cpu_baseline leg: 30.0 of the step's 30.07 GMAC per image
    (21.151 + 0.892 + 0.364 + 7.584; the GCN and the 1x1 regressors, < 0.01, are left out)."""

    def __init__(self):
        super().__init__()
        self.iuv_est = HRNet(part_out_dim=7)
        self.body_net = nn.Sequential(nn.Conv2d(75, 64, 1, bias=False), nn.BatchNorm2d(64), nn.ReLU(), SmplResNet(18, 64, 13))
        self.limb_net = nn.Sequential(nn.Conv2d(21, 64, 1, bias=False), nn.BatchNorm2d(64), nn.ReLU(), SmplResNet(18, 64, 0, truncate=1))
        self.limb_reslayer = LimbResLayers(256, 128, 24)

    def forward(self, img):
        out = self.iuv_est(img)
        B = img.shape[0]
        xd = out['xd']
        S = xd.shape[-1]
        theta = torch.tensor([[0.5, 0., 0.], [0., 0.5, 0.]]).repeat(B, 24, 1, 1)
        part = self.iuv_est.final_pred.predict_partial_iuv(stn_part_maps(xd, theta, True))          # [B,504,S,S]
        idx = out['predict_uv_index']
        onehot = F.one_hot(idx.argmax(1), 25).permute(0, 3, 1, 2).float()
        iuv_map = torch.cat([out['predict_u'] * onehot, out['predict_v'] * onehot, onehot], 1)
        cam_shape, _ = self.body_net(iuv_map)
        _, x4 = self.limb_net(part.reshape(B * 24, 21, S, S))
        rot = self.limb_reslayer(x4.reshape(B, 24 * 256, x4.shape[-2], x4.shape[-1]))
        return out, part, cam_shape, rot


def train_step_cpu(B, size, reps=1):
    """Seconds per fwd + bwd + Adam step of StepNets on the host cores (fp32, stock torch)."""
    import time
    torch.manual_seed(0)
    net = StepNets().train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    img = torch.randn(B, 3, size, size)

    def step():
        opt.zero_grad(set_to_none=True)
        out, part, cam_shape, rot = net(img)
        loss = sum(out[k].float().mean() for k in ('predict_u', 'predict_v', 'predict_uv_index', 'predict_ann_index', 'predict_hm'))
        (loss + part.mean() + cam_shape.mean() + rot.mean()).backward()
        opt.step()
    step()                                                                               # warm-up (allocator, thread pool)
    t0 = time.time()
    for _ in range(reps):
        step()
    return (time.time() - t0) / reps, sum(p.numel() for p in net.parameters())


def train_step_cpu_sweep(B, size, thread_counts, repeats_best=2):
    """[(threads, seconds per fwd + bwd + Adam step of StepNets)] for every thread count: ONE network and one warm-up step, then
    one timed step per setting, then `repeats_best` more steps at the fastest setting -- its entry becomes the BEST of those
    draws and `spread` their (min, max) (bench.py's cpu_baseline reports both); restores the thread count.
    Returns (results, parameter count, spread)."""
    import time
    torch.manual_seed(0)
    net = StepNets().train()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    img = torch.randn(B, 3, size, size)

    def step():
        opt.zero_grad(set_to_none=True)
        out, part, cam_shape, rot = net(img)
        loss = sum(out[k].float().mean() for k in ('predict_u', 'predict_v', 'predict_uv_index', 'predict_ann_index', 'predict_hm'))
        (loss + part.mean() + cam_shape.mean() + rot.mean()).backward()
        opt.step()
    prev = torch.get_num_threads()
    res = []
    try:
        step()                                                                           # warm-up (allocator, thread pool)
        for n in thread_counts:
            torch.set_num_threads(int(n))
            t0 = time.time()
            step()
            res.append((int(n), time.time() - t0))
        best = min(range(len(res)), key=lambda i: res[i][1])
        draws = [res[best][1]]
        torch.set_num_threads(res[best][0])
        for _ in range(int(repeats_best)):
            t0 = time.time()
            step()
            draws.append(time.time() - t0)
        res[best] = (res[best][0], min(draws))
    finally:
        torch.set_num_threads(prev)
    return res, sum(p.numel() for p in net.parameters()), (min(draws), max(draws))
