"""Graph convolution over the 24 SMPL joints (/root/reference/models/module/GCN.py:12-92) and the
skeleton adjacencies (/root/reference/utils/graph.py).  Tiny [24,24]@[B,24,C]@[C,C'] products:
kept as torch matmuls (latency-bound, <0.01 GMAC), but without the reference's 24 host syncs per
forward in normalize_undigraph (graph.py:251-253)."""
import numpy as np
import torch
import torch.nn as nn

SMPL_LINKS = [(0, 1), (1, 4), (4, 7), (7, 10), (0, 2), (2, 5), (5, 8), (8, 11), (0, 3), (3, 6), (6, 9),
              (9, 13), (13, 16), (16, 18), (18, 20), (20, 22), (9, 14), (14, 17), (17, 19), (19, 21), (21, 23),
              (9, 12), (12, 15)]
SMPL_2NEIGH_EXTRA1 = [(12, 17), (12, 16)]
SMPL_2NEIGH_LINKS2 = [(0, 4), (0, 5), (0, 6), (2, 8), (1, 7), (5, 11), (4, 10), (3, 9), (6, 12), (9, 15),
                      (6, 13), (9, 16), (13, 18), (16, 20), (18, 22), (6, 14), (9, 17), (14, 19), (17, 21), (19, 23)]


def adjacency(layout):
    """graph.py:74-106 with max_hop=1, strategy 'uniform', norm 'none': self links + neighbours."""
    A = np.eye(24)
    if layout == 'smpl':
        links = SMPL_LINKS
    elif layout == 'smpl_2neigh':
        links = SMPL_LINKS + SMPL_2NEIGH_EXTRA1 + SMPL_2NEIGH_LINKS2
    else:
        raise ValueError('Do Not Exist This Layout.')
    for i, j in links:
        A[i, j] = 1
        A[j, i] = 1
    return A[None]


def normalize_undigraph(A):
    """D^-1/2 A D^-1/2 with D = column sums (graph.py:232-261); tensor [...,N,N] or numpy [N,N]."""
    if isinstance(A, np.ndarray):
        Dl = A.sum(0)
        d = np.where(Dl > 0, np.power(np.where(Dl > 0, Dl, 1.0), -0.5), 0.0)
        return d[:, None] * A * d[None, :]
    Dl = A.sum(dim=-2)
    d = torch.where(Dl > 0, Dl.clamp(min=1e-30).pow(-0.5), torch.zeros_like(Dl))
    return d.unsqueeze(-1) * A * d.unsqueeze(-2)


def normalize_digraph(A, AD_mode=True):
    """graph.py:176-229 (numpy path)."""
    Dl = A.sum(0 if AD_mode else 1)
    d = np.where(Dl > 0, 1.0 / np.where(Dl > 0, Dl, 1.0), 0.0)
    return A * d[None, :] if AD_mode else d[:, None] * A


class GraphConvFunction(torch.autograd.Function):
    """y = (adj @ x) @ W (+ bias) for x [B,N,C], adj [N,N] (GCN.py:29-41).  The forward products are what autograd would
    run; the backward is written out because autograd's weight and adjacency gradients are single GEMMs with a tiny
    output and a long reduction (e.g. [256, B*24] x [B*24, 256], [24, B*C] x [B*C, 24]) that the BLAS library runs on ONE
    workgroup (50-180 us each at B = 32): here the reduction is split over the batch (bmm) and summed."""

    @staticmethod
    def forward(ctx, x, adj, weight, bias):
        ax = torch.matmul(adj, x)
        y = torch.matmul(ax, weight)
        if bias is not None:
            y = y + bias
        ctx.save_for_backward(x, adj, weight, ax)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        x, adj, weight, ax = ctx.saved_tensors
        gy = gy.contiguous()
        dx = dadj = dw = db = None
        dax = torch.matmul(gy, weight.t())                                   # [B,N,C]
        if ctx.needs_input_grad[2]:
            dw = torch.bmm(ax.transpose(1, 2), gy).sum(0)                    # split-K over the batch
        if ctx.has_bias and ctx.needs_input_grad[3]:
            db = gy.sum(dim=(0, 1))
        if ctx.needs_input_grad[0]:
            dx = torch.matmul(adj.t(), dax)
        if ctx.needs_input_grad[1]:
            dadj = torch.bmm(dax, x.transpose(1, 2)).sum(0)
        return dx, dadj, dw, db


class GraphConv(nn.Module):
    def __init__(self, input_dim, output_dim, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(input_dim, output_dim))
        self.bias = nn.Parameter(torch.empty(output_dim)) if bias else None

    def forward(self, x, adj):
        if x.dim() == 3 and adj.dim() == 2:
            return GraphConvFunction.apply(x, adj, self.weight, self.bias)
        y = torch.matmul(torch.matmul(adj, x), self.weight)
        return y if self.bias is None else y + self.bias


class GCN(nn.Module):
    def __init__(self, input_dim, hidden_dim, out_dim, num_layers, num_nodes, bn=True, normalize=False):
        super().__init__()
        self.num_layers = num_layers
        dims = [input_dim] + [hidden_dim] * (num_layers - 1) + [out_dim]
        self.gc = nn.ModuleList([GraphConv(dims[i], dims[i + 1]) for i in range(num_layers)])
        self.act = nn.ModuleList([nn.Sequential(nn.BatchNorm1d(num_nodes), nn.ReLU(inplace=True)) if bn else nn.ReLU(inplace=True)
                                  for _ in range(num_layers)])
        for m in self.gc:
            nn.init.xavier_uniform_(m.weight, gain=nn.init.calculate_gain('relu'))
            if m.bias is not None:
                nn.init.constant_(m.bias, 0.0)

    def forward(self, x, A):
        for gc, act in zip(self.gc, self.act):
            x = act(gc(x, A))
        return x
