"""world_size-2 gloo tests (CPU) of the data-parallel gradient reducer: N-rank averaged gradients ==
mean of the single-process gradients on the same shards; unused parameters keep the collective
plan static; parameters/buffers are broadcast from rank 0."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 16)
        self.bn = nn.BatchNorm1d(16)
        self.b = nn.Linear(16, 4)
        self.unused = nn.Linear(3, 3)            # never used in forward (like rot2pos / pos2rot)

    def forward(self, x):
        return self.b(torch.relu(self.bn(self.a(x))))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from danet_densepose2smpl_amd.distributed import GradReducer
    torch.manual_seed(100 + rank)                # different init per rank: broadcast must fix it
    net = _Net()
    red = GradReducer(net, bucket_mb=0.0002, device=torch.device('cpu'))     # tiny buckets -> several collectives
    assert len(red.buckets) > 2
    red.broadcast_parameters()
    g = torch.Generator().manual_seed(7)
    data = torch.randn(world, 6, 8, generator=g)
    for step in range(2):
        net.zero_grad(set_to_none=True)
        red.prepare()
        net(data[rank]).pow(2).mean().backward()
        red.finish()
    torch.save({'grads': {k: p.grad.clone() for k, p in net.named_parameters()},
                'params': {k: p.detach().clone() for k, p in net.named_parameters()}}, os.path.join(tmp, 'r%d.pt' % rank))
    # hook-free path used after a hipGraph replay
    net.zero_grad(set_to_none=True)
    net(data[rank]).pow(2).mean().backward()
    red.remove()
    red2 = GradReducer(net, bucket_mb=1.0, device=torch.device('cpu'))
    red2.remove()                                 # no hooks: gradients already there
    red2.reduce_now()
    torch.save({k: p.grad.clone() for k, p in net.named_parameters()}, os.path.join(tmp, 'n%d.pt' % rank))
    dist.destroy_process_group()


def test_two_rank_gradient_average_matches_single_process(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / 'r0.pt'), torch.load(tmp_path / 'r1.pt')
    for k in r0['params']:
        assert torch.equal(r0['params'][k], r1['params'][k]), k          # broadcast from rank 0
    # single-process reference on the same shards, same (rank-0) parameters
    net = _Net()
    net.load_state_dict({**net.state_dict(), **r0['params']})
    g = torch.Generator().manual_seed(7)
    data = torch.randn(world, 6, 8, generator=g)
    grads = []
    for r in range(world):
        net.zero_grad(set_to_none=True)
        net(data[r]).pow(2).mean().backward()
        grads.append({k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for k, p in net.named_parameters()})
    for k in r0['grads']:
        mean = (grads[0][k] + grads[1][k]) / 2
        assert torch.allclose(r0['grads'][k], mean, atol=1e-6), k
        assert torch.allclose(r1['grads'][k], mean, atol=1e-6), k
    assert r0['grads']['unused.weight'].abs().max() == 0               # unused parameters reduced as zeros
    n0 = torch.load(tmp_path / 'n0.pt')
    for k in n0:
        assert torch.allclose(n0[k], (grads[0][k] + grads[1][k]) / 2, atol=1e-6), k
