"""world_size-2 gloo tests (CPU) of the flat gradient store / bucketed all-reduce (distributed.GradStore): N-rank
averaged gradients == mean of the single-process gradients on the same shards (fp32, 1e-6); unused parameters are
reduced as zeros, the collectives run in bucket order on every rank; parameters/buffers are broadcast from rank 0.
The real Trainer path on two processes is tests/test_gpu_models.py::test_two_process_trainer_gradient_allreduce."""
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
    from danet_densepose2smpl_amd.distributed import GradStore
    torch.manual_seed(100 + rank)                # different init per rank: broadcast must fix it
    net = _Net()
    st = GradStore(net.parameters(), bucket_mb=0.0002, device=torch.device('cpu'))     # tiny buckets -> several collectives
    assert len(st.buckets) > 2 and st.world == world and st.grad_scale == 1.0 / world
    st.broadcast_parameters(net)
    g = torch.Generator().manual_seed(7)
    data = torch.randn(world, 6, 8, generator=g)
    for step in range(2):
        net.zero_grad(set_to_none=True)
        st.begin_step()
        net(data[rank]).pow(2).mean().backward()
        # bucket by bucket, in bucket order on every rank (the trainer interleaves the weight-gradient launches here)
        for bi in range(len(st.buckets)):
            st.collect(bi)
            st.reduce_bucket(bi)
        st.wait()
        st.scale_()
        st.attach_all()
    for p in net.parameters():                   # every gradient is a view of the one flat buffer
        assert p.grad.data_ptr() == st.grad_ptr(p)
    torch.save({'grads': {k: p.grad.clone() for k, p in net.named_parameters()},
                'params': {k: p.detach().clone() for k, p in net.named_parameters()}}, os.path.join(tmp, 'r%d.pt' % rank))
    # the one-call form used after a hipGraph replay whose communication could not be captured
    net.zero_grad(set_to_none=True)
    st.begin_step()
    net(data[rank]).pow(2).mean().backward()
    st.reduce_all()
    st.scale_()
    st.attach_all()
    torch.save({k: p.grad.clone() for k, p in net.named_parameters()}, os.path.join(tmp, 'n%d.pt' % rank))
    # segmented backward (segments.py): complete buckets are released BETWEEN two segments, strictly in bucket order, from this
    # thread; never-used parameters are not waited for after the first step; the last bucket (it carries the usage mask) always
    # goes out in the tail; and the bf16 wire format
    from danet_densepose2smpl_amd import segments
    # grouped: the buckets that are complete together go out as ONE run (release_ready_group / reduce_buckets: one collective over their
    # contiguous slice of the store) -- what the Trainer does since round 6; the results must not depend on the grouping
    for wire, grouped in ((torch.float32, False), (torch.bfloat16, False), (torch.float32, True), (torch.bfloat16, True)):
        torch.manual_seed(300 + rank)                # the same initial parameters for the four runs
        net2 = _Net()
        st2 = GradStore(net2.parameters(), bucket_mb=0.0002, device=torch.device('cpu'), wire_dtype=wire)
        st2.broadcast_parameters(net2)
        ncoll = []

        def release(bi, st2=st2):
            st2.collect(bi)
            st2.reduce_bucket(bi)
            ncoll.append(1)

        def release_run(b0, b1, st2=st2):
            st2.collect(range(b0, b1))
            st2.reduce_buckets(b0, b1)
            ncoll.append(b1 - b0)
        log = []
        for step in range(3):
            net2.zero_grad(set_to_none=True)
            st2.begin_step()
            segments.begin(True)
            h = segments.cut(torch.relu(net2.bn(net2.a(data[rank]))))           # segment 0: a, bn | segment 1: b
            loss = {'l': net2.b(h).pow(2).mean()}
            assert segments.level() == 1
            st2.backward_scope(True)
            del ncoll[:]
            if grouped:
                segments.backward(loss, lambda k: st2.release_ready_group(release_run))
            else:
                segments.backward(loss, lambda k: st2.release_ready(release))
            st2.backward_scope(False)
            segments.end()
            if grouped:
                release_run(st2.next_bucket(), len(st2.buckets))
                assert sum(ncoll) == len(st2.buckets) and len(ncoll) <= 2          # one run between the segments (from step 1 on), one in the tail
            else:
                for bi in range(st2.next_bucket(), len(st2.buckets)):
                    release(bi)
            st2.wait()
            used = st2.used.clone()
            st2.scale_()
            st2.attach_all()
            log.append((list(st2.issued), st2.issued_early, used))
        torch.save({'log': log, 'nb': len(st2.buckets), 'grads': {k: p.grad.clone() for k, p in net2.named_parameters()},
                    'params': {k: p.detach().clone() for k, p in net2.named_parameters()},
                    'order': [k for p in st2.params for k, q in net2.named_parameters() if q is p]},
                   os.path.join(tmp, 'e%d_%s%s.pt' % (rank, 'bf16' if wire == torch.bfloat16 else 'fp32', '_grouped' if grouped else '')))
    # a parameter that receives a gradient on ONE rank only: the mask is summed with the last bucket, so both ranks see it in use
    net3 = _Net()
    st3 = GradStore(net3.parameters(), bucket_mb=0.0002, device=torch.device('cpu'))
    st3.broadcast_parameters(net3)
    net3.zero_grad(set_to_none=True)
    st3.begin_step()
    st3.backward_scope(True, early=False)
    y = net3(data[rank]).pow(2).mean()
    if rank == 1:
        y = y + net3.unused(torch.ones(2, 3)).sum()
    y.backward()
    st3.backward_scope(False)
    st3.reduce_all()
    torch.save({'used': st3.used.clone(), 'order': [k for p in st3.params for k, q in net3.named_parameters() if q is p]},
               os.path.join(tmp, 'u%d.pt' % rank))
    # the poison word: rank 1's one-pass BatchNorm barrier "timed out" in step 1 and stays set (the word is sticky);
    # the word rides in the tail of the last bucket, so after that bucket's all-reduce BOTH ranks read a sum > 0 -- which is what
    # the Adam kernel tests (csrc/adam.hip poison_sum) -- for both wire formats; step 0 reads 0 on both
    sums = {}
    for wire in (torch.float32, torch.bfloat16):
        net4 = _Net()
        st4 = GradStore(net4.parameters(), bucket_mb=0.0002, device=torch.device('cpu'), wire_dtype=wire)
        st4.broadcast_parameters(net4)
        word = torch.zeros(1, dtype=torch.int32)
        st4.poison_src = word
        seen = []
        for step in range(3):
            net4.zero_grad(set_to_none=True)
            st4.begin_step()
            st4.backward_scope(True, early=False)
            net4(data[rank]).pow(2).mean().backward()
            if rank == 1 and step == 1:
                word.fill_(1)
            st4.backward_scope(False)
            st4.stamp_poison()
            st4.reduce_all()
            seen.append(float(st4.poison))
        sums['bf16' if wire == torch.bfloat16 else 'fp32'] = seen
    torch.save(sums, os.path.join(tmp, 'p%d.pt' % rank))
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
    for r in (0, 1):                                                   # in use on rank 1 only -> in use everywhere (count 1), the rest count 2
        u = torch.load(tmp_path / ('u%d.pt' % r))
        for name, c in zip(u['order'], u['used'].tolist()):
            assert c == (1.0 if name.startswith('unused') else 2.0), (r, name, c)
    for r in (0, 1):                                                   # the poison word is a global fact as well: set on rank 1 from step 1 on
        assert torch.load(tmp_path / ('p%d.pt' % r)) == {'fp32': [0.0, 1.0, 1.0], 'bf16': [0.0, 1.0, 1.0]}, r
    n0 = torch.load(tmp_path / 'n0.pt')
    for k in n0:
        assert torch.allclose(n0[k], (grads[0][k] + grads[1][k]) / 2, atol=1e-6), k
    # segmented backward (fp32 and bf16 wire): every step issues all buckets in index order on both ranks; the first step cannot
    # release past the never-used module, later steps release the buckets of segment 1 (the layer behind the cut) between the
    # segments; the usage mask counts the ranks
    for wire, tol in (('fp32', 1e-6), ('bf16', 2e-2), ('fp32_grouped', 1e-6), ('bf16_grouped', 2e-2)):
        e0, e1 = torch.load(tmp_path / ('e0_%s.pt' % wire)), torch.load(tmp_path / ('e1_%s.pt' % wire))
        nb = e0['nb']
        for e in (e0, e1):
            for issued, early, used in e['log']:
                assert issued == list(range(nb))
                for name, u in zip(e['order'], used.tolist()):
                    assert u == (0.0 if name.startswith('unused') else 2.0), (name, u)
            assert e['log'][0][1] == 0 and 1 <= e['log'][1][1] < nb and e['log'][2][1] == e['log'][1][1]
        assert e0['log'][1][1] == e1['log'][1][1]
        net = _Net()
        net.load_state_dict({**net.state_dict(), **e0['params']})
        ref = []
        for r in range(world):
            net.zero_grad(set_to_none=True)
            net(data[r]).pow(2).mean().backward()
            ref.append({k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for k, p in net.named_parameters()})
        for k in e0['grads']:
            mean = (ref[0][k] + ref[1][k]) / 2
            assert torch.allclose(e0['grads'][k], mean, atol=tol * float(mean.abs().max() + 1e-6) if wire.startswith('bf16') else tol), (wire, k)
            assert torch.equal(e0['grads'][k], e1['grads'][k]), (wire, k)
    # grouping changes how many collectives carry the buckets, not one bit of the result
    for wire in ('fp32', 'bf16'):
        a, b = torch.load(tmp_path / ('e0_%s.pt' % wire)), torch.load(tmp_path / ('e0_%s_grouped.pt' % wire))
        assert all(torch.equal(a['grads'][k], b['grads'][k]) for k in a['grads']), wire


def test_grad_store_usage_mask_and_single_handout():
    """GradStore on CPU tensors (no process group): `used` marks the parameters that received a gradient in the last backward
    pass (the optimizer skips the others, optim.FusedAdam), `take` hands a parameter's slot out once per step (shared weights)."""
    sys.path.insert(0, ROOT)
    from danet_densepose2smpl_amd.distributed import GradStore
    net = _Net()
    st = GradStore(net.parameters(), device=torch.device('cpu'), world=1)
    names = {id(p): k for k, p in net.named_parameters()}
    for step in range(2):
        net.zero_grad(set_to_none=True)
        st.begin_step()
        st.backward_scope(True, early=False)
        net(torch.randn(6, 8)).pow(2).mean().backward()
        st.backward_scope(False)
        used = {names[id(p)]: int(st.used[st.index_of(p)]) for p in st.params}
        assert used['unused.weight'] == 0 and used['unused.bias'] == 0
        assert all(v == 1 for k, v in used.items() if not k.startswith('unused')), used
        w = net.a.weight
        v = st.take(w)
        assert v is not None and v.data_ptr() == st.grad_ptr(w) and st.take(w) is None      # the second request of a step gets no slot
    st.begin_step()
    assert st.take(net.a.weight) is not None                                              # ... a new step does

