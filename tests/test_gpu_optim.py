"""GPU parity of the one-launch Adam (csrc/adam.hip) against torch.optim.Adam on the same seeded tensors."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fused_adam_matches_torch_adam():
    from danet_densepose2smpl_amd.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(48, 48, 3, 3), (7,), (100003,), (64, 21, 1, 1), (3, 5), (40000, 3)]
    pa = [torch.nn.Parameter(torch.randn(s, device='cuda')) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    lr = torch.tensor(1e-2, device='cuda')
    oa = FusedAdam(pa, lr=lr.clone())
    ob = torch.optim.Adam(pb, lr=1e-2, betas=(0.9, 0.999), eps=1e-8)
    for step in range(5):
        gs = [torch.randn(s, device='cuda') * (0.1 + step) for s in shapes]
        for i, (a, b, g) in enumerate(zip(pa, pb, gs)):
            a.grad = None if (i == 1 and step == 2) else g.clone()       # one parameter skips one step
            b.grad = None if (i == 1 and step == 2) else g.clone()
        v0 = [p._version for p in pa]
        oa.step()
        ob.step()
        assert all(p._version > v for p, v, q in zip(pa, v0, pa) if q.grad is not None)
        if step == 3:
            oa.param_groups[0]['lr'].mul_(0.1)                             # the manual decay of trainer.py:120-128
            ob.param_groups[0]['lr'] *= 0.1
    torch.cuda.synchronize()
    for a, b in zip(pa, pb):
        if a.numel() == 7:
            continue          # torch keeps a per-parameter step count (the skipped step shifts its bias correction); ours is global
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6), (a.shape, (a - b).abs().max().item())


def test_fused_adam_state_dict_is_torch_adam_compatible():
    """state_dict() uses torch.optim.Adam's layout: a torch Adam continues from it exactly like we do, and back."""
    from danet_densepose2smpl_amd.optim import FusedAdam
    torch.manual_seed(1)
    shapes = [(16, 8, 3, 3), (33,), (5, 7)]
    pa = [torch.nn.Parameter(torch.randn(s, device='cuda')) for s in shapes]
    oa = FusedAdam(pa, lr=3e-3)
    for _ in range(3):
        for p in pa:
            p.grad = torch.randn_like(p)
        oa.step()
    sd = oa.state_dict()
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    ob = torch.optim.Adam(pb, lr=1.0)
    ob.load_state_dict(sd)
    pc = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oc = FusedAdam(pc, lr=1.0)
    oc.load_state_dict(ob.state_dict())
    gs = [torch.randn_like(p) for p in pa]
    for ps, o in ((pa, oa), (pb, ob), (pc, oc)):
        for p, g in zip(ps, gs):
            p.grad = g.clone()
        o.step()
    torch.cuda.synchronize()
    for a, b, c in zip(pa, pb, pc):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6) and torch.allclose(a, c, rtol=1e-6, atol=1e-7)
