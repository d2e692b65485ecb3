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
