"""Segmented backward pass: how the data-parallel step overlaps its gradient all-reduces with the backward pass proper
(the reference is single-device, /root/reference/train/base_trainer.py:20).

The forward pass marks CUT points (`cut(tensors)`): the tensors are replaced by detached leaves, so the autograd graph of
a step falls apart into independent segments chained only through those leaves -- for HRNet one segment per
HighResolutionModule (hrnet.py), one for the IUV heads + estimator losses, one for the IUV -> SMPL regressor (danet.py).
`backward(losses, between)` then walks the segments from the last one to the first on the CALLING thread: segment k is
`torch.autograd.backward` from the losses that hang off it plus the tensors behind cut k + 1 (seeded with the gradients
that arrived at that cut's leaves); after each segment `between(k)` runs -- the trainer launches the queued weight gradients
of every gradient bucket that is now complete and starts its all-reduce on the communication stream, while the next
segment's backward kernels go to the compute stream.

Why not post-accumulate-grad hooks (rounds 2-3 released buckets from hooks inside ONE loss.backward())?  Hooks run on
autograd's device thread; collectives issued from there raced the process group's own threads (a sporadic SIGABRT, one
start in eight, never explained) and cannot be ordered against a hipGraph capture.  Here every collective, every kernel
launch and every capture-time graph node is issued by one thread in program order: the order is the same on every rank by
construction, a capture records exactly what an eager step issues, and there is nothing to race.

The registry is per forward pass (`begin()`); with `ACTIVE` false `cut` is the identity and `backward` is one ordinary
backward call."""
import torch

ACTIVE = False            # set by the trainer around the forward pass of a step that wants segments

_cuts = []                # [(originals, leaves)] in forward order
_loss_level = {}          # loss key -> number of cuts recorded when the loss was produced (= index of its segment)


def begin(active):
    global ACTIVE
    ACTIVE = bool(active)
    del _cuts[:]
    _loss_level.clear()


def end():
    """Forget the step's cut tensors (they keep their graphs alive)."""
    global ACTIVE
    ACTIVE = False
    del _cuts[:]
    _loss_level.clear()


def level():
    return len(_cuts)


def cut(tensors):
    """Replace every tensor that requires grad by a detached leaf carrying the same data (no copy) and the same host-side
    attributes (fused-operand hand-overs such as `_nhwc_padded` / `_bn_ctx` are plain Python attributes, not graph edges).
    Returns the list to continue the forward pass with."""
    single = torch.is_tensor(tensors)
    ts = [tensors] if single else list(tensors)
    if not (ACTIVE and torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in ts)):
        return tensors
    origs, leaves, out = [], [], []
    for t in ts:
        if torch.is_tensor(t) and t.requires_grad:
            c = t.detach().requires_grad_(True)
            for k, v in t.__dict__.items():
                setattr(c, k, v)
            origs.append(t)
            leaves.append(c)
            out.append(c)
        else:
            out.append(t)
    _cuts.append((origs, leaves))
    return out[0] if single else out


def note_losses(keys):
    """The losses `keys` were produced by the segment that is open now."""
    lv = len(_cuts)
    for k in keys:
        _loss_level.setdefault(k, lv)


def backward(losses, between=None):
    """losses: dict key -> tensor (each a sum term of the step's objective, gradient seed 1).  Runs the segments last to
    first; between(k) is called after segment k (k = number of cuts .. 1; not after segment 0, the caller's tail follows)."""
    n = len(_cuts)
    by = [[] for _ in range(n + 1)]
    for k, v in losses.items():
        by[min(_loss_level.get(k, n), n)].append(v.reshape(-1))
    for k in range(n, -1, -1):
        roots, seeds = [], []
        if by[k]:
            roots.append(torch.cat(by[k]).sum() if len(by[k]) > 1 else by[k][0].sum())
            seeds.append(None)
        if k < n:
            for o, c in zip(*_cuts[k]):
                if c.grad is not None:
                    roots.append(o)
                    seeds.append(c.grad)
                    c.grad = None
        if roots:
            torch.autograd.backward(roots, seeds)
        if between is not None and k > 0:
            between(k)
