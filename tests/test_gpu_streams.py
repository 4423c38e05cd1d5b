"""The stream-ordering contract of include/lrvb_hip.h for entry points that take DEVICE pointers.

Round 2's driver run went red on exactly this: a test cloned a statistics buffer on torch's stream while
`lrvb_hessian_partial_dev` was still writing it on the context's private (then non-blocking) stream.  The contract
is now written down and the default is safe: (1) the private stream is a BLOCKING stream, ordered against the legacy
default stream in both directions; (2) any other caller stream needs `set_stream`, or the event hand-offs
`wait_stream` / `stream_wait`.  Each test delays the producer behind ~100 ms of matrix products so that a missing
ordering edge shows up as wrong data rather than passing by luck.
"""
import numpy as np
import pytest

from oracle import models as om
from helpers import make_par, glm_data, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vb():
    import lrvb_amd
    return lrvb_amd


def _setup(vb, N=30000, P=256):
    rng = np.random.default_rng(5)
    spec = [('box', 'u', P - 64, -np.inf, np.inf), ('box', 'pos', 64, 0.0, np.inf)]
    par, lay = make_par(vb, spec)
    x, y, w = glm_data(rng, N, P, om.LOGISTIC)
    fun = vb.GLMObjective(par, x, y, loss='logistic', prior_info=0.3, weights=w)
    model = om.DeclaredModel(lay, loss=om.LOGISTIC, x=x, y=y, w=w, quad_A=np.full(P, 0.3))
    theta = rng.normal(size=P) * 0.2
    return fun.ctx, model, theta, P


def _busy(torch, dev, ms_target=100):
    """Queue roughly ms_target of work on the current stream (fp64 4096^3 products, ~2.5 ms each at 55 TFLOP/s)."""
    a = torch.ones((4096, 4096), dtype=torch.float64, device=dev) * 1e-4
    for _ in range(max(1, ms_target // 3)):
        a = a @ a
    return a


def test_default_private_stream_orders_against_the_default_stream(vb):
    """Rule 1: nothing but the library's default.  theta is WRITTEN on the default stream behind a long queue, read by
    hessian_dev on the context's stream, and the result is consumed on the default stream at once -- no host sync, no
    set_stream, no hand-off."""
    import torch
    dev = torch.device('cuda', 0)
    ctx, model, theta, P = _setup(vb)
    want = model.hessian(theta)
    th = torch.full((P,), 7.0, dtype=torch.float64, device=dev)           # a wrong point until the copy below lands
    H = torch.full((P, P), float('nan'), dtype=torch.float64, device=dev)
    src = torch.tensor(theta, device=dev)
    torch.cuda.synchronize()
    keep = _busy(torch, dev)
    th.copy_(src)                                                        # queued behind ~100 ms of products
    ctx.hessian_dev(th.data_ptr(), H.data_ptr(), P)                       # private stream: must wait for the copy
    snap = H.clone()                                                     # default stream: must wait for the build
    th.fill_(7.0)                                                        # ... and this write must not overtake the build's read
    torch.cuda.synchronize()
    assert rel_err(snap.cpu().numpy(), want) < 1e-11
    assert keep.shape == (4096, 4096)


def test_side_stream_with_event_hand_offs(vb):
    """Rule 2: the caller works on a non-default (non-blocking) torch stream; lrvb_ctx_wait_stream before and
    lrvb_stream_wait_ctx after order the context against it without a host synchronisation."""
    import torch
    dev = torch.device('cuda', 0)
    ctx, model, theta, P = _setup(vb)
    want = model.hessian(theta)
    src = torch.tensor(theta, device=dev)
    th = torch.full((P,), 7.0, dtype=torch.float64, device=dev)
    H = torch.full((P, P), float('nan'), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        keep = _busy(torch, dev)
        th.copy_(src)
        ctx.wait_stream(side.cuda_stream)
        ctx.hessian_dev(th.data_ptr(), H.data_ptr(), P)
        ctx.stream_wait(side.cuda_stream)
        snap = H.clone()
    side.synchronize()
    assert rel_err(snap.cpu().numpy(), want) < 1e-11
    # and the third form of rule 2: the context adopted onto the caller's stream
    H.fill_(float('nan')); th.fill_(7.0)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        ctx.set_stream(side.cuda_stream)
        keep = _busy(torch, dev)
        th.copy_(src)
        ctx.hessian_dev(th.data_ptr(), H.data_ptr(), P)
        snap2 = H.clone()
    side.synchronize()
    ctx.set_stream(None)
    assert torch.equal(snap2, snap)
    assert keep.shape == (4096, 4096)


def test_host_pointer_entry_points_are_complete_on_return(vb):
    """Rule 3: host-pointer calls synchronise the context's stream themselves -- also right after `_dev` work."""
    import torch
    dev = torch.device('cuda', 0)
    ctx, model, theta, P = _setup(vb)
    th = torch.tensor(theta, device=dev)
    H = torch.empty((P, P), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    ctx.hessian_dev(th.data_ptr(), H.data_ptr(), P)
    g = ctx.grad(theta)                                                  # queued after the build, complete on return
    assert rel_err(g, model.grad(theta)) < 1e-11
    ctx.sync()
    assert rel_err(H.cpu().numpy(), model.hessian(theta)) < 1e-11
