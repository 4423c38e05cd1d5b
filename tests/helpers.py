"""Shared builders for the parity tests: the same seeded inputs go to the oracle (numpy) and to
the product (C ABI through lrvb_amd)."""
import numpy as np

from oracle import packing as opk
from oracle import models as om

LOSS_NAME = {om.GAUSSIAN: 'gaussian', om.LOGISTIC: 'logistic', om.POISSON: 'poisson'}


def make_par(vb, spec):
    """spec: list of tuples ('box', name, n, lb, ub) | ('psd', name, k, diag_lb) | ('simplex', name, rows, K).
    Returns (product ModelParamsDict, oracle Layout)."""
    par = vb.ModelParamsDict('par')
    blocks = []
    for s in spec:
        if s[0] == 'box':
            _, name, n, lb, ub = s
            par.push_param(vb.VectorParam(name, n, lb=lb, ub=ub))
            blocks.append(opk.box_block(n, lb, ub, name))
        elif s[0] == 'psd':
            _, name, k, diag_lb = s
            par.push_param(vb.PosDefMatrixParam(name, k, diag_lb=diag_lb))
            blocks.append(opk.psd_block(k, diag_lb, name))
        else:
            _, name, rows, K = s
            par.push_param(vb.SimplexParam(name, (rows, K)))
            blocks.append(opk.simplex_block(rows, K, name))
    return par, opk.Layout(blocks)


def glm_data(rng, N, P, loss, scale=None):
    scale = (1.0 / np.sqrt(P)) if scale is None else scale
    x = rng.normal(size=(N, P)) * scale
    w = rng.uniform(0.5, 1.5, size=N)
    if loss == om.GAUSSIAN:
        y = rng.normal(size=N)
    elif loss == om.LOGISTIC:
        y = rng.integers(0, 2, size=N).astype(np.float64)
    else:
        y = rng.poisson(1.0, size=N).astype(np.float64)
    return x, y, w


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    denom = max(np.max(np.abs(b)), 1e-300)
    return float(np.max(np.abs(a - b)) / denom)


def on_torch_stream(ctx, torch_device=None):
    """Run `ctx` on torch's current stream, so that tensors torch produces and the `_dev` entry points that read or
    write them through raw pointers are ordered on ONE stream (include/lrvb_hip.h, "stream ordering", rule 2).  Every
    GPU test that hands a torch pointer to the library calls this (or brackets the calls with wait_stream /
    stream_wait); tests/test_gpu_streams.py covers the default private-stream rule on its own."""
    import torch
    ctx.set_stream(torch.cuda.current_stream(torch_device).cuda_stream)
    return ctx
