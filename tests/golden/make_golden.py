#!/usr/bin/env python3
"""Generates tests/golden/reference_vectors.npz by RUNNING the reference's own code, in the build
container only (the reference does not exist on the GPU box; only the .npz travels).

Only the two reference modules that import without the absent third-party `autograd` package are
used -- LinearResponseVariationalBayes/ConjugateGradient.py and OptimizationUtils.py -- loaded by
file path so that the package __init__ (which imports autograd) is not executed.  No stand-in for
autograd is installed.  One compatibility shim: the reference calls scipy's cg(..., tol=) which
scipy >= 1.14 spells rtol= (legacy `tol` was relative to ||b||, i.e. rtol=tol, atol=0).

Usage: python tests/golden/make_golden.py   (writes next to itself)
"""
import importlib.util
import os
import sys

import numpy as np
import scipy.sparse.linalg

REF = '/root/reference/LinearResponseVariationalBayes'
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def load(name):
    spec = importlib.util.spec_from_file_location('ref_' + name, os.path.join(REF, name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    cg = load('ConjugateGradient')
    opt = load('OptimizationUtils')
    out = {}

    # masks and splits (ConjugateGradient.py:19-57)
    masks = cg.get_masks(20, 3)
    out['masks_20_3'] = np.array(masks)
    vec = np.zeros(23, dtype=bool)
    vec[[1, 2, 5, 8, 13, 14, 20]] = True
    s1, s2 = cg.split_vector(vec)
    out['split_in'], out['split_1'], out['split_2'] = vec, s1, s2
    res = []
    big = np.zeros(64, dtype=bool)
    big[::2] = True
    cg.recursive_split(big, results=res, terminate_len=5)
    out['rsplit_in'], out['rsplit_out'] = big, np.array(res)

    # get_sym_matrix_inv_sqrt with eigenvalue clamping (OptimizationUtils.py:6-20)
    rng = np.random.default_rng(20240)
    a = rng.normal(size=(6, 6))
    h = a @ a.T + 0.1 * np.eye(6) + 0.01 * rng.normal(size=(6, 6))        # slightly asymmetric on purpose
    out['invsqrt_in'] = h
    for tag, kw in (('plain', {}), ('min', {'ev_min': 1.0}), ('max', {'ev_max': 5.0}), ('both', {'ev_min': 1.0, 'ev_max': 5.0})):
        isq, corr = opt.get_sym_matrix_inv_sqrt(h, **kw)
        out['invsqrt_' + tag], out['invsqrt_corr_' + tag] = isq, corr

    # ConjugateGradientSolver on the test_cg problem (test_objectives.py:524-554), closed-form HVP 2 mat v
    orig_cg = scipy.sparse.linalg.cg

    def cg_compat(A, b, x0=None, tol=1e-5, M=None, **kw):
        return orig_cg(A, b, x0=x0, rtol=tol, atol=0.0, M=M, **kw)
    cg.sp.sparse.linalg.cg = cg_compat
    try:
        K = 50
        mat = rng.random((K, K))
        mat = 0.5 * (mat + mat.T) + 10 * np.eye(K)
        loc = np.array([k / 7. for k in range(K)])
        x = loc + 0.1 * rng.random(K)
        solver = cg.ConjugateGradientSolver(lambda x0, v: 2.0 * (mat @ v), loc)
        cg_masks = cg.get_masks(K, 10)
        solver.get_hinv_vec_subsets(x, cg_masks)
        out['cg_mat'], out['cg_loc'], out['cg_x'] = mat, loc, x
        out['cg_masks'] = np.array(cg_masks)
        out['cg_vecs'] = np.array(solver.vecs)
        out['cg_hinv_vecs'] = np.array(solver.hinv_vecs)
        out['cg_infos'] = np.array(solver.cg_infos)
    finally:
        cg.sp.sparse.linalg.cg = orig_cg

    path = os.path.join(HERE, 'reference_vectors.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, {k: np.shape(v) for k, v in out.items()})


if __name__ == '__main__':
    main()
