"""Gauss-Legendre rules on the spans of a knot vector.

Host-side plumbing only: the device builds its own copy of the nodes from the same reference rule
(``igx_patch_desc.gauss_x/gauss_w``) with the same ``half * x + mid`` arithmetic, so host and device agree
bit for bit with the reference's nodes (pyiga/quadrature.py:3-23).
"""
import functools

import numpy as np


@functools.lru_cache(maxsize=None)
def _reference_rule(npoints):
    x, w = np.polynomial.legendre.leggauss(npoints)
    x.setflags(write=False)
    w.setflags(write=False)
    return x, w


def gauss_rule(deg, a, b):
    """`deg`-point rule on every interval ``(a[k], b[k])``; flat arrays, interval-major."""
    lo, hi = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    half, mid = (hi - lo) * 0.5, (lo + hi) * 0.5
    x, w = _reference_rule(int(deg))
    return (half[:, None] * x[None, :] + mid[:, None]).reshape(-1), (half[:, None] * w[None, :]).reshape(-1)


def make_iterated_quadrature(intervals, nqp):
    """Rule over all spans of the break points `intervals` with `nqp` points per span."""
    pts = np.asarray(intervals, dtype=float)
    return gauss_rule(nqp, pts[:-1], pts[1:])


def make_tensor_quadrature(meshes, nqp):
    """Per-axis nodes and weights of the tensor rule: ``(nodes_0, ...), (weights_0, ...)``."""
    rules = [make_iterated_quadrature(m, nqp) for m in meshes]
    return tuple(r[0] for r in rules), tuple(r[1] for r in rules)
