"""Iterated Gauss-Legendre quadrature (pyiga/quadrature.py:3-23): integer/array plumbing on the
host; the same numpy rule is handed to libigx so both sides use identical nodes."""
import numpy as np


def gauss_rule(deg, a, b):
    """Nodes and weights of the `deg`-point Gauss rule on each interval ``(a[k], b[k])``."""
    m = 0.5 * (a + b)
    h = 0.5 * (b - a)
    x, w = np.polynomial.legendre.leggauss(deg)
    nodes = np.outer(h, x) + m[:, np.newaxis]
    weights = np.outer(h, w)
    return nodes.ravel(), weights.ravel()


def make_iterated_quadrature(intervals, nqp):
    return gauss_rule(nqp, intervals[:-1], intervals[1:])


def make_tensor_quadrature(meshes, nqp):
    gauss = tuple(make_iterated_quadrature(mesh, nqp) for mesh in meshes)
    return tuple(g[0] for g in gauss), tuple(g[1] for g in gauss)
