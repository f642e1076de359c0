"""Host-side helpers with the names and behaviour of the few ``pyiga.utils`` functions this path uses
(pyiga/utils.py:8-60): sampling a function on a tensor grid and reading the test fixtures."""
import numpy as np
import scipy.sparse


def read_sparse_matrix(fname):
    """Reference fixture format: a header line, then 1-based ``row col value`` triples; returns CSR."""
    triples = np.loadtxt(fname, skiprows=1)
    r = triples[:, 0].astype(np.int64) - 1
    c = triples[:, 1].astype(np.int64) - 1
    return scipy.sparse.coo_matrix((triples[:, 2], (r, c))).tocsr()


def _on_grid(values, shape):
    """Array of grid samples ``shape + component axes`` from whatever a user function returned: a scalar, an
    array that is missing the axes of arguments the function ignored, or a tuple of those (vector-valued)."""
    if isinstance(values, tuple):
        comps = [np.broadcast_to(np.asanyarray(c), shape) for c in values]
        return np.stack(comps, axis=-1)
    values = np.asanyarray(values)
    trailing = values.shape[len(shape):]
    return values if values.shape == shape + trailing else np.broadcast_to(values, shape + trailing)


def grid_eval(f, grid):
    """Samples of `f` on the tensor grid `grid` (axes in (z, y, x) order).  Spline-like objects evaluate
    themselves; plain callables take the coordinates as ``f(x, y[, z])`` -- last grid axis first."""
    if hasattr(f, 'grid_eval'):
        return f.grid_eval(grid)
    shape = tuple(len(ax) for ax in grid)
    d = len(grid)
    coords = [np.asarray(ax).reshape((1,) * k + (-1,) + (1,) * (d - 1 - k)) for k, ax in enumerate(grid)]
    return _on_grid(f(*coords[::-1]), shape)


def grid_eval_transformed(f, grid, geo):
    """Samples of `f` at the images of the grid points under the geometry map `geo`."""
    shape = tuple(len(ax) for ax in grid)
    pts = grid_eval(geo, grid)                       # shape + (dim,)
    return _on_grid(f(*np.moveaxis(pts, -1, 0)), shape)
