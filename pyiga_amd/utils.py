"""Small helpers mirrored from pyiga/utils.py."""
import numpy as np
import scipy.sparse


def read_sparse_matrix(fname):
    """Read the reference's text fixture format (pyiga/utils.py:54-60): first line skipped,
    then 1-based ``i j value`` triples."""
    I, J, vals = np.loadtxt(fname, skiprows=1, unpack=True)
    I = I.astype(int) - 1
    J = J.astype(int) - 1
    return scipy.sparse.coo_matrix((vals, (I, J))).tocsr()


def _broadcast_to_grid(X, grid_shape):
    X = np.asanyarray(X)
    target_shape = grid_shape + X.shape[len(grid_shape):]
    if X.shape != target_shape:
        X = np.broadcast_to(X, target_shape)
    return X


def _ensure_grid_shape(values, grid):
    """Tuples become a trailing component axis; results that ignore some arguments are broadcast to
    the full grid (pyiga/utils.py:17-31)."""
    grid_shape = tuple(len(g) for g in grid)
    if isinstance(values, tuple):
        values = np.stack(tuple(_broadcast_to_grid(v, grid_shape) for v in values), axis=-1)
    return _broadcast_to_grid(values, grid_shape)


def grid_eval(f, grid):
    """Evaluate `f` over the tensor grid `grid` (axes in (z, y, x) order; `f` takes (x, y, z)).
    pyiga/utils.py:33-41."""
    if hasattr(f, 'grid_eval'):
        return f.grid_eval(grid)
    mesh = list(np.meshgrid(*grid, sparse=True, indexing='ij'))
    mesh.reverse()
    return _ensure_grid_shape(f(*mesh), grid)


def grid_eval_transformed(f, grid, geo):
    """Evaluate `f` at the images of the grid points under `geo` (pyiga/utils.py:43-52)."""
    trf_grid = grid_eval(geo, grid)
    X = tuple(trf_grid[..., i] for i in range(trf_grid.shape[-1]))
    return _ensure_grid_shape(f(*X), grid)
