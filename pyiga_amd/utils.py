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
