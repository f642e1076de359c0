import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    """Lazy loader for tests/golden/golden_<name>.npz (made by make_golden.py from the reference)."""
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, 'golden_%s.npz' % name))
        return cache[name]
    return load


@pytest.fixture(scope='session')
def oracle():
    from oracle import iga_oracle
    iga_oracle.build()
    return iga_oracle


def golden_csr(g, name):
    import scipy.sparse
    shape = tuple(int(x) for x in g[name + '_shape'])
    return scipy.sparse.csr_matrix((g[name + '_data'], g[name + '_indices'], g[name + '_indptr']), shape=shape)


def rel_maxdiff(A, B):
    """max_ij |A-B| / max_ij |B|  -- the parity norm of SURVEY section 8c."""
    D = abs(A - B)
    return (D.max() if D.nnz else 0.0) / abs(B).max()
