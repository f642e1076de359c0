"""N>1 path on CPU: two gloo ranks shard the rows, each produces its block, rank 0 gathers and the
result equals the single-process matrix.  The block producer is the CPU oracle here (no GPU in
this container); on the GPU box tests/test_distributed_gpu.py runs the same worker with the HIP path."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_rank_row_sharding(tmp_path, oracle):
    """Same worker as the GPU test (tests/_dist_worker.py): slab ranges, assemble_rows, gather_matrix are the product's."""
    import subprocess
    import sys
    out = str(tmp_path / 'full.npz')
    port = str(_free_port())
    here = os.path.dirname(os.path.abspath(__file__))
    procs = [subprocess.Popen([sys.executable, os.path.join(here, '_dist_worker.py'), str(r), '2', port, 'cpu', out])
             for r in range(2)]
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    z = np.load(out)
    full = scipy.sparse.csr_matrix((z['data'], z['indices'], z['indptr']), shape=tuple(z['shape']))
    kv = oracle.make_knots(2, 0.0, 1.0, 5)
    ref = oracle.assemble('stiffness', (kv,) * 3, oracle.geo_cylinder())
    assert full.shape == ref.shape and full.nnz == ref.nnz
    assert abs(full - ref).max() == 0.0


def test_slab_partition_tiles_rows():
    from pyiga_amd import bspline, distributed
    kv0 = bspline.make_knots(4, 0.0, 1.0, 128 * 8)
    kv = bspline.make_knots(4, 0.0, 1.0, 128)
    for world in (1, 2, 3, 4, 8):
        edges = [distributed.slab_range(kv0.numdofs, r, world) for r in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == kv0.numdofs
        assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
        sizes = [b - a for a, b in edges]
        assert max(sizes) - min(sizes) <= 1
        rows = [distributed.row_range((kv0, kv, kv), r, world) for r in range(world)]
        assert rows[-1][1] == kv0.numdofs * kv.numdofs ** 2
    with pytest.raises(AssertionError):
        distributed.slab_range(4, 0, 8)
