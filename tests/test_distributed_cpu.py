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


def test_balanced_slabs_by_work():
    """distributed.balanced_slabs (the split bench.py and assemble_rows use in 3D): the slabs tile the axis, the modelled
    cost grows with every plane, the bottleneck is within 10 % of the lightest slab at the BASELINE sizes and never worse
    than the even split; scaling_model reports what the bench line carries."""
    from pyiga_amd import distributed as D
    for N, p in ((132, 4), (101, 5), (66, 2), (40, 3), (9, 4)):
        whole = D.slab_cost(N, p, 0, N)
        costs = [D.slab_cost(N, p, 0, b) for b in range(1, N + 1)]
        assert all(c1 > c0 for c0, c1 in zip(costs, costs[1:]))                # monotone in the planes added
        assert abs(costs[-1] - whole) < 1e-9
        for W in (1, 2, 3, 4, 8):
            if W > N:
                continue
            e = D.balanced_slabs(N, W, p)
            assert len(e) == W + 1 and e[0] == 0 and e[-1] == N
            assert all(b > a for a, b in zip(e, e[1:]))                          # every rank owns at least one plane
            assert [D.slab_range(N, r, W, p) for r in range(W)] == list(zip(e, e[1:]))
            c = [D.slab_cost(N, p, a, b) for a, b in zip(e, e[1:])]
            even = [D.slab_cost(N, p, *D.slab_range(N, r, W)) for r in range(W)]
            assert max(c) <= max(even) + 1e-9                                    # never worse than the even split
            m = D.scaling_model(N, W, p)
            assert m['edges'] == [int(x) for x in e]
            assert abs(m['predicted_speedup'] - whole / max(c)) < 1e-3
            assert 0.0 <= m["halo_fraction"] <= (W - 1) * p / (N - p) + 1e-4
    # BASELINE config 4 (132 dof planes, p = 4) and config 5 (101, p = 5)
    for N, p in ((132, 4), (101, 5)):
        for W in (2, 4, 8):
            e = D.balanced_slabs(N, W, p)
            c = [D.slab_cost(N, p, a, b) for a, b in zip(e, e[1:])]
            assert max(c) / min(c) <= 1.10, (N, p, W, e, c)
    m8 = D.scaling_model(132, 8, 4)
    assert m8['edges'] == [0, 18, 34, 50, 66, 82, 98, 114, 132]                  # 18, 16 x 6, 18 planes
    assert abs(m8["halo_fraction"] - 7 * 4 / 128) < 1e-4
    assert 6.0 < m8['predicted_speedup'] < 8.0
