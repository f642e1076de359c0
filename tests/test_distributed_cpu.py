"""N>1 path on CPU: two gloo ranks shard the rows, each produces its block, rank 0 gathers and the
result equals the single-process matrix.  The block producer is the CPU oracle here (no GPU in
this container); on the GPU box test_gpu_parity.py::test_row_slabs_equal_full checks that the HIP
path produces exactly these blocks."""
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


def _oracle_block(kind, kvs, geo, row0, device, algo):
    """CPU stand-in for the device block: rows of the oracle matrix."""
    from oracle import iga_oracle as orc
    okvs = tuple(orc.KnotVector(kv.kv, kv.p) for kv in kvs)
    A = orc.assemble(kind, okvs, orc.geo_cylinder() if len(kvs) == 3 else orc.geo_quarter_annulus())
    plane = int(np.prod([kv.numdofs for kv in kvs[1:]]))
    return A[row0[0] * plane:row0[1] * plane].tocsr()


def _worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pyiga_amd import bspline, distributed
    kv = bspline.make_knots(2, 0.0, 1.0, 5)
    kvs = (kv, kv, kv)
    blk = distributed.assemble_rows('stiffness', kvs, None, rank, world, block_fn=_oracle_block)
    lo, hi = distributed.row_range(kvs, rank, world)
    assert blk.shape == (hi - lo, kv.numdofs ** 3)
    full = distributed.gather_matrix(blk, dst=0)
    dist.barrier()
    if rank == 0:
        np.savez(out, data=full.data, indices=full.indices, indptr=full.indptr, shape=np.array(full.shape))
    dist.destroy_process_group()


def test_two_rank_row_sharding(tmp_path, oracle):
    import torch.multiprocessing as mp
    out = str(tmp_path / 'full.npz')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
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
