"""Worker of the two-rank tests (CPU: gloo + oracle block producer; GPU: gloo rendezvous + the HIP path of
pyiga_amd.distributed.assemble_rows on device 0).  Run as  python _dist_worker.py RANK WORLD PORT MODE OUT."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def oracle_block(kind, kvs, geo, row0, device, algo):
    """CPU stand-in for the device block: rows of the oracle matrix."""
    from oracle import iga_oracle as orc
    okvs = tuple(orc.KnotVector(kv.kv, kv.p) for kv in kvs)
    A = orc.assemble(kind, okvs, orc.geo_cylinder() if len(kvs) == 3 else orc.geo_quarter_annulus())
    plane = int(np.prod([kv.numdofs for kv in kvs[1:]]))
    return A[row0[0] * plane:row0[1] * plane].tocsr()


def main():
    rank, world, port, mode, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = port
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pyiga_amd import bspline, distributed, geometry
    p, n = (2, 5) if mode == 'cpu' else (3, 11)
    kv = bspline.make_knots(p, 0.0, 1.0, n)
    kvs = (kv, kv, kv)
    if mode == 'cpu':
        blk = distributed.assemble_rows('stiffness', kvs, None, rank, world, block_fn=oracle_block)
    else:
        geo = geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
        blk = distributed.assemble_rows('stiffness', kvs, geo, rank, world, device=0)
    lo, hi = distributed.row_range(kvs, rank, world)
    assert blk.shape == (hi - lo, kv.numdofs ** 3)
    full = distributed.gather_matrix(blk, dst=0)
    dist.barrier()
    if rank == 0:
        np.savez(out, data=full.data, indices=full.indices, indptr=full.indptr, shape=np.array(full.shape))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
