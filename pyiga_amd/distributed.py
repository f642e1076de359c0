"""Multi-GPU assembly: one process per GPU, rows sharded in slabs of axis-0 dof planes.

The path shards without any exchange ("zero-communication
alternative"): rank r owns the dof planes ``[N0*r/W, N0*(r+1)/W)`` of axis 0, i.e. a
contiguous block of CSR rows, and computes every entry of those rows itself -- the lower
triangle directly and the upper triangle as the mirror of lower entries it (re)computes for
the p planes above its slab.  Each rank evaluates the quadrature fields only on the spans its
rows touch, straight from the control net, so no input has to be communicated either.  The
result is bit-for-bit the row block of the single-GPU matrix (tests/test_gpu_parity.py::
test_row_slabs_equal_full), independent of the number of ranks.

``torch.distributed`` (RCCL when the backend is "nccl") is used only to gather results or
timings; there is no collective on the data path.
"""
import numpy as np
import scipy.sparse


def slab_range(ndofs0, rank, world):
    """Dof planes [lo, hi) of axis 0 owned by `rank`; the slabs tile [0, ndofs0) exactly."""
    assert 0 <= rank < world
    assert world <= ndofs0, 'more ranks than dof planes along axis 0'
    return (ndofs0 * rank) // world, (ndofs0 * (rank + 1)) // world


def row_range(kvs, rank, world):
    """Global CSR rows [lo, hi) owned by `rank`."""
    lo, hi = slab_range(kvs[0].numdofs, rank, world)
    plane = int(np.prod([kv.numdofs for kv in kvs[1:]]))
    return lo * plane, hi * plane


def _device_block(kind, kvs, geo, row0, device, algo):
    from . import assemblers
    patch = assemblers.DevicePatch(kvs, geo, device=device, row0=row0)
    try:
        return patch.csr(kind, algo=algo)
    finally:
        patch.close()


def assemble_rows(kind, kvs, geo, rank, world, device=None, algo='auto', block_fn=None):
    """CSR block (owned rows x all columns) of the `kind` matrix for this rank.

    `block_fn(kind, kvs, geo, row0, device, algo)` produces the block; the default runs the HIP
    path on `device` (default: LOCAL_RANK).  Tests inject a CPU producer to exercise the
    sharding logic without a GPU.
    """
    row0 = slab_range(kvs[0].numdofs, rank, world)
    fn = _device_block if block_fn is None else block_fn
    blk = fn(kind, tuple(kvs), geo, row0, device, algo)
    lo, hi = row_range(kvs, rank, world)
    assert blk.shape[0] == hi - lo, 'block has the wrong number of rows'
    return blk


def gather_matrix(block, dst=0):
    """Stack the row blocks of all ranks on rank `dst` (verification / small problems only:
    at 3D p=4 n=128 the matrix is 19 GB).  Returns the full CSR on `dst`, None elsewhere."""
    import torch.distributed as dist
    world = dist.get_world_size()
    rank = dist.get_rank()
    payload = (block.data, block.indices, block.indptr, block.shape)
    parts = [None] * world if rank == dst else None
    dist.gather_object(payload, parts, dst=dst)
    if rank != dst:
        return None
    blocks = [scipy.sparse.csr_matrix((d, i, p), shape=s) for (d, i, p, s) in parts]
    return scipy.sparse.vstack(blocks).tocsr()
