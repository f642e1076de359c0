"""Multi-GPU assembly: one process per GPU, rows sharded in slabs of axis-0 dof planes.

The path shards without any exchange ("zero-communication
alternative"): rank r owns a slab of dof planes of axis 0 (``slab_range``: balanced by work), i.e. a
contiguous block of CSR rows, and computes every entry of those rows itself.  The second kernel
(k_bf3) processes the outer pairs (i0, j0 <= i0) with the row OR the column owned and stores the
direct rows of a block if it owns i0, the transposed rows if it owns j0 -- both triangles come from
the same element matrices, there is no mirror pass.  Each rank evaluates the quadrature fields only
on the spans its rows touch, straight from the control net, so no input has to be communicated
either.  The result is bit-for-bit the row block of the single-GPU matrix
(tests/test_gpu_parity.py::test_row_slabs_equal_full), independent of the number of ranks.

``torch.distributed`` (RCCL when the backend is "nccl") is used only to gather results or
timings; there is no collective on the data path.
"""
import numpy as np
import scipy.sparse


# Cost model of one slab of the default 3D chain k_geoA + k_bf3 (microseconds on MI355X, C4 kernels of round 6 -- whole patch:
# 8.2 ms of k_bf3 over 650 processed outer pairs, 3.7 ms of k_geoA over 128 swept spans; only the RATIO matters for the split):
# the second kernel works per processed outer pair (i0, j0 <= i0) with the row or the column owned, the geometry + axis-0 sweep
# per resident span (own spans + p warm-up spans).  There is no per-target term any more (round 5 removed the mirror pass).
_COST_PAIR, _COST_SPAN = 12.6, 28.9


def slab_cost(ndofs0, p, lo, hi):
    """Modelled cost of the slab of dof planes [lo, hi) of an axis with single interior knots and degree p."""
    own = np.arange(lo, hi)
    halo = np.arange(hi, min(hi + p, ndofs0))                 # rows above the slab whose lower pairs have an owned column
    pairs = int(np.sum(np.minimum(own, p) + 1) + np.sum(np.maximum(hi - np.maximum(halo - p, lo), 0)))
    spans = min(hi - 1, ndofs0 - p - 1) - max(lo - p, 0) + 1     # spans the rows lo .. hi-1 touch
    return _COST_PAIR * pairs + _COST_SPAN * spans


def balanced_slabs(ndofs0, world, p):
    """Edges e[0] = 0 < e[1] < ... < e[world] = ndofs0 of the slabs with the smallest maximal modelled cost: the cost of
    a slab grows with every plane added, so the smallest feasible bottleneck is found by bisection (greedy packing)."""
    assert world <= ndofs0, 'more ranks than dof planes along axis 0'

    def pack(limit):
        """edges of a greedy packing with slab cost <= limit, or None if it needs more than `world` slabs"""
        edges, a = [0], 0
        for w in range(world):
            left = world - 1 - w                            # slabs still to come: leave them one plane each
            lo_b, hi_b = a + 1, ndofs0 - left
            if slab_cost(ndofs0, p, a, lo_b) > limit:
                return None
            while lo_b < hi_b:                              # largest b with cost(a, b) <= limit
                mid = (lo_b + hi_b + 1) // 2
                if slab_cost(ndofs0, p, a, mid) <= limit:
                    lo_b = mid
                else:
                    hi_b = mid - 1
            a = lo_b
            edges.append(a)
            if a == ndofs0:
                break
        if edges[-1] != ndofs0:
            return None
        while len(edges) < world + 1:                       # fewer slabs than ranks: split the largest ones
            k = int(np.argmax(np.diff(edges)))
            edges.insert(k + 1, (edges[k] + edges[k + 1]) // 2)
        return edges

    lo_t, hi_t = 0.0, slab_cost(ndofs0, p, 0, ndofs0)
    best = pack(hi_t)
    for _ in range(60):
        mid = 0.5 * (lo_t + hi_t)
        e = pack(mid)
        if e is None:
            lo_t = mid
        else:
            best, hi_t = e, mid
        if hi_t - lo_t < 1e-6 * hi_t:
            break
    # the greedy packing fills the first slabs to the limit and leaves the rest to the last one: even the slack out from the
    # right as well (same bottleneck, smaller spread)
    return best


def scaling_model(ndofs0, world, p):
    """What the cost model expects of a `world`-way split of the axis (for the bench line, next to the measured curve):
    the slab edges, the share of axis-0 spans that are swept twice (the p warm-up spans below every slab but the first),
    the modelled cost of every slab relative to the whole patch and the speed-up that follows (whole / slowest slab).
    Fixed per-launch costs and the block quantisation of the fused stage are NOT in the model: the measured speed-up of
    the round-3 slab emulation was 6.2-6.8 against 6.9 modelled at 8 ranks."""
    e = balanced_slabs(ndofs0, world, p) if world > 1 else [0, ndofs0]
    whole = slab_cost(ndofs0, p, 0, ndofs0)
    costs = [slab_cost(ndofs0, p, e[r], e[r + 1]) for r in range(world)]
    nspans = ndofs0 - p
    swept = sum(min(e[r + 1] - 1, ndofs0 - p - 1) - max(e[r] - p, 0) + 1 for r in range(world))
    return {'edges': [int(x) for x in e], 'halo_fraction': round(swept / nspans - 1.0, 4),
            'slab_cost_rel': [round(c / whole, 4) for c in costs],
            'max_over_min': round(max(costs) / min(costs), 4),
            'predicted_speedup': round(whole / max(costs), 3)}


_EDGES = {}


def slab_range(ndofs0, rank, world, p=None):
    """Dof planes [lo, hi) of axis 0 owned by `rank`; the slabs tile [0, ndofs0) exactly.  With the degree `p` of the
    axis the slabs are balanced by WORK (modelled cost of the default 3D chain: the first and the last slab own more
    planes, they have no halo on one side and fewer pairs per plane); without it by planes."""
    assert 0 <= rank < world
    assert world <= ndofs0, 'more ranks than dof planes along axis 0'
    if p is None or world == 1:
        return (ndofs0 * rank) // world, (ndofs0 * (rank + 1)) // world
    key = (ndofs0, world, p)
    if key not in _EDGES:
        _EDGES[key] = balanced_slabs(ndofs0, world, p)
    e = _EDGES[key]
    return e[rank], e[rank + 1]


def _balance_degree(kvs, balance):
    """degree to balance the slabs with, or None (even split): the cost model is the one of the 3D chain"""
    return int(kvs[0].p) if balance and len(kvs) == 3 else None


def row_range(kvs, rank, world, balance=True):
    """Global CSR rows [lo, hi) owned by `rank`."""
    lo, hi = slab_range(kvs[0].numdofs, rank, world, _balance_degree(kvs, balance))
    plane = int(np.prod([kv.numdofs for kv in kvs[1:]]))
    return lo * plane, hi * plane


def _device_block(kind, kvs, geo, row0, device, algo):
    from . import assemblers
    patch = assemblers.DevicePatch(kvs, geo, device=device, row0=row0)
    try:
        return patch.csr(kind, algo=algo)
    finally:
        patch.close()


def assemble_rows(kind, kvs, geo, rank, world, device=None, algo='auto', block_fn=None, balance=True):
    """CSR block (owned rows x all columns) of the `kind` matrix for this rank.

    `block_fn(kind, kvs, geo, row0, device, algo)` produces the block; the default runs the HIP
    path on `device` (default: LOCAL_RANK).  Tests inject a CPU producer to exercise the
    sharding logic without a GPU.
    """
    row0 = slab_range(kvs[0].numdofs, rank, world, _balance_degree(kvs, balance))
    fn = _device_block if block_fn is None else block_fn
    blk = fn(kind, tuple(kvs), geo, row0, device, algo)
    lo, hi = row_range(kvs, rank, world, balance)
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
