#!/usr/bin/env python3
"""Throughput of the assembly hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c4|c3|c2|c1|c5|tiny] [--algo auto|sumfact|entrywise]

Workload (BASELINE.json): stiffness assembly of a tensor-product B-spline patch over the NURBS
quarter-annulus cylinder; default C4 = 3D, p=4, 128^3 spans (2.1 M elements, 1.59 G nonzeros).
One "step" = one complete assembly of the CSR values on the device (quadrature fields from the
control net, the sum-factorisation stages, CSR write-out + mirror), inputs resident in HBM.

Multi-GPU (launched by torch.distributed.run, one rank per GPU): WEAK scaling -- the patch grows
along axis 0 to N*128 spans and rank r owns the r-th slab of axis-0 dof planes.  The row-owner
computes everything it needs (DESIGN.md "multi-GPU"), so there is no collective on the data
path; torch.distributed (RCCL) is only used for the timing barrier / max-reduce.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (dim, p, n, kind, geometry)
    'c4': (3, 4, 128, 'stiffness', 'cylinder'),
    'c3': (3, 2, 64, 'stiffness', 'cylinder'),
    'c2': (2, 3, 256, 'stiffness', 'quarter_annulus'),
    'c1': (2, 3, 15, 'stiffness', 'bspline_quarter_annulus'),
    'tiny': (3, 2, 12, 'stiffness', 'cylinder'),
    # BASELINE config 5: the run-time compiled (vform) convection-diffusion form, non-symmetric
    'c5': (3, 5, 96, 'convdiff', 'cylinder'),
}
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # vector = MFMA FP64 peak on MI355X


def algorithmic_bytes_per_element(dim, p, nnz, nelem, kind='stiffness'):
    """Jacobian in (8 d^2 q^d) + CSR values out (8 nnz / n^d); the
    convection-diffusion form also reads its coefficient (8 q^d)."""
    q = p + 1
    return 8.0 * (dim * dim + (1 if kind == 'convdiff' else 0)) * q ** dim + 8.0 * nnz / nelem


def measured_traffic(config, world):
    """HBM bytes per assembly from the committed rocprofv3 PMC passes (profiles/), or None."""
    try:
        t = json.load(open(os.path.join(ROOT, 'profiles', 'r01_traffic.json')))[config]
        return t['chain_bytes'] if world == 1 else None
    except Exception:
        return None


def COEFF(x, y, z):
    return 1.0 + x


def make_geo(geometry, name):
    if name == 'cylinder':
        return geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
    return getattr(geometry, name)()


def cpu_baseline(dim, p, kind):
    """Oracle (C port of the reference's entry-wise loops, reference compile flags) timed on the
    host cores on a bounded sample of the same workload."""
    from oracle import iga_oracle as orc
    orc.build()
    n = {(3, 4): 24, (3, 2): 48, (2, 3): 256}.get((dim, p), 12)
    cores = os.cpu_count() or 1
    kv = orc.make_knots(p, 0.0, 1.0, n)
    geo = orc.geo_cylinder() if dim == 3 else orc.geo_quarter_annulus()
    timing = {}
    t0 = time.perf_counter()
    if kind == 'convdiff':
        A = orc.assemble_nonsymmetric(kind, (kv,) * dim, geo, coeff=COEFF, nthreads=cores)
        timing['entries'] = time.perf_counter() - t0
    else:
        A = orc.assemble(kind, (kv,) * dim, geo, nthreads=cores, fast=True, return_timing=timing)
    dt = time.perf_counter() - t0
    nel = n ** dim
    return {
        'value': nel / dt, 'unit': 'elements/s', 'cores': cores, 'kind': 'port',
        'sample': '%dD p=%d n=%d %s, quarter-annulus %s, full assemble() incl. setup+CSR: %.2f s '
                  '(entry kernel %.2f s = %.3g el/s)' % (dim, p, n, kind, 'cylinder' if dim == 3 else '', dt,
                                                          timing['entries'], nel / timing['entries']),
        'nnz': int(A.nnz),
    }


def bench_rhs(args, patch, kvs, geo, dim, p, n0, n, nel_total, world, rank, dist, barrier):
    """Load vector of the same patch.  The function values are host data by definition (a Python
    callable sampled on the Gauss grid), so a call includes their upload; `value` uses the device time of the
    three contractions (inputs resident, HIP events inside igx_load_vector), the PCIe-inclusive rate is noted."""
    import pyiga_amd
    from pyiga_amd import utils
    grid = tuple(patch.gauss(k)[0] for k in range(dim))
    fvals = np.ascontiguousarray(utils.grid_eval(lambda *x: np.cos(x[0]) * np.exp(x[1]) * (np.sin(x[2]) if dim == 3 else 1.0), grid))
    for _ in range(args.warmup):
        patch.load_vector(fvals)
    barrier()
    dev_ms, t0 = 0.0, time.perf_counter()
    for _ in range(args.steps):
        patch.load_vector(fvals)
        dev_ms += patch.timing()['total_ms']
    barrier()
    wall = time.perf_counter() - t0
    if rank != 0:
        return
    dev_ms /= args.steps
    q = p + 1
    b_el = 8.0 * q ** dim + 8.0 * np.prod([k.numdofs for k in kvs]) / nel_total      # f in + vector out (W is resident)
    nel_rank = nel_total / world
    achieved = b_el * nel_rank / (dev_ms * 1e-3) / 1e9
    out = {'metric': 'load-vector elements/sec', 'value': nel_total / (dev_ms * 1e-3), 'unit': 'elements/s', 'n_gpus': world,
           'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dev_ms, 'higher_is_better': True, 'scaling': 'weak',
           'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': '%dD p=%d load vector (inner_products), %s spans' % (dim, p, 'x'.join(str(x) for x in (n0,) + (n,) * (dim - 1))),
                      'config': args.config, 'elements': nel_total,
                      'note': 'value = device time of the contractions; a whole call incl. the upload of the function values took %.1f ms' % (1e3 * wall / args.steps)},
           'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                        'traffic': None, 'kernel': 'k_contract_axis x %d' % dim, 'algorithmic_bytes_per_element': b_el}}
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='c4', choices=sorted(CONFIGS))
    ap.add_argument('--n', type=int, default=0, help='override spans per axis (testing)')
    ap.add_argument('--algo', default='auto', choices=['auto', 'sumfact', 'entrywise'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--op', default='matrix', choices=['matrix', 'rhs'], help='rhs: the load vector (inner_products) of the same patch instead of the matrix')
    ap.add_argument('--strong', action='store_true', help='strong scaling: keep the patch fixed, split its rows')
    ap.add_argument('--emulate', default='', help='R/W: assemble the slab of rank R of a W-rank weak-scaling run on this one GPU (no collectives)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    emu = None
    if args.emulate:
        emu = tuple(int(x) for x in args.emulate.split('/'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    dist = None
    if world > 1 or os.environ.get('BENCH_FORCE_DIST'):      # BENCH_FORCE_DIST: exercise the RCCL path on one GPU
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))

    import pyiga_amd
    from pyiga_amd import bspline, geometry, assemblers, distributed
    os.environ['IGX_DEVICE'] = str(local_rank)

    dim, p, n, kind, gname = CONFIGS[args.config]
    if args.n:
        n = args.n
    geo = make_geo(geometry, gname)
    # weak scaling: axis 0 grows with the number of ranks; each rank owns one slab of dof planes
    part_rank, part_world = (rank, world) if emu is None else emu
    n0 = n if (args.strong or part_world == 1) else n * part_world
    kv0 = bspline.make_knots(p, 0.0, 1.0, n0)
    kv = bspline.make_knots(p, 0.0, 1.0, n)
    kvs = (kv0,) + (kv,) * (dim - 1)
    row0 = distributed.slab_range(kv0.numdofs, part_rank, part_world)
    if kind == 'convdiff':
        patch = assemblers.ConvDiffAssembler3D(kvs, geo, COEFF, device=local_rank, row0=row0 if part_world > 1 else None).patch
    else:
        patch = assemblers.DevicePatch(kvs, geo, device=local_rank, row0=row0 if part_world > 1 else None)
    if rank == 0 and os.environ.get('BENCH_VERBOSE'):
        print('rank 0 slab', row0, 'nnz', patch.nnz, 'rows', patch.row_range, file=sys.stderr)
    nel_total = n0 * n ** (dim - 1)
    if emu is not None:
        nel_total //= part_world        # one slab's share
    nnz_local = patch.nnz

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()
        patch.ctx.sync()

    if args.op == 'rhs':
        return bench_rhs(args, patch, kvs, geo, dim, p, n0, n, nel_total, world, rank, dist, barrier)
    for _ in range(args.warmup):
        patch.assemble(kind, algo=args.algo, to_host=False)
    stage_ms = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        patch.assemble(kind, algo=args.algo, to_host=False)     # returns after the ctx stream has drained
        for k, v in patch.timing().items():
            if k.endswith('_ms'):
                stage_ms[k] = stage_ms.get(k, 0.0) + v
        algo_used = patch.timing()['algo_used']
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        z = torch.tensor([float(nnz_local)], dtype=torch.float64, device='cuda')
        dist.all_reduce(z, op=dist.ReduceOp.SUM)
        nnz_total = int(z.item())
    else:
        nnz_total = nnz_local
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = 1e3 * dt / args.steps
    value = nel_total * args.steps / dt
    stage_ms = {k: v / args.steps for k, v in stage_ms.items()}
    # roofline on rank 0's slab: algorithmic bytes of one assembly / device time of the kernel chain
    nel_rank = nel_total / world
    b_el = algorithmic_bytes_per_element(dim, p, nnz_total, nel_total, kind)
    chain_ms = stage_ms.get('total_ms', ms_per_step)
    achieved = b_el * nel_rank / (chain_ms * 1e-3) / 1e9
    names = {'fields_ms': 'k_geo_fields', 'stage0_ms': 'k_stageA', 'stage1_ms': 'k_stageB', 'final_ms': 'k_final',
             'entry_ms': 'k_entries_csr'}
    parts = {names[k]: round(v, 4) for k, v in stage_ms.items() if k in names and v > 0}
    dominant = max(parts, key=parts.get) if parts else None
    # the dominant kernel by itself: the part of the algorithmic bytes that passes through it
    # (final stage / entry-wise kernel: the CSR values out; stage A: the quadrature input in) over its own time
    dom = None
    if dominant in ('k_final', 'k_entries_csr', 'k_stageA', 'k_geo_fields'):
        q = p + 1
        if dominant in ('k_final', 'k_entries_csr'):
            dom_bytes = 8.0 * nnz_total / world
        else:
            dom_bytes = 8.0 * (dim * dim + (1 if kind == 'convdiff' else 0)) * q ** dim * nel_rank
        dom_ach = dom_bytes / (parts[dominant] * 1e-3) / 1e9
        dom = {'kernel': dominant, 'ms': parts[dominant], 'algorithmic_bytes': dom_bytes, 'achieved': dom_ach,
               'unit': 'GB/s', 'frac': dom_ach / HBM_PEAK_GBS}
    out = {
        'metric': 'assembled elements/sec (3D p=4, 128^3 spans) + HBM-roofline %; 1/2/4/8 GPU'
                  if args.config == 'c4' else 'assembled elements/sec',
        'value': value, 'unit': 'elements/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': 'strong' if args.strong else 'weak',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': '%dD p=%d %s, %s spans, NURBS quarter-annulus %s, uniform open knots'
                               % (dim, p, kind, 'x'.join(str(x) for x in (n0,) + (n,) * (dim - 1)),
                                  'cylinder' if dim == 3 else gname),
                   'config': args.config, 'elements': nel_total, 'nnz': nnz_total, 'dofs': int(np.prod([k.numdofs for k in kvs])),
                   'algo': {1: 'entrywise', 2: 'sumfact'}.get(algo_used, str(algo_used)),
                   'parallelism': 'row slabs of axis-0 dof planes, %d rank(s), no data-path collective' % world},
        'roofline': {
            'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
            'traffic': measured_traffic(args.config, world),
            'kernel': 'assembly chain (k_geo_fields + k_stageA x fields + k_stageB + k_final), HIP events on the igx stream',
            'algorithmic_bytes_per_element': b_el, 'chain_ms': chain_ms, 'kernel_ms': parts, 'dominant_kernel': dominant,
            'dominant': dom,
        },
    }
    if not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(dim, p, kind)
    if dist is not None:
        dist.destroy_process_group()        # before the result line: anything RCCL prints comes first
    # RCCL writes its version banner through C stdio, which is block-buffered on a pipe and would surface
    # after this line at exit: flush it first so that the result is the LAST line of stdout
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
