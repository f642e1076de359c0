#!/usr/bin/env python3
"""Throughput of the assembly hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c4|c3|c2|c1|c5|tiny] [--algo auto|sumfact|entrywise]
                    [--weak] [--emulate R/W] [--op matrix|rhs|entries]

Workload (BASELINE.json): stiffness assembly of a tensor-product B-spline patch over the NURBS quarter-annulus
cylinder; default C4 = 3D, p=4, 128^3 spans (2.1 M elements, 1.59 G nonzeros).  One "step" = one complete assembly of
the CSR values on the device (quadrature fields from the control net, the sum-factorisation stages, CSR write-out +
mirror), inputs resident in HBM.

Multi-GPU: one process per GPU (torch.distributed, backend "nccl" = RCCL).  Launched by torch.distributed.run the
script reads RANK/LOCAL_RANK/WORLD_SIZE; started plainly with --gpus N > 1 it spawns the N ranks itself BEFORE
anything touches the GPU and relays rank 0's line.  The default is STRONG scaling of the named patch (BASELINE config 4
fixes 128^3 spans): rank r owns the r-th slab of axis-0 dof planes and computes everything it needs itself
(DESIGN.md "multi-GPU"), so there is no collective on the data path; RCCL carries the barrier and the reductions of
the timing.  --weak grows axis 0 with the rank count instead (each rank a 128-span slab) and says so in `scaling`.

Prints ONE JSON line on rank 0 (the last line of stdout).
"""
import argparse
import json
import os
import subprocess
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
    # what a patch OUTSIDE the fused chain pays: C4 with double interior knots on the mid axis (C^2 instead of C^3
    # there) -- the stage kernels with 64-bit addresses; and C4 at degree 4 x 3 x 4 (unequal degrees on the mid / last axis)
    'c4k': (3, 4, 128, 'stiffness', 'cylinder'),
    'c4m': (3, 4, 128, 'stiffness', 'cylinder'),
    # round 6: C4 with double interior knots on the LAST axis (through the axis-exchanged twin patch), C4 at degree 4 x 2 x 4 (a degree gap
    # of two), the symmetric form at C5's size
    'c4l': (3, 4, 128, 'stiffness', 'cylinder'),
    'c4g': (3, 4, 128, 'stiffness', 'cylinder'),
    'c5s': (3, 5, 96, 'stiffness', 'cylinder'),
    # the mass form of BASELINE config 3 ("3D p=2 n=64 mass+stiffness"), and at C4's size
    'c3mass': (3, 2, 64, 'mass', 'cylinder'),
    'c4mass': (3, 4, 128, 'mass', 'cylinder'),
    # BASELINE config 5: the run-time compiled (vform) convection-diffusion form, non-symmetric
    'c5': (3, 5, 96, 'convdiff', 'cylinder'),
}
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6        # vector = MFMA FP64 peak on MI355X


ALGORITHMIC_BYTES_DEF = ('SURVEY.md 8d: 8 d^2 q^d (Jacobian in) + 8 nnz / n^d (CSR values out); the convection-diffusion form adds '
                         '8 (1 + d) q^d (diff_coeff and the d coordinate fields of the convection direction: SURVEY 8d quotes C5 = '
                         '8 (9 + 1 + 3) 216 + 8 * 1427.8 = 33.9 KB/el; rounds 1-2 of this repository used 8 (9 + 1) 216 for C5)')


def algorithmic_bytes_per_element(dim, p, nnz, nelem, kind='stiffness'):
    """Jacobian in (8 d^2 q^d) + CSR values out (8 nnz / n^d); the convection-diffusion form adds
    its coefficient and the d coordinate fields of the convection direction (8 (1 + d) q^d): C5 = 33.9 KB per element,
    the figure SURVEY 8d itself quotes.  The definition travels with the number (`algorithmic_bytes_def`)."""
    q = p + 1
    return 8.0 * (dim * dim + ((1 + dim) if kind == 'convdiff' else 0)) * q ** dim + 8.0 * nnz / nelem


def algorithmic_flops(dim, p, kvs, kind):
    """FP64 operations of ONE assembly with the global sum factorisation, by stage -- the
    useful ones: no tile halos, no warm-up spans.  Counts follow the kernels: fields ~150 flop per Gauss point;
    stage A: 2 flops per (array, lower pair of the span, point); stage B in rank-1 form (fused stage) 2 * 170 per
    (axis-0 pair, g1, g2) for the 9 stiffness terms; last contraction 2 * 70 q per (line, row)."""
    P = p + 1
    q = P
    N = [kv.numdofs for kv in kvs]
    n = [kv.numspans for kv in kvs]
    G = [k * q for k in n]
    npts = float(np.prod(G))
    S = [Nk * (2 * p + 1) - p * (p + 1) for Nk in N]          # 1D pairs (single interior knots)
    Slow = (S[0] + N[0]) // 2                                   # lower pairs of axis 0
    sym = kind != 'convdiff'
    nX = {('stiffness', 3): 8, ('stiffness', 2): 3, ('convdiff', 3): 11}.get((kind, dim), 1)
    nterm = {('stiffness', 3): 9, ('stiffness', 2): 4, ('convdiff', 3): 12}.get((kind, dim), 1)
    out = {'fields': 150.0 * npts}
    pairs_span = P * (P + 1) / 2 if sym else P * P
    out['stage_a'] = 2.0 * nX * pairs_span * npts
    np0 = Slow if sym else S[0]
    if dim == 3:
        per_pt = 2.0 * {9: 170, 12: 225, 1: 30}.get(nterm, 25 * nterm)
        out['stage_b'] = per_pt * np0 * G[1] * G[2]
        lines = np0 * S[1]
        out['contraction'] = 2.0 * (70 if nterm > 1 else 30) * q * lines * N[2]
    else:
        out['contraction'] = 2.0 * (70 if nterm > 1 else 30) * q * np0 * N[1]
    return out


TRAFFIC_FILE = 'profiles/r06_traffic.json'


def measured_traffic(config, world, op='matrix', kernels_now=None):
    """HBM bytes per assembly from the committed rocprofv3 PMC passes (TRAFFIC_FILE, tools/make_traffic.py) -- or None
    when the device code of the library has changed since they were taken, or when this run launched other kernels than the
    profiled one (stale numbers are not reported; the key is the hash of the code objects inside libigx.so plus the set of kernel
    names of the chain).  The counters need
    passes of their own under rocprofv3, so this is never measured by the run that prints it: `measured_in_this_run`
    says so on the line."""
    try:
        import glob
        import hashlib
        key = config if op == 'matrix' else '%s_%s' % (config, op)
        t = json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))[key]
        if world != 1:
            return None
        from pyiga_amd import _lib
        if _lib.device_code_sha() != t.get('kernels_sha'):     # hash of the device code objects inside libigx.so
            return None
        # ... and the CHAIN that ran must be the one that was profiled: the host side picks the kernels (path knobs, shapes), and a
        # host-only change that sends the config down another chain leaves the device code -- and the hash -- untouched
        if kernels_now is not None and set(kernels_now) != set(t['kernels']):
            return None
        return {'bytes': t['chain_bytes'], 'measured_in_this_run': False, 'kernels_sha': t['kernels_sha'], 'source': TRAFFIC_FILE,
                'kernels': {k: round(v['read_bytes'] + v['write_bytes']) for k, v in t['kernels'].items()}}
    except Exception:
        return None


MIXED_STREAM_GBS = 5300.0      # what a stream of reads and writes side by side reaches on this part (a copy: 6.3 TB/s)
ISSUE_CYCLES, N_SIMD, CLOCK_GHZ = 5.0, 1024, 2.0   # a wave64 FP64 instruction holds its SIMD 5 cycles (profiles/r03_ubench_valu_f64.txt); the chain clocks 1.8-2.0 GHz


def design_floor(config, dim, kvs, kind, nnz_total, alg_bytes, world):
    """Where the TWO-KERNEL design stops, printed beside the measured fraction so that the gap to north_star's 0.70 is stated where
    the number is: the chain must write the axis-0 intermediate K1 once, read it once and write the CSR values once (`floor_bytes`);
    at the rate a mixed stream reaches that is `floor_ms`, i.e. `ceiling_frac` of the HBM roof in ALGORITHMIC bytes -- and the
    vector instructions the two kernels issue (SQ_INSTS_VALU of the committed PMC pass) take `issue_ms` on 1024 SIMDs by themselves."""
    if dim != 3 or world != 1 or kind not in ('stiffness', 'mass', 'convdiff', 'form'):
        return None
    p0, N0 = kvs[0].p, kvs[0].numdofs
    sym = kind != 'convdiff'
    pairs0 = N0 * (p0 + 1) - p0 * (p0 + 1) // 2 if sym else N0 * (2 * p0 + 1) - p0 * (p0 + 1)     # single interior knots on axis 0
    nq = max(k.p for k in kvs) + 1
    npl = (kvs[1].numspans * nq) * (kvs[2].numspans * nq)
    k1 = 8.0 * (1 if kind == 'mass' else 8) * pairs0 * npl
    floor_bytes = 2 * k1 + 8.0 * nnz_total
    floor_ms = floor_bytes / (MIXED_STREAM_GBS * 1e9) * 1e3
    out = {'floor_bytes': floor_bytes, 'floor_ms': round(floor_ms, 3), 'ceiling_frac': round(alg_bytes / (floor_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           'target_frac': 0.70, 'issue_ms': None,
           'note': 'two-kernel global sum factorisation: K1 written + read once, CSR values written once, at %.1f TB/s (mixed stream); '
                   'north_star asks 0.70 -- not reachable by this design' % (MIXED_STREAM_GBS / 1e3)}
    try:
        t = json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))[config]
        insts = sum(k['valu_insts'] for k in t['kernels'].values())
        out['issue_ms'] = round(insts * ISSUE_CYCLES / N_SIMD / (CLOCK_GHZ * 1e9) * 1e3, 3)
        out['valu_insts'] = insts
    except Exception:
        pass
    return out


FORM_STRING = '(inner(grad(u), grad(v)) + u * v) * dx'


def COEFF(x, y, z):
    return 1.0 + x


def make_geo(geometry, name):
    if name == 'cylinder':
        return geometry.tensor_product(geometry.line_segment(0.0, 1.0), geometry.quarter_annulus())
    return getattr(geometry, name)()


def host_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota (cpu.max, or the v1 files)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    quota = None
    try:
        a, b = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if a != 'max':
            quota = float(a) / float(b)
    except Exception:
        try:
            q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline(dim, p, kind):
    """Oracle (C port of the reference's entry-wise loops, reference compile flags) timed on the
    host cores on a bounded sample of the same workload.  Runs BEFORE anything touches the GPU (it may start gcc).
    `cores` = threads actually used = affinity mask capped by the cgroup quota; the entry kernel is also timed at 1 and
    8 threads on seeded subsamples of the same index pairs, so that the line shows how the port scales on this host
    (pyiga/genericasm.pxi:722-758 is the loop being timed; the reference measured 1.55e3 el/s on 8 cores at p = 4)."""
    from oracle import iga_oracle as orc
    orc.build()
    cores, quota = host_cores()
    # bounded sample (about 10-30 s of CPU work): a few more spans on a many-core host, so that the threads have work
    n = {(3, 4): 32 if cores > 32 else 24, (3, 2): 64 if cores > 32 else 48, (2, 3): 256, (3, 5): 18 if cores > 32 else 14}.get((dim, p), 12)
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
    # thread scaling of the entry kernel alone: seeded subsamples of the pattern sized for ~2 s each
    scaling = {}
    try:
        asm = orc.Assembler(kind, (kv,) * dim, geo=geo, coeff=COEFF if kind == 'convdiff' else None)
        I, J = (orc.full_pattern if kind == 'convdiff' else orc.lower_pattern)((kv,) * dim)
        M = I.shape[0]
        rate_all = M / timing['entries']                         # entries/s with `cores` threads
        rng = np.random.default_rng(11)
        for nt in sorted({1, min(8, cores)}):
            if nt >= cores:
                continue
            m = int(min(M, max(2000, 2.0 * rate_all * nt / cores)))
            sel = np.sort(rng.choice(M, m, replace=False))
            idx = np.column_stack((I[sel], J[sel]))
            t1 = time.perf_counter()
            asm.multi_entries(idx, nthreads=nt, fast=kind != 'convdiff')
            scaling[nt] = nel * (m / M) / (time.perf_counter() - t1)
        scaling[cores] = nel / timing['entries']
    except Exception as e:                                      # the scaling note is optional; the baseline itself is above
        scaling = {'error': str(e)[:80]}
    return {
        'value': nel / dt, 'unit': 'elements/s', 'cores': cores, 'kind': 'port',
        'sample': '%dD p=%d n=%d %s, quarter-annulus %s, full assemble() incl. setup+CSR: %.2f s '
                  '(entry kernel %.2f s = %.3g el/s); entry kernel el/s by threads: %s'
                  % (dim, p, n, kind, 'cylinder' if dim == 3 else '', dt, timing['entries'], nel / timing['entries'],
                     ', '.join('%s: %.3g' % (k, v) if not isinstance(v, str) else '%s: %s' % (k, v) for k, v in scaling.items())),
        'entry_kernel_el_s_by_threads': {str(k): (round(v, 1) if not isinstance(v, str) else v) for k, v in scaling.items()},
        'host': {'os_cpu_count': os.cpu_count(), 'affinity': cores if quota is None else None, 'cgroup_quota_cores': quota},
        'nnz': int(A.nnz),
    }


def api_call(args, kvs, geo, kind, nnz):
    """assemble.stiffness()/mass() end to end, as a user of the reference API sees it: patch creation, assembly, pattern,
    D2H of values + indices, scipy wrap (pyiga/assemble.py:1017-1049).  Skipped (None) when the host copy would not fit
    comfortably into the free memory of the box."""
    need = nnz * 12.0 + 4.0 * float(np.prod([k.numdofs for k in kvs]))
    try:
        import psutil
        if psutil.virtual_memory().available < 2.5 * need:
            return None
    except Exception:
        if need > 8e9:
            return None
    from pyiga_amd import assemble
    tries = os.environ.pop('IGX_PLACEMENT_TRIES', None)     # the plain call of a user of the reference API: no opt-in
    t0 = time.perf_counter()
    A = getattr(assemble, kind)(kvs, geo)
    dt = time.perf_counter() - t0
    if tries is not None:
        os.environ['IGX_PLACEMENT_TRIES'] = tries
    assert A.nnz == nnz
    del A
    return round(dt, 3)


def flush_c_stdio():
    # RCCL writes its banner through C stdio, which is block-buffered on a pipe: flush it so that the result is the LAST line
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def _tdev():
    """device of the tensors that go through collectives (gloo in the one-GPU test mode)"""
    return 'cpu' if os.environ.get('BENCH_SHARE_GPU') else 'cuda'


def self_launch(args):
    """--gpus N without a launcher: start the N ranks as children (nothing here has touched the GPU), relay rank 0."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs, logs = [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        import tempfile
        log = tempfile.TemporaryFile(mode='w+')              # stderr (and, for ranks > 0, stdout) of the rank: shown if it fails
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else log, stderr=log, text=True))
    # rank 0's stdout is drained by a thread; the parent polls ALL ranks: the first one that fails takes the others down
    # (a survivor would sit in the rendezvous or a barrier until the backend times out)
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + float(os.environ.get('BENCH_LAUNCH_TIMEOUT', '1500'))
    failed = None
    while True:
        rcs = [p.poll() for p in procs]
        bad = [i for i, rc in enumerate(rcs) if rc not in (None, 0)]
        if bad:
            failed = 'rank %d exited with %d' % (bad[0], rcs[bad[0]])
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.monotonic() > deadline:
            failed = 'timeout'
            break
        time.sleep(0.05)
    if failed:
        for p in procs:                                      # exactly the children started here
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    if not failed:                                           # warnings of a passing run (RCCL, HIP) are not lost: rank 0's stderr
        logs[0].seek(0)
        err0 = logs[0].read()
        if err0.strip():
            sys.stderr.write(err0[-8000:])
            sys.stderr.flush()
    sys.stdout.write(''.join(c for c in chunks if c))
    sys.stdout.flush()
    if failed:
        for r, log in enumerate(logs):
            log.seek(0)
            tail = log.read()[-2000:]
            if tail.strip():
                sys.stderr.write('--- rank %d ---\n%s\n' % (r, tail))
        raise SystemExit('bench.py: %s (exit codes %s)' % (failed, [p.returncode for p in procs]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', default='c4', choices=sorted(CONFIGS))
    ap.add_argument('--n', type=int, default=0, help='override spans per axis (testing)')
    ap.add_argument('--algo', default='auto', choices=['auto', 'sumfact', 'entrywise'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-api-call', action='store_true', help='skip the end-to-end assemble.stiffness() call (host copy of the matrix)')
    ap.add_argument('--op', default='matrix', choices=['matrix', 'rhs', 'entries', 'fast', 'structured', 'form'],
                    help='rhs: the load vector (inner_products) of the same patch; entries: batched multi_entries on random in-pattern pairs; '
                         'fast: the low-rank (ACA) assembler against the exact assembly (use --config c2 / c3: the reordered tensor lives on the host); '
                         'structured: the opt-in Kronecker expansion for geometries that are separable along axis 0 (the cylinder of the 3D configs)')
    ap.add_argument('--weak', action='store_true', help='weak scaling: axis 0 grows to N * n spans, one n-span slab per rank')
    ap.add_argument('--strong', action='store_true', help='(default) strong scaling: the patch is fixed, its rows are split')
    ap.add_argument('--emulate', default='', help='R/W: assemble the slab of rank R of a W-rank run on this one GPU (no collectives)')
    ap.add_argument('--placement-tries', type=int, default=1,
                    help='candidate buffers for the CSR values, timed under the mirror pass at the FIRST assembly (outside the timed region; '
                         'IGX_PLACEMENT_TRIES: the pass follows where the driver put the buffer); 1: plain allocation')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1 and args.gpus > 1 and 'RANK' not in os.environ:
        return self_launch(args)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    emu = tuple(int(x) for x in args.emulate.split('/')) if args.emulate else None
    dim, p, n, kind, gname = CONFIGS[args.config]
    if args.n:
        n = args.n

    # the CPU baseline first: it may compile the oracle (child processes only before the GPU is initialised)
    cpu = None
    if rank == 0 and emu is None and not args.no_cpu_baseline and args.op == 'matrix':
        cpu = cpu_baseline(dim, p, kind)

    dist = None
    if world > 1 or os.environ.get('BENCH_FORCE_DIST'):      # BENCH_FORCE_DIST: exercise the RCCL path on one GPU
        import torch
        import torch.distributed as dist
        # BENCH_SHARE_GPU=1 (testing on a one-GPU box): every rank uses device 0 and the rendezvous runs over gloo --
        # RCCL refuses two ranks on one device; the slab logic, the reductions and the result line are the same
        if os.environ.get('BENCH_SHARE_GPU'):
            local_rank = 0
            dist.init_process_group(backend='gloo')
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit('RCCL world size %d != --gpus %d' % (dist.get_world_size(), args.gpus))

    import pyiga_amd
    from pyiga_amd import bspline, geometry, assemblers, distributed
    os.environ['IGX_DEVICE'] = str(local_rank)
    if args.placement_tries > 1 and 'IGX_PLACEMENT_TRIES' not in os.environ:
        os.environ['IGX_PLACEMENT_TRIES'] = str(args.placement_tries)      # read when a patch is created

    geo = make_geo(geometry, gname)
    part_rank, part_world = (rank, world) if emu is None else emu
    weak = args.weak and part_world > 1
    n0 = n * part_world if weak else n
    # test hook (tests/test_bench_launch_cpu.py): a stand-in for DevicePatch from the named file, so that the launcher, the
    # rendezvous, the reductions and the result line can be exercised on a box without a GPU.  Never set in a measurement.
    stub = os.environ.get('BENCH_PATCH_STUB')
    if stub:
        import importlib.util
        spec = importlib.util.spec_from_file_location('bench_patch_stub', stub)
        stubmod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(stubmod)
        assemblers = stubmod
    t_init = time.perf_counter()
    if not stub:
        from pyiga_amd import _lib
        _lib.context(local_rank).sync()                     # HIP runtime + device context + stream: first GPU call of the process
    # (with several ranks torch / RCCL have created the runtime and the context before this point: the figure is the cost of
    # the first GPU call of a process only when world == 1, and is reported as null otherwise)
    runtime_init_s = time.perf_counter() - t_init if dist is None else None
    t_setup = time.perf_counter()
    kv0 = bspline.make_knots(p, 0.0, 1.0, n0)
    kv = bspline.make_knots(p, 0.0, 1.0, n)
    kvs = (kv0,) + (kv,) * (dim - 1)
    if args.config == 'c4k':
        kvs = (kv0, bspline.make_knots(p, 0.0, 1.0, n // 2, mult=2), kv)
    elif args.config == 'c4m':
        kvs = (kv0, bspline.make_knots(p - 1, 0.0, 1.0, n), kv)
    elif args.config == 'c4l':
        kvs = (kv0, kv, bspline.make_knots(p, 0.0, 1.0, n // 2, mult=2))
    elif args.config == 'c4g':
        kvs = (kv0, bspline.make_knots(p - 2, 0.0, 1.0, n), kv)
    row0 = distributed.slab_range(kv0.numdofs, part_rank, part_world, p if dim == 3 else None)      # balanced by work
    if args.op == 'form':
        # the matrix of the config's patch from a FORM STRING through the front-end (pyiga.assemble.assemble: one generated kernel per
        # form in the reference, pyiga/codegen/cython.py:325-387): stiffness + mass, a symmetric table of constants
        assert dim == 3 and kind == 'stiffness' and not stub, '--op form: the 3D stiffness configs'
        kind = 'form'
        patch = assemblers.GeneralFormAssembler3D(kvs, geo, FORM_STRING, device=local_rank, row0=row0 if part_world > 1 else None).patch
    elif kind == 'convdiff':
        patch = assemblers.ConvDiffAssembler3D(kvs, geo, assemblers.AffineCoefficient(1.0, 1.0), device=local_rank, row0=row0 if part_world > 1 else None).patch
    else:
        patch = assemblers.DevicePatch(kvs, geo, device=local_rank, row0=row0 if part_world > 1 else None)
    patch.ctx.sync()
    setup_s = time.perf_counter() - t_setup                 # knots, tables, plan, coefficient sampling + upload (context warm)
    nel_total = int(np.prod([k.numspans for k in kvs]))
    if emu is not None:
        nel_total //= part_world        # one slab's share
    nnz_local = patch.nnz

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            if not os.environ.get('BENCH_PATCH_STUB'):
                torch.cuda.synchronize()
        patch.ctx.sync()

    if args.op == 'rhs':
        return bench_rhs(args, patch, kvs, dim, p, n0, n, nel_total, world, rank, dist, barrier)
    if args.op == 'entries':
        return bench_entries(args, patch, kvs, dim, p, kind, n0, n, rank)
    if args.op == 'fast':
        return bench_fast(args, patch, kvs, dim, p, kind, n0, n, nel_total)
    if args.op == 'structured':
        return bench_structured(args, patch, kvs, geo, dim, p, kind, nel_total, assemblers)
    # cold assembly: the first one of the patch allocates the workspaces (K1, CSR values) and builds the per-plane table
    t_cold = time.perf_counter()
    patch.assemble(kind, algo=args.algo, to_host=False)
    cold_ms = 1e3 * (setup_s + time.perf_counter() - t_cold)       # patch creation + first assembly, result resident on the device
    for _ in range(args.warmup):
        patch.assemble(kind, algo=args.algo, to_host=False)
    stage_ms, steps_ms = {}, []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        patch.assemble(kind, algo=args.algo, to_host=False)     # returns after the ctx stream has drained
        tm = patch.timing()
        for k, v in tm.items():
            if k.endswith('_ms'):
                stage_ms[k] = stage_ms.get(k, 0.0) + v
        steps_ms.append(tm['total_ms'])
        algo_used = tm['algo_used']
    barrier()
    dt = time.perf_counter() - t0
    slab_ms = [float(np.median(steps_ms))]
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device=_tdev())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        z = torch.tensor([float(nnz_local)], dtype=torch.float64, device=_tdev())
        dist.all_reduce(z, op=dist.ReduceOp.SUM)
        nnz_total = int(z.item())
        sl = [torch.zeros(3, dtype=torch.float64, device=_tdev()) for _ in range(world)]
        dist.all_gather(sl, torch.tensor([slab_ms[0], setup_s, cold_ms], dtype=torch.float64, device=_tdev()))
        slab_ms = [float(x[0].item()) for x in sl]
        setup_all = [round(float(x[1].item()), 4) for x in sl]
        cold_all = [round(float(x[2].item()), 2) for x in sl]
    else:
        nnz_total = nnz_local
        setup_all, cold_all = [round(setup_s, 4)], [round(cold_ms, 2)]
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # 2D and small 3D chains run without events between their kernels (a marker costs ~5 us of stream time, as much as a
    # 2D kernel): the per-kernel times then come from a separate pass over a second patch created with IGX_STAGE_EVENTS=1
    kernel_ms_source = 'same pass'
    if not stub and kind in ('stiffness', 'mass') and 'single' not in patch.last_path() and not any(stage_ms.get(k, 0.0) > 0 for k in ('stage0_ms', 'stage1_ms', 'final_ms', 'entry_ms')):
        prev_ev = os.environ.get('IGX_STAGE_EVENTS')
        os.environ['IGX_STAGE_EVENTS'] = '1'
        p2 = assemblers.DevicePatch(kvs, geo, device=local_rank, row0=row0 if part_world > 1 else None)
        if prev_ev is None:
            del os.environ['IGX_STAGE_EVENTS']
        else:
            os.environ['IGX_STAGE_EVENTS'] = prev_ev
        stage_ms = {}
        for it in range(args.warmup + args.steps):
            p2.assemble(kind, algo=args.algo, to_host=False)
            if it >= args.warmup:
                for k, v in p2.timing().items():
                    if k.endswith('_ms'):
                        stage_ms[k] = stage_ms.get(k, 0.0) + v
        del p2
        kernel_ms_source = 'separate pass with stage events (its chain: %.4f ms per step)' % (stage_ms.pop('total_ms') / args.steps)

    api_call_s = None
    if world == 1 and emu is None and not args.no_api_call and kind in ('stiffness', 'mass'):
        api_call_s = api_call(args, kvs, geo, kind, nnz_total)
    ms_per_step = 1e3 * dt / args.steps
    value = nel_total * args.steps / dt
    stage_ms = {k: v / args.steps for k, v in stage_ms.items()}
    # roofline on rank 0's slab: algorithmic bytes of one assembly / device time of the kernel chain (median step)
    nel_rank = nel_total / world
    mkind = 'stiffness' if kind == 'form' else kind          # (the form of --op form is stiffness + mass: the stiffness models)
    b_el = algorithmic_bytes_per_element(dim, p, nnz_total, nel_total, mkind)
    chain_ms = float(np.median(steps_ms))
    achieved = b_el * nel_rank / (chain_ms * 1e-3) / 1e9
    path = patch.last_path()
    fused = 'fused' in path
    names = {'fields_ms': 'k_geo_fields', 'stage0_ms': 'k_geoA' if 'geoA' in path else 'k_stageA',
             'stage1_ms': 'k_single2d' if 'single' in path else ('k_bf3' if 'bf3' in path else 'k_bf2') if fused else 'k_stageB',
             'final_ms': ('k_mirror2' if dim == 3 and p in (2, 3, 4) else 'k_mirror') if 'mirror' in path else 'k_final', 'entry_ms': 'k_entries_csr'}
    if 'geoA' in path or 'single' in path:
        stage_ms.pop('fields_ms', None)         # no field kernel: the geometry is evaluated inside k_geoA / k_single2d
    if 'single' in path:
        stage_ms = {k: v for k, v in stage_ms.items() if k == 'stage1_ms'}
    if 'bf3' in path or 'both' in path:
        stage_ms.pop('final_ms', None)          # k_bf3 writes both triangles (or the form has one): no kernel behind it
    if dim == 2 and not fused and 'single' not in path:
        stage_ms.pop('stage1_ms', None)         # 2D stage chain: axis-0 sweep -> final stage, nothing between the two events
    parts = {names[k]: round(v, 4) for k, v in stage_ms.items() if k in names and v > 0}
    dominant = max(parts, key=parts.get) if parts else None
    flops = algorithmic_flops(dim, p, kvs, mkind) if algo_used == 2 else None
    fp64 = None
    if flops:
        share = part_world if emu is not None else world    # flops of ONE slab (approximately: halo planes not counted)
        tot = sum(flops.values()) / share
        nel_patch = nel_total * (part_world if emu is not None else 1)
        fp64 = {'flops_per_element': {k: round(v / nel_patch, 1) for k, v in flops.items()}, 'total_gflop': round(tot / 1e9, 2),
                'achieved': tot / (chain_ms * 1e-3) / 1e12, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                'frac': tot / (chain_ms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS, 'algorithm': 'global sum factorisation, useful flops only'}
    out = {
        'metric': 'assembled elements/sec (3D p=4, 128^3 spans) + HBM-roofline %; 1/2/4/8 GPU'
                  if args.config == 'c4' and args.op == 'matrix' else 'assembled elements/sec',
        'value': value, 'unit': 'elements/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': 'weak' if weak else 'strong',
        'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic' if not stub else 'STUB (launcher test, no device work)',
        'config': {'workload': '%dD p=%s %s, %s spans, NURBS quarter-annulus %s, %s'
                               % (dim, p if len({k.p for k in kvs}) == 1 else 'x'.join(str(k.p) for k in kvs), kind if kind != 'form' else 'form string %r' % FORM_STRING,
                                  'x'.join(str(k.numspans) for k in kvs), 'cylinder' if dim == 3 else gname,
                                  'double interior knots on the mid axis' if args.config == 'c4k' else 'double interior knots on the last axis' if args.config == 'c4l' else 'uniform open knots'),
                   'config': args.config, **({'emulated_slab': args.emulate} if emu is not None else {}), 'elements': nel_total, 'nnz': nnz_total, 'dofs': int(np.prod([k.numdofs for k in kvs])),
                   'algo': {1: 'entrywise', 2: 'sumfact'}.get(algo_used, str(algo_used)),
                   'path': ' + '.join(parts) + (' (on the twin patch: mid and last axis exchanged, values stored to this layout)' if 'twin' in path else ''),
                   'parallelism': 'row slabs of axis-0 dof planes, %d rank(s), no data-path collective' % world},
        'step_ms': {'median': chain_ms, 'min': float(np.min(steps_ms)), 'max': float(np.max(steps_ms))},
        'slab_ms': [round(x, 3) for x in slab_ms],
        # what the work model of the slabs expects of this split (distributed.scaling_model): the first SCALE curve explains itself
        **({'scaling_model': distributed.scaling_model(kv0.numdofs, part_world, p)} if part_world > 1 and dim == 3 and not weak else {}),
        'setup_s': round(max(setup_all), 4), 'setup_s_ranks': setup_all,      # patch creation per rank (device context warm)
        'runtime_init_s': None if runtime_init_s is None else round(runtime_init_s, 3),                            # HIP runtime + context + stream of rank 0: once per process
        'cold_ms': round(max(cold_all), 2), 'cold_ms_ranks': cold_all,         # patch creation + FIRST assembly (workspace allocation), device-resident result
        # opt-in of this caller (--placement-tries): the CSR buffer is the fastest of n allocations under the mirror pass, chosen at the
        # first assembly (its cost is part of cold_ms); ms of the pass on the kept / the slowest candidate
        'placement': (patch.placement() if hasattr(patch, 'placement') else None),
        'api_call_s': api_call_s,                                              # assemble.stiffness() end to end: + pattern, D2H of values and indices, scipy
        'roofline': {
            'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
            'traffic': measured_traffic(args.config, world, op=args.op, kernels_now=list(parts)),
            'design': design_floor(args.config if args.op == 'matrix' else '%s_%s' % (args.config, args.op), dim, kvs, mkind, nnz_total, b_el * nel_total, world),
            'kernel': 'assembly chain (' + ' + '.join(parts) + '), HIP events on the igx stream; median step',
            'algorithmic_bytes_per_element': b_el, 'algorithmic_bytes_def': ALGORITHMIC_BYTES_DEF, 'chain_ms': chain_ms, 'kernel_ms': parts, 'kernel_ms_source': kernel_ms_source, 'dominant_kernel': dominant,
            'fp64': fp64,
        },
    }
    if cpu is not None:
        out['cpu_baseline'] = cpu
    if dist is not None:
        dist.destroy_process_group()        # before the result line: anything RCCL prints comes first
    flush_c_stdio()
    print(json.dumps(out), flush=True)


def bench_rhs(args, patch, kvs, dim, p, n0, n, nel_total, world, rank, dist, barrier):
    """Load vector of the same patch.  `value` uses the device time of the contractions with the
    function values resident in HBM (igx_load_vector_d); the whole host-pointer call incl. the upload is noted."""
    from pyiga_amd import utils
    from pyiga_amd import symbolic
    grid = tuple(patch.gauss(k)[0] for k in range(dim))
    f = lambda *x: np.cos(x[0]) * np.exp(x[1]) * (np.sin(x[2]) if dim == 3 else 1.0)
    fvals = np.ascontiguousarray(utils.grid_eval(f, grid))
    t0 = time.perf_counter()
    patch.load_vector(fvals)
    host_call_ms = 1e3 * (time.perf_counter() - t0)
    # the same function traced into C and evaluated at the Gauss points by a run-time compiled kernel (what
    # assemble.inner_products does with a plain callable): second call = code object from the cache
    compiled_call_ms = compiled_dev_ms = None
    src = symbolic.trace_function(f, dim)
    if src is not None:
        for _ in range(2):
            t0 = time.perf_counter()
            patch.load_vector_expr(src, parametric=True)
            compiled_call_ms = 1e3 * (time.perf_counter() - t0)
        compiled_dev_ms = patch.timing()['total_ms']
    patch.upload_function(fvals)                      # resident from here on
    for _ in range(args.warmup):
        patch.load_vector_resident()
    barrier()
    dev, t0 = [], time.perf_counter()
    for _ in range(args.steps):
        patch.load_vector_resident()
        dev.append(patch.timing()['total_ms'])
    barrier()
    wall_ms = 1e3 * (time.perf_counter() - t0) / args.steps
    dev_ms = float(np.median(dev))
    if dist is not None:
        import torch
        t = torch.tensor([dev_ms, wall_ms], dtype=torch.float64, device=_tdev())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dev_ms, wall_ms = float(t[0].item()), float(t[1].item())
    if rank != 0:
        return
    q = p + 1
    b_el = 8.0 * q ** dim + 8.0 * np.prod([k.numdofs for k in kvs]) / nel_total      # f in + vector out (W is resident)
    nel_rank = nel_total / world
    achieved = b_el * nel_rank / (dev_ms * 1e-3) / 1e9
    # (3D, equal degrees 2..5: the last two axes are contracted by one kernel -- two launches; else one launch per axis)
    kernel_names = 'k_lv12 + k_contract_axis' if dim == 3 and patch.timing().get('n_launches') == 2 else 'k_contract_axis x %d' % dim
    out = {'metric': 'load-vector elements/sec', 'value': nel_total / (dev_ms * 1e-3), 'unit': 'elements/s', 'n_gpus': world,
           'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dev_ms, 'higher_is_better': True, 'scaling': 'weak' if args.weak else 'strong',
           'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': '%dD p=%d load vector (inner_products), %s spans' % (dim, p, 'x'.join(str(x) for x in (n0,) + (n,) * (dim - 1))),
                      'config': args.config, **({'emulated_slab': args.emulate} if args.emulate else {}), 'elements': nel_total,
                      'note': 'value = device time of the contractions, function values resident (igx_load_vector_d); whole resident call '
                              '%.2f ms wall; a host-pointer call incl. the upload of the function values %.1f ms%s' % (
                                  wall_ms, host_call_ms, '' if compiled_call_ms is None else
                                  '; the whole call with the function compiled for the device (traced callable, igx_load_vector_expr: the function is evaluated inside a generated variant of the fused contraction kernel, no array of function values), result on the host: %.1f ms, of which on the device %.2f ms' % (compiled_call_ms, compiled_dev_ms))},
           'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                        'traffic': measured_traffic(args.config, world, 'rhs'), 'kernel': kernel_names, 'algorithmic_bytes_per_element': b_el}}
    flush_c_stdio()
    print(json.dumps(out), flush=True)


def bench_entries(args, patch, kvs, dim, p, kind, n0, n, rank):
    """Batched multi_entries: 1 M seeded in-pattern pairs, index pairs and results resident on the device."""
    rng = np.random.default_rng(5)
    nd = [k.numdofs for k in kvs]
    M = 1 << 20
    I = np.stack([rng.integers(0, x, M) for x in nd], 1)
    J = np.clip(I + rng.integers(-p, p + 1, (M, dim)), 0, np.array(nd) - 1)
    idx = np.stack([np.ravel_multi_index(I.T, nd), np.ravel_multi_index(J.T, nd)], 1).astype(np.uintp)
    for _ in range(max(1, args.warmup)):
        patch.entries(kind, idx[:1024])
    t0 = time.perf_counter()
    patch.entries(kind, idx)
    host_ms = 1e3 * (time.perf_counter() - t0)
    patch.upload_pairs(idx)
    dev = []
    for _ in range(args.steps):
        patch.entries_resident(kind)
        dev.append(patch.timing()['entry_ms'])
    dev_ms = float(np.median(dev))
    q = p + 1
    # arithmetic of the request: Gauss points of every pair's support intersection (uniform open knots: p + 1 - |i - j| spans per
    # axis, fewer at the ends) x the reference's flops per point of its entry sum (33 for 3D stiffness; 2 d^2 + 3 d + 6 else)
    pts = np.ones(M)
    for k, kv in enumerate(kvs):
        supp = kv.mesh_support_idx_all()
        lo = np.maximum(supp[I[:, k], 0], supp[J[:, k], 0])
        hi = np.minimum(supp[I[:, k], 1], supp[J[:, k], 1])
        pts *= np.maximum(hi - lo, 0) * q
    flop_pt = {('stiffness', 3): 33.0, ('stiffness', 2): 17.0, ('mass', 3): 7.0, ('mass', 2): 5.0}.get((kind, dim), 2.0 * dim * dim + 3 * dim + 6)
    tflops = float(pts.sum()) * flop_pt / (dev_ms * 1e-3) / 1e12
    nfld = {'stiffness': dim * (dim + 1) // 2, 'mass': 1}.get(kind, 9)
    gbs = float(pts.sum()) * 8.0 * nfld / (dev_ms * 1e-3) / 1e9
    # bytes one entry must read: its fields on the support intersection (avg ((p+1) q / 2)^d points x d(d+1)/2 fields) -- L2-resident reuse aside
    out = {'metric': 'multi_entries pairs/sec', 'value': M / (dev_ms * 1e-3), 'unit': 'entries/s', 'n_gpus': 1, 'steps': args.steps,
           'warmup': args.warmup, 'ms_per_step': dev_ms, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64',
           'data': 'synthetic',
           'config': {'workload': '%dD p=%d %s multi_entries, %d random in-pattern pairs, %s spans' % (dim, p, kind, M, 'x'.join(str(x) for x in (n0,) + (n,) * (dim - 1))),
                      'config': args.config, 'note': 'device time with resident pairs; host-pointer call (upload pairs, download values) %.1f ms' % host_ms},
           'roofline': {'bound': 'hbm', 'note': 'every Gauss point of a pair\'s support intersection reads its %d field values (random pairs: no reuse '
                        'between pairs beyond L2 / Infinity Cache); %.3g points in this request' % (nfld, float(pts.sum())),
                        'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'traffic': measured_traffic(args.config, 1, 'entries'),
                        'fp64': {'flops_per_point': flop_pt, 'achieved': tflops, 'peak': FP64_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tflops / FP64_PEAK_TFLOPS}}}
    print(json.dumps(out), flush=True)


def bench_structured(args, patch, kvs, geo, dim, p, kind, nel_total, assemblers):
    """Opt-in structured path: for a geometry that is separable along axis 0 the matrix is
    M0 (x) K2D + K0 (x) M2D; the 2D matrices of the cross-section are assembled by the regular 2D path and expanded into the CSR
    values by one store-bound kernel.  NOT the headline: the general quadrature chain is (it serves every geometry)."""
    from pyiga_amd import assemble
    assert dim == 3 and kind in ('mass', 'stiffness'), 'the structured path knows the 3D mass and stiffness forms'
    t0 = time.perf_counter()
    terms = assemble.separable_terms(kvs, geo, patch)
    if terms is None:
        raise SystemExit('bench.py --op structured: the geometry of this config is not separable along axis 0')
    geo2, m0, k0 = terms
    p2 = assemblers.DevicePatch(kvs[1:], geo2, nqp=patch.nqp)
    setup_s = time.perf_counter() - t0
    for _ in range(max(1, args.warmup)):
        patch.assemble_kron(kind, p2, m0, k0, to_host=False)
    dev, t0 = [], time.perf_counter()
    for _ in range(args.steps):
        patch.assemble_kron(kind, p2, m0, k0, to_host=False)
        dev.append(patch.timing()['total_ms'])
    wall_ms = 1e3 * (time.perf_counter() - t0) / args.steps
    dev_ms = float(np.median(dev))
    nnz = patch.nnz
    gbs = 8.0 * nnz / (dev_ms * 1e-3) / 1e9
    b_el = algorithmic_bytes_per_element(dim, p, nnz, nel_total, kind)
    out = {'metric': 'assembled elements/sec, separable geometry (Kronecker expansion, opt-in)', 'value': nel_total / (wall_ms * 1e-3), 'unit': 'elements/s',
           'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': wall_ms, 'higher_is_better': True, 'scaling': 'strong',
           'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': '%dD p=%d %s, %s spans, geometry separable along axis 0' % (dim, p, kind, 'x'.join(str(k.numspans) for k in kvs)),
                      'config': args.config, 'elements': nel_total, 'nnz': nnz,
                      'note': 'a step = both 2D assemblies of the cross-section + the expansion (wall clock per call); k_kron3 alone %.3f ms; '
                              'set-up (separability test on the control net, 1D matrices on the host, 2D patch) %.3f s' % (dev_ms, setup_s)},
           'roofline': {'bound': 'hbm', 'achieved': gbs, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS, 'traffic': None,
                        'kernel': 'k_kron3 (store stream of the CSR values)', 'kernel_ms': {'k_kron3': round(dev_ms, 4)},
                        'algorithmic_bytes_per_element': 8.0 * nnz / nel_total,
                        'note': 'bytes = the CSR values written (this algorithm reads no Jacobians); with the bytes of SURVEY 8d (%.0f B/el) the '
                                'same time would read as %.2f of the roof' % (b_el, b_el * nel_total / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)}}
    p2.close()
    print(json.dumps(out), flush=True)


def bench_fast(args, patch, kvs, dim, p, kind, n0, n, nel_total):
    """Low-rank (ACA) assembler (pyiga/fastasm.cc, mass_fast / stiffness_fast) as a consumer of batched entries:
    wall time of a whole call (host control flow + device requests + inflation to CSR) beside the exact assembly."""
    assert kind in ('mass', 'stiffness'), 'the low-rank assembler knows the mass and stiffness forms'
    S = [kv.numdofs * (2 * kv.p + 1) - kv.p * (kv.p + 1) for kv in kvs]
    if 8.0 * float(np.prod(S)) > 16e9:
        raise SystemExit('bench.py --op fast: the reordered tensor of this config needs %.0f GB of host memory; use --config c2 or c3'
                         % (8e-9 * float(np.prod(S))))
    t0 = time.perf_counter()
    exact = patch.assemble(kind, algo=args.algo, to_host=True)
    exact_wall = time.perf_counter() - t0
    t0 = time.perf_counter()
    exact = patch.assemble(kind, algo=args.algo, to_host=True)
    exact_wall = min(exact_wall, time.perf_counter() - t0)
    exact_dev_ms = patch.timing()['total_ms']
    patch.pattern()
    walls = []
    for _ in range(max(1, min(args.steps, 3))):
        t0 = time.perf_counter()
        A = patch.fast_assemble(kind, tol=1e-10)
        walls.append(time.perf_counter() - t0)
    st = patch.aca_stats
    err = float(np.abs(A.data - exact).max() / np.abs(exact).max())
    wall = float(np.median(walls))
    out = {'metric': 'low-rank (ACA) assembly, elements/sec of a whole call', 'value': nel_total / wall, 'unit': 'elements/s', 'n_gpus': 1,
           'steps': len(walls), 'warmup': 0, 'ms_per_step': 1e3 * wall, 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
           'dtype': 'f64', 'data': 'synthetic',
           'config': {'workload': '%dD p=%d %s, %s spans, tol 1e-10' % (dim, p, kind, 'x'.join(str(x) for x in (n0,) + (n,) * (dim - 1))),
                      'config': args.config},
           'aca': {'crosses': st['rank'], 'requests': st['requests'], 'entries_evaluated': st['entries'], 'nnz': st['nnz'],
                   'max_rel_error_vs_exact': err},
           'exact': {'wall_s_with_copy_to_host': round(exact_wall, 4), 'device_ms': round(exact_dev_ms, 3)},
           'roofline': {'bound': 'host', 'note': 'control flow, rank-1 updates of the reordered tensor and the inflation run on the host; the device '
                        'serves %d requests' % st['requests'], 'achieved': None, 'peak': None, 'unit': None, 'frac': None, 'traffic': None}}
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
