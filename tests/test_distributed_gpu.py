"""Two ranks on ONE MI355X (both on device 0; the GPU box has a single GPU): each runs the product path
pyiga_amd.distributed.assemble_rows for its row slab, rank 0 gathers, and the stacked result equals the single-process
matrix bit for bit.  The ranks are child processes started before this process touches the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest
import scipy.sparse

from test_distributed_cpu import _free_port

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_one_gpu(tmp_path):
    out = str(tmp_path / 'full.npz')
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, '_dist_worker.py'), str(r), '2', port, 'gpu', out], env=env)
             for r in range(2)]
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0, 0]
    z = np.load(out)
    full = scipy.sparse.csr_matrix((z['data'], z['indices'], z['indptr']), shape=tuple(z['shape']))
    import pyiga_amd as iga
    kv = iga.bspline.make_knots(3, 0.0, 1.0, 11)
    geo = iga.geometry.tensor_product(iga.geometry.line_segment(0.0, 1.0), iga.geometry.quarter_annulus())
    ref = iga.assemble.stiffness((kv,) * 3, geo)
    assert full.shape == ref.shape and full.nnz == ref.nnz
    assert np.array_equal(full.indptr, ref.indptr) and np.array_equal(full.indices, ref.indices)
    assert np.array_equal(full.data, ref.data)


def test_bench_two_ranks_share_one_gpu():
    """bench.py --gpus 2 launches its own ranks; BENCH_SHARE_GPU=1 puts both on device 0 with a gloo rendezvous (RCCL refuses
    two ranks on one device): the slab split, the max/sum reductions and the result line of the N > 1 path."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--config', 'tiny', '--steps', '2', '--warmup', '1',
                          '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and len(d['slab_ms']) == 2
    assert d['config']['elements'] == 12 ** 3 and d['value'] > 0


def test_bench_eight_ranks_share_one_gpu():
    """The world size of the round-end SCALE run: bench.py --gpus 8 starts eight ranks (all on device 0, gloo rendezvous), each
    takes its work-balanced slab of the tiny patch, and the eight slabs tile its pattern."""
    import json
    root = os.path.dirname(HERE)
    env = dict(os.environ, BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--config', 'tiny', '--steps', '2', '--warmup', '1',
                          '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 8 and d['scaling'] == 'strong' and len(d['slab_ms']) == 8 and all(x > 0 for x in d['slab_ms'])
    N, p = 14, 2
    S = sum(min(i + p, N - 1) + 1 - max(i - p, 0) for i in range(N))
    assert d['config']['elements'] == 12 ** 3 and d['config']['nnz'] == S ** 3 and d['value'] > 0
