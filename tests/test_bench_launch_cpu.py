"""bench.py --gpus 2 without a launcher and without a GPU: the script starts its two ranks itself, they meet over gloo
(BENCH_SHARE_GPU), each takes its work-balanced slab, the timing is reduced and rank 0 prints ONE valid JSON line.  The
device patch is replaced by tests/_bench_stub.py -- this test is about the launcher, not about arithmetic."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, extra_env=None, timeout=300):
    env = dict(os.environ, BENCH_PATCH_STUB=os.path.join(ROOT, 'tests', '_bench_stub.py'), BENCH_SHARE_GPU='1',
               BENCH_LAUNCH_TIMEOUT='120')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_self_launch_two_ranks_prints_one_valid_line():
    r = _run(['--gpus', '2', '--steps', '3', '--warmup', '1', '--config', 'tiny', '--no-cpu-baseline'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1 and r.stdout.rstrip().splitlines()[-1] == lines[0]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['warmup'] == 1 and d['scaling'] == 'strong' and d['unit'] == 'elements/s'
    assert d['value'] > 0 and d['ms_per_step'] > 0 and d['higher_is_better'] is True
    assert len(d['slab_ms']) == 2 and len(d['setup_s_ranks']) == 2 and len(d['cold_ms_ranks']) == 2
    assert d['config']['elements'] == 12 ** 3 and 'roofline' in d
    # the slabs tile the patch: nnz of the two ranks add up to the whole pattern
    N, p = 14, 2
    S = sum(min(i + p, N - 1) + 1 - max(i - p, 0) for i in range(N))
    assert d['config']['nnz'] == S ** 3


def test_single_rank_line_and_failed_rank_is_reported():
    r = _run(['--steps', '2', '--warmup', '0', '--config', 'tiny', '--no-cpu-baseline', '--no-api-call'])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.rstrip().splitlines()[-1])
    assert d['n_gpus'] == 1 and d['slab_ms'] and d['setup_s_ranks'] and d['api_call_s'] is None
    # a rank that dies takes the run down quickly (no hang in the rendezvous) and the parent says which one
    bad = _run(['--gpus', '2', '--steps', '1', '--config', 'tiny', '--no-cpu-baseline'],
               extra_env={'BENCH_PATCH_STUB': os.path.join(ROOT, 'tests', 'does_not_exist.py')}, timeout=120)
    assert bad.returncode != 0 and 'rank' in (bad.stderr + bad.stdout)


def test_self_launch_eight_ranks():
    """world size 8 (what the driver launches for the scaling curve): eight ranks meet, eight slabs tile the patch."""
    r = _run(['--gpus', '8', '--steps', '2', '--warmup', '1', '--config', 'tiny', '--no-cpu-baseline'], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 8 and len(d['slab_ms']) == 8 and len(d['setup_s_ranks']) == 8
    N, p = 14, 2
    S = sum(min(i + p, N - 1) + 1 - max(i - p, 0) for i in range(N))
    assert d['config']['nnz'] == S ** 3 and 'scaling_model' in d
