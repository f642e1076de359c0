"""Stand-in for pyiga_amd.assemblers.DevicePatch used by tests/test_bench_launch_cpu.py (BENCH_PATCH_STUB): no device, no
arithmetic -- it only has the interface bench.py touches, so that the launch / rendezvous / reduction / result-line logic of
bench.py runs on a box without a GPU."""
import time

import numpy as np


class _Ctx:
    def sync(self):
        pass


class DevicePatch:
    def __init__(self, kvs, geo, device=None, row0=None, jacobian=None):
        self.kvs, self.ctx = tuple(kvs), _Ctx()
        N = [kv.numdofs for kv in self.kvs]
        lo, hi = row0 if row0 is not None else (0, N[0])
        self.row0 = (lo, hi)
        p = self.kvs[0].p
        cnt = lambda i, n: min(i + p, n - 1) + 1 - max(i - p, 0)
        S = [sum(cnt(i, n) for i in range(n)) for n in N]
        self.nnz = sum(cnt(i, N[0]) for i in range(lo, hi)) * int(np.prod(S[1:]))
        self.shape = ((hi - lo) * int(np.prod(N[1:])), int(np.prod(N)))

    def assemble(self, kind, algo='auto', to_host=False):
        time.sleep(0.002 * (self.row0[1] - self.row0[0]) / self.kvs[0].numdofs)

    def timing(self):
        return {'total_ms': 2.0, 'fields_ms': 0.0, 'stage0_ms': 0.5, 'stage1_ms': 1.0, 'final_ms': 0.5, 'entry_ms': 0.0, 'algo_used': 2}

    def last_path(self):
        return {'geoA', 'fused', 'both', 'bf3'}

    def close(self):
        pass
