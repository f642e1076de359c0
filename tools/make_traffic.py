#!/usr/bin/env python3
"""profiles/r06_traffic.json from the PMC passes of tools/prof_r06.sh pmc (gpurun_out/r06/pmc_<key>/{fetch,write}): memory-side
bytes per kernel of ONE pass of the timed operation -- keys c2, c3, c4, c5 (one assembly), c4_rhs (one load vector),
c4_entries (one batched multi_entries request) -- with the hash of the kernel sources they were measured on (bench.py drops
the figure when the sources have changed since).  usage: python3 tools/make_traffic.py gpurun_out/r06
The vector-instruction counts of the SQ passes (tools/prof_r06.sh sq -> pmc_<key>sq/sq1: SQ_INSTS_VALU) are added per kernel where
present: bench.py turns them into the issue-time floor it prints beside the roofline (roofline.design.issue_ms).

Reads: 2 * FETCH_SIZE * 1024 for EVERY kernel -- a read request is a whole 128-byte line and the counter tallies it at 64
bytes, for coalesced streams and for the 72-byte gathers of the mirror pass alike (profiles/r03_fetch_calibration.txt,
tools/ubench/fetch_calib.hip).  Writes: WRITE_SIZE * 1024 (exact on the K1 stream of k_geoA).  Per kernel the MEAN over its
dispatches in the profiled process (the cold pass and the timed step launch the same kernels on the same data)."""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# set-up kernels (basis tables, pattern, coefficient sampling, per-plane records) are not part of a timed step; every other
# igx kernel of the profiled process is
SETUP = {'k_basis_tables', 'k_pi_tables', 'k_pattern', 'k_coeff_affine', 'k_geoa_table', 'k_grid_geo', 'k_box_pairs'}


def kernels_sha():
    """hash of the device code objects inside pyiga_amd/libigx.so (the build the passes were taken on)"""
    sys.path.insert(0, ROOT)
    from pyiga_amd import _lib
    return _lib.device_code_sha(os.path.join(ROOT, 'pyiga_amd', 'libigx.so'))


def one(root, key):
    """Per kernel the LAST dispatch of the profiled process: the timed step (earlier dispatches are the cold pass of the same
    operation -- same kernels, same data; for the batched entries the warm-up request is a small one)."""
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, 'pmc_' + key) + '/**/*counter_collection.csv', recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Dispatch_Id']))
        for row in rows:
            name = re.sub(r'[<(].*', '', row['Kernel_Name']).replace('void igx::', '').replace('igx::', '')
            if name.startswith('k_') and name not in SETUP and not (key.endswith(('_rhs', '_entries')) and name.startswith('k_geo_fields')):
                acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
    kernels, total = {}, 0.0
    for name, c in sorted(acc.items()):
        if not c['FETCH_SIZE'] or not c['WRITE_SIZE']:
            continue
        fetch, write = c['FETCH_SIZE'][-1] * 1024, c['WRITE_SIZE'][-1] * 1024
        kernels[name] = {'write_bytes': write, 'read_bytes': 2 * fetch, 'dispatches_seen': len(c['FETCH_SIZE'])}
        total += write + 2 * fetch
    if not kernels:
        return None
    # vector instructions per kernel (wave-level) from the SQ pass of the same configuration, if it was taken
    for f in glob.glob(os.path.join(root, 'pmc_' + key.split('_')[0] + 'sq') + '/**/*counter_collection.csv', recursive=True) if '_' not in key else []:
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Dispatch_Id']))
        for row in rows:
            name = re.sub(r'[<(].*', '', row['Kernel_Name']).replace('void igx::', '').replace('igx::', '')
            if name in kernels and row['Counter_Name'] == 'SQ_INSTS_VALU':
                kernels[name]['valu_insts'] = float(row['Counter_Value'])
    return {'config': key,
            'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE in separate passes (tools/prof_r06.sh pmc), plain buffer allocation',
            'correction': 'read bytes = 2 * FETCH_SIZE * 1024 (128-byte requests tallied at 64 bytes; calibrated for streams and '
                          'for 72-byte gathers: profiles/r03_fetch_calibration.txt), write bytes = WRITE_SIZE * 1024',
            'kernels': kernels, 'chain_bytes': total, 'kernels_sha': kernels_sha()}


def main():
    root = sys.argv[1]
    out = {}
    for key in ('c4', 'c5', 'c3', 'c2', 'c4_rhs', 'c4_entries', 'c4_form'):
        r = one(root, key)
        if r:
            out[key] = r
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'r06_traffic.json'), 'w'), indent=1)
    print(json.dumps({k: {'chain_GB': round(v['chain_bytes'] / 1e9, 3), 'kernels': {n: [round(x['read_bytes'] / 1e9, 3), round(x['write_bytes'] / 1e9, 3)] for n, x in v['kernels'].items()}}
                      for k, v in out.items()}, indent=1))


if __name__ == '__main__':
    main()
