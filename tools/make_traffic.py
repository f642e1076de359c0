#!/usr/bin/env python3
"""profiles/r02_traffic.json from the PMC passes of tools/prof_r02.sh (gpurun_out/r02/pmc/*): HBM-side bytes per kernel
of one C4 assembly, with the hash of the kernel sources they were measured on (bench.py drops the figure when the
sources have changed since).  usage: python3 tools/make_traffic.py gpurun_out/r02/pmc [config]"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernels_sha():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, 'pyiga_amd', 'csrc', '*.hip')) + glob.glob(os.path.join(ROOT, 'pyiga_amd', 'csrc', '*.h'))):
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def main():
    root, config = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'c4')
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r'[<(].*', '', row['Kernel_Name']).replace('void igx::', '').replace('igx::', '')
            if name in ('k_geoA', 'k_bf', 'k_mirror', 'k_geo_fields', 'k_stageA', 'k_stageB', 'k_final', 'k_final_q'):
                acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
    kernels, lo, hi = {}, 0.0, 0.0
    for name, c in sorted(acc.items()):
        fetch = sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE']) * 1024
        write = sum(c['WRITE_SIZE']) / len(c['WRITE_SIZE']) * 1024
        # coalesced reads (whole 128-byte lines per request) are tallied at 64 B: x2, calibrated on k_bf, whose only input
        # is K1 (17.04 GB, every value read once: 2 * FETCH_SIZE = 17.24 GB).  The mirror gathers 72-byte runs: its
        # request size is not calibrated, so both readings are given.
        kernels[name] = {'write_bytes': write, 'read_bytes_x2': 2 * fetch, 'read_bytes_x1': fetch}
        lo += write + (fetch if name == 'k_mirror' else 2 * fetch)
        hi += write + 2 * fetch
    out = {config: {'config': config,
                    'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE in separate passes (tools/prof_r02.sh), '
                              'summary in profiles/r02_c4_pmc_summary.txt',
                    'correction': 'read bytes = 2 * FETCH_SIZE * 1024 for coalesced reads (gfx950), write bytes = WRITE_SIZE * 1024',
                    'kernels': kernels, 'chain_bytes': hi, 'chain_bytes_low': lo, 'kernels_sha': kernels_sha()}}
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'r02_traffic.json'), 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
