#!/usr/bin/env python3
"""profiles/r03_traffic.json from the PMC passes of tools/prof_r03.sh (gpurun_out/r03/pmc/*): memory-side bytes per kernel
of one C4 assembly, with the hash of the kernel sources they were measured on (bench.py drops the figure when the
sources have changed since).  usage: python3 tools/make_traffic.py gpurun_out/r03/pmc [config]

Reads: 2 * FETCH_SIZE * 1024 for EVERY kernel -- a read request is a whole 128-byte line and the counter tallies it at 64
bytes, for coalesced streams and for the 72-byte gathers of the mirror pass alike (profiles/r03_fetch_calibration.txt,
tools/ubench/fetch_calib.hip).  Writes: WRITE_SIZE * 1024 (exact on the K1 stream of k_geoA)."""
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
            name = {'k_bf2': 'k_bf', 'k_mirror2': 'k_mirror'}.get(name, name)
            if name in ('k_geoA', 'k_bf', 'k_mirror', 'k_geo_fields', 'k_stageA', 'k_stageB', 'k_final', 'k_final_q'):
                acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
    kernels, total = {}, 0.0
    for name, c in sorted(acc.items()):
        if not c['FETCH_SIZE'] or not c['WRITE_SIZE']:
            continue
        fetch = sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE']) * 1024
        write = sum(c['WRITE_SIZE']) / len(c['WRITE_SIZE']) * 1024
        kernels[name] = {'write_bytes': write, 'read_bytes': 2 * fetch}
        total += write + 2 * fetch
    out = {config: {'config': config,
                    'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE in separate passes (tools/prof_r03.sh), '
                              'summary in profiles/r03_c4_pmc_summary.txt',
                    'correction': 'read bytes = 2 * FETCH_SIZE * 1024 (128-byte requests tallied at 64 bytes; calibrated for streams and '
                                  'for 72-byte gathers: profiles/r03_fetch_calibration.txt), write bytes = WRITE_SIZE * 1024',
                    'kernels': kernels, 'chain_bytes': total, 'kernels_sha': kernels_sha()}}
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'r03_traffic.json'), 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
