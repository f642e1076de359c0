#!/usr/bin/env python3
"""Compile one .hip file for gfx950 and print VGPR/SGPR/scratch/occupancy per kernel
(hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel."""
import re
import subprocess
import sys

src = sys.argv[1]
out = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Iinclude', '-I../../include',
                      '--cuda-device-only', '-c', src, '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r'\(.*', '', cur)
        rows[cur] = {}
        continue
    m = re.search(r'remark: \s*(SGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)', line)
    if m and cur:
        rows[cur][m.group(1).split(' ')[0]] = int(m.group(2))
print('%-60s %5s %5s %8s %5s' % ('kernel', 'VGPR', 'SGPR', 'scratch', 'occ'))
for k, v in rows.items():
    print('%-60s %5d %5d %8d %5d' % (k[:60], v.get('VGPRs', -1), v.get('SGPRs', -1), v.get('ScratchSize', -1), v.get('Occupancy', -1)))
