#!/usr/bin/env python3
"""Predicted strong-scaling speed-up of C4 from the emulated slabs (tools/prof_r03.sh slabs): per W the slowest slab by
MEDIAN and by per-step MAXIMUM against the whole patch.  usage: python3 tools/slab_table.py <dir with c4_*bench.json>"""
import glob
import json
import os
import re
import sys

d = sys.argv[1]


def line(path):
    for l in open(path):
        if l.startswith('{'):
            return json.loads(l)
    return None


full = line(os.path.join(d, 'c4_bench.json'))
if full is None:
    sys.exit('no c4_bench.json in ' + d)
fm, fx = full['step_ms']['median'], full['step_ms']['max']
print('whole patch: median %.3f ms, max %.3f ms (cold %.1f ms, set-up %.4f s)' % (fm, fx, full.get('cold_ms', 0), full.get('setup_s', 0)))
print('%3s  %-44s %9s %9s %8s %8s %9s' % ('W', 'slab medians (ms)', 'max med', 'max max', 'x(med)', 'x(max)', 'max/min'))
for W in (2, 4, 8):
    med, mx = [], []
    for r in range(W):
        p = os.path.join(d, 'c4_slab%dof%d_bench.json' % (r, W))
        if not os.path.exists(p):
            break
        j = line(p)
        med.append(j['step_ms']['median']); mx.append(j['step_ms']['max'])
    if len(med) == W:
        print('%3d  %-44s %9.3f %9.3f %8.2f %8.2f %9.3f' % (W, ' '.join('%.2f' % x for x in med), max(med), max(mx), fm / max(med), fx / max(mx), max(med) / min(med)))
