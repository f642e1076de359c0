#!/usr/bin/env python3
"""FETCH_SIZE (rocprofv3 --pmc, KB) of tools/ubench/fetch_calib against the bytes each pattern touches.
usage: python3 tools/fetch_calib_summary.py <rocprof output dir> <stdout of fetch_calib>"""
import csv
import glob
import re
import sys
from collections import defaultdict

known = {}
for line in open(sys.argv[2]):
    if line.startswith('CALIB'):
        w = line.split()
        known[w[1]] = {w[i]: int(w[i + 1]) for i in range(2, len(w), 2)}
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if row['Counter_Name'] == 'FETCH_SIZE':
            name = re.sub(r'[<(].*', '', row['Kernel_Name']).replace('k_', '')
            acc[name].append(float(row['Counter_Value']) * 1024)
for name, vals in sorted(acc.items()):
    fetch = sum(vals) / len(vals)
    print('%-9s FETCH_SIZE*1024 = %.4e B' % (name, fetch))
    for k, v in known.get(name, {}).items():
        print('          %-20s %.4e B   counter / bytes = %.3f   correction factor = %.3f' % (k, v, fetch / v, v / fetch))
