#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean of each counter."""
import csv
import glob
import re
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        name = re.sub(r'\(.*', '', row['Kernel_Name'])
        name = name.replace('void igx::', '').replace('igx::', '')
        if not name.startswith('k_'):
            continue
        acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
# kernel durations
dur = defaultdict(list)
for f in glob.glob(root + '/**/*kernel_trace.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        name = re.sub(r'\(.*', '', row['Kernel_Name']).replace('void igx::', '').replace('igx::', '')
        if name.startswith('k_'):
            dur[name].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6)
for name in sorted(acc):
    d = dur.get(name, [0])
    print('== %s   calls/pass=%d  mean %.3f ms' % (name, len(d), sum(d) / max(1, len(d))))
    for c in sorted(acc[name]):
        v = acc[name][c]
        print('   %-28s %.6g' % (c, sum(v) / len(v)))
