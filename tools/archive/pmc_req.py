#!/usr/bin/env python3
"""mean of every counter per kernel from rocprofv3 counter_collection.csv files.  usage: python3 tools/pmc_req.py <dir>"""
import csv, glob, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        acc[re.sub(r'[<(].*', '', row['Kernel_Name'])][row['Counter_Name']].append(float(row['Counter_Value']))
for k, v in sorted(acc.items()):
    print(k, {c: round(sum(x) / len(x), 1) for c, x in sorted(v.items())})
