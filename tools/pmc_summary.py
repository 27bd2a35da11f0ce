#!/usr/bin/env python
"""Sum rocprofv3 counter_collection.csv values per (kernel, counter): tools/pmc_summary.py dir [dir ...]"""
import csv, glob, sys, collections, re

def short(name):
    name = name.replace('(anonymous namespace)::', '')
    name = re.sub(r'\(.*', '', name)
    return name.replace('void ', '')[:40]

for d in sys.argv[1:]:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        tot = collections.defaultdict(float); n = collections.defaultdict(int)
        for row in csv.DictReader(open(f)):
            k = (short(row['Kernel_Name']), row['Counter_Name'])
            tot[k] += float(row['Counter_Value']); n[k] += 1
        for (k, c), v in sorted(tot.items()):
            print(f'{k:40s} {c:28s} launches {n[(k, c)]:5d} total {v:.6g}')
