#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc output (…_counter_collection.csv files under a directory) into
per-kernel, per-counter means per launch.   python tools/pmc_summary.py DIR [kernel-substring]"""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row['Kernel_Name'].split('(')[0]
            if want and want not in name:
                continue
            a = acc[(name, row['Counter_Name'])]
            a[0] += float(row['Counter_Value'])
            a[1] += 1
for (name, ctr), (tot, n) in sorted(acc.items()):
    print('%-40s %-28s launches %4d  mean %16.1f' % (name[:40], ctr, n, tot / n))
