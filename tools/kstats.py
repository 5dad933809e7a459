#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run: per kernel calls / total / avg / min / max (us), libspalign
kernels first.   python tools/kstats.py DIR [substring ...]"""
import csv
import glob
import os
import sys

root = sys.argv[1]
want = sys.argv[2:]
rows = {}
for path in glob.glob(os.path.join(root, '**', '*kernel_stats.csv'), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r['Name'].split('(')[0]
            rows[name] = (int(r['Calls']), float(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3,
                          float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3)
keys = sorted(rows, key=lambda k: -rows[k][1])
print('| kernel | calls | total ms | avg us | min us | max us |\n|---|---|---|---|---|---|')
for k in keys:
    if want and not any(w in k for w in want):
        continue
    if not want and not k.startswith(('k_', 'void k_')):
        continue
    c, tot, avg, mn, mx = rows[k]
    print('| %s | %d | %.2f | %.1f | %.1f | %.1f |' % (k[:60], c, tot / 1e3, avg, mn, mx))
