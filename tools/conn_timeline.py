#!/usr/bin/env python3
"""Timeline (us) of the connectivity kernels of the last profiled pass in a rocprofv3 --kernel-trace csv directory:
which kernels overlap on the side streams and where the pass waits.   python tools/conn_timeline.py DIR"""
import csv
import glob
import os
import sys

f = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r['Kernel_Name'].startswith(('k_run', 'k_conn', 'k_small'))]
idx = max(i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_conn_init_misc'))
t0 = int(rows[idx]['Start_Timestamp'])
for r in sorted(rows[idx:], key=lambda r: int(r['Start_Timestamp'])):
    a, b = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print('%-22s start %8.1f  end %8.1f  dur %7.1f  queue %s  lds %s' % (
        r['Kernel_Name'].split('(')[0], a / 1e3, b / 1e3, (b - a) / 1e3, r['Queue_Id'], r['LDS_Block_Size']))
