#!/usr/bin/env python3
"""Derived SQ-counter table from the means of tools/pmc_summary.py (the output of tools/pmc_drn_split.sh, gpurun_out/sq_drn_split.txt):
cycles = SQ_BUSY_CYCLES / 32 (one SQ per shader engine, 8 XCDs x 4); MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles);
VALU/MFMA and SALU/MFMA = other vector (SQ_INSTS_VALU - SQ_INSTS_MFMA) and scalar instructions per matrix instruction;
waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES.      python tools/sq_table.py gpurun_out/sq_drn_split.txt"""
import collections
import re
import sys

vals = collections.defaultdict(dict)
launches = {}
for line in open(sys.argv[1]):
    m = re.match(r'(\S.*?)\s+(SQ_\w+)\s+launches\s+(\d+)\s+mean\s+([\d.]+)', line)
    if m:
        vals[m.group(1)][m.group(2)] = float(m.group(4))
        launches[m.group(1)] = int(m.group(3))
print('%-42s %8s %10s %9s %10s %10s %9s %9s %9s' % ('kernel', 'launches', 'cycles M', 'MFMA M', 'MFMA busy', 'VALU/MFMA', 'SALU/MFMA', 'LDS M', 'waiting'))
for k in sorted(vals):
    v = vals[k]
    cyc = v.get('SQ_BUSY_CYCLES', 0.0) / 32.0
    mf = v.get('SQ_INSTS_MFMA', 0.0)
    busy = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (1024.0 * cyc) if cyc else 0.0
    wait = v.get('SQ_WAIT_ANY', 0.0) / v['SQ_WAVE_CYCLES'] if v.get('SQ_WAVE_CYCLES') else 0.0
    print('%-42s %8d %10.2f %9.1f %10s %10s %9s %9.1f %8.0f%%' % (
        k[:42], launches[k], cyc / 1e6, mf / 1e6, ('%.1f%%' % (100 * busy)) if mf else '-',
        ('%.2f' % ((v.get('SQ_INSTS_VALU', 0.0) - mf) / mf)) if mf else '-', ('%.2f' % (v.get('SQ_INSTS_SALU', 0.0) / mf)) if mf else '-',
        v.get('SQ_INSTS_LDS', 0.0) / 1e6, 100 * wait))
