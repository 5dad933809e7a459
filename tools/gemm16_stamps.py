#!/usr/bin/env python3
"""Development aid (round 5): in-kernel stamps of the staggered GEMM (k_gemm_f16x3_stag<..., XP & 16>): per period and wave (0 =
early half, 4 = late half) the cycles between the stamps  0 top | 1 staged (early) | 2 R done | 3 late's wait done | 4 mid
barrier passed | 5 M done | 6 end wait done | 7 end barrier passed.
    (libspalign built with make EXTRA=-DSPA_DIAG)   SPA_GEMM16_STAGGER=33 python tools/gemm16_stamps.py        (32 + RS: stamps of the production form, RS = 1 or 2; 41: no split, 49: no loads)"""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
eng = importlib.import_module('superpixel-align_amd.engine').Engine()
torch.manual_seed(0)
Cin = Cout = 512
B, H, W = 30, 128, 256
x = torch.relu(torch.randn((B, Cin, H, W), device='cuda')).contiguous(memory_format=torch.channels_last)
w = torch.randn((Cout, Cin, 3, 3), device='cuda') * (2.0 / (9 * Cin)) ** 0.5
bias = torch.randn((Cout,), device='cuda')
u2, cs = eng.winograd_weights_split(w)
am = eng.amax(x)
for _ in range(3):
    y, ao = eng.conv3x3_wino_f16s(x, u2, cs, bias, None, True, 4, amax_in=am)
torch.cuda.synchronize()
NQ = 40
buf = np.zeros((2, NQ, 8), np.uint32)
rc = eng._lib.spa_debug_peek(eng._ctx, -1, 0, buf.nbytes, buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
names = ['stage(e)', 'R', 'wait(l)', 'barrier', 'stage(l)+M', 'epi+wait(e)', 'barrier']
for wv in range(2):
    t = buf[wv].astype(np.int64)
    d = (t[:, 1:] - t[:, :-1]) & 0xffffffff
    per = (t[1:, 0] - t[:-1, 0]) & 0xffffffff
    print('wave %d (%s half): period %.0f cycles (median; min %d max %d)' % (4 * wv, 'late' if wv else 'early', np.median(per), per.min(), per.max()))
    print('   median cycles: ' + ' | '.join('%s %.0f' % (n, np.median(d[:, i])) for i, n in enumerate(names)))
    print('   mean cycles:   ' + ' | '.join('%s %.0f' % (n, d[:, i].mean()) for i, n in enumerate(names)))
# skew between the halves: late's mid barrier vs early's end barrier of the same wall-clock event
e, l = buf[0].astype(np.int64), buf[1].astype(np.int64)
print('early end-barrier passed - late mid-barrier passed (same event, cycles):', np.median((e[:, 7] - l[:, 4]) & 0xffffffff))
