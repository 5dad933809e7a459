#!/usr/bin/env python3
"""Development aid (round 6): in-kernel stamps of k_conv3x3_p16 (a libspalign built with EXTRA=-DSPA_CP_STAMPS): per group the cycles
between ten points of waves 0 and 4 (the two waves of one SIMD) of one workgroup:
 0 top | 1 reads of taps 0, 1 + staging issued | 2 tap 0's matrix instructions issued | 3 wait for the next segment done | 4 mid
 barrier passed | 5 taps 1, 2 + conversion issued | 6 leftover pairs | 7 epilogue (if any) | 8 end wait done | 9 end barrier passed
    python tools/convp_stamps.py [C]        (C = 64 or 128)"""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
eng_mod = importlib.import_module('superpixel-align_amd.engine')
eng = eng_mod.default_engine()
torch.manual_seed(0)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B, H, W = (30, 256, 512) if C == 64 else (30, 128, 256)
x = torch.randn(B, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
w = torch.randn(C, C, 3, 3, device='cuda') * 0.05
b = torch.randn(C, device='cuda')
wt2, inv_t = eng_mod.Engine.split_planes(w.permute(0, 2, 3, 1).reshape(C, 9, C).contiguous())
am = eng.amax(x)
for _ in range(3):
    y, a2 = eng.conv3x3_f16s(x, wt2, inv_t, b, None, True, 1, amax_in=am)
torch.cuda.synchronize()
NQ = 48
buf = np.zeros((2, NQ, 10), np.uint32)
rc = eng._lib.spa_debug_peek(eng._ctx, -1, 0, buf.nbytes, buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
names = ['reads+stage', 'mfma tap0', 'wait seg', 'barrier 1', 'taps 1,2 + conv', 'leftover', 'epilogue', 'wait w', 'barrier 2']
for wv in range(2):
    t = buf[wv].astype(np.int64)
    d = (t[:, 1:] - t[:, :-1]) & 0xffffffff
    per = (t[1:, 0] - t[:-1, 0]) & 0xffffffff
    print('C %d wave %d: group period %.0f cycles (median; min %d max %d)' % (C, 4 * wv, np.median(per), per.min(), per.max()))
    print('   median: ' + ' | '.join('%s %.0f' % (n, np.median(d[:, i])) for i, n in enumerate(names)))
    print('   mean:   ' + ' | '.join('%s %.0f' % (n, d[:, i].mean()) for i, n in enumerate(names)))
