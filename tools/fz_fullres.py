#!/usr/bin/env python3
"""Development aid: felzenszwalb timing at full resolution for several workgroups-per-image
settings (SPA_FZ_GROUP) and batch sizes.   python tools/fz_fullres.py"""
import importlib
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import numpy as np
    import torch
    spa = importlib.import_module('superpixel-align_amd')
    eng = importlib.import_module('superpixel-align_amd.engine').default_engine()
    B = int(sys.argv[1])
    imgs = torch.from_numpy(np.stack([spa.synth.synth_image(i, 1024, 2048) for i in range(B)])).cuda()
    eng.felzenszwalb(imgs, 300.0, 0.8, 20)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(2):
        lab, n = eng.felzenszwalb(imgs, 300.0, 0.8, 20)
    torch.cuda.synchronize()
    print('B=%d G=%s: %.1f ms per batch, %.1f ms per image, segments %d' % (
        B, os.environ.get('SPA_FZ_GROUP', '1'), (time.time() - t0) / 2 * 1e3, (time.time() - t0) / 2 / B * 1e3, int(n[0])))
    # pass diagnostics of the last call (k_fz_pass_tab): windows, chunks, rounds summed over images and both passes
    import ctypes, re
    lib_mod = importlib.import_module('superpixel-align_amd._lib')
    src = open(os.path.join(os.path.dirname(lib_mod.__file__), 'csrc', 'spa_common.h')).read()
    names = re.findall(r'^\s*(WS_[A-Z_0-9]+)\s*(?:=\s*0)?,', src, re.M)
    host = (ctypes.c_int32 * 13)()
    lib_mod.check(lib_mod.lib().spa_debug_peek(eng._ctx, names.index('WS_CONNMISC'), 256 * 32 + 256 * 4, 52, host))
    print('   per image and both passes: %.0f windows, %.0f chunks of 1024 sorted edges, %.0f full rounds, %.0f tail rounds' % tuple(v / float(B) for v in host[:4]))
    print('   kilo-cycles per image: flatten %.0f | collect %.0f | window set-up + write-back %.0f | full rounds %.0f | tail %.0f' % tuple(v / float(B) for v in host[4:9]))
    print('   of the rounds: propagation %.0f kilo-cycles in %.0f steps | hub chains %.0f kilo-cycles' % (host[9] / float(B), host[12] / float(B), host[10] / float(B)))
    print('   of the set-up: table reset + cost loads %.0f kilo-cycles (the rest: table entry)' % (host[11] / float(B)))
else:
    for B in (1, 8, 30):
        for G in (1, 2, 4, 8):
            env = dict(os.environ, SPA_FZ_GROUP=str(G))
            subprocess.call([sys.executable, __file__, str(B)], env=env)
