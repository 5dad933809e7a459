#!/usr/bin/env python3
"""Development aid: reservation rounds of the felzenszwalb passes (diagnostic counters)."""
import ctypes, importlib, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
spa = importlib.import_module('superpixel-align_amd'); engine = importlib.import_module('superpixel-align_amd.engine')
eng = engine.Engine()
L = spa._lib.lib()
L.spa_debug_peek.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
WS_CONNMISC = int(sys.argv[3]) if len(sys.argv) > 3 else 10
H, W, B = int(sys.argv[1]), int(sys.argv[2]), 8
for gen in ('scene', 'image'):
    imgs = np.stack([(spa.synth.synth_scene if gen == 'scene' else spa.synth.synth_image)(s, H, W) for s in range(B)])
    x = torch.from_numpy(imgs).cuda()
    eng.felzenszwalb(x, 300.0, 0.8, 20); torch.cuda.synchronize()
    t = time.time(); lab, nl = eng.felzenszwalb(x, 300.0, 0.8, 20); torch.cuda.synchronize(); dt = time.time() - t
    buf = np.zeros(8 * B, np.int32)
    L.spa_debug_peek(eng._ctx, WS_CONNMISC, 0, buf.nbytes, buf.ctypes.data_as(ctypes.c_void_p))
    st = buf.reshape(B, 8)
    print(gen, '%dx%d B=%d: %.2f ms; segments %s; rounds/prop-steps/windows per image (both passes):' % (H, W, B, dt * 1e3, nl.cpu().tolist()[:4]),
          st[:, 5].tolist()[:4], st[:, 6].tolist()[:4], st[:, 7].tolist()[:4])
