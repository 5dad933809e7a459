#!/usr/bin/env python3
"""Development aid: felzenszwalb time at the reference operating point (30 images of 224x224)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
spa = importlib.import_module('superpixel-align_amd')
eng = importlib.import_module('superpixel-align_amd.engine').default_engine()
x = torch.from_numpy(np.stack([spa.synth.synth_scene(s, 224, 224) for s in range(30)])).cuda()
eng.felzenszwalb(x, 300.0, 0.8, 20); torch.cuda.synchronize()
t = time.time()
for _ in range(5):
    lab, nl = eng.felzenszwalb(x, 300.0, 0.8, 20)
torch.cuda.synchronize()
print('felzenszwalb 30 x 224x224: %.2f ms per batch; segments %s' % ((time.time() - t) / 5 * 1e3, nl[:4].tolist()))
