#!/usr/bin/env python3
"""Development aid (round 6): how many SLIC centres / labels are bit-stable from one Lloyd sweep to the next — what a
'skip the tiles whose candidate centres did not change' scheme could save."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
eng = importlib.import_module('superpixel-align_amd.engine').Engine()
synth = importlib.import_module('superpixel-align_amd.synth')
H, W, n = 1024, 2048, 200
x = torch.from_numpy(synth.synth_batch([11, 12, 13, 14], H, W, integer_valued=True)).cuda()
lab = eng.rgb2lab(x, 10.0) if 'compactness' in eng.rgb2lab.__code__.co_varnames else eng.rgb2lab(x)
prev_l = prev_c = None
for k in range(1, 11):
    l, c = eng.slic_core(lab, n, max_iter=k, want_centres=True)
    if prev_l is not None:
        ch = (l != prev_l).float().mean().item()
        same = (c.view(torch.int32) == prev_c.view(torch.int32)).all(dim=2).float().mean().item()
        print('sweep %2d: %.4f %% of the labels changed; %.1f %% of the centres bit-identical to the sweep before' % (k, 100 * ch, 100 * same))
    prev_l, prev_c = l, c
