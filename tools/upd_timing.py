#!/usr/bin/env python3
"""Development aid for k_slic_update: with libspalign built with -DSPA_UPD_TIMING, print the
wave cycles each phase of the update kernel took (summed over waves and sweeps).
    make -C superpixel-align_amd/csrc EXTRA=-DSPA_UPD_TIMING && python tools/upd_timing.py --batch 1"""
import argparse
import ctypes
import importlib
import os
import re
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--n', type=int, default=200)
a = ap.parse_args()
spa = importlib.import_module('superpixel-align_amd')
eng_mod = importlib.import_module('superpixel-align_amd.engine')
lib_mod = importlib.import_module('superpixel-align_amd._lib')
bench = importlib.import_module('bench')
eng = eng_mod.default_engine()
imgs_h, _ = bench.make_batch(spa.synth, a.batch, 1024, 2048)
imgs = torch.from_numpy(imgs_h).cuda()
src = open(os.path.join(os.path.dirname(lib_mod.__file__), 'csrc', 'spa_common.h')).read()
names = re.findall(r'^\s*(WS_[A-Z_0-9]+)\s*(?:=\s*0)?,', src, re.M)
which = names.index('WS_SLIC_ORDER')
eng.slic(imgs, a.n)
torch.cuda.synchronize()
total = a.batch * lib_mod.make_plan(1024, 2048, a.n).n_centroids
host = (ctypes.c_uint64 * 6)()
lib_mod.check(lib_mod.lib().spa_debug_peek(eng._ctx, which, (total + 8) * 4 + 32, 48, host))
lab = ['piece list', 'index expansion', 'stage rows (incl. load waits) + publish', 'wait for a ring slot (chain wave)', 'round loop rest', 'issue rounds (index reads, addresses, loads)']
tot = sum(host)
for name, v in zip(lab, host):
    print('%-32s %14d cycles  %5.1f %%' % (name, v, 100.0 * v / max(tot, 1)))
print('per segment-sweep: %.0f cycles' % (tot / (total * 9.0)))
